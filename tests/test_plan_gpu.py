"""Replaying a recorded launch plan (ops.LaunchPlan: the C-ABI calls + stream-event operations of one step with their
arguments already converted) is the same computation as launching the step eagerly: bit-identical losses, logits and
variables over several steps, for the training step, the data-parallel backward-only step and eval scoring."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd.net import CLSRNet  # noqa: E402


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _feed(golden_dir, name, b=0):
    g = np.load(os.path.join(golden_dir, name))
    pre = "b%d_" % b
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _run(hp, golden_dir, use_plans, steps=5):
    net = CLSRNet(hp, _dims(hp), device="cuda:0", seed=3)
    net.use_plans = use_plans
    net.sorted_hist_grad = True
    feeds = [_feed(golden_dir, "iterator_train_sa.npz", b) for b in (0, 1)]
    ev = _feed(golden_dir, "iterator_eval_sa.npz", 0)
    st = [None, None]
    ev_st = None
    trace = []
    for i in range(steps):
        for k in (0, 1):            # two feeds of the same shape share ONE static device feed, like CLSRModel does
            st[0] = net.upload(feeds[k], True, into=st[0])
            out = net.train_step(st[0])
            trace.append((net.losses.clone(), out["logit"].clone()))
        ev_st = net.upload(ev, False, into=ev_st)
        trace.append((net.forward(ev_st, False)["logit"].clone(),))
    torch.cuda.synchronize()
    replayed = sum(1 for e in net._step_plans.values() if e[1] is not None)
    return trace, net.state_dict(), replayed


def test_replayed_steps_equal_eager_steps(golden_dir, golden_hparams):
    hp = copy.deepcopy(golden_hparams)
    hp.learning_rate = 1e-5     # Adam moves every weight by ~lr whatever its gradient: keeps the last-bit noise of the
    a, sd_a, n_a = _run(hp, golden_dir, False)      # float atomics from growing into visible differences
    b, sd_b, n_b = _run(hp, golden_dir, True)
    assert n_a == 0 and n_b == 2          # one plan for the training step, one for eval scoring
    assert len(a) == len(b)
    def far(x, y):
        x, y = x.double().cpu(), y.double().cpu()
        return float(((x - y).abs() - (1e-5 + 1e-3 * y.abs())).max())

    # float atomics in the embedding gradient make two runs differ in the last bits (eager vs eager as well);
    # a missing or mis-ordered launch would be orders of magnitude away
    bad = [(i, j, far(x, y)) for i, (ta, tb) in enumerate(zip(a, b)) for j, (x, y) in enumerate(zip(ta, tb))
           if far(x, y) > 0]
    assert not bad, bad[:6]
    # (biases in front of a batch-norm and the softmax-invariant output bias have an exactly-zero gradient: what they
    # receive is fp32 noise, and Adam turns its sign into +-lr steps -- not comparable between any two runs)
    bad = [(k, far(sd_a[k], sd_b[k])) for k in sd_a if "/b_nn_" not in k and far(sd_a[k], sd_b[k]) > 0]
    assert not bad, bad[:6]


def test_plan_is_not_reused_across_changed_scalars(golden_dir, golden_hparams):
    hp = copy.deepcopy(golden_hparams)
    net = CLSRNet(hp, _dims(hp), device="cuda:0", seed=3)
    f = net.upload(_feed(golden_dir, "iterator_train_sa.npz", 0), True)
    for _ in range(3):
        net.train_step(f)
    assert sum(1 for e in net._step_plans.values() if e[1] is not None) == 1
    hp.learning_rate = 0.5          # a different Adam clock argument: a new key, the old plan is not replayed
    before = net.state_dict()["sequential/logit_fcn/nn_part/w_nn_output"].clone()
    net.train_step(f)
    torch.cuda.synchronize()
    after = net.state_dict()["sequential/logit_fcn/nn_part/w_nn_output"]
    assert float((after - before).abs().max()) > 0.05      # an lr = 0.5 Adam step, not an lr = 1e-3 one
    assert len(net._step_plans) == 2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hipgraph_capture_of_the_step_equals_eager(golden_dir, golden_hparams, precision):
    """The whole training step (four streams: fork / join through events, the dense path finishing on the weight-gradient
    stream) captured as ONE hipGraph and replayed == the same steps launched eagerly (bench.py --graph)."""
    from clsr_amd import ops

    hp = copy.deepcopy(golden_hparams)
    hp.learning_rate = 1e-5
    feed = _feed(golden_dir, "iterator_train_sa.npz", 0)
    res = []
    for graph in (False, True):
        net = CLSRNet(hp, _dims(hp), device="cuda:0", seed=3, precision=precision)
        net.use_plans = False
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            f = net.upload(feed, True)
            net.train_step(f)                      # buffers allocated, weights packed
            stream.synchronize()
            if graph:
                ops.graph_begin()
                net.train_step(f)
                g = ops.graph_end()
                run = lambda: ops.graph_launch(g)
            else:
                run = lambda: net.train_step(f)
            for _ in range(3):
                run()
            stream.synchronize()
        res.append((net.losses.clone().cpu(), {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (la, sa), (lb, sb) = res
    assert float((la - lb).abs().max()) <= 1e-6 * float(la.abs().max())
    for k in sa:
        if sa[k].dtype.is_floating_point:
            d = float((sa[k].double() - sb[k].double()).abs().max())
            assert d <= 1e-5 + 1e-3 * float(sa[k].abs().max()), (k, d)
