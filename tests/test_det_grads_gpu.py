"""Deterministic embedding gradients (csrc/segsum.hip): the stable radix sort of the lookup ids, the segmented sums that
write every gradient row once in a fixed order, and the whole training step run twice from the same state -- the gradient
tables, clip norms and updated tables must be BIT-identical (reference semantics of the sums: tf.train.AdamOptimizer's
unsorted_segment_sum over the IndexedSlices of tf.nn.embedding_lookup, models/base_model.py:263-276,
models/sequential/sequential_base_model.py:381-452; the reference itself makes no ordering promise -- this is a property
of the build, asked for by the reviews)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402

DEV = "cuda"


def _stable_sort(tables):
    """tables: list of (ids tensor [nrows, row_stride] int32 on the device, nrows, ncols, vocab[, second id tensor [n2]])"""
    rows, outs, total = [], [], 0
    for tab in tables:
        ids, nrows, ncols, V = tab[:4]
        ids2 = tab[4] if len(tab) > 4 else None
        n = nrows * ncols + (ids2.numel() if ids2 is not None else 0)
        k = torch.full((n,), -7, dtype=torch.int32, device=DEV)
        p = torch.full((n,), -7, dtype=torch.int32, device=DEV)
        outs.append((k, p))
        row = (ids.data_ptr(), k.data_ptr(), p.data_ptr(), nrows, ids.stride(0), ncols, max(1, (V - 1).bit_length()))
        if ids2 is not None:
            row += (ids2.data_ptr(), ids2.numel(), 1)
        rows.append(row)
        total += n
    ws = torch.empty(query("clsr_sort_ids_stable_workspace_bytes", total, len(rows)), dtype=torch.uint8, device=DEV)
    ops.sort_ids_stable_multi(rows, ws)
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("shapes", [
    [(4096, 50, 64138, True), (4096, 50, 4096, True), (20480, 1, 64138, True), (20480, 1, 4096, False), (4096, 1, 36915, False)],
    [(1024, 250, 292286, True)],
    [(4096, 50, 100_000_000, False), (4096, 50, 10000, False)],
    [(3, 1, 7, False), (1, 1, 2, False), (17, 3, 1 << 20, False)],
    [(700, 3, 300, True)]])
def test_stable_radix_sort_of_lookup_ids(shapes):
    """clsr_sort_ids_stable_multi == torch.sort(stable=True) of the same ids: ascending ids, equal ids in position order, for
    several tables of different sizes / vocabularies in one launch set, strided id rows included."""
    g = torch.Generator().manual_seed(len(shapes) * 7 + shapes[0][0])
    tables, flat = [], []
    for nrows, ncols, V, zipf in shapes:
        n = nrows * ncols
        if zipf:
            ids = (torch.rand(n, generator=g).pow(6.0) * (V - 1)).long() + 1
            ids[::97] = 0
        else:
            ids = torch.randint(0, V, (n,), generator=g)
        wide = torch.full((nrows, ncols + 3), -1, dtype=torch.int32)        # a row stride that is not the column count
        wide[:, :ncols] = ids.view(nrows, ncols).int()
        d = wide.to(DEV)
        tables.append((d[:, :ncols], nrows, ncols, V))
        flat.append(ids)
    outs = _stable_sort(tables)
    for (k, p), ids in zip(outs, flat):
        ek, ep = torch.sort(ids, stable=True)
        assert torch.equal(k.cpu().long(), ek), "keys"
        assert torch.equal(p.cpu().long(), ep), "positions (stable order)"


def _segsum(sites, V_C):
    rows = []
    for s in sites:
        rows.append(s)
    ws = torch.zeros(ops.segsum_workspace_bytes(rows), dtype=torch.uint8, device=DEV)      # (zero when first used: launch epoch / counters)
    ops.segsum_multi(rows, ws)
    torch.cuda.synchronize()
    assert ops.query("clsr_segsum_error", ws.data_ptr()) == 0, "a tail gave up waiting for an earlier chunk's partial"


@pytest.mark.parametrize("Hn,T,Di,Dc,V,zipf", [(64, 50, 32, 8, 40, False), (33, 10, 32, 8, 5000, False), (16, 7, 96, 32, 300, True),
                                               (512, 50, 96, 32, 1_000_000, False), (4096, 50, 32, 8, 64138, True),
                                               (40, 50, 128, 8, 77, False), (5, 1, 20, 4, 3, False)])
def test_deterministic_segmented_history_gradient(Hn, T, Di, Dc, V, zipf):
    """clsr_segsum_multi on the stable-sorted ids == float64 index_add of every slice (mean / recent-k terms of the history
    prologue included, second addend included), item and category tables in ONE call; squared norms of the un-summed
    slices; and two calls give bit-identical tables."""
    g = torch.Generator().manual_seed(Hn + V % 97)
    D, k, n = Di + Dc, 3, Hn * T
    Vc = max(2, min(V, 37))
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    if zipf:
        ii = ((torch.rand(n, generator=g).pow(5.0) * (V - 1)).long() + 1).view(Hn, T)
    else:
        ii = torch.randint(0, V, (Hn, T), generator=g)
    ci = torch.randint(0, Vc, (Hn, T), generator=g)
    d_ii, d_ci = ii.int().to(DEV), ci.int().to(DEV)
    (ki, pi), (kc, pc) = _stable_sort([(d_ii, Hn, T, V), (d_ci, Hn, T, Vc)])
    dh, dh2 = torch.randn(Hn, T, D, generator=g), torch.randn(Hn, T, D, generator=g)
    dm, dr = torch.randn(Hn, D, generator=g), torch.randn(Hn, D, generator=g)
    d_len = lens.int().to(DEV)
    dev = lambda t: t.to(DEV).contiguous()
    a, b, m_, r_ = dev(dh), dev(dh2), dev(dm), dev(dr)
    res = []
    for _ in range(2):
        gi, gc = torch.zeros(V, Di, device=DEV), torch.zeros(Vc, Dc, device=DEV)
        gi[min(3, V - 1)] = 1.5                                  # rows are ADDED to (another site may have written them)
        ss = torch.zeros(2, dtype=torch.float64, device=DEV)
        rows = [(a.data_ptr(), b.data_ptr(), m_.data_ptr(), r_.data_ptr(), ki.data_ptr(), pi.data_ptr(), d_len.data_ptr(),
                 gi.data_ptr(), ss[0:].data_ptr(), n, 0, 1, T, D, 0, Di, k, Di, 0, 0),
                (a.data_ptr(), b.data_ptr(), m_.data_ptr(), r_.data_ptr(), kc.data_ptr(), pc.data_ptr(), d_len.data_ptr(),
                 gc.data_ptr(), ss[1:].data_ptr(), n, 0, 1, T, D, Di, Dc, k, Dc, 0, 0)]
        _segsum(rows, None)
        res.append((gi, gc, ss))
    m = (torch.arange(T)[None, :] < lens[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    gfull = dh.double() + dh2.double() + m[..., None] * (dm.double() / m.sum(1, keepdim=True))[:, None, :] \
        + rec[..., None] * (dr.double() / rec.sum(1, keepdim=True))[:, None, :]
    ei = torch.zeros(V, Di, dtype=torch.float64).index_add_(0, ii.reshape(-1), gfull[..., :Di].reshape(-1, Di))
    ei[min(3, V - 1)] += 1.5
    ec = torch.zeros(Vc, Dc, dtype=torch.float64).index_add_(0, ci.reshape(-1), gfull[..., Di:].reshape(-1, Dc))
    gi, gc, ss = res[0]
    scale = float(ei.abs().max())
    err = (gi.double().cpu() - ei).abs().max()
    assert float(err) <= 2e-6 * scale + 1e-5, "item gradient: max err %.3e at scale %.3e" % (float(err), scale)
    errc = (gc.double().cpu() - ec).abs().max()
    assert float(errc) <= 2e-6 * float(ec.abs().max()) + 1e-5, "category gradient: max err %.3e" % float(errc)
    exp_ss = torch.tensor([float((gfull[..., :Di] ** 2).sum()), float((gfull[..., Di:] ** 2).sum())])
    assert torch.allclose(ss.cpu(), exp_ss.double(), rtol=1e-5)
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y), "two runs of the segmented sums must be bit-identical"


def test_deterministic_row_level_sites():
    """target / user lookups as sites with T = 1: slices = rows of a [B, ld] gradient matrix, column slices, duplicates"""
    g = torch.Generator().manual_seed(9)
    B, V, ld, col0, C = 20480, 5000, 40, 32, 8
    idx = (torch.rand(B, generator=g).pow(3.0) * (V - 1)).long()
    src = torch.randn(B, ld, generator=g)
    d_idx = idx.int().to(DEV).view(B, 1)
    (k, p), = _stable_sort([(d_idx, B, 1, V)])
    grad = torch.zeros(V, C, device=DEV)
    ss = torch.zeros(1, dtype=torch.float64, device=DEV)
    d_src = src.to(DEV)
    _segsum([(d_src.data_ptr(), 0, 0, 0, k.data_ptr(), p.data_ptr(), 0, grad.data_ptr(), ss.data_ptr(), B, 0, 0, 1, ld, col0, C,
              1, C, 0, 0)], None)
    exp = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx, src.double()[:, col0:col0 + C])
    assert float((grad.double().cpu() - exp).abs().max()) <= 1e-5 * float(exp.abs().max())
    assert abs(float(ss) - float((src[:, col0:col0 + C].double() ** 2).sum())) <= 1e-6 * float(ss)


@pytest.mark.parametrize("Hn,T,B,Di,Dc,V", [(64, 50, 320, 32, 8, 500), (4096, 50, 20480, 32, 8, 64138), (128, 50, 640, 96, 32, 1_000_000)])
def test_history_and_target_sites_in_one_list_stored_once(Hn, T, B, Di, Dc, V):
    """The CLSR step's form: the target rows' ids ride behind the history lookup's ids in ONE sorted list (second id source
    of the sort), the segmented sums take the target rows' slices from a second gradient matrix, keep the two sites' squared
    norms apart, and STORE every row total (assign: the rows start as zeros, garbage here to prove they are not read)."""
    g = torch.Generator().manual_seed(B + V % 13)
    D, k, n = Di + Dc, 3, Hn * T
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    ii = ((torch.rand(n, generator=g).pow(3.0) * (V - 1)).long()).view(Hn, T)
    it = (torch.rand(B, generator=g).pow(3.0) * (V - 1)).long()
    d_ii, d_it = ii.int().to(DEV), it.int().to(DEV)
    (ks, ps), = _stable_sort([(d_ii, Hn, T, V, d_it)])
    allids = torch.cat([ii.reshape(-1), it])
    ek, ep = torch.sort(allids, stable=True)
    assert torch.equal(ks.cpu().long(), ek) and torch.equal(ps.cpu().long(), ep)
    dh = torch.randn(Hn, T, D, generator=g)
    dm, dr = torch.randn(Hn, D, generator=g), torch.randn(Hn, D, generator=g)
    dt = torch.randn(B, D, generator=g)
    a, m_, r_, t_ = dh.to(DEV), dm.to(DEV), dr.to(DEV), dt.to(DEV)
    d_len = lens.int().to(DEV)
    grad = torch.full((V, Di), 777.0, device=DEV)          # (assign mode never reads the rows)
    ss = torch.zeros(4, dtype=torch.float64, device=DEV)
    rows = [(a.data_ptr(), 0, m_.data_ptr(), r_.data_ptr(), ks.data_ptr(), ps.data_ptr(), d_len.data_ptr(), grad.data_ptr(),
             ss[0:].data_ptr(), n + B, 0, 1, T, D, 0, Di, k, Di, 0, 1, t_.data_ptr(), ss[2:].data_ptr(), n, D, 0)]
    _segsum(rows, None)
    m = (torch.arange(T)[None, :] < lens[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    gfull = dh.double() + m[..., None] * (dm.double() / m.sum(1, keepdim=True))[:, None, :] \
        + rec[..., None] * (dr.double() / rec.sum(1, keepdim=True))[:, None, :]
    exp = torch.zeros(V, Di, dtype=torch.float64).index_add_(0, ii.reshape(-1), gfull[..., :Di].reshape(-1, Di))
    exp.index_add_(0, it, dt.double()[:, :Di])
    touched = torch.zeros(V, dtype=torch.bool)
    touched[allids] = True
    got = grad.double().cpu()
    assert float((got[touched] - exp[touched]).abs().max()) <= 2e-6 * float(exp.abs().max()) + 1e-5
    assert bool((got[~touched] == 777.0).all()), "rows without a slice are not written"
    assert abs(float(ss[0]) - float((gfull[..., :Di] ** 2).sum())) <= 1e-5 * float(ss[0])
    assert abs(float(ss[2]) - float((dt.double()[:, :Di] ** 2).sum())) <= 1e-5 * float(ss[2])
    assert float(ss[1]) == 0.0 and float(ss[3]) == 0.0


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("Hn,T,B,Di,Dc,V,site", [(128, 50, 640, 96, 32, 1_000_000, "item"), (64, 50, 320, 32, 8, 500, "item"),
                                                  (256, 50, 1280, 96, 32, 3000, "cate"), (33, 7, 165, 32, 8, 40, "cate"),
                                                  (512, 50, 2560, 96, 32, 20000, "item")])
def test_lean_walk_with_a_second_source_fp32_and_bf16_slices(Hn, T, B, Di, Dc, V, site, bf16):
    """The LEAN instantiation (no second gradient tensor, no mean / recent shares, rows stored once) that BASELINE configs[4]
    and bench.py's gather_bwd run: software-pipelined walk with its keys staged in LDS, fp32 d(hist) rows or bf16 rows read as
    ONE 16-byte load per entry (the pair-lane form), the target rows' fp32 slices as the second source -- against an
    index_add in float64, with the two sites' squared norms, twice (bit-identical)."""
    g = torch.Generator().manual_seed(Hn + V % 17 + int(bf16))
    D, n = Di + Dc, Hn * T
    col0, C = (0, Di) if site == "item" else (Di, Dc)
    ii = ((torch.rand(n, generator=g).pow(2.0) * (V - 1)).long()).view(Hn, T)
    it = (torch.rand(B, generator=g).pow(2.0) * (V - 1)).long()
    (ks, ps), = _stable_sort([(ii.int().to(DEV), Hn, T, V, it.int().to(DEV))])
    dh = torch.randn(Hn, T, D, generator=g)
    if bf16:
        dh = dh.to(torch.bfloat16)
    dt = torch.randn(B, D, generator=g)
    a, t_ = dh.to(DEV), dt.to(DEV)
    d_len = torch.full((Hn,), T, dtype=torch.int32, device=DEV)
    outs = []
    for _ in range(2):
        grad = torch.full((V, C), 777.0, device=DEV)
        ss = torch.zeros(4, dtype=torch.float64, device=DEV)
        rows = [(a.data_ptr(), 0, 0, 0, ks.data_ptr(), ps.data_ptr(), d_len.data_ptr(), grad.data_ptr(), ss[0:].data_ptr(), n + B,
                 int(bf16), 1, T, D, col0, C, 3, C, 0, 1, t_.data_ptr(), ss[2:].data_ptr(), n, D, col0)]
        _segsum(rows, None)
        outs.append((grad.clone(), ss.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    grad, ss = outs[0]
    src = dh.double()[..., col0:col0 + C].reshape(-1, C)
    exp = torch.zeros(V, C, dtype=torch.float64).index_add_(0, ii.reshape(-1), src)
    exp.index_add_(0, it, dt.double()[:, col0:col0 + C])
    touched = torch.zeros(V, dtype=torch.bool)
    touched[torch.cat([ii.reshape(-1), it])] = True
    got = grad.double().cpu()
    assert float((got[touched] - exp[touched]).abs().max()) <= 2e-6 * float(exp.abs().max()) + 1e-5
    assert bool((got[~touched] == 777.0).all()), "rows without a slice are not written"
    assert abs(float(ss[0]) - float((src ** 2).sum())) <= 1e-5 * float(ss[0])
    assert abs(float(ss[2]) - float((dt.double()[:, col0:col0 + C] ** 2).sum())) <= 1e-5 * float(ss[2])


def test_training_step_is_bit_reproducible(golden_dir, golden_hparams):
    """The same step from the same state, twice: every trained embedding table, its Adam moments, the dense variables and the
    clip norms come out bit-identical."""
    import pickle

    from clsr_amd.net import CLSRNet
    from oracle import clsr_oracle as O

    hp = golden_hparams
    dims = dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))
    gfeed = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = {k[3:]: gfeed[k] for k in gfeed.files if k.startswith("b0_")}
    params = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    runs = []
    for _ in range(2):
        net = CLSRNet(hp, dims, device="cuda:0", seed=0)
        assert net.det_grads
        net.load_state_dict(copy.deepcopy(sd))
        net.capture_grads = True
        f = net.upload(feed, True)
        for _ in range(2):
            net.train_step(f)
        torch.cuda.synchronize()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        state.update({"m." + k: t.clone() for k, t in net.tab_m.items()})
        # squared norms of the un-summed slices of the six lookup sites (the other slots are regulariser norms)
        state["sumsq"] = net.captured["table_sumsq"][[0, 1, 2, 3, 6, 7]].clone()
        state.update({"g." + k: t.clone() for k, t in net.captured["tables"].items()})
        runs.append(state)
    a, b = runs
    tab = [k for k in a if "embedding" in k or k.startswith(("m.", "g.")) or k == "sumsq"]
    assert len(tab) >= 9
    for k in tab:
        assert torch.equal(a[k], b[k]), "%s differs between two runs of the same step" % k


@pytest.mark.parametrize("n,V,C,power", [(225280, 64138, 32, 5.0), (100000, 7, 96, 1.0), (40000, 1, 8, 1.0), (70001, 3000, 40, 3.0),
                                        (9, 2, 8, 1.0), (33, 1, 96, 1.0)])
def test_long_runs_one_launch_and_a_workspace_that_is_never_cleared(n, V, C, power):
    """Runs that span many chunks (one id for the whole list, seven ids, a Zipf head of ~800 chunks) are combined INSIDE the
    one launch by the chunk in which they end (decoupled look-back over the earlier chunks' partials), short continuations are
    walked by the chunk they come from; the workspace is zeroed once and then reused by launches on DIFFERENT lists (its ready
    words are launch epochs): every launch == float64 index_add, the squared norms are right, the error word stays 0."""
    g = torch.Generator().manual_seed(n + V)
    ws = None
    for rep in range(4):
        idx = (torch.rand(n, generator=g).pow(power) * V).long().clamp_(max=V - 1)
        src = torch.randn(n, C, generator=g)
        d_idx = idx.int().to(DEV).view(n, 1)
        (k, p), = _stable_sort([(d_idx, n, 1, V)])
        assign = rep % 2
        grad = torch.full((V, C), 0.25 if not assign else 777.0, device=DEV)
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        d_src = src.to(DEV)
        rows = [(d_src.data_ptr(), 0, 0, 0, k.data_ptr(), p.data_ptr(), 0, grad.data_ptr(), ss.data_ptr(), n, 0, 0, 1, C, 0, C, 1, C,
                 0, assign)]
        if ws is None:
            ws = torch.zeros(ops.segsum_workspace_bytes(rows), dtype=torch.uint8, device=DEV)
        ops.segsum_multi(rows, ws)
        torch.cuda.synchronize()
        assert ops.query("clsr_segsum_error", ws.data_ptr()) == 0
        exp = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx, src.double())
        touched = torch.zeros(V, dtype=torch.bool)
        touched[idx] = True
        got = grad.double().cpu()
        base = 0.0 if assign else 0.25
        err = float((got[touched] - exp[touched] - base).abs().max())
        assert err <= 3e-6 * float(exp.abs().max()) + 1e-5, (rep, err)
        assert bool((got[~touched] == (777.0 if assign else 0.25)).all())
        assert abs(float(ss) - float((src.double() ** 2).sum())) <= 1e-6 * float(ss)
        grad2 = torch.full((V, C), 0.25 if not assign else 777.0, device=DEV)
        rows2 = [rows[0][:7] + (grad2.data_ptr(),) + rows[0][8:]]
        ops.segsum_multi(rows2, ws)
        torch.cuda.synchronize()
        assert torch.equal(grad, grad2), "two launches on one list: bit-identical rows"
