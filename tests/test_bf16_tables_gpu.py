"""GPU parity tests of the bf16 embedding-table kernels ("bf16 tables + activations";
reference lookups: models/sequential/sequential_base_model.py:381-452, sparse optimiser models/base_model.py:263-276):
the history gather with bf16 rows (a pure copy: bit-exact), the sorted segmented gradient with a bf16 d(hist), and the
lazy-Adam row update of a bf16 table -- each against a bf16-ROUNDING restatement of the same arithmetic (float64 sums of
the widened values; the update computed in fp32 and rounded to nearest-even).  Also the wide-row (16-byte) form of the
fp32 segmented gradient against the scalar form and the float64 index_add."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd.ops import call, query  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def _close(got, exp, rtol, atol, name):
    got, exp = got.double().cpu().reshape(-1), exp.double().cpu().reshape(-1)
    err = (got - exp).abs()
    excess = float((err - (atol + rtol * exp.abs())).max())
    assert excess <= 0, "%s: max abs err %.3e (max |exp| %.3e)" % (name, float(err.max()), float(exp.abs().max()))


@pytest.mark.parametrize("Hn,T,Di,Dc,G,out_bf16", [(37, 10, 32, 8, 5, 1), (64, 50, 32, 8, 1, 0), (9, 7, 96, 32, 3, 1),
                                                   (130, 50, 96, 32, 5, 0), (5, 3, 8, 8, 2, 1)])
def test_gather_hist_fwd_bf16_tables(Hn, T, Di, Dc, G, out_bf16):
    g = torch.Generator().manual_seed(Hn + T)
    Vi, Vc, k = 301, 23, 3
    item_tbl = torch.randn(Vi, Di, generator=g).to(DEV).to(BF)
    cate_tbl = torch.randn(Vc, Dc, generator=g).to(DEV).to(BF)
    B = Hn * G
    lens_h = torch.randint(1, T + 1, (Hn,), generator=g)
    lens_h[0], lens_h[-1] = 1, T
    lens = lens_h.repeat_interleave(G)
    valid = torch.arange(T)[None, :] < lens[:, None]
    ii = torch.where(valid, torch.randint(1, Vi, (B, T), generator=g), torch.zeros(B, T, dtype=torch.long))
    ci = torch.where(valid, torch.randint(1, Vc, (B, T), generator=g), torch.zeros(B, T, dtype=torch.long))
    ii = ii.view(Hn, G, T)[:, :1].expand(Hn, G, T).reshape(B, T).contiguous()
    ci = ci.view(Hn, G, T)[:, :1].expand(Hn, G, T).reshape(B, T).contiguous()
    D = Di + Dc
    hist = torch.full((Hn, T, D), 7.0, device=DEV, dtype=BF if out_bf16 else torch.float32)
    hm, hr = torch.empty(Hn, D, device=DEV), torch.empty(Hn, D, device=DEV)
    d_ii, d_ci = ii.int().to(DEV), ci.int().to(DEV)
    call("clsr_gather_hist_fwd_h", item_tbl, cate_tbl, d_ii, d_ci, G * T, lens.int().to(DEV), G, Hn, T, Di, Dc, k, hist,
         out_bf16, hm, hr)
    torch.cuda.synchronize()
    iih, cih, lh = ii[::G].to(DEV), ci[::G].to(DEV), lens[::G]
    exp = torch.cat([item_tbl[iih], cate_tbl[cih]], -1)            # bf16 rows
    assert torch.equal(hist.to(BF), exp), "the gather is a copy: every gathered value must be bit-exact"
    e64 = exp.double().cpu()
    m = (torch.arange(T)[None, :] < lh[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    _close(hm, (e64 * m[..., None]).sum(1) / m.sum(1, keepdim=True), 1e-5, 1e-6, "hist_mean (fp32 sums of the widened rows)")
    _close(hr, (e64 * rec[..., None]).sum(1) / rec.sum(1, keepdim=True), 1e-5, 1e-6, "hist_recent")
    # == the fp32 kernel on the widened tables
    hist2 = torch.empty(Hn, T, D, device=DEV)
    hm2, hr2 = torch.empty(Hn, D, device=DEV), torch.empty(Hn, D, device=DEV)
    call("clsr_gather_hist_fwd", item_tbl.float(), cate_tbl.float(), d_ii, d_ci, G * T, lens.int().to(DEV), G, Hn, T, Di,
         Dc, k, hist2, hm2, hr2)
    torch.cuda.synchronize()
    assert torch.equal(hist2, hist.float())
    _close(hm, hm2, 1e-6, 1e-7, "hist_mean vs fp32 kernel on the widened table")


def _sorted_inputs(Hn, T, V, g, G=1):
    n = Hn * T
    idx = torch.randint(0, V, (Hn, T), generator=g).repeat_interleave(G, 0).contiguous()
    nbytes = query("clsr_sort_ids_workspace_bytes", n, V)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    keys = torch.empty(n, dtype=torch.int32, device=DEV)
    perm = torch.empty(n, dtype=torch.int32, device=DEV)
    call("clsr_sort_ids", idx.int().to(DEV), Hn, T, G * T, V, keys, perm, ws, nbytes)
    return idx, keys, perm


@pytest.mark.parametrize("Hn,T,Di,Dc,V", [(64, 50, 32, 8, 40), (33, 10, 32, 8, 5000), (16, 7, 96, 32, 300),
                                          (512, 50, 96, 32, 1_000_000), (40, 50, 128, 8, 77)])
def test_wide_row_segmented_gradient(Hn, T, Di, Dc, V):
    """16-byte form of clsr_gather_bwd_sorted2 (one launch for a 96- or 128-float row slice): == float64 index_add of every
    slice, == the scalar form (CLSR_GBS_SCALAR path is exercised by the narrow-column tests of test_kernels_gpu.py);
    fp32 and bf16 d(hist)."""
    g = torch.Generator().manual_seed(Hn + V % 97)
    D, k, n = Di + Dc, 3, Hn * T
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    idx, keys, perm = _sorted_inputs(Hn, T, V, g)
    dh = torch.randn(Hn, T, D, generator=g)
    dh2 = torch.randn(Hn, T, D, generator=g)
    dm, dr = torch.randn(Hn, D, generator=g), torch.randn(Hn, D, generator=g)
    assert query("clsr_gather_bwd_sorted_max_cols", D, 0, Di, Di, 0) == 256
    m = (torch.arange(T)[None, :] < lens[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    for bf in (0, 1):
        a, b = (dh.to(BF), dh2.to(BF)) if bf else (dh, dh2)
        grad = torch.zeros(V, Di, device=DEV)
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        call("clsr_gather_bwd_sorted2_h" if bf else "clsr_gather_bwd_sorted2", a.to(DEV), b.to(DEV), dm.to(DEV), dr.to(DEV),
             keys, perm, lens.int().to(DEV), 1, n, T, D, 0, Di, k, grad, Di, 0, ss)
        torch.cuda.synchronize()
        gfull = a.double() + b.double() + m[..., None] * (dm.double() / m.sum(1, keepdim=True))[:, None, :] \
            + rec[..., None] * (dr.double() / rec.sum(1, keepdim=True))[:, None, :]
        exp = torch.zeros(V, Di, dtype=torch.float64).index_add_(0, idx.reshape(-1), gfull[..., :Di].reshape(-1, Di))
        _close(grad, exp, 1e-4, 1e-4, "sorted grad (bf16 d(hist): %d)" % bf)
        _close(ss, (gfull[..., :Di] ** 2).sum().reshape(1), 1e-5, 0.0, "sumsq")
    # the category columns of the same rows: a slice that starts at column Di
    grad_c = torch.zeros(V, Dc, device=DEV)
    call("clsr_gather_bwd_sorted2", dh.to(DEV), None, None, None, keys, perm, lens.int().to(DEV), 1, n, T, D, Di, Dc, k,
         grad_c, Dc, 0, None)
    torch.cuda.synchronize()
    exp_c = torch.zeros(V, Dc, dtype=torch.float64).index_add_(0, idx.reshape(-1), dh.double()[..., Di:].reshape(-1, Dc))
    _close(grad_c, exp_c, 1e-4, 1e-4, "column slice")


@pytest.mark.parametrize("V,C,nrows", [(5000, 96, 777), (300, 32, 300), (100000, 128, 4096)])
def test_lazy_adam_rows_on_a_bf16_table(V, C, nrows):
    """clsr_table_adam_rows_h == the fp32 row update (clsr_table_adam_rows) applied to the WIDENED rows and rounded to
    nearest-even when stored; moments and the cleared gradient rows identical to the fp32 kernel's."""
    g = torch.Generator().manual_seed(V)
    tbl_h = (torch.randn(V, C, generator=g) * 0.05).to(DEV).to(BF)
    tbl_f = tbl_h.float()
    ids = torch.randperm(V, generator=g)[:nrows].sort()[0].int().to(DEV)
    count = torch.tensor([nrows], dtype=torch.int32, device=DEV)
    grad = torch.zeros(V, C, device=DEV)
    grad[ids.long()] = torch.randn(nrows, C, generator=g).to(DEV) * 1e-2
    m0, v0 = torch.randn(V, C, generator=g).to(DEV) * 1e-3, torch.rand(V, C, generator=g).to(DEV) * 1e-5
    flags = torch.zeros(V, dtype=torch.uint8, device=DEV)
    flags[ids.long()] = 1
    sumsq = torch.tensor([float((grad.double() ** 2).sum())], dtype=torch.float64, device=DEV)
    state = torch.tensor([3.0, 0.9 ** 3, 0.999 ** 3, 1e-3 * (1 - 0.999 ** 3) ** 0.5 / (1 - 0.9 ** 3), 0.0],
                         dtype=torch.float64, device=DEV)
    res = []
    for h in (1, 0):
        t = (tbl_h if h else tbl_f).clone()
        gr, m, v, fl = grad.clone(), m0.clone(), v0.clone(), flags.clone()
        call("clsr_table_adam_rows_h" if h else "clsr_table_adam_rows", t, gr, m, v, fl, ids, count, nrows, C, sumsq, 1, 1,
             2.0, state, 0.9, 0.999, 1e-8)
        torch.cuda.synchronize()
        res.append((t, gr, m, v, fl))
    (th, gh, mh, vh, fh), (tf, gf, mf, vf, ff) = res
    # (the two kernels are compiled separately: their fused-multiply-add contractions can differ in the last bit of the
    #  moments; an update that lands within that of a bf16 rounding boundary may then round the other way)
    d_ulp = (th.view(torch.int16).int() - tf.to(BF).view(torch.int16).int()).abs()
    assert int(d_ulp.max()) <= 1 and float((d_ulp > 0).float().mean()) < 1e-3, \
        "bf16 rows = the fp32 update of the widened rows, rounded to nearest-even"
    _close(mh, mf, 1e-6, 2e-9, "first moments")
    _close(vh, vf, 1e-6, 1e-12, "second moments")
    assert torch.equal(gh, gf) and torch.equal(fh, ff)
    assert float(gh.abs().max()) == 0.0 and int(fh.sum()) == 0
    untouched = torch.ones(V, dtype=torch.bool, device=DEV)
    untouched[ids.long()] = False
    assert torch.equal(th[untouched], tbl_h[untouched])
    assert not torch.equal(th[ids.long()], tbl_h[ids.long()])


# ------------------------------------------------------------------------------- the whole step with bf16 tables
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("lazy", [False, True])
def test_step_with_bf16_tables(golden_dir, golden_hparams, precision, lazy):
    """CLSRNet(table_dtype="bf16"): the step on bf16 tables == the step of a net whose fp32 tables hold the SAME (bf16-
    representable) values -- logits, losses and gradient tables bit for bit (the lookups return identical fp32 values and
    everything behind them is deterministic) -- and its updated tables are that net's updated tables rounded to
    nearest-even; against the float64 oracle on the rounded tables: logits at the exact-mode / speed-mode bars."""
    import copy
    import os
    import pickle

    import numpy as np

    from clsr_amd.net import CLSRNet
    from clsr_amd.params import TABLES
    from oracle import clsr_oracle as O

    hp = copy.deepcopy(golden_hparams)
    hp.item_embedding_dim, hp.cate_embedding_dim = 32, 8
    if lazy:
        hp.optimizer = "lazyadam"
    dims = dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))
    gf = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = {k[3:]: gf[k] for k in gf.files if k.startswith("b1_")}
    params = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    for name in TABLES.values():                      # bf16-representable table values
        params[name] = params[name].to(BF).float()
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    nets = {}
    for td in ("bf16", "fp32"):
        net = CLSRNet(hp, dims, device="cuda:0", seed=0, precision=precision, table_dtype=td)
        net.load_state_dict(copy.deepcopy(sd))
        net.capture_grads = True
        out = net.train_step(net.upload(feed, True))
        torch.cuda.synchronize()
        nets[td] = (net, out["logit"].clone(), net.read_losses(), {k: v.clone() for k, v in net.captured["tables"].items()})
    (nh, lh, lsh, gh), (nf, lf, lsf, gfp) = nets["bf16"], nets["fp32"]
    assert nh.tables["item"].dtype == BF and nf.tables["item"].dtype == torch.float32
    assert torch.equal(lh, lf), "same looked-up values -> identical logits"
    for k in gh:
        assert torch.equal(gh[k], gfp[k]), "gradient table %s" % k
    for k in lsh:
        assert abs(lsh[k] - lsf[k]) <= 1e-9 * max(1.0, abs(lsf[k])), k
    changed = 0
    for k in nh.tables:
        a, b = nh.tables[k], nf.tables[k].to(BF)
        d = (a.view(torch.int16).int() - b.view(torch.int16).int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3, "updated table %s" % k
        changed += int((nf.tables[k] != params[TABLES[k]].to("cuda")).sum())
    assert changed > 0
    # the checkpoint payload is fp32 and round-trips
    sd2 = nh.state_dict()
    assert all(v.dtype != BF for v in sd2.values())
    n2 = CLSRNet(hp, dims, device="cuda:0", seed=1, precision=precision, table_dtype="bf16")
    n2.load_state_dict(sd2)
    for k in nh.tables:
        assert torch.equal(n2.tables[k], nh.tables[k])
    # against the oracle on the rounded tables
    tf = O.to_torch_feed(feed, dtype=torch.float64)
    p64 = type(params)((k, v.double()) for k, v in params.items())
    _, _, _, ls, _, _, oout = O.train_step(p64, O.init_bn_state(p64), O.init_adam(p64), 1, tf, hp)
    tol = 1e-4 if precision == "fp32" else 1e-3
    _close(lh, oout["logit"].reshape(-1), tol, tol, "logit vs oracle")
