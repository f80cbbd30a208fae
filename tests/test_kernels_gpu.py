"""GPU parity tests of every HIP kernel against torch-CPU float64 restatements of the same op
(and the oracle's cells for the recurrent kernels).  All calls go through the C ABI
(clsr_amd.ops.call -> libclsr_hip.so).  fp32 kernels: tolerances ~1e-5 relative."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def close(got, exp, rtol=2e-5, atol=2e-6, name=""):
    got = got.detach().double().cpu()
    exp = exp.detach().double().cpu()
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    err = (got - exp).abs()
    tol = atol + rtol * exp.abs()
    worst = float((err - tol).max())
    assert worst <= 0, "%s: max abs err %.3e (max |exp| %.3e), worst excess %.3e" % (
        name, float(err.max()), float(exp.abs().max()), worst)


def rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, dtype=torch.float64) * scale)


# ------------------------------------------------------------------------------- embedding
@pytest.mark.parametrize("Hn,T,Di,Dc,G", [(37, 10, 32, 8, 5), (64, 50, 32, 8, 1), (9, 7, 96, 32, 3)])
def test_gather_hist_fwd_bwd(Hn, T, Di, Dc, G):
    g = torch.Generator().manual_seed(1)
    Vi, Vc, k = 301, 23, 3
    item_tbl, cate_tbl = rnd(g, Vi, Di), rnd(g, Vc, Dc)
    B = Hn * G
    lens_h = torch.randint(1, T + 1, (Hn,), generator=g)
    lens_h[0] = 1
    lens_h[-1] = T
    lens = lens_h.repeat_interleave(G)
    valid = torch.arange(T)[None, :] < lens[:, None]
    ii = torch.where(valid, torch.randint(1, Vi, (B, T), generator=g), torch.zeros(B, T, dtype=torch.long))
    ci = torch.where(valid, torch.randint(1, Vc, (B, T), generator=g), torch.zeros(B, T, dtype=torch.long))
    # replicate group rows like the iterator does
    ii = ii.view(Hn, G, T)[:, :1].expand(Hn, G, T).reshape(B, T).contiguous()
    ci = ci.view(Hn, G, T)[:, :1].expand(Hn, G, T).reshape(B, T).contiguous()
    D = Di + Dc
    hist = torch.empty(Hn, T, D, device="cuda")
    hm = torch.empty(Hn, D, device="cuda")
    hr = torch.empty(Hn, D, device="cuda")
    d_it, d_ct = dev(item_tbl, torch.float32), dev(cate_tbl, torch.float32)
    d_ii, d_ci, d_len = dev(ii, torch.int32), dev(ci, torch.int32), dev(lens, torch.int32)
    call("clsr_gather_hist_fwd", d_it, d_ct, d_ii, d_ci, G * T, d_len, G, Hn, T, Di, Dc, k, hist, hm, hr)
    iih, cih, lh = ii[::G], ci[::G], lens[::G]
    exp = torch.cat([item_tbl[iih], cate_tbl[cih]], -1)
    m = (torch.arange(T)[None, :] < lh[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    close(hist, exp.float(), name="hist")
    close(hm, (exp * m[..., None]).sum(1) / m.sum(1, keepdim=True), name="hist_mean")
    close(hr, (exp * rec[..., None]).sum(1) / rec.sum(1, keepdim=True), name="hist_recent")
    # backward
    dh, dm, dr = rnd(g, Hn, T, D), rnd(g, Hn, D), rnd(g, Hn, D)
    gi = torch.zeros(Vi, Di, device="cuda")
    gc = torch.zeros(Vc, Dc, device="cuda")
    ss = torch.zeros(2, dtype=torch.float64, device="cuda")
    call("clsr_gather_hist_bwd", dev(dh, torch.float32), dev(dm, torch.float32), dev(dr, torch.float32),
         d_ii, d_ci, G * T, d_len, G, Hn, T, Di, Dc, k, gi, gc, ss)
    gfull = dh + m[..., None] * (dm / m.sum(1, keepdim=True))[:, None, :] \
        + rec[..., None] * (dr / rec.sum(1, keepdim=True))[:, None, :]
    egi = torch.zeros(Vi, Di, dtype=torch.float64).index_add_(0, iih.reshape(-1), gfull[..., :Di].reshape(-1, Di))
    egc = torch.zeros(Vc, Dc, dtype=torch.float64).index_add_(0, cih.reshape(-1), gfull[..., Di:].reshape(-1, Dc))
    close(gi, egi, rtol=1e-4, atol=1e-5, name="item_grad")
    close(gc, egc, rtol=1e-4, atol=1e-4, name="cate_grad")
    close(ss, torch.stack([(gfull[..., :Di] ** 2).sum(), (gfull[..., Di:] ** 2).sum()]), rtol=1e-5, name="sumsq")


def test_gather_scatter_rows_and_flags():
    g = torch.Generator().manual_seed(2)
    V, C, N, G = 97, 40, 55, 5
    tbl = rnd(g, V, C)
    idx = torch.randint(0, V, (N * G,), generator=g)
    out = torch.zeros(N, 48, device="cuda")
    call("clsr_gather_rows", dev(tbl, torch.float32), dev(idx, torch.int32), G, N, C, out, 48, 8)
    close(out[:, 8:], tbl[idx[::G]].float(), name="gather_rows")
    assert float(out[:, :8].abs().max()) == 0.0
    src = rnd(g, N, 48)
    grad = torch.zeros(V, C, device="cuda")
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    call("clsr_scatter_add_rows", dev(src, torch.float32), 48, 8, dev(idx, torch.int32), G, N, C, grad, ss)
    exp = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx[::G], src[:, 8:])
    close(grad, exp, rtol=1e-4, atol=1e-5, name="scatter")
    close(ss, (src[:, 8:] ** 2).sum().reshape(1), name="scatter sumsq")
    flags = torch.zeros(V, dtype=torch.uint8, device="cuda")
    ids = torch.randint(0, V, (7 * G, 6), generator=g)
    call("clsr_mark_rows", dev(ids, torch.int32), 7, 6, G * 6, flags)
    exp_f = torch.zeros(V, dtype=torch.uint8)
    exp_f[ids[::G].reshape(-1)] = 1
    assert torch.equal(flags.cpu(), exp_f)
    cnt = torch.zeros(1, device="cuda")
    call("clsr_count_flags", flags, V, cnt)
    assert float(cnt) == float(exp_f.sum())


@pytest.mark.parametrize("Hn,T,Di,Dc,G,V", [(64, 50, 32, 8, 5, 40), (33, 10, 32, 8, 1, 5000), (16, 7, 96, 32, 2, 300)])
def test_sorted_segmented_history_gradient(Hn, T, Di, Dc, G, V):
    """sort (id, position) pairs + run-length sums == index_add of every slice (heavy duplicates: V small)."""
    g = torch.Generator().manual_seed(Hn)
    D, k = Di + Dc, 3
    B = Hn * G
    lens = torch.randint(1, T + 1, (Hn,), generator=g).repeat_interleave(G)
    idx = torch.randint(0, V, (Hn, T), generator=g).repeat_interleave(G, 0).contiguous()
    dh, dm, dr = rnd(g, Hn, T, D), rnd(g, Hn, D), rnd(g, Hn, D)
    n = Hn * T
    nbytes = query("clsr_sort_ids_workspace_bytes", n, V)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    d_idx = dev(idx, torch.int32)
    call("clsr_sort_ids", d_idx, Hn, T, G * T, V, keys, perm, ws, nbytes)
    flat = idx[::G].reshape(-1)
    ks, pm = keys.cpu().long(), perm.cpu().long()
    assert torch.equal(ks, torch.sort(flat)[0]) and torch.equal(flat[pm], ks)
    grad = torch.zeros(V, Di, device="cuda")
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    d_len = dev(lens, torch.int32)
    for c0 in range(0, Di, 64):
        call("clsr_gather_bwd_sorted", dev(dh, torch.float32), dev(dm, torch.float32), dev(dr, torch.float32), keys,
             perm, d_len, G, n, T, D, c0, min(64, Di - c0), k, grad, Di, c0, ss)
    lh = lens[::G]
    m = (torch.arange(T)[None, :] < lh[:, None]).double()
    pos = torch.flip(torch.cumsum(torch.flip(m, [1]), 1), [1])
    rec = ((pos >= 1) & (pos <= k)).double()
    gfull = dh + m[..., None] * (dm / m.sum(1, keepdim=True))[:, None, :] \
        + rec[..., None] * (dr / rec.sum(1, keepdim=True))[:, None, :]
    exp = torch.zeros(V, Di, dtype=torch.float64).index_add_(0, flat, gfull[..., :Di].reshape(-1, Di))
    close(grad, exp, rtol=1e-4, atol=1e-4, name="sorted grad")
    close(ss, (gfull[..., :Di] ** 2).sum().reshape(1), rtol=1e-5, name="sumsq")


@pytest.mark.parametrize("Hn,T,V,zipf", [(4096, 50, 64138, True), (1024, 250, 292286, True), (4096, 50, 100_000_000, False),
                                         (3, 1, 7, False)])
def test_counting_sort_groups_ids_at_benchmark_sizes(Hn, T, V, zipf):
    """clsr_sort_ids at the BASELINE shapes: a permutation, keys == ids[perm], ascending ids up to 2^18-id vocabularies
    and grouped by (id mod 2^18) beyond (100M-item catalogue), Zipf-head duplicates included."""
    g = torch.Generator().manual_seed(V % 1000)
    n = Hn * T
    if zipf:
        ids = (torch.rand(n, generator=g).pow(6.0) * (V - 1)).long() + 1     # heavy head: thousands of copies of id 1
        ids[::97] = 0                                                       # padding row
    else:
        ids = torch.randint(0, V, (n,), generator=g)
    d_idx = dev(ids.reshape(Hn, T), torch.int32)
    nbytes = query("clsr_sort_ids_workspace_bytes", n, V)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    keys = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    perm = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    call("clsr_sort_ids", d_idx, Hn, T, T, V, keys, perm, ws, nbytes)
    ks, pm = keys.cpu().long(), perm.cpu().long()
    assert torch.equal(torch.sort(pm)[0], torch.arange(n))
    assert torch.equal(ids[pm], ks)
    bucket = ks & ((1 << query("clsr_sort_ids_bits", V)) - 1)
    assert bool((bucket[1:] >= bucket[:-1]).all())
    if V <= (1 << 18):
        assert torch.equal(ks, torch.sort(ids)[0])


# ------------------------------------------------------------------------------- pgemm
def _pgemm(X, W, bias=None, T=0, G=0, Xmul=None, in_scale=None, in_shift=None, relu=0, addU=None,
           addV=None, Y=None, accumulate=0, stats=False, M=None, ldx=None):
    K, N = W.shape
    M = X.shape[0] if M is None else M
    Wt, Kp = ops.pack_weight(dev(W, torch.float32), N, K)
    if Y is None:
        Y = torch.zeros(M, N, device="cuda")
    st = None
    if stats:
        st = torch.zeros(query("clsr_pgemm_stats_parts", M), 2, N, dtype=torch.float64, device="cuda")
    f = lambda t: None if t is None else dev(t, torch.float32)
    dX = f(X)
    call("clsr_pgemm", dX, ldx or X.shape[1], T, G, f(Xmul), 0 if Xmul is None else Xmul.shape[1],
         f(in_scale), f(in_shift), relu, Wt, Kp, f(bias), f(addU), 0 if addU is None else addU.shape[1],
         f(addV), 0 if addV is None else addV.shape[1], Y, Y.shape[1], accumulate, st, M, K, N)
    return Y, st


@pytest.mark.parametrize("M,K,N", [(1000, 80, 80), (333, 40, 40), (77, 164, 80), (515, 100, 64), (2304, 256, 256), (768, 256, 256),
                                   (260, 40, 240), (129, 80, 100), (50, 64, 4), (2000, 120, 120),
                                   (200, 516, 80), (300, 1536, 128), (90, 320, 80)])   # wide K: chunked launches
def test_pgemm_plain_bias_stats(M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    X, W, b = rnd(g, M, K), rnd(g, K, N, scale=0.3), rnd(g, N)
    Y, st = _pgemm(X, W, b, stats=True)
    exp = X @ W + b
    tol = 1e-5 * max(1.0, (K / 100.0) ** 0.5)
    close(Y, exp, rtol=1e-5, atol=tol, name="Y")
    tot = st.sum(0).cpu()
    close(tot[0], exp.sum(0), rtol=1e-5, atol=10 * tol, name="colsum")
    close(tot[1], (exp ** 2).sum(0), rtol=1e-5, atol=10 * tol * max(1.0, K / 100.0), name="colsumsq")
    Y2, _ = _pgemm(X, W, None, Y=Y.clone(), accumulate=1)
    close(Y2, 2 * exp - b, rtol=1e-5, atol=2 * tol, name="accumulate")


@pytest.mark.parametrize("M,K,N", [(32768, 1536, 128), (40000, 512, 96), (33000, 260, 40), (32768 + 77, 1000, 128)])
def test_pgemm_wide_k_in_kernel_loop(M, K, N):
    """Wide inputs on many positions (K >= 256, N <= 128, M >= 32 768: d(hist) = dPin . W_x^T of BASELINE configs[4]) take the
    kernel that loops over K itself (pgemm_kloop_kernel) instead of a chain of launches over K ranges: plain product, bias,
    accumulation, ragged last workgroup, K not a multiple of the 32-wide stage, a strided input."""
    g = torch.Generator().manual_seed(M + K + N)
    X, W, b = rnd(g, M, K + 8), rnd(g, K, N, scale=0.3), rnd(g, N)
    Xs = X[:, :K]
    exp = Xs @ W + b
    tol = 1e-5 * max(1.0, (K / 100.0) ** 0.5)
    Y, _ = _pgemm(X, W, b, M=M, ldx=K + 8)
    close(Y, exp, rtol=1e-5, atol=tol, name="Y")
    Y2, _ = _pgemm(X, W, None, Y=Y.clone(), accumulate=1, M=M, ldx=K + 8)
    close(Y2, 2 * exp - b, rtol=2e-5, atol=4 * tol, name="accumulate")     # (two fp32 roundings of sums of ~1 500 products)


def test_pgemm_rowmap_mul_affine_adds():
    g = torch.Generator().manual_seed(5)
    Hn, G, T, K, N = 13, 5, 10, 80, 80
    R = Hn * G
    a = rnd(g, Hn * T, K)          # history-level keys projections
    q = rnd(g, R, K)               # row-level queries
    W = rnd(g, K, N, scale=0.2)
    U, V = rnd(g, Hn * T, N), rnd(g, R, N)
    Y, st = _pgemm(a, W, None, T=T, G=G, Xmul=q, addU=U, addV=V, stats=True, M=R * T)
    rows = torch.arange(R * T)
    r, t = rows // T, rows % T
    xrow = (r // G) * T + t
    exp = (a[xrow] * q[r]) @ W + U[xrow] + V[r]
    close(Y, exp, rtol=1e-5, atol=2e-5, name="att z0")
    close(st.sum(0)[0].cpu(), exp.sum(0), rtol=1e-5, atol=1e-3, name="stats")
    sc, sh = rnd(g, K), rnd(g, K)
    X = rnd(g, 301, K)
    Y, _ = _pgemm(X, W, None, in_scale=sc, in_shift=sh, relu=1)
    close(Y, torch.relu(X * sc + sh) @ W, rtol=1e-5, atol=2e-5, name="bn-relu prologue")
    # transposed pack (dX = dY . W^T) and strided output into a wider buffer
    dY = rnd(g, 200, N)
    Wt, Kp = ops.pack_weight(dev(W, torch.float32), K, N, transposed=True)
    out = torch.zeros(200, K + 8, device="cuda")
    call("clsr_pgemm", dev(dY, torch.float32), N, 0, 0, None, 0, None, None, 0, Wt, Kp, None, None, 0,
         None, 0, out[:, 4:], K + 8, 0, None, 200, N, K)
    close(out[:, 4:4 + K], dY @ W.T, rtol=1e-5, atol=2e-5, name="dX")
    assert float(out[:, :4].abs().max()) == 0 and float(out[:, 4 + K:].abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(1000, 80, 80), (4097, 40, 120), (300, 164, 80), (64, 100, 64), (999, 80, 40),
                                   (2304, 256, 256), (768, 256, 256)])
def test_pgemm_dw(M, K, N):
    g = torch.Generator().manual_seed(M)
    X, dY = rnd(g, M, K), rnd(g, M, N)
    ws = torch.empty(query("clsr_pgemm_dw_workspace_floats", M, K, N), device="cuda")
    dW = torch.zeros(K, N, device="cuda")
    db = torch.zeros(N, device="cuda")
    call("clsr_pgemm_dw", dev(X, torch.float32), K, 0, 0, None, 0, None, None, 0, dev(dY, torch.float32), N,
         M, K, N, 1.0, dW, N, db, 0, ws)
    close(dW, X.T @ dY, rtol=2e-5, atol=2e-4, name="dW")
    close(db, dY.sum(0), rtol=2e-5, atol=2e-4, name="db")
    call("clsr_pgemm_dw", dev(X, torch.float32), K, 0, 0, None, 0, None, None, 0, dev(dY, torch.float32), N,
         M, K, N, 0.5, dW, N, db, 1, ws)
    close(dW, 1.5 * (X.T @ dY), rtol=2e-5, atol=3e-4, name="dW accumulate")


@pytest.mark.parametrize("entry,rel", [("clsr_pgemm_dw_wide", 3e-6), ("clsr_pgemm_dw_wide_x3", 1e-4), ("clsr_pgemm_dw_wide_x6", 4e-6)])
@pytest.mark.parametrize("M,K,N,bias", [(40000, 128, 1536, True), (32768 + 5, 256, 384, False), (33000, 100, 96, True),
                                         (70000, 128, 128, True)])
def test_pgemm_dw_wide(M, K, N, bias, entry, rel):
    """Weight gradients of wide layers (csrc/dwwide.hip: 128 x 128 output tiles over position ranges, partial tiles summed in
    range order): against X^T dY in float64, strided operands, ragged position ranges / tiles, accumulation, and bit-identical
    results of two runs."""
    g = torch.Generator().manual_seed(M + K + N)
    X, dY = rnd(g, M, K + 4), rnd(g, M, N + 8, scale=0.1)
    dX, ddY = dev(X, torch.float32), dev(dY, torch.float32)
    assert query("clsr_pgemm_dw_wide_supported", M, K, N) == 1
    ws = torch.empty(query("clsr_pgemm_dw_wide_workspace_floats", M, K, N), device="cuda")
    exp = X[:, :K].double().T @ dY[:, :N].double()
    expb = dY[:, :N].double().sum(0)
    outs = []
    for _ in range(2):
        dW = torch.full((K, N + 3), 7.0, device="cuda")
        db = torch.full((N,), 7.0, device="cuda") if bias else None
        call(entry, dX, K + 4, None, 0, ddY, N + 8, M, K, N, ws, dW, N + 3, db, 0)
        outs.append((dW.clone(), None if db is None else db.clone()))
    tol = rel * float(exp.abs().max()) + 1e-5      # (x3: 2^-16 relative per product term)
    close(outs[0][0][:, :N], exp, rtol=2e-5, atol=tol, name="dW")
    assert float((outs[0][0][:, N:] - 7.0).abs().max()) == 0.0, "columns past N must stay untouched"
    if bias:
        close(outs[0][1], expb, rtol=2e-5, atol=3e-6 * float(expb.abs().max()) + 1e-5, name="db")
        assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0]), "two runs must agree bit for bit"
    dW2 = outs[0][0].clone()
    call(entry, dX, K + 4, None, 0, ddY, N + 8, M, K, N, ws, dW2, N + 3, None, 1)
    close(dW2[:, :N], 2 * exp, rtol=2e-5, atol=2 * tol, name="accumulate")
    assert query("clsr_pgemm_dw_wide_supported", 1000, K, N) == 0 and query("clsr_pgemm_dw_wide_supported", M, 80, N) == 0
    # element-wise multiplier on X (the candidate kernel of a GRU: (r * h)^T d(candidate))
    Xm = rnd(g, M, K + 12)
    dW3 = torch.zeros(K, N, device="cuda")
    call(entry, dX, K + 4, dev(Xm, torch.float32), K + 12, ddY, N + 8, M, K, N, ws, dW3, N, None, 0)
    exp3 = (X[:, :K].double() * Xm[:, :K].double()).T @ dY[:, :N].double()
    close(dW3, exp3, rtol=2e-5, atol=rel * float(exp3.abs().max()) + 1e-5, name="dW with multiplier")


def test_pgemm_dw_prologues():
    g = torch.Generator().manual_seed(8)
    Hn, G, T, K, N = 11, 5, 10, 80, 80
    R = Hn * G
    a, q, dz = rnd(g, Hn * T, K), rnd(g, R, K), rnd(g, R * T, N)
    ws = torch.empty(query("clsr_pgemm_dw_workspace_floats", R * T, K, N), device="cuda")
    dW = torch.zeros(K, N, device="cuda")
    call("clsr_pgemm_dw", dev(a, torch.float32), K, T, G, dev(q, torch.float32), K, None, None, 0,
         dev(dz, torch.float32), N, R * T, K, N, 1.0, dW, N, None, 0, ws)
    rows = torch.arange(R * T)
    r, t = rows // T, rows % T
    xrow = (r // G) * T + t
    close(dW, (a[xrow] * q[r]).T @ dz, rtol=2e-5, atol=3e-4, name="dW0p")
    sc, sh = rnd(g, K), rnd(g, K)
    X = rnd(g, 500, K)
    dz = rnd(g, 500, 40)
    dW = torch.zeros(K, 40, device="cuda")
    call("clsr_pgemm_dw", dev(X, torch.float32), K, 0, 0, None, 0, dev(sc, torch.float32),
         dev(sh, torch.float32), 1, dev(dz, torch.float32), 40, 500, K, 40, 1.0, dW, 40, None, 0, ws)
    close(dW, torch.relu(X * sc + sh).T @ dz, rtol=2e-5, atol=3e-4, name="dW1")


# ------------------------------------------------------------------------------- batch norm
@pytest.mark.parametrize("M,C", [(1200, 80), (777, 40), (300, 100), (64, 64), (2304, 256)])
def test_bn_forward_backward(M, C):
    g = torch.Generator().manual_seed(C)
    z = (rnd(g, M, C) * 1.5 + 0.3).float().double().requires_grad_(True)
    gamma, beta = (rnd(g, C) * 0.5 + 1).requires_grad_(True), rnd(g, C).requires_grad_(True)
    W = rnd(g, C, 1)
    mean = z.mean(0)
    var = ((z - mean) ** 2).mean(0)
    y = (z - mean) / torch.sqrt(var + 1e-4) * gamma + beta
    h = torch.relu(y)
    up = rnd(g, M, C)
    (h * up).sum().backward()
    # kernel path: stats come from pgemm with identity weights
    eye = torch.eye(C, dtype=torch.float64)
    zz, st = _pgemm(z.detach(), eye, None, stats=True)
    mm, mv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    sc, sh, mu, istd = (torch.empty(C, device="cuda") for _ in range(4))
    call("clsr_bn_finalize", st, st.shape[0], C, float(M), dev(gamma.detach(), torch.float32),
         dev(beta.detach(), torch.float32), mm, mv, 0.95, 1e-4, 1, sc, sh, mu, istd)
    close(mu, mean, name="mean")
    close(istd, 1 / torch.sqrt(var + 1e-4), name="invstd")
    close(mm, 0.05 * mean, name="moving_mean")
    close(mv, 0.95 + 0.05 * var, name="moving_var")
    close(zz * sc + sh, y, rtol=1e-5, atol=1e-5, name="bn out")
    dh = dev(up, torch.float32)
    part = torch.zeros(query("clsr_colred_parts", M, C), 2, C, dtype=torch.float64, device="cuda")
    call("clsr_bn_relu_bwd_reduce", dh, zz, sc, sh, mu, istd, M, C, part)
    coef = torch.empty(3, C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    call("clsr_bn_bwd_coef", part, part.shape[0], C, float(M), dev(gamma.detach(), torch.float32), mu, istd,
         coef, dg, db, 0)
    call("clsr_bn_bwd_apply", dh, zz, coef, M, C)
    close(dh, z.grad, rtol=1e-4, atol=1e-5, name="dz")
    close(dg, gamma.grad, rtol=1e-4, atol=1e-4, name="dgamma")
    close(db, beta.grad, rtol=1e-4, atol=1e-4, name="dbeta")
    # eval mode uses the moving statistics
    call("clsr_bn_finalize", None, 0, C, 0.0, dev(gamma.detach(), torch.float32),
         dev(beta.detach(), torch.float32), mm, mv, 0.95, 1e-4, 0, sc, sh, None, None)
    close(sc, gamma.detach() / torch.sqrt(mv.double().cpu() + 1e-4), name="eval scale")


# ------------------------------------------------------------------------------- attention tail
@pytest.mark.parametrize("Hn,G,T,C1,Dk", [(19, 5, 10, 40, 40), (7, 1, 50, 40, 40), (3, 2, 130, 40, 40),
                                             (5, 10, 10, 40, 40), (2, 21, 7, 12, 24), (3, 3, 256, 256, 40),
                                             (9, 1, 256, 256, 128)])
def test_att_out_fwd_bwd(Hn, G, T, C1, Dk):
    g = torch.Generator().manual_seed(T)
    R = Hn * G
    z1 = rnd(g, R * T, C1).float().double().requires_grad_(True)
    keys = rnd(g, Hn, T, Dk).float().double().requires_grad_(True)
    sc, sh = rnd(g, C1).abs() * 0.5 + 0.5, rnd(g, C1) * 0.3
    mu, istd = rnd(g, C1) * 0.1, rnd(g, C1).abs() + 0.5
    w_out = rnd(g, C1).requires_grad_(True)
    b_out = rnd(g, 1).requires_grad_(True)
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0] = T
    lens_rows = lens.repeat_interleave(G)
    h1 = torch.relu(z1 * sc + sh)
    score = (h1 @ w_out + b_out).view(R, T)
    mask = torch.arange(T)[None, :] < lens_rows[:, None]
    score = torch.where(mask, score, torch.full_like(score, float(-(2 ** 32) + 1)))
    w = torch.softmax(score, -1)
    kr = keys.repeat_interleave(G, 0)
    out = (kr * w[..., None]).sum(1)
    up = rnd(g, R, Dk)
    (out * up).sum().backward()

    d_z1, d_keys = dev(z1.detach(), torch.float32), dev(keys.detach(), torch.float32)
    d_sc, d_sh, d_mu, d_is = (dev(x, torch.float32) for x in (sc, sh, mu, istd))
    d_w, d_b = dev(w_out.detach(), torch.float32), dev(b_out.detach(), torch.float32)
    d_len = dev(lens_rows, torch.int32)
    wts = torch.empty(R, T, device="cuda")
    o = torch.empty(R, Dk, device="cuda")
    call("clsr_att_out_fwd", d_z1, d_sc, d_sh, d_w, d_b, d_len, G, d_keys, Hn, G, T, C1, Dk, wts, o)
    close(wts, w, rtol=1e-4, atol=1e-6, name="weights")
    close(o, out, rtol=1e-4, atol=1e-5, name="out")
    # backward: score/softmax backward, then the two streaming passes (dy1 never materialised)
    nparts = query("clsr_att_score_bwd_parts", Hn)
    ds = torch.empty(R * T, device="cuda")
    dkeys = torch.zeros(Hn, T, Dk, device="cuda")
    bp = torch.zeros(nparts, device="cuda")
    call("clsr_att_score_bwd", dev(up, torch.float32), wts, d_len, G, d_keys, Hn, G, T, Dk, ds, dkeys, bp)
    close(dkeys, keys.grad, rtol=1e-4, atol=1e-5, name="dkeys")
    close(bp.sum().reshape(1), b_out.grad, rtol=1e-4, atol=1e-4, name="db_out")
    dy_exp = z1.grad / sc  # gradient wrt the BN output y1 (relu mask applied); z1 -> y1 has slope sc
    np2 = query("clsr_att_dy1_parts", R * T, C1)
    bnp = torch.zeros(np2, 2, C1, dtype=torch.float64, device="cuda")
    wp = torch.zeros(np2, C1, device="cuda")
    call("clsr_att_dy1_stats", d_z1, ds, d_sc, d_sh, d_mu, d_is, d_w, R * T, C1, bnp, wp)
    tot = bnp.sum(0).cpu()
    xhat = ((z1 - mu) * istd).detach()
    close(tot[0], dy_exp.sum(0), rtol=1e-4, atol=1e-4, name="sum dy")
    close(tot[1], (dy_exp * xhat).sum(0), rtol=1e-4, atol=1e-4, name="sum dy xhat")
    close(wp.sum(0), w_out.grad, rtol=1e-4, atol=1e-4, name="dw_out")
    coef = rnd(g, 3, C1)
    dz1 = torch.empty(R * T, C1, device="cuda")
    call("clsr_att_dy1_apply", d_z1, ds, d_sc, d_sh, d_w, dev(coef, torch.float32), R * T, C1, dz1)
    close(dz1, coef[0] * dy_exp + coef[1] * z1.detach() + coef[2], rtol=1e-4, atol=1e-5, name="dz1")


# ------------------------------------------------------------------------------- recurrent cells
def _oracle():
    from oracle import clsr_oracle as O
    return O


@pytest.mark.parametrize("Hn,G,T,Q,col0,ld", [(19, 5, 10, 80, 0, 80), (7, 1, 50, 40, 0, 40), (11, 5, 13, 40, 40, 80),
                                               (5, 9, 6, 128, 128, 256)])
def test_att_prod_bwd_on_column_blocks(Hn, G, T, Q, col0, ld):
    """Backward of the product feature a[h,t] * q[r]: da[h,t] = sum_g daq[r,t] * q[r], dq[r] = sum_t daq[r,t] * a[h,t]
    (clsr.py:368-370) -- the plain entry point and the leading-dimension one on a column block [col0, col0+Q) of
    wider a / q / da / dq tensors, overwriting and accumulating into dq."""
    g = torch.Generator().manual_seed(Hn * 100 + Q)
    R = Hn * G
    a, q = rnd(g, Hn * T, ld), rnd(g, R, ld)
    daq = rnd(g, R * T, Q)
    ab, qb = a[:, col0:col0 + Q], q[:, col0:col0 + Q]
    d4 = daq.view(Hn, G, T, Q)
    da_exp = (d4 * qb.view(Hn, G, 1, Q)).sum(1).reshape(Hn * T, Q)
    dq_exp = (d4 * ab.view(Hn, 1, T, Q)).sum(2).reshape(R, Q)
    A, Qd, D = dev(a, torch.float32), dev(q, torch.float32), dev(daq, torch.float32)
    for acc in (0, 1):
        da = torch.full((Hn * T, ld), 7.0, device="cuda")
        dq = torch.full((R, ld), 3.0, device="cuda")
        call("clsr_att_prod_bwd_ld", D, Q, A[:, col0:], ld, Qd[:, col0:], ld, Hn, G, T, Q, da[:, col0:], ld,
             dq[:, col0:], ld, acc)
        close(da[:, col0:col0 + Q], da_exp, name="da")
        close(dq[:, col0:col0 + Q], dq_exp + (3.0 if acc else 0.0), name="dq acc=%d" % acc)
        if ld > Q:   # the other columns are untouched
            other = [c for c in range(ld) if not col0 <= c < col0 + Q]
            assert float((da[:, other] - 7.0).abs().max()) == 0 and float((dq[:, other] - 3.0).abs().max()) == 0
    if ld == Q:
        da, dq = torch.empty(Hn * T, Q, device="cuda"), torch.empty(R, Q, device="cuda")
        call("clsr_att_prod_bwd", D, A, Qd, Hn, G, T, Q, da, dq)
        close(da, da_exp, name="da plain")
        close(dq, dq_exp, name="dq plain")


# The recurrences come in two forms (clsr_gru_desc.products, csrc/rnn.hip): "fp32" = fp32-input MFMAs, bit-exact fp32
# products -- the tolerances below; "x3" = split-bf16 products (16 significand bits per operand, errors compound over the
# steps; relative to the float64 oracle on weights of scale 0.3) -- twenty times those tolerances.  The plain entry points (clsr_gru_fwd ...) always run the fp32 form.
FORMS = [("fp32", 1.0), ("x3", 20.0)]


def _scaled_close(tf, exact=()):
    """``close`` with both tolerances multiplied by the form's factor (tensors named in ``exact`` keep theirs)"""
    def scaled(got, exp, rtol=2e-5, atol=2e-6, name=""):
        f = 1.0 if name in exact else tf
        close(got, exp, rtol=rtol * f, atol=atol * f, name=name)
    return scaled


@pytest.mark.parametrize("form,tf", FORMS)
@pytest.mark.parametrize("Hn,T,n,use_h0,seq_out", [(37, 10, 40, True, False), (16, 50, 40, False, True),
                                                   (5, 7, 40, True, True), (21, 9, 128, True, True),
                                                   (4, 5, 64, False, False)])
def test_gru_fwd_bwd(Hn, T, n, use_h0, seq_out, form, tf):
    O = _oracle()
    close = _scaled_close(tf)
    g = torch.Generator().manual_seed(T + Hn)
    D = 40
    x = rnd(g, Hn, T, D).float().double().requires_grad_(True)
    Wg = (rnd(g, D + n, 2 * n) * 0.3).requires_grad_(True)
    bg = (rnd(g, 2 * n) * 0.1 + 1).requires_grad_(True)
    Wc = (rnd(g, D + n, n) * 0.3).requires_grad_(True)
    bc = (rnd(g, n) * 0.1).requires_grad_(True)
    h0 = (rnd(g, Hn, n) * 0.5).requires_grad_(True) if use_h0 else None
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0] = T
    params = {"gates/kernel": Wg, "gates/bias": bg, "candidate/kernel": Wc, "candidate/bias": bc}
    h_init = h0 if use_h0 else torch.zeros(Hn, n, dtype=torch.float64)
    outs, hT = O.dynamic_gru(x, lens, h_init, "", params)
    up_T, up_seq = rnd(g, Hn, n), rnd(g, Hn, T, n)
    loss = (hT * up_T).sum() + ((outs * up_seq).sum() if seq_out else 0)
    loss.backward()
    # kernel: Pin = x . [Wg_x | Wc_x] + [bg | bc]
    Win = torch.cat([Wg[:D], Wc[:D]], 1).detach()
    Pin = (x.detach().reshape(-1, D) @ Win + torch.cat([bg, bc]).detach()).float()
    dPin_exp = None
    f32 = torch.float32
    d_Wg, d_Wc = dev(Wg.detach(), f32), dev(Wc.detach(), f32)
    d_len = dev(lens, torch.int32)
    hT_k = torch.empty(Hn, n, device="cuda")
    out_k = torch.full((Hn, T, n), 7.0, device="cuda") if seq_out else None
    hprev = torch.zeros(Hn, T, n, device="cuda")
    gates = torch.zeros(Hn, T, 3 * n, device="cuda")
    h0d = None if h0 is None else dev(h0.detach(), f32)
    if form == "fp32" and Hn % 2:      # (the plain entry point: always the fp32 form)
        call("clsr_gru_fwd", dev(Pin), 3 * n, d_Wg[D:], 2 * n, d_Wc[D:], n, h0d, n, d_len, 1, Hn, T, n, hT_k, out_k, hprev, gates)
    else:
        ops.rnn_multi("clsr_rnn_fwd_multi", [ops.gru_desc(n, Pin=dev(Pin), ldp=3 * n, Wgh=d_Wg[D:], ldg=2 * n, Wch=d_Wc[D:],
                                                          ldc=n, h0=h0d, h0_stride=n, hT=hT_k, out_seq=out_k, hprev=hprev,
                                                          gates=gates, products=form)], None, d_len, 1, Hn, T)
    close(hT_k, hT, rtol=1e-4, atol=2e-5, name="hT")
    if seq_out:
        close(out_k, outs, rtol=1e-4, atol=2e-5, name="out_seq")
    dPin = torch.full((Hn, T, 3 * n), 5.0, device="cuda")
    dh0 = torch.empty(Hn, n, device="cuda")
    if form == "fp32" and Hn % 2:
        call("clsr_gru_bwd", gates, hprev, d_Wg[D:], 2 * n, d_Wc[D:], n, d_len, 1, Hn, T, n, dev(up_T, f32),
             dev(up_seq, f32) if seq_out else None, dPin, dh0)
    else:
        ops.rnn_multi("clsr_rnn_bwd_multi", [ops.gru_desc(n, Wgh=d_Wg[D:], ldg=2 * n, Wch=d_Wc[D:], ldc=n, hprev=hprev,
                                                          gates=gates, dhT=dev(up_T, f32),
                                                          dout_seq=dev(up_seq, f32) if seq_out else None, dPin=dPin,
                                                          lddp=3 * n, dh0=dh0, products=form)], None, d_len, 1, Hn, T)
    # check through the implied parameter / input gradients
    dP = dPin.double().cpu().reshape(-1, 3 * n)
    close(dP @ Win.T, x.grad.reshape(-1, D), rtol=2e-4, atol=2e-5, name="dx")
    xf = x.detach().reshape(-1, D)
    close(xf.T @ dP[:, :2 * n], Wg.grad[:D], rtol=2e-4, atol=1e-4, name="dWg_x")
    close(dP[:, :2 * n].sum(0), bg.grad, rtol=2e-4, atol=1e-4, name="dbg")
    close(dP[:, 2 * n:].sum(0), bc.grad, rtol=2e-4, atol=1e-4, name="dbc")
    hp = hprev.double().cpu().reshape(-1, n)
    rh = (gates[..., :n] * hprev).double().cpu().reshape(-1, n)
    close(hp.T @ dP[:, :2 * n], Wg.grad[D:], rtol=2e-4, atol=1e-4, name="dWg_h")
    close(rh.T @ dP[:, 2 * n:], Wc.grad[D:], rtol=2e-4, atol=1e-4, name="dWc_h")
    if use_h0:
        close(dh0, h0.grad, rtol=2e-4, atol=2e-5, name="dh0")


@pytest.mark.parametrize("form,tf", FORMS)
@pytest.mark.parametrize("Hn,T,n", [(37, 10, 40), (16, 50, 40), (19, 8, 128)])
def test_t4lstm_fwd_bwd(Hn, T, n, form, tf):
    O = _oracle()
    close = _scaled_close(tf, exact=("TT",))
    g = torch.Generator().manual_seed(T)
    D = 40
    x = rnd(g, Hn, T, D).float().double().requires_grad_(True)
    names = {"_time_input_w1": (n,), "_time_input_bias1": (n,), "_time_input_w2": (n,), "_time_input_bias2": (n,),
             "_time_kernel_w1": (D, n), "_time_kernel_t1": (n, n), "_time_bias1": (n,),
             "_time_kernel_w2": (D, n), "_time_kernel_t2": (n, n), "_time_bias2": (n,),
             "_o_kernel_t1": (n, n), "_o_kernel_t2": (n, n), "kernel": (D + n, 4 * n), "bias": (4 * n,)}
    P = {k: (rnd(g, *s) * 0.3).requires_grad_(True) for k, s in names.items()}
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0] = T
    tfirst, tnow = rnd(g, Hn, T).float().double(), rnd(g, Hn, T).float().double()
    out = O.time4lstm(x, tfirst, tnow, lens, "", P, n)
    up = rnd(g, Hn, T, n)
    (out * up).sum().backward()
    f32 = torch.float32
    d = {k: dev(v.detach(), f32) for k, v in P.items()}
    TT = torch.empty(Hn * T, 2 * n, device="cuda")
    call("clsr_t4_time_inputs_fwd", dev(tnow, f32), dev(tfirst, f32), T, d["_time_input_w1"], d["_time_input_bias1"],
         d["_time_input_w2"], d["_time_input_bias2"], Hn, T, n, TT)
    tn = torch.tanh(tnow[..., None] * P["_time_input_w1"] + P["_time_input_bias1"]).detach()
    tl = torch.tanh(tfirst[..., None] * P["_time_input_w2"] + P["_time_input_bias2"]).detach()
    close(TT, torch.cat([tn, tl], -1).reshape(-1, 2 * n), rtol=1e-5, atol=1e-6, name="TT")
    xd = x.detach()
    Pd = {k: v.detach() for k, v in P.items()}
    z = xd @ Pd["kernel"][:D] + Pd["bias"]
    z[..., 3 * n:] += tn @ Pd["_o_kernel_t1"] + tl @ Pd["_o_kernel_t2"]
    tns = xd @ Pd["_time_kernel_w1"] + tn @ Pd["_time_kernel_t1"] + Pd["_time_bias1"]
    tls = xd @ Pd["_time_kernel_w2"] + tl @ Pd["_time_kernel_t2"] + Pd["_time_bias2"]
    Pin = torch.cat([z, tns, tls], -1).float()
    d_len = dev(lens, torch.int32)
    out_k = torch.full((Hn, T, n), 3.0, device="cuda")
    act = torch.zeros(Hn, T, 6 * n, device="cuda")
    cst = torch.zeros(Hn, T, n, device="cuda")
    mprev = torch.zeros(Hn, T, n, device="cuda")
    tiled = form == "x3" and Hn % 2 == 1        # (x3 only: the private tile-major image of the saved activations)
    if tiled:
        act, cst = torch.zeros(query("clsr_t4_act_tiled_floats", Hn, T, n), device="cuda"), None
    if form == "fp32" and Hn % 2:               # (the plain entry points: always the fp32 form)
        call("clsr_t4lstm_fwd", dev(Pin), 6 * n, d["kernel"][D:], 4 * n, d_len, 1, Hn, T, n, out_k, act, cst, mprev)
    else:
        ops.rnn_multi("clsr_rnn_fwd_multi", [], ops.t4_desc(n, Pin=dev(Pin), ldp=6 * n, Wm=d["kernel"][D:], ldm=4 * n,
                                                            out_seq=out_k, act=act, cst=cst, mprev=mprev, products=form,
                                                            act_tiled=tiled), d_len, 1, Hn, T)
    close(out_k, out, rtol=1e-4, atol=2e-5, name="rnn_out")
    dPin = torch.full((Hn, T, 6 * n), 9.0, device="cuda")
    if form == "fp32" and Hn % 2:
        call("clsr_t4lstm_bwd", act, cst, d["kernel"][D:], 4 * n, d_len, 1, Hn, T, n, dev(up, f32), dPin)
    else:
        ops.rnn_multi("clsr_rnn_bwd_multi", [], ops.t4_desc(n, Wm=d["kernel"][D:], ldm=4 * n, act=act, cst=cst,
                                                            dout_seq=dev(up, f32), dPin=dPin, lddp=6 * n, products=form,
                                                            act_tiled=tiled), d_len, 1, Hn, T)
    dP = dPin.double().cpu().reshape(-1, 6 * n)
    xf = xd.reshape(-1, D)
    close(xf.T @ dP[:, :4 * n], P["kernel"].grad[:D], rtol=2e-4, atol=1e-4, name="dkernel_x")
    close(mprev.double().cpu().reshape(-1, n).T @ dP[:, :4 * n], P["kernel"].grad[D:], rtol=2e-4, atol=1e-4,
          name="dkernel_m")
    close(dP[:, :4 * n].sum(0), P["bias"].grad, rtol=2e-4, atol=1e-4, name="dbias")
    close(xf.T @ dP[:, 4 * n:5 * n], P["_time_kernel_w1"].grad, rtol=2e-4, atol=1e-4, name="d_time_kernel_w1")
    close(xf.T @ dP[:, 5 * n:], P["_time_kernel_w2"].grad, rtol=2e-4, atol=1e-4, name="d_time_kernel_w2")
    tnf, tlf = tn.reshape(-1, n), tl.reshape(-1, n)
    close(tnf.T @ dP[:, 3 * n:4 * n], P["_o_kernel_t1"].grad, rtol=2e-4, atol=1e-4, name="d_o_kernel_t1")
    close(tlf.T @ dP[:, 5 * n:], P["_time_kernel_t2"].grad, rtol=2e-4, atol=1e-4, name="d_time_kernel_t2")
    dx = dP[:, :4 * n] @ Pd["kernel"][:D].T + dP[:, 4 * n:5 * n] @ Pd["_time_kernel_w1"].T \
        + dP[:, 5 * n:] @ Pd["_time_kernel_w2"].T
    close(dx, x.grad.reshape(-1, D), rtol=2e-4, atol=2e-5, name="dx")
    # time-input parameter gradients
    dTT = torch.cat([dP[:, 3 * n:4 * n] @ Pd["_o_kernel_t1"].T + dP[:, 4 * n:5 * n] @ Pd["_time_kernel_t1"].T,
                     dP[:, 3 * n:4 * n] @ Pd["_o_kernel_t2"].T + dP[:, 5 * n:] @ Pd["_time_kernel_t2"].T], 1)
    nparts = query("clsr_t4_time_inputs_bwd_parts", Hn, T, n)
    part = torch.zeros(nparts, 2, 2 * n, device="cuda")
    call("clsr_t4_time_inputs_bwd", dev(dTT, f32), TT, dev(tnow, f32), dev(tfirst, f32), T, Hn, T, n, part)
    tot = part.sum(0).double().cpu()
    close(tot[0, :n], P["_time_input_w1"].grad, rtol=2e-4, atol=1e-4, name="d_time_input_w1")
    close(tot[0, n:], P["_time_input_w2"].grad, rtol=2e-4, atol=1e-4, name="d_time_input_w2")
    close(tot[1, :n], P["_time_input_bias1"].grad, rtol=2e-4, atol=1e-4, name="d_time_input_bias1")
    close(tot[1, n:], P["_time_input_bias2"].grad, rtol=2e-4, atol=1e-4, name="d_time_input_bias2")


# ------------------------------------------------------------------------------- heads
def test_alpha_concat_fuse_roundtrip():
    g = torch.Generator().manual_seed(3)
    Hn, G, D, n, T = 21, 5, 40, 40, 10
    B = Hn * G
    fs = rnd(g, Hn, n).requires_grad_(True)
    target, S = rnd(g, B, D).requires_grad_(True), rnd(g, B, D).requires_grad_(True)
    L = rnd(g, Hn, D).requires_grad_(True)
    tnow = rnd(g, B, T)
    f32 = torch.float32
    ld = 164
    out = torch.full((B, ld), 5.0, device="cuda")
    call("clsr_alpha_concat", dev(fs.detach(), f32), n, dev(target.detach(), f32), dev(L.detach(), f32),
         dev(S.detach(), f32), dev(tnow, f32), T, T - 1, 1, B, G, D, out, ld)
    exp = torch.cat([fs.repeat_interleave(G, 0), target, L.repeat_interleave(G, 0), S, tnow[:, -1:]], 1)
    close(out[:, :161], exp, name="concat")
    assert float(out[:, 161:].abs().max()) == 0
    up = rnd(g, B, ld)
    (exp * up[:, :161]).sum().backward()
    dfs, dL = torch.zeros(Hn, n, device="cuda"), torch.zeros(Hn, D, device="cuda")
    dt, dS = torch.zeros(B, D, device="cuda"), torch.zeros(B, D, device="cuda")
    call("clsr_alpha_concat_bwd", dev(up, f32), ld, n, Hn, G, D, dfs, dt, dL, dS)
    close(dfs, fs.grad, rtol=1e-5, atol=1e-5, name="dfs")
    close(dL, L.grad, rtol=1e-5, atol=1e-5, name="dL")
    close(dt, target.grad, name="dtarget")
    close(dS, S.grad, name="dS")
    # fusion
    for t in (fs, target, S, L):
        t.grad = None
    al = rnd(g, B).requires_grad_(True)
    alpha = torch.sigmoid(al)
    mo = torch.cat([alpha[:, None] * L.repeat_interleave(G, 0) + (1 - alpha[:, None]) * S, target], 1)
    up = rnd(g, B, 2 * D)
    (mo * up).sum().backward()
    a_k, mo_k = torch.empty(B, device="cuda"), torch.empty(B, 2 * D, device="cuda")
    call("clsr_alpha_fuse_fwd", dev(al.detach(), f32), 0.0, dev(L.detach(), f32), dev(S.detach(), f32),
         dev(target.detach(), f32), B, G, D, a_k, mo_k)
    close(a_k, alpha, name="alpha")
    close(mo_k, mo, name="model_output")
    dal = torch.empty(B, device="cuda")
    dL, dS, dt = torch.zeros(Hn, D, device="cuda"), torch.zeros(B, D, device="cuda"), torch.zeros(B, D, device="cuda")
    call("clsr_alpha_fuse_bwd", dev(up, f32), a_k, 0.0, dev(L.detach(), f32), dev(S.detach(), f32), Hn, G, D,
         dal, dL, dS, dt)
    close(dal, al.grad, rtol=1e-4, atol=1e-5, name="dalpha_logit")
    close(dL, L.grad, rtol=1e-4, atol=1e-5, name="dL fuse")
    close(dS, S.grad, rtol=1e-4, atol=1e-5, name="dS fuse")
    close(dt, target.grad, name="dtarget fuse")


@pytest.mark.parametrize("B,C1", [(1000, 40), (333, 64)])
def test_mlp_out_fwd_bwd(B, C1):
    g = torch.Generator().manual_seed(C1)
    z1 = rnd(g, B, C1).float().double().requires_grad_(True)
    sc, sh, mu, istd = rnd(g, C1).abs() * 0.5 + 0.5, rnd(g, C1) * 0.3, rnd(g, C1) * 0.1, rnd(g, C1).abs() + 0.5
    w, b = rnd(g, C1).requires_grad_(True), rnd(g, 1).requires_grad_(True)
    logit = torch.relu(z1 * sc + sh) @ w + b
    up = rnd(g, B)
    (logit * up).sum().backward()
    f32 = torch.float32
    dz, dsc, dsh, dmu, dis = (dev(x, f32) for x in (z1.detach(), sc, sh, mu, istd))
    lg = torch.empty(B, device="cuda")
    call("clsr_mlp_out_fwd", dz, dsc, dsh, dev(w.detach(), f32), dev(b.detach(), f32), B, C1, lg)
    close(lg, logit, rtol=1e-5, atol=1e-5, name="logit")
    nparts = query("clsr_mlp_out_bwd_parts", B, C1)
    dy1 = torch.empty(B, C1, device="cuda")
    bnp = torch.zeros(nparts, 2, C1, dtype=torch.float64, device="cuda")
    wp = torch.zeros(nparts, C1 + 4, device="cuda")
    call("clsr_mlp_out_bwd", dev(up, f32), dz, dsc, dsh, dmu, dis, dev(w.detach(), f32), B, C1, dy1, bnp, wp)
    dy_exp = z1.grad / sc
    close(dy1, dy_exp, rtol=1e-5, atol=1e-6, name="dy1")
    tot = bnp.sum(0).cpu()
    close(tot[0], dy_exp.sum(0), rtol=1e-4, atol=1e-4, name="sum dy")
    close(tot[1], (dy_exp * ((z1 - mu) * istd).detach()).sum(0), rtol=1e-4, atol=1e-4, name="sum dy xhat")
    ws = wp.sum(0).cpu()
    close(ws[:C1], w.grad, rtol=1e-4, atol=1e-4, name="dw")
    close(ws[C1:C1 + 1], b.grad, rtol=1e-4, atol=1e-4, name="db")


def test_softmax_loss():
    g = torch.Generator().manual_seed(4)
    P, G = 777, 5
    logit = (rnd(g, P * G) * 2).requires_grad_(True)
    labels = torch.zeros(P, G)
    labels[:, 0] = 1
    labels = labels.reshape(-1).double()
    sm = torch.softmax(logit.view(P, G), -1)
    pos = torch.where(labels.view(P, G) == 1, sm, torch.ones_like(sm))
    loss = -G * torch.log(pos).mean()
    loss.backward()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    dl = torch.empty(P * G, device="cuda")
    call("clsr_softmax_loss", dev(logit.detach(), torch.float32), dev(labels, torch.float32), P, G, 1.0 / P, out, dl)
    close(out, loss.detach().reshape(1), rtol=1e-5, name="data loss")
    close(dl, logit.grad, rtol=1e-4, atol=1e-7, name="dlogit")


@pytest.mark.parametrize("mode,D", [(1, 40), (0, 40), (1, 128), (0, 128), (0, 200)])
def test_contrastive(mode, D):
    g = torch.Generator().manual_seed(6 + mode)
    Hn, G, thr = 40, 5, 5
    B = Hn * G
    L, M, R = (rnd(g, Hn, D).requires_grad_(True) for _ in range(3))
    S = rnd(g, B, D).requires_grad_(True)
    lens = torch.randint(1, 11, (Hn,), generator=g)
    cm = (lens > thr).double().repeat_interleave(G)
    Lr, Mr, Rr = (t.repeat_interleave(G, 0) for t in (L, M, R))
    if mode == 0:
        sp = torch.nn.functional.softplus
        terms = [sp((Lr * (-Mr + Rr)).sum(-1)), sp((S * (-Rr + Mr)).sum(-1)), sp((Mr * (-Lr + S)).sum(-1)),
                 sp((Rr * (-S + Lr)).sum(-1))]
    else:
        dLM, dLR, dSM, dSR = (Lr - Mr) ** 2, (Lr - Rr) ** 2, (S - Mr) ** 2, (S - Rr) ** 2
        terms = [torch.relu(dLM - dLR + 1.0).sum(-1), torch.relu(dSR - dSM + 1.0).sum(-1),
                 torch.relu(dLM - dSM + 1.0).sum(-1), torch.relu(dSR - dLR + 1.0).sum(-1)]
    loss = 0.1 * sum((cm * t).sum() / cm.sum() for t in terms)
    loss.backward()
    f32 = torch.float32
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    dL, dM, dR = (torch.zeros(Hn, D, device="cuda") for _ in range(3))
    dS = torch.zeros(B, D, device="cuda")
    denom = dev(cm.sum().reshape(1), f32)
    call("clsr_contrastive", dev(L.detach(), f32), dev(S.detach(), f32), dev(M.detach(), f32), dev(R.detach(), f32),
         dev(lens.repeat_interleave(G), torch.int32), G, Hn, G, D, thr, mode, 1.0, 0.1, denom, out, dL, dS, dM, dR)
    close(out, loss.detach().reshape(1), rtol=1e-5, name="contrastive loss")
    close(dL, L.grad, rtol=1e-4, atol=1e-7, name="dL")
    close(dS, S.grad, rtol=1e-4, atol=1e-7, name="dS")
    close(dM, M.grad, rtol=1e-4, atol=1e-7, name="dM")
    close(dR, R.grad, rtol=1e-4, atol=1e-7, name="dR")


# ------------------------------------------------------------------------------- optimiser
def test_dense_reg_clip_adam():
    g = torch.Generator().manual_seed(9)
    sizes = [6400, 80, 1, 3200, 40]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(off[-1])
    p, gr = rnd(g, n), rnd(g, n) * 0.05
    gr[:6400] *= 40  # first tensor exceeds the clip norm
    m, v = rnd(g, n) * 0.01, rnd(g, n).abs() * 0.01
    seg_of = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    f32 = torch.float32
    dp, dg, dm, dv = dev(p, f32), dev(gr, f32), dev(m, f32), dev(v, f32)
    ss = torch.zeros(len(sizes), dtype=torch.float64, device="cuda")
    reg = torch.zeros(1, dtype=torch.float64, device="cuda")
    l2, l1, clip = 1e-3, 3e-4, 2.0
    call("clsr_dense_reg_norm", dp, dg, dev(off), len(sizes), l2, l1, ss, reg)
    g2 = gr + l2 * p + l1 * torch.sign(p)
    close(reg, (0.5 * l2 * (p ** 2).sum() + l1 * p.abs().sum()).reshape(1), rtol=1e-5, name="reg loss")
    exp_ss = torch.stack([(g2[off[i]:off[i + 1]] ** 2).sum() for i in range(len(sizes))])
    close(ss, exp_ss, rtol=1e-5, name="sumsq")
    st = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.0], dtype=torch.float64, device="cuda")
    call("clsr_adam_tick", st, 1e-3, 0.9, 0.999)
    call("clsr_adam_tick", st, 1e-3, 0.9, 0.999)
    lr_t = 1e-3 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert abs(float(st[3]) - lr_t) < 1e-12 and float(st[0]) == 2.0
    call("clsr_dense_adam", dp, dg, dm, dv, dev(seg_of), ss, clip, st, 0.9, 0.999, 1e-8, n)
    fac = torch.cat([torch.full((s,), clip / max(math.sqrt(float(exp_ss[i])), clip), dtype=torch.float64)
                     for i, s in enumerate(sizes)])
    gc = g2 * fac
    m2 = 0.9 * m + 0.1 * gc
    v2 = 0.999 * v + 0.001 * gc * gc
    close(dm, m2, rtol=1e-5, atol=1e-8, name="m")
    close(dv, v2, rtol=1e-5, atol=1e-10, name="v")
    close(dp, p - lr_t * m2 / (torch.sqrt(v2) + 1e-8), rtol=1e-5, atol=1e-6, name="param")
    assert float(dg.abs().max()) == 0.0


@pytest.mark.parametrize("lazy", [0, 1])
def test_table_reg_adam(lazy):
    g = torch.Generator().manual_seed(10 + lazy)
    V, C = 500, 40
    UL, US = rnd(g, V, C), rnd(g, V, C)
    flags = (torch.rand(V, generator=g) < 0.3)
    G_l = torch.where(flags[:, None], rnd(g, V, C) * 0.01, torch.zeros(V, C, dtype=torch.float64))
    m, v = rnd(g, V, C) * 0.01, rnd(g, V, C).abs() * 0.01
    l2, l1, wd, clip = 1e-3, 2e-4 * lazy, 0.01, 0.05
    f32 = torch.float32
    dUL, dUS, dG, dm, dv = dev(UL, f32), dev(US, f32), dev(G_l, f32), dev(m, f32), dev(v, f32)
    dfl = dev(flags.to(torch.uint8))
    cnt = torch.zeros(1, device="cuda")
    call("clsr_count_flags", dfl, V, cnt)
    ss = torch.tensor([float((G_l ** 2).sum()), 0.0], dtype=torch.float64, device="cuda")
    reg = torch.zeros(1, dtype=torch.float64, device="cuda")
    disc = torch.zeros(1, dtype=torch.float64, device="cuda")
    call("clsr_table_reg", dUL, dUS, dfl, V, C, l2, l1, -2 * wd, -wd, cnt, dG, ss[1:], reg, disc)
    nu = float(flags.sum())
    fm = flags[:, None].double()
    g_reg = fm * (l2 * UL + l1 * torch.sign(UL) + (-2 * wd / (nu * C)) * (UL - US))
    close(reg, (0.5 * l2 * (fm * UL ** 2).sum() + l1 * (fm * UL.abs()).sum()).reshape(1), rtol=1e-5, name="reg")
    close(disc, (-wd * (fm * (UL - US) ** 2).sum() / (nu * C)).reshape(1), rtol=1e-5, name="disc")
    close(ss[1:], (g_reg ** 2).sum().reshape(1), rtol=1e-5, name="reg sumsq")
    st = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.0], dtype=torch.float64, device="cuda")
    call("clsr_adam_tick", st, 1e-3, 0.9, 0.999)
    call("clsr_table_adam", dUL, dG, dm, dv, dfl, V, C, ss, 1, 2, clip, st, 0.9, 0.999, 1e-8, lazy)
    tot = float((G_l ** 2).sum() + (g_reg ** 2).sum())
    fac = clip / max(math.sqrt(tot), clip)
    gt = (G_l + g_reg) * fac
    m2 = 0.9 * m + 0.1 * gt
    v2 = 0.999 * v + 0.001 * gt * gt
    newp = UL - float(st[3]) * m2 / (torch.sqrt(v2) + 1e-8)
    if lazy:
        m2 = torch.where(flags[:, None], m2, m)
        v2 = torch.where(flags[:, None], v2, v)
        newp = torch.where(flags[:, None], newp, UL)
    close(dm, m2, rtol=1e-5, atol=1e-8, name="m")
    close(dv, v2, rtol=1e-5, atol=1e-10, name="v")
    close(dUL, newp, rtol=1e-5, atol=1e-6, name="table")
    assert float(dG.abs().max()) == 0.0 and int(dfl.sum()) == 0


@pytest.mark.parametrize("V,C,frac", [(1000, 8, 0.1), (70001, 32, 0.03), (5000, 40, 1.0), (64, 4, 0.0)])
def test_touched_row_compaction_pack_unpack(V, C, frac):
    """clsr_flags_compact / clsr_rows_pack / clsr_rows_unpack (touched-row exchange): bit exact."""
    g = torch.Generator().manual_seed(V)
    flags = (torch.rand(V, generator=g) < frac).to(torch.uint8)
    table = torch.randn(V, C, generator=g)
    exp_ids = torch.nonzero(flags).reshape(-1).to(torch.int32)
    n = int(exp_ids.numel())
    cap = max(n, 1) + 5
    d_flags, d_table = flags.cuda(), table.cuda()
    ids = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    count = torch.zeros(2, dtype=torch.int32, device="cuda")
    nws = ops.query("clsr_flags_compact_workspace_bytes", V)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    call("clsr_flags_compact", d_flags, V, ids, cap, count, ws, nws)
    assert count.tolist() == [n, 0]
    assert torch.equal(ids[:n].cpu(), exp_ids) and bool((ids[n:] == -7).all())
    rows = torch.zeros(cap, C, device="cuda")
    call("clsr_rows_pack", d_table, ids, count, cap, C, rows)
    assert torch.equal(rows[:n].cpu(), table[exp_ids.long()])
    # truncation is reported, never written past cap
    if n > 3:
        small = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        call("clsr_flags_compact", d_flags, V, small, n - 3, count, ws, nws)
        assert count.tolist() == [n - 3, 1]
        assert torch.equal(small[:n - 3].cpu(), exp_ids[:n - 3]) and bool((small[n - 3:] == -7).all())
        call("clsr_flags_compact", d_flags, V, ids, cap, count, ws, nws)
    # unpack: clear, then add twice -> 2 * rows on the touched rows, untouched rows unchanged
    dst = d_table.clone()
    f2 = torch.zeros(V, dtype=torch.uint8, device="cuda")
    call("clsr_rows_unpack", ids, None, count, cap, C, 0, dst, None)
    call("clsr_rows_unpack", ids, rows, count, cap, C, 1, dst, f2)
    call("clsr_rows_unpack", ids, rows, count, cap, C, 1, dst, f2)
    exp = table.clone()
    exp[exp_ids.long()] *= 2
    assert torch.equal(dst.cpu(), exp) and torch.equal(f2.cpu(), flags)


# ------------------------------------------------------------------------------- sibling-model kernels
@pytest.mark.parametrize("form,tf", FORMS)
@pytest.mark.parametrize("Hn,G,T,n", [(7, 5, 10, 40), (5, 1, 50, 40), (3, 4, 9, 128)])
def test_attentional_gru_fwd_bwd(Hn, G, T, n, form, tf):
    """clsr_rnn_*_multi with clsr_gru_desc.att (DIEN's VecAttGRUCell): one sequence per candidate row reading the
    input projections / length of its history (in_div = G), update gate scaled by 1 - att; backward incl. d att."""
    from oracle import sibling_oracle as S
    close = _scaled_close(tf)

    g = torch.Generator().manual_seed(T + Hn + n)
    B, Hin = Hn * G, 24
    x = rnd(g, Hn, T, Hin).requires_grad_(True)
    Wg = (rnd(g, Hin + n, 2 * n) * 0.3).requires_grad_(True)
    bg = (rnd(g, 2 * n) * 0.1 + 1).requires_grad_(True)
    Wc = (rnd(g, Hin + n, n) * 0.3).requires_grad_(True)
    bc = (rnd(g, n) * 0.1).requires_grad_(True)
    att = torch.rand(B, T, generator=g, dtype=torch.float64).requires_grad_(True)
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0] = T
    params = {"gates/kernel": Wg, "gates/bias": bg, "candidate/kernel": Wc, "candidate/bias": bc}
    xr = x.repeat_interleave(G, 0)
    hT = S.dynamic_augru(xr, att, lens.repeat_interleave(G), "", params, n)
    up = rnd(g, B, n)
    (hT * up).sum().backward()
    f32 = torch.float32
    Win = torch.cat([Wg[:Hin], Wc[:Hin]], 1).detach()
    Pin = dev((x.detach().reshape(-1, Hin) @ Win + torch.cat([bg, bc]).detach()).float())     # history level
    d_Wg, d_Wc, d_len, d_att = dev(Wg.detach(), f32), dev(Wc.detach(), f32), dev(lens, torch.int32), dev(att.detach(), f32)
    hT_k = torch.empty(B, n, device="cuda")
    hprev, gates = torch.zeros(B, T, n, device="cuda"), torch.zeros(B, T, 3 * n, device="cuda")
    d = ops.gru_desc(n, Pin=Pin, ldp=3 * n, Wgh=d_Wg[Hin:], ldg=2 * n, Wch=d_Wc[Hin:], ldc=n, hT=hT_k, hprev=hprev,
                     gates=gates, att=d_att, in_div=G, products=form)
    ops.rnn_multi("clsr_rnn_fwd_multi", [d], None, d_len, 1, B, T)
    close(hT_k, hT, rtol=1e-4, atol=2e-5, name="final state")
    dPin = torch.full((B, T, 3 * n), 5.0, device="cuda")
    datt = torch.zeros(B, T, device="cuda")
    db = ops.gru_desc(n, Wgh=d_Wg[Hin:], ldg=2 * n, Wch=d_Wc[Hin:], ldc=n, hprev=hprev, gates=gates, dhT=dev(up, f32),
                      dPin=dPin, lddp=3 * n, att=d_att, datt=datt, in_div=G, products=form)
    ops.rnn_multi("clsr_rnn_bwd_multi", [db], None, d_len, 1, B, T)
    close(datt, att.grad, rtol=2e-4, atol=2e-5, name="d att")
    dP = dPin.double().cpu().reshape(Hn, G, T, 3 * n).sum(1).reshape(-1, 3 * n)      # rows of a group share the inputs
    close(dP @ Win.T, x.grad.reshape(-1, Hin), rtol=2e-4, atol=2e-5, name="dx")
    close(x.detach().reshape(-1, Hin).T @ dP[:, :2 * n], Wg.grad[:Hin], rtol=2e-4, atol=1e-4, name="dWg_x")
    close(dP[:, 2 * n:].sum(0), bc.grad, rtol=2e-4, atol=1e-4, name="dbc")
    dPr = dPin.double().cpu().reshape(-1, 3 * n)
    hp = hprev.double().cpu().reshape(-1, n)
    rh = (gates[..., :n] * hprev).double().cpu().reshape(-1, n)
    close(hp.T @ dPr[:, :2 * n], Wg.grad[Hin:], rtol=2e-4, atol=1e-4, name="dWg_h")
    close(rh.T @ dPr[:, 2 * n:], Wc.grad[Hin:], rtol=2e-4, atol=1e-4, name="dWc_h")


@pytest.mark.parametrize("Hn,T,D", [(9, 10, 40), (4, 130, 40), (3, 7, 128)])
def test_asvd_attention_fwd_bwd(Hn, T, D):
    """A2SVD attention (base_model.py:595-625): unmasked softmax over T of (x.A).query, output sum_t w x."""
    g = torch.Generator().manual_seed(Hn * T)
    x = rnd(g, Hn, T, D).requires_grad_(True)
    A = (rnd(g, D, D) * 0.2).requires_grad_(True)
    q = (rnd(g, D) * 0.5).requires_grad_(True)
    ai = x @ A
    w = torch.softmax(ai @ q, -1)
    out = (x * w.unsqueeze(-1)).sum(1)
    up = rnd(g, Hn, D)
    ai.retain_grad()
    (out * up).sum().backward()
    f32 = torch.float32
    d_x, d_ai, d_q = dev(x.detach(), f32), dev(ai.detach(), f32), dev(q.detach(), f32)
    wk, ok = torch.empty(Hn, T, device="cuda"), torch.empty(Hn, D, device="cuda")
    call("clsr_asvd_att_fwd", d_ai, d_q, d_x, Hn, T, D, wk, ok)
    close(wk, w, rtol=1e-4, atol=1e-6, name="weights")
    close(ok, out, rtol=1e-4, atol=1e-5, name="output")
    parts = query("clsr_asvd_att_bwd_parts", Hn)
    dai = torch.empty(Hn * T, D, device="cuda")
    dx = torch.full((Hn * T, D), 2.0, device="cuda")                # accumulated into
    qp = torch.empty(parts, D, device="cuda")
    call("clsr_asvd_att_bwd", dev(up, f32), wk, d_ai, d_q, d_x, Hn, T, D, dai, dx, qp)
    close(dai.view(Hn, T, D), ai.grad, rtol=2e-4, atol=2e-6, name="d att_inputs")
    close(qp.sum(0), q.grad, rtol=2e-4, atol=2e-5, name="d query")
    direct = (w.detach().unsqueeze(-1) * up.unsqueeze(1))           # the x * w path; the x.A path goes through dai
    close(dx.view(Hn, T, D) - 2.0, direct, rtol=2e-4, atol=2e-5, name="d x (direct part)")


def test_weights_softmax_bwd_row_scaling_and_products():
    g = torch.Generator().manual_seed(3)
    Hn, G, T, D = 6, 5, 12, 40
    R = Hn * G
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0] = T
    score = rnd(g, R, T).requires_grad_(True)
    mask = (torch.arange(T)[None, :] < lens.repeat_interleave(G)[:, None])
    w = torch.softmax(torch.where(mask, score, torch.full_like(score, -4294967295.0)), -1)
    dw = rnd(g, R, T)
    (w * dw).sum().backward()
    f32 = torch.float32
    parts = query("clsr_softmax_weights_bwd_parts", R)
    ds, bp = torch.empty(R, T, device="cuda"), torch.empty(parts, device="cuda")
    call("clsr_softmax_weights_bwd", dev(dw, f32), dev(w.detach(), f32), dev(lens, torch.int32), 1, Hn, G, T, ds, bp)
    close(ds, score.grad, rtol=2e-4, atol=2e-6, name="d score")
    close(bp.sum(), score.grad.sum(), rtol=1e-3, atol=1e-5, name="d b_out partials")
    # hist_sum = hist_mean * len (0 for an empty history), accumulate flag
    m = rnd(g, Hn, D)
    lens0 = lens.clone()
    lens0[1] = 0
    out = torch.full((Hn, D), 1.0, device="cuda")
    call("clsr_scale_rows_by_len", dev(m, f32), dev(lens0, torch.int32), 1, Hn, D, out, 1)
    close(out, 1.0 + m * lens0[:, None], name="scale rows (accumulate)")
    # a[r] * b[r / G] into a column block of a wider output
    a, b = rnd(g, R, D), rnd(g, Hn, D)
    wide = torch.full((R, 3 * D), 9.0, device="cuda")
    call("clsr_mul_rows", dev(a, f32), D, dev(b, f32), D, G, R, D, wide[:, D:], 3 * D)
    close(wide[:, D:2 * D], a * b.repeat_interleave(G, 0), name="row products")
    assert float((wide[:, :D] - 9.0).abs().max()) == 0 and float((wide[:, 2 * D:] - 9.0).abs().max()) == 0


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 80, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 36), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (130, 2, 33, 48, 80)])
def test_fused_layer0_backward_reductions_fp32(Hn, G, T, Q, A0):
    """clsr_att_l0_bwd: da, dq, dU, dV of the re-associated first attention layer from one pass over dz0 on the fp32
    matrix pipe (daq = dz0 . Wp^T never stored) == float64, and == the three kernels it replaces."""
    assert query("clsr_att_l0_bwd_supported", G, Q, A0) == 1
    assert query("clsr_att_l0_bwd_supported", 9, Q, A0) == 0 and query("clsr_att_l0_bwd_supported", G, 96, A0) == 0
    g = torch.Generator().manual_seed(Hn * 7 + T)
    R, M = Hn * G, Hn * G * T
    dz0, Wp = rnd(g, M, A0, scale=0.5), rnd(g, Q, A0, scale=0.2)       # Wp: [Q, A0] block of the layer-0 weights
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    Wt, Kp = ops.pack_weight(dev(Wp, torch.float32), Q, A0, transposed=True)   # rows = out features of dz0 . Wp^T
    f = lambda t: dev(t, torch.float32)
    da, dq = torch.full((Hn * T, Q), 7.0, device="cuda"), torch.full((R, Q), 7.0, device="cuda")
    dU, dV = torch.full((Hn * T, A0), 7.0, device="cuda"), torch.full((R, A0), 7.0, device="cuda")
    ddz0, da_, dq_ = f(dz0), f(a), f(q)
    call("clsr_att_l0_bwd", ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0)
    torch.cuda.synchronize()
    d = dz0.double().view(Hn, G, T, A0)
    daq = d @ Wp.double().t()
    a4, q4 = a.double().view(Hn, 1, T, Q), q.double().view(Hn, G, 1, Q)
    close(da, (daq * q4).sum(1).reshape(Hn * T, Q), rtol=1e-5, atol=2e-5, name="da")
    close(dq, (daq * a4).sum(2).reshape(R, Q), rtol=1e-5, atol=1e-4, name="dq")
    close(dU, d.sum(1).reshape(Hn * T, A0), rtol=1e-5, atol=1e-5, name="dU")
    close(dV, d.sum(2).reshape(R, A0), rtol=1e-5, atol=2e-5, name="dV")
    # dU = NULL (G == 1 callers alias dU with dz0): nothing else changes
    da2, dq2, dV2 = torch.zeros_like(da), torch.zeros_like(dq), torch.zeros_like(dV)
    call("clsr_att_l0_bwd", ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da2, Q, dq2, Q, None, 0, dV2, A0)
    torch.cuda.synchronize()
    assert torch.equal(da2, da) and torch.equal(dq2, dq) and torch.equal(dV2, dV)
    # the path it replaces
    daq_d = torch.zeros(M, Q, device="cuda")
    call("clsr_pgemm", ddz0, A0, 0, 0, None, 0, None, None, 0, Wt, Kp, None, None, 0, None, 0, daq_d, Q, 0, None, M, A0, Q)
    da3, dq3, dU3, dV3 = torch.zeros_like(da), torch.zeros_like(dq), torch.zeros_like(dU), torch.zeros_like(dV)
    call("clsr_att_prod_bwd", daq_d, da_, dq_, Hn, G, T, Q, da3, dq3)
    call("clsr_att_z0_bwd_reduce", ddz0, Hn, G, T, A0, dU3, dV3)
    torch.cuda.synchronize()
    close(da, da3, rtol=1e-5, atol=2e-5, name="da vs three kernels")
    close(dq, dq3, rtol=1e-5, atol=1e-4, name="dq vs three kernels")
    close(dU, dU3, rtol=1e-6, atol=1e-6, name="dU vs three kernels")
    close(dV, dV3, rtol=1e-5, atol=2e-5, name="dV vs three kernels")


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 80, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 36), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (2100, 2, 33, 48, 80), (11, 8, 18, 80, 80), (5, 4, 20, 40, 80),
                                         (7, 5, 16, 80, 40), (3, 2, 8, 16, 16)])   # packed / unpacked ragged tiles
def test_attention_layer0_forward_one_wave_per_history(Hn, G, T, Q, A0):
    """clsr_att_l0_fwd: z0 = U[h,t] + V[r] + (a[h,t] * q[r]) . Wp with the batch-norm column sums == float64, and ==
    clsr_pgemm with the Xmul prologue / addU + addV epilogue (the kernel it replaces)."""
    assert query("clsr_att_l0_fwd_supported", G, Q, A0) == 1
    assert query("clsr_att_l0_fwd_supported", 9, Q, A0) == 0 and query("clsr_att_l0_fwd_supported", G, 132, A0) == 0
    g = torch.Generator().manual_seed(Hn * 3 + T)
    R, M = Hn * G, Hn * G * T
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    U, V, Wp = rnd(g, Hn * T, A0), rnd(g, R, A0), rnd(g, Q, A0, scale=0.2)
    Wt, Kp = ops.pack_weight(dev(Wp, torch.float32), A0, Q)
    f = lambda t: dev(t, torch.float32)
    parts = query("clsr_att_l0_fwd_stats_parts", Hn)
    st = torch.full((parts, 2, A0), 7.0, dtype=torch.float64, device="cuda")
    z0 = torch.full((M, A0 + 4), 7.0, device="cuda")          # strided output
    da, dq, dU, dV = f(a), f(q), f(U), f(V)
    call("clsr_att_l0_fwd", da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z0, A0 + 4, st, Hn, G, T, Q, A0)
    torch.cuda.synchronize()
    a4, q4 = a.view(Hn, 1, T, Q), q.view(Hn, G, 1, Q)
    exp = ((a4 * q4) @ Wp + U.view(Hn, 1, T, A0) + V.view(Hn, G, 1, A0)).reshape(M, A0)
    close(z0[:, :A0], exp, rtol=1e-5, atol=2e-5, name="z0")
    assert float((z0[:, A0:] - 7.0).abs().max()) == 0
    got = z0[:, :A0].double()
    close(st.sum(0)[0], got.sum(0), rtol=1e-6, atol=1e-4, name="column sums")
    close(st.sum(0)[1], (got * got).sum(0), rtol=1e-6, atol=1e-4, name="column sums of squares")
    # no statistics (scoring)
    z1 = torch.zeros(M, A0, device="cuda")
    call("clsr_att_l0_fwd", da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z1, A0, None, Hn, G, T, Q, A0)
    # the position-tiled kernel
    z2 = torch.zeros(M, A0, device="cuda")
    call("clsr_pgemm", da, Q, T, G, dq, Q, None, None, 1, Wt, Kp, None, dU, A0, dV, A0, z2, A0, 0, None, M, Q, A0)
    torch.cuda.synchronize()
    assert torch.equal(z1, z0[:, :A0].contiguous())
    close(z1, z2, rtol=1e-6, atol=2e-6, name="z0 vs clsr_pgemm")


@pytest.mark.parametrize("entry", ["clsr_pgemm_dw_partial_multi", "clsr_hdw_partial_multi"])
def test_weight_gradient_multi_job_launch(entry):
    """Several weight-gradient products in one launch == the same products launched one by one (same partial chunks):
    plain, X * Xmul[r] and relu(X * scale + shift) jobs of different shapes, column slices of wider matrices."""
    g = torch.Generator().manual_seed(11)
    M = 3000
    f = lambda t: dev(t, torch.float32)
    Xw, dYw = f(rnd(g, M, 200)), f(rnd(g, M, 480))
    mul = f(rnd(g, M, 120))
    sc, sh = f(rnd(g, 80)), f(rnd(g, 80))
    single = "clsr_hdw_partial" if "hdw" in entry else "clsr_pgemm_dw_partial"
    #        X col0, K, dY col0, N, Xmul, affine
    specs = [(0, 40, 0, 480, None, False), (40, 40, 0, 160, None, False), (80, 80, 160, 120, None, False),
             (160, 40, 280, 80, None, False), (160, 40, 360, 40, mul, False), (80, 80, 400, 40, None, True),
             (0, 36, 440, 24, None, False),
             (0, 128, 0, 256, None, False), (64, 120, 100, 128, mul, False), (0, 200, 40, 440, None, False)]   # several K chunks
    jobs, ws_multi, ws_single = [], [], []
    for x0, K, y0, N, xm, aff in specs:
        need = query("clsr_pgemm_dw_workspace_floats", M, K, N)
        wa, wb = torch.zeros(need, device="cuda"), torch.zeros(need, device="cuda")
        ws_multi.append(wa)
        ws_single.append(wb)
        X, dY = Xw[:, x0:], dYw[:, y0:]
        jobs.append((X.data_ptr(), xm.data_ptr() if xm is not None else 0, sc.data_ptr() if aff else 0,
                     sh.data_ptr() if aff else 0, dY.data_ptr(), wa.data_ptr(), 0, 200, 0, 0, 120 if xm is not None else 0,
                     1, 0, 480, M, K, N, 0))
        if "hdw" in entry:
            call(single, X, 0, 200, 0, 0, xm, 120 if xm is not None else 0, sc if aff else None, sh if aff else None, 1, dY,
                 0, 480, M, K, N, wb)
        else:
            call(single, X, 200, 0, 0, xm, 120 if xm is not None else 0, sc if aff else None, sh if aff else None, 1, dY,
                 480, M, K, N, wb)
    ops.dw_multi(entry, jobs)
    torch.cuda.synchronize()
    for i, (wa, wb) in enumerate(zip(ws_multi, ws_single)):
        assert torch.equal(wa, wb), "job %d differs from its single launch" % i
    assert float(ws_multi[0].abs().max()) > 0
    if "hdw" in entry:      # bf16 dY (speed mode: dPin stored as bf16): plain and X * Xmul jobs
        dYh = dYw.to(torch.bfloat16)
        jobs, pairs = [], []
        for x0, K, y0, N, xm, aff in specs:
            if aff:
                continue
            need = query("clsr_pgemm_dw_workspace_floats", M, K, N)
            wa, wb = torch.zeros(need, device="cuda"), torch.zeros(need, device="cuda")
            pairs.append((wa, wb))
            X, dY = Xw[:, x0:], dYh[:, y0:]
            jobs.append((X.data_ptr(), xm.data_ptr() if xm is not None else 0, 0, 0, dY.data_ptr(), wa.data_ptr(), 0, 200, 0,
                         0, 120 if xm is not None else 0, 1, 1, 480, M, K, N, 0))
            call(single, X, 0, 200, 0, 0, xm, 120 if xm is not None else 0, None, None, 1, dY, 1, 480, M, K, N, wb)
        ops.dw_multi(entry, jobs)
        torch.cuda.synchronize()
        for i, (wa, wb) in enumerate(pairs):
            assert torch.equal(wa, wb), "bf16-dY job %d differs from its single launch" % i


def test_rnn_backward_with_bf16_input_projection_gradients():
    """clsr_rnn_bwd_multi with dPin as a bf16 tensor (speed mode): the same values as the fp32 run, rounded once when
    stored -- GRU + Time4LSTM in one launch, ragged lengths (zeros past the length), column slices of one wide buffer."""
    g = torch.Generator().manual_seed(21)
    Hn, T, n = 37, 9, 40
    NX = 3 * n + 6 * n
    f = lambda t: dev(t, torch.float32)
    lens = dev(torch.randint(1, T + 1, (Hn,), generator=g).int())
    gates, hprev = f(torch.rand(Hn, T, 3 * n, generator=g)), f(rnd(g, Hn, T, n))
    Wg, Wc = f(rnd(g, n, 2 * n, scale=0.3)), f(rnd(g, n, n, scale=0.3))
    act, cst = f(torch.rand(Hn, T, 6 * n, generator=g)), f(rnd(g, Hn, T, n))
    Wm = f(rnd(g, n, 4 * n, scale=0.3))
    dseq, dhT = f(rnd(g, Hn, T, n)), f(rnd(g, Hn, n))
    outs = []
    for dt in (torch.float32, torch.bfloat16):
        dP = torch.full((Hn * T, NX), 3.0, device="cuda", dtype=dt)
        gd = ops.gru_desc(n, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, hprev=hprev, gates=gates, dhT=dhT, dout_seq=None,
                          dPin=dP[:, :3 * n], lddp=NX)
        td = ops.t4_desc(n, Wm=Wm, ldm=4 * n, act=act, cst=cst, dout_seq=dseq, dPin=dP[:, 3 * n:], lddp=NX)
        ops.rnn_multi("clsr_rnn_bwd_multi", [gd], td, lens, 1, Hn, T)
        torch.cuda.synchronize()
        outs.append(dP)
    assert float(outs[0].abs().max()) > 0
    assert torch.equal(outs[1], outs[0].to(torch.bfloat16))
    valid = (torch.arange(T, device="cuda")[None, :] < lens[:, None]).reshape(-1)
    assert float(outs[1][~valid].abs().max()) == 0


@pytest.mark.parametrize("M,C1,C0", [(2000, 40, 80), (515, 40, 80), (300, 48, 128), (70, 12, 20), (4099, 40, 36)])
def test_attention_layer1_backward_two_passes_fp32(M, C1, C0):
    """clsr_att_l1_bwd: dz1 recomputed from (z1, ds) -> dh0 = dz1 . W1^T -> ReLU / batch-norm backward of layer 0: the
    statistics pass and the apply pass == float64, and == the three kernels they replace (dy1-apply, GEMM with the
    fused BN sums, bn-apply)."""
    assert query("clsr_att_l1_bwd_supported", C1, C0) == 1 and query("clsr_att_l1_bwd_supported", 64, C0) == 0
    g = torch.Generator().manual_seed(3)
    f = lambda t: dev(t, torch.float32)
    z1, z0, ds = rnd(g, M, C1), rnd(g, M, C0), rnd(g, M)
    W1 = rnd(g, C0, C1, scale=0.3)                            # layer-1 weight [in = C0, out = C1]
    sc1, sh1 = torch.rand(C1, generator=g, dtype=torch.float64) + 0.5, rnd(g, C1, scale=0.3)
    wo, coef1 = rnd(g, C1), rnd(g, 3 * C1, scale=0.5)
    sc0, sh0 = torch.rand(C0, generator=g, dtype=torch.float64) + 0.5, rnd(g, C0, scale=0.3)
    mean0, inv0 = rnd(g, C0, scale=0.1), torch.rand(C0, generator=g, dtype=torch.float64) + 0.5
    coef0 = rnd(g, 3 * C0, scale=0.5)
    Wt, Kp = ops.pack_weight(dev(W1, torch.float32), C0, C1, transposed=True)   # dh0 = dz1 . W1^T: out = C0, in = C1
    d = {k: f(v) for k, v in dict(z1=z1, z0=z0, ds=ds, sc1=sc1, sh1=sh1, wo=wo, coef1=coef1, sc0=sc0, sh0=sh0, mean0=mean0,
                                  inv0=inv0, coef0=coef0).items()}
    parts = query("clsr_att_l1_bwd_stats_parts", M)
    st = torch.full((parts, 2, C0), 7.0, dtype=torch.float64, device="cuda")
    call("clsr_att_l1_bwd", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], d["mean0"], d["inv0"], None, None, 0, None, 0, st, M, C1, C0)
    dz1 = torch.full((M, C1), 7.0, device="cuda")
    dz0 = torch.full((M, C0 + 4), 7.0, device="cuda")
    call("clsr_att_l1_bwd", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], None, None, d["coef0"], dz1, C1, dz0, C0 + 4, None, M, C1, C0)
    torch.cuda.synchronize()
    y1 = z1 * sc1 + sh1
    a1, a2, a3 = coef1[:C1], coef1[C1:2 * C1], coef1[2 * C1:]
    x = torch.where(y1 > 0, (a1 * wo) * ds[:, None], torch.zeros_like(y1)) + a2 * z1 + a3
    close(dz1, x, rtol=1e-5, atol=1e-5, name="dz1")
    dh0 = x @ W1.t()
    y0 = z0 * sc0 + sh0
    dy0 = torch.where(y0 > 0, dh0, torch.zeros_like(dh0))
    xhat = (z0 - mean0) * inv0
    scale = float(dy0.abs().sum(0).max())
    close(st.sum(0)[0], dy0.sum(0), rtol=1e-5, atol=1e-6 * scale, name="sum dy0")
    close(st.sum(0)[1], (dy0 * xhat).sum(0), rtol=1e-5, atol=2e-6 * scale, name="sum dy0 * xhat0")
    c1, c2, c3 = coef0[:C0], coef0[C0:2 * C0], coef0[2 * C0:]
    close(dz0[:, :C0], c1 * dy0 + c2 * z0 + c3, rtol=1e-5, atol=2e-5, name="dz0")
    assert float((dz0[:, C0:] - 7.0).abs().max()) == 0
    # the path it replaces
    dz1_b = torch.zeros(M, C1, device="cuda")
    call("clsr_att_dy1_apply", d["z1"], d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], M, C1, dz1_b)
    p2 = query("clsr_pgemm_stats_parts", M)
    st2 = torch.zeros(p2, 2, C0, dtype=torch.float64, device="cuda")
    dy0_b = torch.zeros(M, C0, device="cuda")
    call("clsr_pgemm_bnbwd", dz1_b, C1, Wt, Kp, dy0_b, C0, d["z0"], C0, d["sc0"], d["sh0"], d["mean0"], d["inv0"], st2, M, C1, C0)
    call("clsr_bn_bwd_apply", dy0_b, d["z0"], d["coef0"], M, C0)
    torch.cuda.synchronize()
    close(dz1, dz1_b, rtol=1e-6, atol=1e-6, name="dz1 vs dy1-apply kernel")
    close(st.sum(0), st2.sum(0), rtol=1e-6, atol=1e-6 * scale, name="BN sums vs clsr_pgemm_bnbwd")
    close(dz0[:, :C0], dy0_b, rtol=1e-5, atol=2e-5, name="dz0 vs GEMM + bn-apply kernels")


# ------------------------------------------------------------------------------- merged small launches (round 2)
def test_zero_multi_ranges():
    """clsr_zero_multi clears exactly the given byte ranges (unaligned starts, odd word counts, empty range)."""
    buf = torch.full((4099,), 7.0, device="cuda")
    d8 = torch.full((24,), 3.0, dtype=torch.float64, device="cuda")
    one = torch.full((3,), 5.0, device="cuda")
    ops.multi("clsr_zero_multi", ops.ZeroDesc, [(buf[1:].data_ptr(), 4093 * 4), (d8.data_ptr(), 24 * 8),
                                                (one[1:].data_ptr(), 4), (one.data_ptr(), 0)])
    torch.cuda.synchronize()
    b = buf.cpu()
    assert float(b[0]) == 7.0 and float(b[1:4094].abs().max()) == 0.0 and bool((b[4094:] == 7.0).all())
    assert float(d8.abs().max()) == 0.0
    assert one.cpu().tolist() == [5.0, 0.0, 5.0]


def test_scatter_add_rows_multi_matches_single_launches():
    g = torch.Generator().manual_seed(21)
    V, N, G = 301, 77, 3
    sites = [(40, 0, 40, 40), (48, 8, 32, 32), (48, 40, 8, 8)]     # (ld_src, col0, C, table width)
    idx = torch.randint(0, V, (N * G,), generator=g)
    d_idx = dev(idx, torch.int32)
    rows, exp, grads, sss = [], [], [], []
    for ld, c0, C, _ in sites:
        src = rnd(g, N, ld)
        dsrc = dev(src, torch.float32)
        grad = torch.zeros(V, C, device="cuda")
        ss = torch.zeros(1, dtype=torch.float64, device="cuda")
        rows.append((dsrc.data_ptr(), d_idx.data_ptr(), grad.data_ptr(), ss.data_ptr(), G, ld, c0, N, C))
        exp.append((torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx[::G], src[:, c0:c0 + C]),
                    (src[:, c0:c0 + C] ** 2).sum().reshape(1)))
        grads.append(grad)
        sss.append(ss)
        rows[-1] = rows[-1] + ()
        exp[-1] = exp[-1] + (dsrc,)    # keep the device source alive until the launch
    ops.multi("clsr_scatter_add_rows_multi", ops.ScatterDesc, rows)
    for grad, ss, (e, es, _) in zip(grads, sss, exp):
        close(grad, e, rtol=1e-4, atol=1e-5, name="scatter multi")
        close(ss, es, name="scatter multi sumsq")


@pytest.mark.parametrize("M,C,parts", [(20480, 100, 160), (333, 64, 7), (4096, 40, 1)])
def test_bn_bwd_coef_apply_fused_equals_two_launches(M, C, parts):
    g = torch.Generator().manual_seed(M)
    f32 = torch.float32
    part = dev(rnd(g, parts, 2, C))                     # float64 partial sums
    gamma, mean, invstd = dev(rnd(g, C), f32), dev(rnd(g, C), f32), dev(rnd(g, C).abs() + 0.5, f32)
    dy, z = rnd(g, M, C), rnd(g, M, C)
    dy_a, dy_b, dz = dev(dy, f32), dev(dy, f32), dev(z, f32)
    coef_a, coef_b = torch.empty(3, C, device="cuda"), torch.empty(3, C, device="cuda")
    dg_a, db_a, dg_b, db_b = (torch.zeros(C, device="cuda") for _ in range(4))
    call("clsr_bn_bwd_coef", part, parts, C, float(M), gamma, mean, invstd, coef_a, dg_a, db_a, 0)
    call("clsr_bn_bwd_apply", dy_a, dz, coef_a, M, C)
    call("clsr_bn_bwd_coef_apply", part, parts, C, float(M), gamma, mean, invstd, coef_b, dg_b, db_b, dy_b, dz, M)
    close(coef_b, coef_a.double().cpu(), rtol=1e-6, atol=1e-7, name="coef")
    close(dg_b, dg_a.double().cpu(), rtol=1e-6, atol=1e-7, name="dgamma")
    close(db_b, db_a.double().cpu(), rtol=1e-6, atol=1e-7, name="dbeta")
    close(dy_b, dy_a.double().cpu(), rtol=1e-5, atol=1e-6, name="dz")


def test_dense_reg_norm_tick_advances_the_adam_clock():
    g = torch.Generator().manual_seed(5)
    sizes = [300, 7]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(off[-1])
    p, gr = rnd(g, n), rnd(g, n)
    f32 = torch.float32
    dp_, dg_ = dev(p, f32), dev(gr, f32)
    ss = torch.zeros(2, dtype=torch.float64, device="cuda")
    st = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.0], dtype=torch.float64, device="cuda")
    call("clsr_dense_reg_norm_tick", dp_, dg_, dev(off), 2, 1e-3, 0.0, ss, None, st, 1e-3, 0.9, 0.999)
    call("clsr_dense_reg_norm_tick", dp_, dg_, dev(off), 2, 0.0, 0.0, ss, None, st, 1e-3, 0.9, 0.999)
    lr_t = 1e-3 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert abs(float(st[3]) - lr_t) < 1e-12 and float(st[0]) == 2.0
    g2 = gr + 1e-3 * p
    close(ss, torch.stack([(g2[:300] ** 2).sum(), (g2[300:] ** 2).sum()]), rtol=1e-5, name="sumsq")
    call("clsr_dense_reg_norm_tick", dp_, dg_, dev(off), 2, 0.0, 0.0, ss, None, None, 0.0, 0.0, 0.0)
    assert float(st[0]) == 2.0


@pytest.mark.parametrize("M1,M2,K,N", [(5000, 700, 80, 80), (300, 9000, 40, 96), (64, 64, 164, 36)])
def test_dw_reduce_batch_two_products_into_one_block(M1, M2, K, N):
    """A reduction descriptor with a second partial buffer: dW = X1^T dY1 - X2^T dY2 in the batched reduction itself
    (the W0d block of the re-associated attention layer), next to ordinary descriptors of the same launch."""
    g = torch.Generator().manual_seed(M1 + K)
    f32 = torch.float32
    X1, Y1, X2, Y2 = rnd(g, M1, K), rnd(g, M1, N), rnd(g, M2, K), rnd(g, M2, N)
    d = [dev(t, f32) for t in (X1, Y1, X2, Y2)]
    ws1 = torch.zeros(query("clsr_pgemm_dw_workspace_floats", M1, K, N), device="cuda")
    ws2 = torch.zeros(query("clsr_pgemm_dw_workspace_floats", M2, K, N), device="cuda")
    call("clsr_pgemm_dw_partial", d[0], K, 0, 0, None, 0, None, None, 1, d[1], N, M1, K, N, ws1)
    call("clsr_pgemm_dw_partial", d[2], K, 0, 0, None, 0, None, None, 1, d[3], N, M2, K, N, ws2)
    dW1, dW2, dWd = (torch.full((K, N), 9.0, device="cuda") for _ in range(3))
    db1 = torch.zeros(N, device="cuda")
    p1, p2 = query("clsr_pgemm_dw_parts", M1), query("clsr_pgemm_dw_parts", M2)
    sig = ((ws1.data_ptr(), dW1.data_ptr(), db1.data_ptr(), 1.0, p1, K, N, N, 0),
           (ws2.data_ptr(), dW2.data_ptr(), 0, 1.0, p2, K, N, N, 0),
           (ws1.data_ptr(), dWd.data_ptr(), 0, 1.0, p1, K, N, N, 0, ws2.data_ptr(), -1.0, p2))
    tab = ops.dw_table(sig, torch.device("cuda"))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    e1, e2 = X1.T @ Y1, X2.T @ Y2
    sc = float(max(e1.abs().max(), e2.abs().max()))
    close(dW1, e1, rtol=1e-5, atol=2e-6 * sc, name="dW1")
    close(db1, Y1.sum(0), rtol=1e-5, atol=2e-6 * sc, name="db1")
    close(dW2, e2, rtol=1e-5, atol=2e-6 * sc, name="dW2")
    close(dWd, e1 - e2, rtol=1e-5, atol=4e-6 * sc, name="dW1 - dW2")


# ------------------------------------------------------------------------------- recurrences / products over time ranges
def _ranges(T, n):
    return [(k * T // n, (k + 1) * T // n) for k in range(n)]


@pytest.mark.parametrize("form", ["fp32", "x3"])
@pytest.mark.parametrize("Hn,T,n,nch", [(37, 10, 40, 5), (16, 50, 40, 5), (21, 9, 128, 2), (5, 23, 40, 4)])
def test_recurrences_as_a_chain_of_time_ranges_equal_one_launch(Hn, T, n, nch, form):
    """clsr_rnn_{fwd,bwd}_multi_range over consecutive ranges (state carried through h0 / hT, st_in / st_out, dhT / dh0,
    dst_in / dst_out) == ONE launch over [0, T): every output bit for bit (same instruction sequence per step), in both
    forms of the hidden-to-hidden products."""
    g = torch.Generator().manual_seed(Hn + T)
    f = lambda t: dev(t, torch.float32)
    ldp = 3 * n + 6 * n
    Pin = f(rnd(g, Hn * T, ldp))
    Wgh, Wch, Wm = f(rnd(g, n, 2 * n) * 0.3), f(rnd(g, n, n) * 0.3), f(rnd(g, n, 4 * n) * 0.3)
    h0 = f(rnd(g, Hn, n) * 0.5)
    lens = torch.randint(1, T + 1, (Hn,), generator=g)
    lens[0], lens[-1] = T, 1
    d_len = dev(lens, torch.int32)
    dhT, dseq_g, dseq_t = f(rnd(g, Hn, n)), f(rnd(g, Hn, T, n)), f(rnd(g, Hn, T, n))

    def run(ranges):
        z = lambda *s: torch.full(s, 3.0, device="cuda")
        o = dict(hT=z(Hn, n), seq=z(Hn, T, n), hprev=torch.zeros(Hn, T, n, device="cuda"),
                 gates=torch.zeros(Hn, T, 3 * n, device="cuda"), out=z(Hn, T, n), act=torch.zeros(Hn, T, 6 * n, device="cuda"),
                 cst=torch.zeros(Hn, T, n, device="cuda"), mprev=torch.zeros(Hn, T, n, device="cuda"),
                 dPin=z(Hn * T, ldp), dh0=z(Hn, n))
        st, dst, dhc = torch.zeros(Hn, 2 * n, device="cuda"), torch.zeros(Hn, 2 * n, device="cuda"), torch.zeros(Hn, n, device="cuda")
        for k, (t0, t1) in enumerate(ranges):
            first, last = k == 0, k == len(ranges) - 1
            gd = ops.gru_desc(n, Pin=Pin, ldp=ldp, Wgh=Wgh, ldg=2 * n, Wch=Wch, ldc=n, h0=h0 if first else o["hT"],
                              h0_stride=n, hT=o["hT"], out_seq=o["seq"], hprev=o["hprev"], gates=o["gates"], products=form)
            td = ops.t4_desc(n, Pin=Pin[:, 3 * n:], ldp=ldp, Wm=Wm, ldm=4 * n, out_seq=o["out"], act=o["act"], cst=o["cst"],
                             mprev=o["mprev"], st_in=None if first else st, st_out=None if last else st, products=form)
            ops.rnn_multi("clsr_rnn_fwd_multi", [gd], td, d_len, 1, Hn, T, t_range=(t0, t1))
        for k, (t0, t1) in enumerate(reversed(ranges)):
            first, last = k == 0, k == len(ranges) - 1
            gd = ops.gru_desc(n, Wgh=Wgh, ldg=2 * n, Wch=Wch, ldc=n, hprev=o["hprev"], gates=o["gates"],
                              dhT=dhT if first else dhc, dout_seq=dseq_g, dPin=o["dPin"], lddp=ldp,
                              dh0=o["dh0"] if last else dhc, products=form)
            td = ops.t4_desc(n, Wm=Wm, ldm=4 * n, act=o["act"], cst=o["cst"], dout_seq=dseq_t, dPin=o["dPin"][:, 3 * n:],
                             lddp=ldp, dst_in=None if first else dst, dst_out=None if last else dst, products=form)
            ops.rnn_multi("clsr_rnn_bwd_multi", [gd], td, d_len, 1, Hn, T, t_range=(t0, t1))
        torch.cuda.synchronize()
        return o

    one, chain = run([(0, T)]), run(_ranges(T, nch))
    for k in one:
        assert torch.equal(one[k], chain[k]), k
    assert float(one["dPin"].abs().max()) > 0 and float(one["out"].abs().max()) > 0


@pytest.mark.parametrize("Hn,T,K,N,nch", [(37, 10, 40, 480, 5), (9, 50, 80, 120, 5), (5, 23, 480, 40, 4)])
def test_product_over_time_ranges_equals_the_whole_product(Hn, T, K, N, nch):
    """clsr_pgemm_range over the ranges of [Hn, T, .] tensors == clsr_pgemm over all rows (bias / accumulate forms)."""
    g = torch.Generator().manual_seed(K + N)
    f = lambda t: dev(t, torch.float32)
    X, W, b = f(rnd(g, Hn * T, K + 4)), rnd(g, K, N), f(rnd(g, N))
    Wt, Kp = ops.pack_weight(f(W), N, K)
    Y0 = f(rnd(g, Hn * T, N + 8))
    for acc, bias in ((0, b), (1, None)):
        whole, parts = Y0.clone(), Y0.clone()
        call("clsr_pgemm", X, K + 4, 0, 0, None, 0, None, None, 0, Wt, Kp, bias, None, 0, None, 0, whole, N + 8, acc, None,
             Hn * T, K, N)
        for t0, t1 in _ranges(T, nch):
            call("clsr_pgemm_range", X, K + 4, Wt, Kp, bias, parts, N + 8, acc, Hn, T, t0, t1, K, N)
        torch.cuda.synchronize()
        assert torch.equal(whole, parts), (acc,)
    exp = X[:, :K].double().cpu() @ W + b.double().cpu()
    whole = torch.zeros(Hn * T, N, device="cuda")
    for t0, t1 in _ranges(T, nch):
        call("clsr_pgemm_range", X, K + 4, Wt, Kp, b, whole, N, 0, Hn, T, t0, t1, K, N)
    close(whole, exp, rtol=2e-5, atol=2e-6 * K, name="Y")


@pytest.mark.parametrize("Hn,T,nch", [(37, 10, 5), (64, 50, 5), (9, 23, 4)])
def test_weight_gradients_over_time_ranges_fill_one_workspace(Hn, T, nch):
    """Multi-job weight-gradient launches over time ranges (each writing its own partial slots of ONE workspace), summed by
    one clsr_dw_reduce_batch descriptor == the products over all rows; plain and X * Xmul jobs; + the time-feature sums."""
    g = torch.Generator().manual_seed(T)
    f = lambda t: dev(t, torch.float32)
    M = Hn * T
    Xw, dYw, mul = f(rnd(g, M, 120)), f(rnd(g, M, 480)), f(rnd(g, M, 120))
    specs = [(0, 40, 0, 480, None), (40, 40, 0, 160, None), (80, 40, 360, 40, mul), (0, 80, 160, 120, None)]
    ranges = _ranges(T, nch)
    tcmax = max(b - a for a, b in ranges)
    pgx = query("clsr_pgemm_dw_parts", Hn * tcmax)
    wss, outs, sig = [], [], []
    for x0, K, y0, N, xm in specs:
        ws = torch.full((nch * query("clsr_pgemm_dw_workspace_floats", Hn * tcmax, K, N),), 7.0, device="cuda")
        dW, db = torch.zeros(K, N, device="cuda"), torch.zeros(N, device="cuda")
        wss.append(ws)
        outs.append((dW, db))
        sig.append((ws.data_ptr(), dW.data_ptr(), db.data_ptr(), 1.0, nch * pgx, K, N, N, 0))
    for k, (t0, t1) in enumerate(ranges):
        tc = t1 - t0
        jobs = [(Xw[:, x0:].data_ptr(), xm.data_ptr() if xm is not None else 0, 0, 0, dYw[:, y0:].data_ptr(), ws.data_ptr(),
                 0, 120, 0, 0, 120 if xm is not None else 0, 1, 0, 480, Hn * tc, K, N, 0, tc, T, t0, pgx, nch * pgx, k * pgx)
                for (x0, K, y0, N, xm), ws in zip(specs, wss)]
        ops.dw_multi("clsr_pgemm_dw_partial_multi", jobs)
    tab = ops.dw_table(tuple(sig), torch.device("cuda"))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    Xd, Yd, md = Xw.double().cpu(), dYw.double().cpu(), mul.double().cpu()
    for (x0, K, y0, N, xm), (dW, db) in zip(specs, outs):
        x = Xd[:, x0:x0 + K] * (md[:, :K] if xm is not None else 1.0)
        close(dW, x.T @ Yd[:, y0:y0 + N], rtol=2e-4, atol=2e-4 * math.sqrt(M), name="dW")
        close(db, Yd[:, y0:y0 + N].sum(0), rtol=2e-4, atol=2e-4 * math.sqrt(M), name="db")
    # time-feature gradient sums over ranges == over all steps
    n = 40
    dTT, TT = f(rnd(g, M, 2 * n)), f(torch.tanh(rnd(g, M, 2 * n)))
    tnow, tfirst = f(rnd(g, Hn, T)), f(rnd(g, Hn, T))
    p_all = torch.zeros(query("clsr_t4_time_inputs_bwd_parts", Hn, T, n), 2, 2 * n, device="cuda")
    call("clsr_t4_time_inputs_bwd", dTT, TT, tnow, tfirst, T, Hn, T, n, p_all)
    tparts = [query("clsr_t4_time_inputs_bwd_parts", Hn, b - a, n) for a, b in ranges]
    p_rng = torch.full((sum(tparts), 2, 2 * n), 5.0, device="cuda")
    for k, (t0, t1) in enumerate(ranges):
        call("clsr_t4_time_inputs_bwd_range", dTT, TT, tnow, tfirst, T, Hn, T, t0, t1, n, p_rng[sum(tparts[:k]):])
    close(p_rng.sum(0), p_all.sum(0).double(), rtol=1e-4, atol=1e-4 * math.sqrt(M), name="time-feature sums")


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("M", [16 * 300, 16 * 257 + 5, 37])
def test_fused_encoder_backward_one_pass_over_dpin(M, fold):
    """clsr_enc_bwd_fused (csrc/encbwd.hip): the seven encoder-side weight gradients (+ bias sums) and d(hist) from one
    pass over dPin == float64 products of the same operands, through the same partial layout + clsr_dw_reduce_batch
    that the step uses."""
    g = torch.Generator().manual_seed(M)
    f = lambda t: dev(t, torch.float32)
    n = 40
    dPin, hist = f(rnd(g, M, 480)), f(rnd(g, M, n))
    hp1, hp2, mp, TT = f(rnd(g, M, n)), f(rnd(g, M, n)), f(rnd(g, M, n)), f(torch.tanh(rnd(g, M, 2 * n)))
    g1, g2 = f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64)), f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64))
    Wx = rnd(g, n, 480) * 0.2                     # [in = hist feature, out = projection column]
    Wt, Kp = ops.pack_weight(f(Wx), n, 480, transposed=True)      # rows = hist features, K = 480
    dhist0 = f(rnd(g, M, n))
    dhist = dhist0.clone()
    parts = query("clsr_enc_bwd_fused_parts", M)
    shapes = [(n, 480), (n, 80), (n, 40), (n, 160), (2 * n, 120), (n, 80), (n, 40)]
    wss = [torch.full((query("clsr_enc_bwd_fused_workspace_floats", M, i),), 9.0, device="cuda") for i in range(7)]
    outs = [torch.zeros(K, N, device="cuda") for K, N in shapes]
    db = torch.zeros(480, device="cuda")
    extra = 0.0
    if fold:
        # clsr_enc_bwd_fused_fold: + a second d(hist) tensor + the mean / recent-k shares of the history prologue per (h, t) row
        T, k = (37 if M == 37 else 16), 3
        if M % T:
            pytest.skip("M is not whole histories")
        Hn = M // T
        d2, dm, dr = f(rnd(g, M, n)), f(rnd(g, Hn, n)), f(rnd(g, Hn, n))
        lens = torch.randint(0, T + 1, (Hn,), generator=g)
        seq_len = torch.zeros(Hn * 2, dtype=torch.int32)
        seq_len[::2] = lens.int()                       # (len_stride 2)
        call("clsr_enc_bwd_fused_fold", dPin, hist, hp1, g1, mp, TT, hp2, g2, Wt, Kp, dhist, d2, dm, dr, seq_len.cuda(), 2, T, k,
             *wss, M)
        t = torch.arange(T)[None, :]
        L = lens[:, None].double()
        m_ = (t < lens[:, None]).double()
        r_ = ((t < lens[:, None]) & (t >= lens[:, None] - k)).double()
        extra = (d2.double().cpu().view(Hn, T, n) + (m_ / L.clamp(min=1))[..., None] * dm.double().cpu()[:, None, :]
                 + (r_ / torch.minimum(L, torch.tensor(float(k))).clamp(min=1))[..., None] * dr.double().cpu()[:, None, :]).view(M, n)
    else:
        call("clsr_enc_bwd_fused", dPin, hist, hp1, g1, mp, TT, hp2, g2, Wt, Kp, dhist, *wss, M)
    sig = tuple((ws.data_ptr(), o.data_ptr(), db.data_ptr() if i == 0 else 0, 1.0, parts, K, N, N, 0)
                for i, (ws, o, (K, N)) in enumerate(zip(wss, outs, shapes)))
    tab = ops.dw_table(sig, torch.device("cuda"))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    P = d(dPin)
    exp = [d(hist).T @ P, d(hp1).T @ P[:, 0:80], (d(hp1) * d(g1)[:, :n]).T @ P[:, 80:120], d(mp).T @ P[:, 240:400],
           d(TT).T @ P[:, 360:480], d(hp2).T @ P[:, 120:200], (d(hp2) * d(g2)[:, :n]).T @ P[:, 200:240]]
    tol = 2e-5 * math.sqrt(M)
    for i, (o, e) in enumerate(zip(outs, exp)):
        close(o, e, rtol=2e-4, atol=tol, name="product %d" % i)
    close(db, P.sum(0), rtol=2e-4, atol=tol, name="bias sums")
    close(dhist, d(dhist0) + P @ Wx.T + extra, rtol=2e-4, atol=2e-4, name="d(hist)")


@pytest.mark.parametrize("P,G,C1", [(4096, 5, 64), (37, 3, 40), (5, 8, 4), (700, 1, 64)])
def test_mlp_tail_softmax_equals_the_three_launches(P, G, C1):
    """clsr_mlp_tail_softmax (output layer + group-softmax loss + their backward in one launch) == clsr_mlp_out_fwd ->
    clsr_softmax_loss -> clsr_mlp_out_bwd, and the float64 restatement of the same chain."""
    g = torch.Generator().manual_seed(P * 31 + G)
    f32 = torch.float32
    B = P * G
    z = dev(rnd(g, B, C1), f32)
    sc, sh = dev(rnd(g, C1).abs() + 0.5, f32), dev(rnd(g, C1, scale=0.3), f32)
    mu, isd = dev(rnd(g, C1, scale=0.2), f32), dev(rnd(g, C1).abs() + 0.5, f32)
    w, b = dev(rnd(g, C1, scale=0.5), f32), dev(rnd(g, 1), f32)
    labels = torch.zeros(P, G)
    labels[:, 0] = 1.0
    if G > 2:
        labels[::3, 2] = 1.0          # groups with two positives
    labels = dev(labels.reshape(-1), f32)
    lscale = 1.0 / P
    # the three launches
    logit0, dl0 = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    loss0 = torch.zeros(1, dtype=torch.float64, device="cuda")
    call("clsr_mlp_out_fwd", z, sc, sh, w, b, B, C1, logit0)
    call("clsr_softmax_loss", logit0, labels, P, G, lscale, loss0, dl0)
    n0 = query("clsr_mlp_out_bwd_parts", B, C1)
    dy0 = torch.zeros(B, C1, device="cuda")
    bnp0 = torch.zeros(n0, 2, C1, dtype=torch.float64, device="cuda")
    wp0 = torch.zeros(n0, C1 + 4, device="cuda")
    call("clsr_mlp_out_bwd", dl0, z, sc, sh, mu, isd, w, B, C1, dy0, bnp0, wp0)
    # one launch
    assert query("clsr_mlp_tail_softmax_supported", G, C1)
    n1 = query("clsr_mlp_tail_softmax_parts", P)
    logit1, dl1 = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    loss1 = torch.zeros(1, dtype=torch.float64, device="cuda")
    dy1 = torch.full((B, C1), 7.0, device="cuda")
    bnp1 = torch.full((n1, 2, C1), 7.0, dtype=torch.float64, device="cuda")
    wp1 = torch.full((n1, C1 + 4), 7.0, device="cuda")
    call("clsr_mlp_tail_softmax", z, sc, sh, mu, isd, w, b, labels, P, G, C1, lscale, loss1, logit1, dl1, dy1, bnp1, wp1)
    torch.cuda.synchronize()
    close(logit1, logit0, rtol=1e-5, atol=1e-5, name="logit")
    close(dl1, dl0, rtol=2e-5, atol=1e-7, name="dlogit")
    close(loss1, loss0, rtol=1e-6, atol=1e-9, name="loss")
    close(dy1, dy0, rtol=2e-5, atol=1e-7, name="dy1")
    close(bnp1.sum(0), bnp0.sum(0), rtol=1e-5, atol=1e-6, name="bn sums")
    close(wp1[:, :C1 + 1].sum(0), wp0[:, :C1 + 1].sum(0), rtol=2e-5, atol=2e-5, name="w_out / b_out sums")
    # float64 restatement
    zd, lab = z.double().cpu(), labels.double().cpu().reshape(P, G)
    y = zd * sc.double().cpu() + sh.double().cpu()
    lg = (y.clamp_min(0) * w.double().cpu()).sum(1) + b.double().cpu()
    lsm = torch.log_softmax(lg.reshape(P, G), 1)
    close(loss1, -(lsm * lab).sum().reshape(1) * lscale, rtol=1e-5, atol=1e-7, name="loss vs float64")
    dlg = (lscale * (lab.sum(1, keepdim=True) * lsm.exp() - lab)).reshape(-1)
    close(dy1, (dlg[:, None] * w.double().cpu()) * (y > 0), rtol=1e-4, atol=1e-6, name="dy1 vs float64")


@pytest.mark.parametrize("M", [4800, 4117, 37])
def test_fused_encoder_backward_speed_mode(M):
    """clsr_enc_bwd_fused_h (csrc/encbwd.hip): the seven encoder-side weight gradients + bias sums from a bf16 dPin on the
    bf16 matrix pipe == float64 products of the bf16-ROUNDED operands (products exact, accumulation fp32), through the
    partial layout + clsr_dw_reduce_batch that the step uses."""
    g = torch.Generator().manual_seed(M + 1)
    f = lambda t: dev(t, torch.float32)
    n = 40
    dPin = dev(rnd(g, M, 480), torch.float32).to(torch.bfloat16)
    hist, hp1, hp2, mp = f(rnd(g, M, n)), f(rnd(g, M, n)), f(rnd(g, M, n)), f(rnd(g, M, n))
    TT = f(torch.tanh(rnd(g, M, 2 * n)))
    g1, g2 = f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64)), f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64))
    parts = query("clsr_enc_bwd_fused_h_parts", M)
    shapes = [(n, 480), (n, 80), (n, 40), (n, 160), (2 * n, 120), (n, 80), (n, 40)]
    wss = [torch.full((query("clsr_enc_bwd_fused_h_workspace_floats", M, i),), 9.0, device="cuda") for i in range(7)]
    outs = [torch.zeros(K, N, device="cuda") for K, N in shapes]
    db = torch.zeros(480, device="cuda")
    Wx = rnd(g, n, 480) * 0.2                     # [in = hist feature, out = projection column]
    Wh = torch.zeros(64 * query("clsr_hgemm_kp", 480), dtype=torch.bfloat16, device="cuda")
    Kph = query("clsr_hgemm_kp", 480)
    Wxd = f(Wx)
    tab_h = ops.pack_table([ops.pack_desc(Wxd, n, 480, Wh, Kph, transposed=True)], torch.device("cuda"))
    call("clsr_pack_batch_bf16", tab_h[0], tab_h[1], tab_h[2])
    dhist0 = f(rnd(g, M, n))
    dhist = dhist0.clone()
    call("clsr_enc_bwd_fused_h", dPin, hist, hp1, g1, mp, TT, hp2, g2, *wss, Wh, Kph, dhist, M)
    sig = tuple((ws.data_ptr(), o.data_ptr(), db.data_ptr() if i == 0 else 0, 1.0, parts, K, N, N, 0)
                for i, (ws, o, (K, N)) in enumerate(zip(wss, outs, shapes)))
    tab = ops.dw_table(sig, torch.device("cuda"))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    r = lambda t: t.to(torch.bfloat16).double().cpu()          # what the kernel multiplies
    P = dPin.double().cpu()
    exp = [r(hist).T @ P, r(hp1).T @ P[:, 0:80], r(hp1 * g1[:, :n]).T @ P[:, 80:120], r(mp).T @ P[:, 240:400],
           r(TT).T @ P[:, 360:480], r(hp2).T @ P[:, 120:200], r(hp2 * g2[:, :n]).T @ P[:, 200:240]]
    tol = 2e-5 * math.sqrt(M)
    for i, (o, e) in enumerate(zip(outs, exp)):
        close(o, e, rtol=2e-4, atol=tol, name="product %d" % i)
    close(db, P.sum(0), rtol=2e-4, atol=tol, name="bias sums")
    close(dhist, dhist0.double().cpu() + P @ Wxd.to(torch.bfloat16).double().cpu().T, rtol=2e-4, atol=2e-4, name="d(hist)")
