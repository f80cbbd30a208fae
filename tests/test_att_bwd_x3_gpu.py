"""Split-bf16 backward kernels of the attention MLP with the weight gradients folded in (csrc/attbwdx3.hip) against
float64 restatements and against the exact fp32-MFMA kernels they replace.  Every product is hi*hi + hi*lo + lo*hi of
bf16 pairs with fp32 accumulation: ~2^-16 relative per term, so the tolerances are 1e-4 of the result scale, not 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402


def dev(x, dtype=torch.float32):
    return torch.as_tensor(x).to(dtype).cuda().contiguous()


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float64) * scale


def close(got, exp, rel, name):
    got, exp = got.detach().double().cpu(), exp.detach().double().cpu()
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    scale = float(exp.abs().max()) + 1e-30
    err = float((got - exp).abs().max())
    assert err <= rel * scale, "%s: max abs err %.3e at scale %.3e (allowed %.1e of it)" % (name, err, scale, rel)


def chunks_to_matrix(ws, parts, K, N):
    """sum of ``parts`` partial chunks (layout of clsr_pgemm_dw_partial) -> (dW [K, N], bias sums [N])"""
    C = query("clsr_dw_chunk_floats")
    w = ws.view(parts, C).double().sum(0)
    tiles = w[: 25 * 256].view(5, 5, 16, 16)                 # [kt][nt][k & 15][n & 15]
    full = tiles.permute(0, 2, 1, 3).reshape(80, 80)
    return full[:K, :N].cpu(), w[25 * 256: 25 * 256 + N].cpu(), full


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 40, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 40), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (130, 2, 33, 48, 80), (2100, 2, 5, 40, 80), (5, 5, 50, 8, 16)])
@pytest.mark.parametrize("entry,tol,tols", [("clsr_att_l0_bwd_x3", 1e-4, 3e-5), ("clsr_att_l0_bwd_x6", 3e-6, 3e-6)])
def test_layer0_backward_x3_with_weight_gradient(Hn, G, T, Q, A0, entry, tol, tols):
    """x3: two bf16 pieces per operand (2^-16 per product term); x6: three pieces -- every tensor at fp32 accuracy"""
    assert query("clsr_att_l0_bwd_x3_supported", G, Q, A0) == 1
    assert query("clsr_att_l0_bwd_x3_supported", 9, Q, A0) == 0 and query("clsr_att_l0_bwd_x3_supported", G, Q, 36) == 0
    g = torch.Generator().manual_seed(Hn * 7 + T)
    R, M = Hn * G, Hn * G * T
    dz0, Wp = rnd(g, M, A0, scale=0.5), rnd(g, Q, A0, scale=0.2)
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    Wt, Kp = ops.pack_weight(dev(Wp), Q, A0, transposed=True)
    da, dq = torch.full((Hn * T, Q), 7.0, device="cuda"), torch.full((R, Q), 7.0, device="cuda")
    dU, dV = torch.full((Hn * T, A0), 7.0, device="cuda"), torch.full((R, A0), 7.0, device="cuda")
    parts = query("clsr_att_l0_bwd_x3_parts", Hn)
    C = query("clsr_dw_chunk_floats")
    ws = torch.full((parts * C,), 7.0, device="cuda")
    ddz0, da_, dq_ = dev(dz0), dev(a), dev(q)
    call(entry, ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0, ws)
    torch.cuda.synchronize()
    d = ddz0.double().cpu().view(Hn, G, T, A0)
    a4, q4 = da_.double().cpu().view(Hn, 1, T, Q), dq_.double().cpu().view(Hn, G, 1, Q)
    daq = d @ dev(Wp).double().cpu().t()
    close(da, (daq * q4).sum(1).reshape(Hn * T, Q), tol, "da")
    close(dq, (daq * a4).sum(2).reshape(R, Q), tol, "dq")
    close(dU, d.sum(1).reshape(Hn * T, A0), tols, "dU")
    close(dV, d.sum(2).reshape(R, A0), tols, "dV")
    dWp, _, full = chunks_to_matrix(ws, parts, Q, A0)
    aq = (a4 * q4).reshape(M, Q)
    close(dWp, aq.t() @ d.reshape(M, A0), tol, "dWp")
    # padded rows / columns of the written tiles are exact zeros (the batched reduction never reads them, but a NaN there
    # would mean an operand escaped its mask)
    nf, nz = (3 if Q <= 48 else 5), (3 if A0 <= 48 else 5)
    pad = full[: 16 * nf, : 16 * nz].clone()
    pad[:Q, :A0] = 0
    assert float(pad.abs().max()) == 0
    # against the exact kernel + the weight-gradient kernel it replaces, through the batched reduction
    da2, dq2, dU2, dV2 = torch.zeros_like(da), torch.zeros_like(dq), torch.zeros_like(dU), torch.zeros_like(dV)
    call("clsr_att_l0_bwd", ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da2, Q, dq2, Q, dU2, A0, dV2, A0)
    torch.cuda.synchronize()
    close(da, da2, tol, "da vs exact kernel")
    close(dV, dV2, tols, "dV vs exact kernel")
    out = torch.full((Q, A0), 3.0, device="cuda")
    tab = ops.dw_table(((ws.data_ptr(), out.data_ptr(), 0, 1.0, parts, Q, A0, A0, 0),), "cuda")
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    close(out, dWp, 1e-6, "dWp through clsr_dw_reduce_batch")
    # dU = NULL (G == 1 callers alias dU with dz0), determinism: bit-identical reruns
    da3, dq3, dV3, ws3 = torch.zeros_like(da), torch.zeros_like(dq), torch.zeros_like(dV), torch.zeros_like(ws)
    call(entry, ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da3, Q, dq3, Q, None, 0, dV3, A0, ws3)
    torch.cuda.synchronize()
    assert torch.equal(da3, da) and torch.equal(dq3, dq) and torch.equal(dV3, dV)
    t5 = lambda w: w.view(parts, C)[:, : 25 * 256].view(parts, 5, 5, 256)[:, :nf, :nz]     # the tiles the kernel writes
    assert torch.equal(t5(ws3), t5(ws))


@pytest.mark.parametrize("M,C1,C0", [(2000, 40, 80), (515, 40, 80), (300, 48, 80), (70, 16, 20), (4099, 40, 36),
                                     (40000, 40, 80), (100, 24, 48), (33, 8, 4)])
@pytest.mark.parametrize("entry,tol,tolz", [("clsr_att_l1_bwd_x3", 2e-4, 1e-4), ("clsr_att_l1_bwd_x6", 5e-6, 3e-6)])
def test_layer1_backward_x3_two_passes_with_weight_gradient(M, C1, C0, entry, tol, tolz):
    """x3: two bf16 pieces per operand; x6: three pieces (fp32 accuracy)"""
    assert query("clsr_att_l1_bwd_x3_supported", C1, C0) == 1 and query("clsr_att_l1_bwd_x3_supported", 44, C0) == 0
    g = torch.Generator().manual_seed(3 + M)
    z1, z0, ds = rnd(g, M, C1), rnd(g, M, C0), rnd(g, M)
    W1 = rnd(g, C0, C1, scale=0.3)
    sc1, sh1 = torch.rand(C1, generator=g, dtype=torch.float64) + 0.5, rnd(g, C1, scale=0.3)
    wo, coef1 = rnd(g, C1), rnd(g, 3 * C1, scale=0.5)
    sc0, sh0 = torch.rand(C0, generator=g, dtype=torch.float64) + 0.5, rnd(g, C0, scale=0.3)
    mean0, inv0 = rnd(g, C0, scale=0.1), torch.rand(C0, generator=g, dtype=torch.float64) + 0.5
    coef0 = rnd(g, 3 * C0, scale=0.5)
    # keep every pre-activation away from the ReLU kink: an element within rounding of it may take either branch
    for z, sc, sh in ((z1, sc1, sh1), (z0, sc0, sh0)):
        for _ in range(3):
            y = z.float().double() * sc.float().double() + sh.float().double()
            z[y.abs() < 1e-3] += 0.01
    Wt, Kp = ops.pack_weight(dev(W1), C0, C1, transposed=True)
    d = {k: dev(v) for k, v in dict(z1=z1, z0=z0, ds=ds, sc1=sc1, sh1=sh1, wo=wo, coef1=coef1, sc0=sc0, sh0=sh0, mean0=mean0,
                                    inv0=inv0, coef0=coef0).items()}
    parts = query("clsr_att_l1_bwd_x3_parts", M)
    C = query("clsr_dw_chunk_floats")
    st = torch.full((parts, 2, C0), 7.0, dtype=torch.float64, device="cuda")
    call(entry, d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], d["mean0"], d["inv0"], None, None, 0, None, st, M, C1, C0)
    ld0 = C0 + 4
    dz0 = torch.full((M, ld0), 7.0, device="cuda")
    ws = torch.full((parts * C,), 7.0, device="cuda")
    call(entry, d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], None, None, d["coef0"], dz0, ld0, ws, None, M, C1, C0)
    torch.cuda.synchronize()
    # float64 restatement on the fp32-rounded inputs
    f = {k: v.double().cpu() for k, v in d.items()}
    y1 = f["z1"] * f["sc1"] + f["sh1"]
    a1, a2, a3 = f["coef1"][:C1], f["coef1"][C1:2 * C1], f["coef1"][2 * C1:]
    x = torch.where(y1 > 0, (a1 * f["wo"]).float().double() * f["ds"][:, None], torch.zeros_like(y1)) + a2 * f["z1"] + a3
    dh0 = x @ dev(W1).double().cpu().t()
    y0 = f["z0"] * f["sc0"] + f["sh0"]
    safe = (y0.abs() > 1e-4)
    dy0 = torch.where(y0 > 0, dh0, torch.zeros_like(dh0))
    xhat = (f["z0"] - f["mean0"]) * f["inv0"]
    assert bool(safe.all())
    close(st.sum(0)[0], dy0.sum(0), tol, "sum dy0")
    close(st.sum(0)[1], (dy0 * xhat).sum(0), tol, "sum dy0 * xhat0")
    c1, c2, c3 = f["coef0"][:C0], f["coef0"][C0:2 * C0], f["coef0"][2 * C0:]
    exp = c1 * dy0 + c2 * f["z0"] + c3
    got = dz0[:, :C0].double().cpu()
    assert float(((got - exp).abs() * safe).max()) <= tolz * float(exp.abs().max()), "dz0"
    assert float((dz0[:, C0:] - 7.0).abs().max()) == 0
    x1 = torch.clamp(y0, min=0)
    dW1, db1, _ = chunks_to_matrix(ws, parts, C0, C1)
    close(dW1, x1.t() @ x, tolz, "dW1")
    close(db1, x.sum(0), tolz, "db1")
    # against the exact two-pass kernel
    p2 = query("clsr_att_l1_bwd_stats_parts", M)
    st2 = torch.zeros(p2, 2, C0, dtype=torch.float64, device="cuda")
    call("clsr_att_l1_bwd", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], d["mean0"], d["inv0"], None, None, 0, None, 0, st2, M, C1, C0)
    torch.cuda.synchronize()
    close(st.sum(0), st2.sum(0), tol, "stats vs exact kernel")
    # the reduction the step uses, weights and bias
    outW, outb = torch.full((C0, C1), 3.0, device="cuda"), torch.full((C1,), 3.0, device="cuda")
    tab = ops.dw_table(((ws.data_ptr(), outW.data_ptr(), outb.data_ptr(), 1.0, parts, C0, C1, C1, 0),), "cuda")
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    close(outW, dW1, 1e-6, "dW1 through clsr_dw_reduce_batch")
    close(outb, db1, 1e-6, "db1 through clsr_dw_reduce_batch")
    # determinism
    dz0b, wsb = torch.zeros_like(dz0), torch.zeros_like(ws)
    call(entry, d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], None, None, d["coef0"], dz0b, ld0, wsb, None, M, C1, C0)
    torch.cuda.synchronize()
    assert torch.equal(dz0b[:, :C0], dz0[:, :C0])


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 80, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 36), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (2100, 2, 33, 48, 80), (11, 8, 18, 80, 80), (5, 4, 20, 40, 80),
                                         (7, 5, 16, 80, 40), (3, 2, 8, 16, 16),   # packed / unpacked ragged tiles
                                         (37, 5, 50, 128, 80), (9, 3, 17, 96, 40)])   # K up to 128: the wide layers of configs[4]
@pytest.mark.parametrize("entry,tol", [("clsr_att_l0_fwd_x3", 5e-5), ("clsr_att_l0_fwd_x6", 2e-6)])
def test_layer0_forward_x3(Hn, G, T, Q, A0, entry, tol):
    """clsr_att_l0_fwd_x3: z0 = U[h,t] + V[r] + (a[h,t] * q[r]) . Wp with the product as split-bf16 sums: 2^-16 relative per
    term against float64, statistics consistent with the stored z0, and close to the exact kernel."""
    g = torch.Generator().manual_seed(Hn * 3 + T)
    R, M = Hn * G, Hn * G * T
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    U, V, Wp = rnd(g, Hn * T, A0), rnd(g, R, A0), rnd(g, Q, A0, scale=0.2)
    Wt, Kp = ops.pack_weight(dev(Wp), A0, Q)
    parts = query("clsr_att_l0_fwd_stats_parts", Hn)
    st = torch.full((parts, 2, A0), 7.0, dtype=torch.float64, device="cuda")
    z0 = torch.full((M, A0 + 4), 7.0, device="cuda")
    da, dq, dU, dV = dev(a), dev(q), dev(U), dev(V)
    call(entry, da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z0, A0 + 4, st, Hn, G, T, Q, A0)
    torch.cuda.synchronize()
    a4, q4 = da.double().cpu().view(Hn, 1, T, Q), dq.double().cpu().view(Hn, G, 1, Q)
    exp = ((a4 * q4) @ dev(Wp).double().cpu() + dU.double().cpu().view(Hn, 1, T, A0)
           + dV.double().cpu().view(Hn, G, 1, A0)).reshape(M, A0)
    close(z0[:, :A0], exp, tol, "z0")
    assert float((z0[:, A0:] - 7.0).abs().max()) == 0
    got = z0[:, :A0].double().cpu()
    close(st.sum(0)[0], got.sum(0), 1e-5, "column sums")
    close(st.sum(0)[1], (got * got).sum(0), 1e-5, "column sums of squares")
    z1 = torch.zeros(M, A0, device="cuda")
    call("clsr_att_l0_fwd", da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z1, A0, None, Hn, G, T, Q, A0)
    torch.cuda.synchronize()
    close(z0[:, :A0], z1, tol, "z0 vs the exact kernel")


@pytest.mark.parametrize("Hn,T,Dk,Q,A0,qh", [(37, 50, 40, 80, 80, 40), (64, 50, 40, 40, 80, 0), (9, 7, 24, 44, 36, 0),
                                             (5, 17, 8, 16, 16, 8), (2100, 5, 40, 80, 80, 40), (3, 50, 64, 48, 80, 24),
                                             (2, 16, 32, 80, 40, 48)])
@pytest.mark.parametrize("pieces", [2, 3])
def test_history_level_prologue_x3(Hn, T, Dk, Q, A0, qh, pieces):
    """clsr_att_hist_fwd_x3: a = keys . A, U = a . Wu + (a[:, :qh] * q_hist[h]) . Wp[:qh] in one launch of split-bf16
    products == float64 (2^-16 relative per product term, two chained products)."""
    assert query("clsr_att_hist_fwd_x3_supported", Dk, Q, A0, qh) == 1
    assert query("clsr_att_hist_fwd_x3_supported", Dk, 96, A0, qh) == 0
    g = torch.Generator().manual_seed(Hn + T + qh)
    keys, A, Wu = rnd(g, Hn * T, Dk), rnd(g, Dk, Q, scale=0.3), rnd(g, Q, A0, scale=0.3)
    Wp1, qhist = rnd(g, max(qh, 4), A0, scale=0.3), rnd(g, Hn, max(qh, 4))
    At, Kpa = ops.pack_weight(dev(A), Q, Dk)
    Wut, Kpu = ops.pack_weight(dev(Wu), A0, Q)
    Wpt, Kpp = ops.pack_weight(dev(Wp1), A0, max(qh, 4))
    dk, dqh = dev(keys), dev(qhist)
    a = torch.full((Hn * T, Q + 4), 7.0, device="cuda")
    U = torch.full((Hn * T, A0 + 4), 7.0, device="cuda")
    call("clsr_att_hist_fwd_x3", dk, Dk, At, Kpa, Wut, Kpu, Wpt if qh else None, Kpp if qh else 0, dqh if qh else None,
         max(qh, 4), Hn, T, Dk, Q, A0, qh, pieces, a, Q + 4, U, A0 + 4)
    torch.cuda.synchronize()
    ea = dk.double().cpu() @ dev(A).double().cpu()
    eu = ea @ dev(Wu).double().cpu()
    if qh:
        qrep = dqh.double().cpu()[:, :qh].repeat_interleave(T, 0)
        eu = eu + (ea[:, :qh] * qrep) @ dev(Wp1).double().cpu()[:qh]
    close(a[:, :Q], ea, 5e-5 if pieces == 2 else 1e-6, "a")       # (3 pieces: 2^-23 per product term, fp32 accumulation)
    close(U[:, :A0], eu, 1e-4 if pieces == 2 else 2e-6, "U")
    assert float((a[:, Q:] - 7.0).abs().max()) == 0 and float((U[:, A0:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("Hn,T,Dk,Q,A0,qh", [(37, 50, 40, 80, 80, 40), (64, 50, 40, 40, 80, 0), (9, 7, 24, 44, 40, 0),
                                             (5, 17, 8, 16, 16, 8), (2100, 5, 40, 80, 80, 40), (3, 50, 48, 48, 96, 24),
                                             (2, 16, 32, 80, 40, 48)])
@pytest.mark.parametrize("pieces", [2, 3])
def test_history_level_backward_x3(Hn, T, Dk, Q, A0, qh, pieces):
    """clsr_att_hist_bwd_x3: da = [(dU . Wp[:qh]^T) * q_hist | da] + dU . Wu^T, dq_hist += sum_t (dU . Wp[:qh]^T) * a,
    dkeys += da . A^T in one launch == float64 (two pieces: 2^-16 per product term; three: fp32 accuracy)."""
    tol = 1e-4 if pieces == 2 else 3e-6
    assert query("clsr_att_hist_bwd_x3_supported", Dk, Q, A0, qh) == 1
    g = torch.Generator().manual_seed(Hn + T + qh + 1)
    M = Hn * T
    dU, A, Wu = rnd(g, M, A0, scale=0.5), rnd(g, Dk, Q, scale=0.3), rnd(g, Q, A0, scale=0.3)
    q4 = max(qh, 4)
    Wp1, qhist, a = rnd(g, q4, A0, scale=0.3), rnd(g, Hn, q4), rnd(g, M, Q)
    da0, dqh0, dk0 = rnd(g, M, Q), rnd(g, Hn, q4), rnd(g, M, Dk)
    WuT, Kpu = ops.pack_weight(dev(Wu), Q, A0, transposed=True)
    WpT, Kpp = ops.pack_weight(dev(Wp1), q4, A0, transposed=True)
    AT, Kpa = ops.pack_weight(dev(A), Dk, Q, transposed=True)
    d_dU, d_a, d_qh = dev(dU), dev(a), dev(qhist)
    da = torch.full((M, Q + 4), 7.0, device="cuda")
    da[:, :Q] = dev(da0)
    if qh:
        da[:, :qh] = float("nan")          # (columns < qh are written, never read)
    dqh, dk = dev(dqh0).clone(), torch.full((M, Dk + 4), 7.0, device="cuda")
    dk[:, :Dk] = dev(dk0)
    call("clsr_att_hist_bwd_x3", d_dU, A0, WuT, Kpu, WpT if qh else None, Kpp if qh else 0, AT, Kpa, d_a, Q,
         d_qh if qh else None, q4, Hn, T, Dk, Q, A0, qh, pieces, da, Q + 4, dqh if qh else None, q4, dk, Dk + 4)
    torch.cuda.synchronize()
    f = lambda t: dev(t).double().cpu()
    r1 = f(dU) @ f(Wu).t()
    eda = f(da0) + r1
    if qh:
        r2 = f(dU) @ f(Wp1)[:qh].t()
        qrep = f(qhist)[:, :qh].repeat_interleave(T, 0)
        eda[:, :qh] = r2 * qrep + r1[:, :qh]
        edq = f(dqh0)[:, :qh] + (r2 * f(a)[:, :qh]).view(Hn, T, qh).sum(1)
        close(dqh[:, :qh], edq, tol, "dq_hist")
        if q4 > qh:
            assert torch.equal(dqh[:, qh:].cpu(), dev(dqh0)[:, qh:].cpu())
    close(da[:, :Q], eda, tol, "da")
    close(dk[:, :Dk], f(dk0) + eda @ f(A).t(), tol, "dkeys")
    assert float((da[:, Q:] - 7.0).abs().max()) == 0 and float((dk[:, Dk:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("M,C0,C1", [(2000, 80, 40), (515, 80, 40), (70, 16, 12), (4099, 40, 36), (40000, 80, 40),
                                     (100, 48, 24), (33, 8, 4), (300, 96, 48)])
def test_layer1_forward_three_pieces(M, C0, C1):
    """clsr_att_l1_fwd: z1 = relu(bn0(z0)) . W1 + b1 over three bf16 pieces per operand == float64 at fp32 accuracy, with
    the batch-norm sums of the stored z1."""
    assert query("clsr_att_l1_fwd_supported", C0, C1) == 1 and query("clsr_att_l1_fwd_supported", 84, C1) == 0
    g = torch.Generator().manual_seed(5 + M)
    z0, W1, b1 = rnd(g, M, C0), rnd(g, C0, C1, scale=0.3), rnd(g, C1)
    sc0, sh0 = torch.rand(C0, generator=g, dtype=torch.float64) + 0.5, rnd(g, C0, scale=0.3)
    Wt, Kp = ops.pack_weight(dev(W1), C1, C0)
    parts = query("clsr_att_l1_fwd_stats_parts", M)
    st = torch.full((parts, 2, C1), 7.0, dtype=torch.float64, device="cuda")
    z1 = torch.full((M, C1 + 4), 7.0, device="cuda")
    d0, dsc, dsh, db = dev(z0), dev(sc0), dev(sh0), dev(b1)
    call("clsr_att_l1_fwd", d0, C0, dsc, dsh, Wt, Kp, db, z1, C1 + 4, st, M, C0, C1)
    torch.cuda.synchronize()
    x1 = torch.clamp((d0 * dsc + dsh).double().cpu(), min=0)         # (the prologue in fp32, as the kernel takes it)
    exp = x1 @ dev(W1).double().cpu() + db.double().cpu()
    close(z1[:, :C1], exp, 2e-6, "z1")
    assert float((z1[:, C1:] - 7.0).abs().max()) == 0
    got = z1[:, :C1].double().cpu()
    close(st.sum(0)[0], got.sum(0), 1e-6, "column sums")
    close(st.sum(0)[1], (got * got).sum(0), 1e-6, "column sums of squares")
    z1b = torch.zeros(M, C1, device="cuda")
    call("clsr_att_l1_fwd", d0, C0, dsc, dsh, Wt, Kp, db, z1b, C1, None, M, C0, C1)
    torch.cuda.synchronize()
    assert torch.equal(z1b, z1[:, :C1].contiguous())


@pytest.mark.parametrize("M,K,N", [(2000, 128, 120), (515, 48, 40), (70, 16, 12), (4099, 40, 36), (40000, 128, 120),
                                   (33, 8, 4), (300, 96, 128), (1000, 128, 520), (300, 40, 260), (130, 128, 1536)])   # (N > 128: column blocks)
@pytest.mark.parametrize("pieces", [2, 3])
def test_projection_split_products(M, K, N, pieces):
    """clsr_proj_x3: Y = X . W + b into a column block of a wider tensor == float64 (2^-16 / 2^-23 per product term)."""
    assert query("clsr_proj_x3_supported", M, K, N) == 1 and query("clsr_proj_x3_supported", M, 132, N) == 0
    g = torch.Generator().manual_seed(M + K)
    X, W, b = rnd(g, M, K), rnd(g, K, N, scale=0.3), rnd(g, N)
    Wt, Kp = ops.pack_weight(dev(W), N, K)
    dX, db = dev(X), dev(b)
    Y = torch.full((M, N + 8), 7.0, device="cuda")
    call("clsr_proj_x3", dX, K, Wt, Kp, db, Y[:, 4:], N + 8, M, K, N, pieces)
    torch.cuda.synchronize()
    exp = dX.double().cpu() @ dev(W).double().cpu() + db.double().cpu()
    close(Y[:, 4:4 + N], exp, 5e-5 if pieces == 2 else 2e-6, "Y")
    assert float((Y[:, :4] - 7.0).abs().max()) == 0 and float((Y[:, 4 + N:] - 7.0).abs().max()) == 0
    Y2 = torch.zeros(M, N, device="cuda")
    call("clsr_proj_x3", dX, K, Wt, Kp, None, Y2, N, M, K, N, pieces)
    torch.cuda.synchronize()
    close(Y2, exp - db.double().cpu(), 5e-5 if pieces == 2 else 2e-6, "Y without bias")


@pytest.mark.parametrize("M,NX,D,H,tcol0", [(2000, 480, 40, 40, 360), (515, 480, 40, 40, 360), (70, 96, 8, 8, 40),
                                             (4099, 144, 16, 16, 96), (40000, 480, 40, 40, 360), (33, 512, 48, 40, 392)])
def test_encoder_tail_back_projections(M, NX, D, H, tcol0):
    """clsr_enc_back_x3: dhist += dPin . Wx^T and dTT = dPin[:, tcol0 : tcol0 + 3H] . Wt^T from one pass == float64."""
    H2, H3 = 2 * H, 3 * H
    assert query("clsr_enc_back_x3_supported", M, NX, D, H2, H3, tcol0) == 1
    g = torch.Generator().manual_seed(M + NX)
    dPin, Wx, Wt = rnd(g, M, NX, scale=0.3), rnd(g, D, NX, scale=0.3), rnd(g, H2, H3, scale=0.3)     # Wx: [D, NX] = W_x as stored
    dh0 = rnd(g, M, D)
    WxT, Kpx = ops.pack_weight(dev(Wx), D, NX, transposed=True)      # dhist = dPin . W_x^T: out = D, in = NX
    WtT, Kpt = ops.pack_weight(dev(Wt), H2, H3, transposed=True)
    dP = dev(dPin)
    dhist = torch.full((M, D + 4), 7.0, device="cuda")
    dhist[:, :D] = dev(dh0)
    dTT = torch.full((M, H2 + 4), 7.0, device="cuda")
    call("clsr_enc_back_x3", dP, NX, WxT, Kpx, WtT, Kpt, tcol0, dhist, D + 4, dTT, H2 + 4, M, NX, D, H2, H3)
    torch.cuda.synchronize()
    f = lambda t: dev(t).double().cpu()
    close(dhist[:, :D], f(dh0) + f(dPin) @ f(Wx).t(), 5e-5, "dhist")
    close(dTT[:, :H2], f(dPin)[:, tcol0:tcol0 + H3] @ f(Wt).t(), 5e-5, "dTT")
    assert float((dhist[:, D:] - 7.0).abs().max()) == 0 and float((dTT[:, H2:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(1000, 384, 384), (515, 136, 40), (70, 264, 132), (2000, 128, 120), (3000, 1536, 128), (33, 200, 8)])
@pytest.mark.parametrize("pieces", [2, 3])
def test_projection_wide_k_in_slabs(M, K, N, pieces):
    """clsr_proj_x3_wide: K > 128 as slabs of 128 input features, the later slabs accumulating into Y == float64."""
    assert query("clsr_proj_x3_wide_supported", M, K, N) == 1 and query("clsr_proj_x3_wide_supported", M, K + 4, N) == 0
    g = torch.Generator().manual_seed(M + K)
    X, W, b = rnd(g, M, K), rnd(g, K, N, scale=0.2), rnd(g, N)
    Wt, Kp = ops.pack_weight(dev(W), N, K)
    dX, db = dev(X), dev(b)
    Y = torch.full((M, N + 8), 7.0, device="cuda")
    call("clsr_proj_x3_wide", dX, K, Wt, Kp, db, Y[:, 4:], N + 8, M, K, N, pieces, 0)
    torch.cuda.synchronize()
    exp = dX.double().cpu() @ dev(W).double().cpu() + db.double().cpu()
    close(Y[:, 4:4 + N], exp, 5e-5 if pieces == 2 else 2e-6, "Y")
    assert float((Y[:, :4] - 7.0).abs().max()) == 0 and float((Y[:, 4 + N:] - 7.0).abs().max()) == 0
    call("clsr_proj_x3_wide", dX, K, Wt, Kp, None, Y[:, 4:], N + 8, M, K, N, pieces, 1)       # Y += X . W
    torch.cuda.synchronize()
    close(Y[:, 4:4 + N], 2 * exp - db.double().cpu(), 5e-5 if pieces == 2 else 2e-6, "Y accumulated")
    assert float((Y[:, :4] - 7.0).abs().max()) == 0 and float((Y[:, 4 + N:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("Hn,T,D,H", [(64, 50, 40, 40), (7, 13, 8, 8), (33, 250, 40, 40), (5, 7, 32, 24)])
@pytest.mark.parametrize("pieces", [2, 3])
def test_time_gate_projection_with_the_time_features_in_its_prologue(Hn, T, D, H, pieces):
    """clsr_proj_x3_tt == clsr_t4_time_inputs_fwd2 (the [hist | 0 | TT] image) followed by clsr_proj_x3 on that image, and ==
    float64 (reference rnn_cell_implement.py:200-236)."""
    M, Dp, N = Hn * T, 16 * ((D + 15) // 16), 3 * H
    K = Dp + 2 * H
    assert query("clsr_proj_x3_tt_supported", M, D, Dp, H, N) == 1
    g = torch.Generator().manual_seed(Hn + T)
    hist, W, b = dev(rnd(g, M, D)), dev(rnd(g, K, N, scale=0.2)), dev(rnd(g, N))
    W[D:Dp] = 0
    tnow, tfirst = dev(torch.rand(Hn, T, generator=g, dtype=torch.float64) * 5), dev(torch.rand(Hn, T, generator=g, dtype=torch.float64) * 5)
    w1, b1, w2, b2 = dev(rnd(g, H)), dev(rnd(g, H)), dev(rnd(g, H)), dev(rnd(g, H))
    Wt, Kp = ops.pack_weight(W, N, K)
    TT, XT = torch.zeros(M, 2 * H, device="cuda"), torch.zeros(M, K, device="cuda")
    call("clsr_t4_time_inputs_fwd2", tnow, tfirst, T, w1, b1, w2, b2, Hn, T, H, TT, hist, D, XT, K, Dp)
    Y0 = torch.full((M, N + 8), 7.0, device="cuda")
    call("clsr_proj_x3", XT, K, Wt, Kp, b, Y0[:, 4:], N + 8, M, K, N, pieces)
    Y1 = torch.full((M, N + 8), 7.0, device="cuda")
    call("clsr_proj_x3_tt", hist, D, tnow, tfirst, T, T, w1, b1, w2, b2, H, Dp, Wt, Kp, b, Y1[:, 4:], N + 8, M, N, pieces)
    torch.cuda.synchronize()
    tol = 5e-5 if pieces == 2 else 2e-6
    close(Y1[:, 4:4 + N], Y0[:, 4:4 + N], tol, "fused vs two launches")
    assert float((Y1[:, :4] - 7.0).abs().max()) == 0 and float((Y1[:, 4 + N:] - 7.0).abs().max()) == 0
    tn, tf = tnow.double().cpu().reshape(M, 1), tfirst.double().cpu().reshape(M, 1)
    X = torch.cat([hist.double().cpu(), torch.zeros(M, Dp - D, dtype=torch.float64),
                   torch.tanh(tn * w1.double().cpu() + b1.double().cpu()), torch.tanh(tf * w2.double().cpu() + b2.double().cpu())], 1)
    close(Y1[:, 4:4 + N], X @ W.double().cpu() + b.double().cpu(), tol, "fused vs float64")
