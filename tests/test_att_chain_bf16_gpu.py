"""Speed-mode (bf16) instantiations of the attention chain kernels (csrc/attl0fwd.hip, attl1fwd.hip, attbwdx3.hip with ONE
bf16 piece per operand and bf16 storage of z0 / z1 / dz0: clsr_att_l0_fwd_x1_h, clsr_att_l1_fwd_x1_h, clsr_att_l1_bwd_x1_h,
clsr_att_l0_bwd_x1_h) against float64 restatements computed on the SAME bf16-rounded operands, so that what is left is fp32
accumulation order and the bf16 rounding of a stored output (2^-8 relative).  Reference: clsr.py:343-381,
base_model.py:627-708 at bf16 product precision."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402

BF = torch.bfloat16


def dev(x, dtype=torch.float32):
    return torch.as_tensor(x).to(dtype).cuda().contiguous()


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float64) * scale


def r16(t):
    """fp32 tensor -> rounded to bf16 (RNE) -> float64 on the host"""
    return t.float().to(BF).double().cpu()


def close(got, exp, rel, name, rtol=0.0):
    got, exp = got.detach().double().cpu(), exp.detach().double().cpu()
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    scale = float(exp.abs().max()) + 1e-30
    excess = float(((got - exp).abs() - rtol * exp.abs()).max())
    assert excess <= rel * scale, "%s: excess err %.3e at scale %.3e (allowed %.1e of it + %.1e relative)" % (
        name, excess, scale, rel, rtol)


def chunks_to_matrix(ws, parts, K, N):
    C = query("clsr_dw_chunk_floats")
    w = ws.view(parts, C).double().sum(0)
    full = w[: 25 * 256].view(5, 5, 16, 16).permute(0, 2, 1, 3).reshape(80, 80)
    return full[:K, :N].cpu(), w[25 * 256: 25 * 256 + N].cpu()


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 80, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 40), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (2100, 2, 33, 48, 80), (7, 5, 16, 80, 40), (3, 2, 8, 16, 16)])
def test_layer0_forward_one_piece_bf16_out(Hn, G, T, Q, A0):
    g = torch.Generator().manual_seed(Hn * 3 + T)
    R, M = Hn * G, Hn * G * T
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    U, V, Wp = rnd(g, Hn * T, A0), rnd(g, R, A0), rnd(g, Q, A0, scale=0.2)
    Wt, Kp = ops.pack_weight(dev(Wp), A0, Q)
    parts = query("clsr_att_l0_fwd_stats_parts", Hn)
    st = torch.full((parts, 2, A0), 7.0, dtype=torch.float64, device="cuda")
    ld = A0 + 8
    z0 = torch.full((M, ld), 7.0, dtype=BF, device="cuda")
    da, dq, dU, dV = dev(a), dev(q), dev(U), dev(V)
    call("clsr_att_l0_fwd_x1_h", da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z0, ld, st, Hn, G, T, Q, A0)
    torch.cuda.synchronize()
    x = r16(da.view(Hn, 1, T, Q) * dq.view(Hn, G, 1, Q))             # the product a * q in fp32, then one bf16 piece
    exp = (x @ r16(dev(Wp)) + dU.double().cpu().view(Hn, 1, T, A0) + dV.double().cpu().view(Hn, G, 1, A0)).reshape(M, A0)
    close(z0[:, :A0], exp, 1e-5, "z0", rtol=2.0 ** -8)
    assert float((z0[:, A0:].float() - 7.0).abs().max()) == 0
    got = z0[:, :A0].double().cpu()
    close(st.sum(0)[0], got.sum(0), 1e-5, "column sums (of the stored values)")
    close(st.sum(0)[1], (got * got).sum(0), 1e-5, "column sums of squares")
    z0b = torch.zeros(M, A0, dtype=BF, device="cuda")
    call("clsr_att_l0_fwd_x1_h", da, Q, dq, Q, Wt, Kp, dU, A0, dV, A0, z0b, A0, None, Hn, G, T, Q, A0)
    torch.cuda.synchronize()
    assert torch.equal(z0b, z0[:, :A0].contiguous())


@pytest.mark.parametrize("M,C0,C1", [(2000, 80, 40), (515, 80, 40), (70, 16, 8), (4099, 40, 32), (40000, 80, 40),
                                     (100, 48, 24), (300, 96, 48)])
def test_layer1_forward_one_piece_bf16(M, C0, C1):
    g = torch.Generator().manual_seed(5 + M)
    z0, W1, b1 = rnd(g, M, C0), rnd(g, C0, C1, scale=0.3), rnd(g, C1)
    sc0, sh0 = torch.rand(C0, generator=g, dtype=torch.float64) + 0.5, rnd(g, C0, scale=0.3)
    Wt, Kp = ops.pack_weight(dev(W1), C1, C0)
    parts = query("clsr_att_l1_fwd_stats_parts", M)
    st = torch.full((parts, 2, C1), 7.0, dtype=torch.float64, device="cuda")
    ld1 = C1 + 8
    z1 = torch.full((M, ld1), 7.0, dtype=BF, device="cuda")
    d0, dsc, dsh, db = dev(z0).to(BF), dev(sc0), dev(sh0), dev(b1)
    call("clsr_att_l1_fwd_x1_h", d0, C0, dsc, dsh, Wt, Kp, db, z1, ld1, st, M, C0, C1)
    torch.cuda.synchronize()
    x1 = r16(torch.clamp(d0.float() * dsc + dsh, min=0))             # the prologue in fp32, as the kernel takes it
    exp = x1 @ r16(dev(W1)) + db.double().cpu()
    # (an element of the prologue within fp32 rounding of a bf16 tie may round the other way in the kernel: 2^-9 of ONE term)
    close(z1[:, :C1], exp, 1e-3, "z1", rtol=2.0 ** -8)
    assert float((z1[:, C1:].float() - 7.0).abs().max()) == 0
    got = z1[:, :C1].double().cpu()
    close(st.sum(0)[0], got.sum(0), 1e-5, "column sums (of the stored values)")
    close(st.sum(0)[1], (got * got).sum(0), 1e-5, "column sums of squares")


@pytest.mark.parametrize("M,C1,C0", [(2000, 40, 80), (515, 40, 80), (300, 48, 80), (70, 16, 24), (4099, 40, 40),
                                     (40000, 40, 80), (100, 24, 48)])
def test_layer1_backward_one_piece_bf16(M, C1, C0):
    g = torch.Generator().manual_seed(3 + M)
    z1, z0, ds = rnd(g, M, C1), rnd(g, M, C0), rnd(g, M)
    W1 = rnd(g, C0, C1, scale=0.3)
    sc1, sh1 = torch.rand(C1, generator=g, dtype=torch.float64) + 0.5, rnd(g, C1, scale=0.3)
    wo, coef1 = rnd(g, C1), rnd(g, 3 * C1, scale=0.5)
    sc0, sh0 = torch.rand(C0, generator=g, dtype=torch.float64) + 0.5, rnd(g, C0, scale=0.3)
    mean0, inv0 = rnd(g, C0, scale=0.1), torch.rand(C0, generator=g, dtype=torch.float64) + 0.5
    coef0 = rnd(g, 3 * C0, scale=0.5)
    z1, z0 = z1.to(BF).double(), z0.to(BF).double()
    # keep every pre-activation away from the ReLU kink (steps of a bf16 ulp)
    for z, sc, sh in ((z1, sc1, sh1), (z0, sc0, sh0)):
        for _ in range(40):
            y = z * sc.float().double() + sh.float().double()
            near = y.abs() < 1e-3
            if not bool(near.any()):
                break
            z[near] = torch.where(z[near] == 0, torch.full_like(z[near], 0.5), z[near] * 1.125)
            z.copy_(z.float().to(BF).double())
    Wt, Kp = ops.pack_weight(dev(W1), C0, C1, transposed=True)
    d = {k: dev(v) for k, v in dict(ds=ds, sc1=sc1, sh1=sh1, wo=wo, coef1=coef1, sc0=sc0, sh0=sh0, mean0=mean0, inv0=inv0,
                                    coef0=coef0).items()}
    d["z1"], d["z0"] = dev(z1).to(BF), dev(z0).to(BF)
    assert torch.equal(d["z1"].double().cpu(), z1) and torch.equal(d["z0"].double().cpu(), z0)
    parts = query("clsr_att_l1_bwd_x3_parts", M)
    C = query("clsr_dw_chunk_floats")
    st = torch.full((parts, 2, C0), 7.0, dtype=torch.float64, device="cuda")
    call("clsr_att_l1_bwd_x1_h", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], d["mean0"], d["inv0"], None, None, 0, None, st, M, C1, C0)
    ld0 = C0 + 8
    dz0 = torch.full((M, ld0), 7.0, dtype=BF, device="cuda")
    ws = torch.full((parts * C,), 7.0, device="cuda")
    call("clsr_att_l1_bwd_x1_h", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], None, None, d["coef0"], dz0, ld0, ws, None, M, C1, C0)
    torch.cuda.synchronize()
    f = {k: v.double().cpu() for k, v in d.items()}
    y1 = f["z1"] * f["sc1"] + f["sh1"]
    a1, a2, a3 = f["coef1"][:C1], f["coef1"][C1:2 * C1], f["coef1"][2 * C1:]
    x = torch.where(y1 > 0, (a1 * f["wo"]).float().double() * f["ds"][:, None], torch.zeros_like(y1)) + a2 * f["z1"] + a3
    xr = x.float().to(BF).double()                                    # one bf16 piece
    dh0 = xr @ r16(dev(W1)).t()
    y0 = f["z0"] * f["sc0"] + f["sh0"]
    assert bool((y0.abs() > 1e-4).all())
    dy0 = torch.where(y0 > 0, dh0, torch.zeros_like(dh0))
    xhat = (f["z0"] - f["mean0"]) * f["inv0"]
    # (an element of x within fp32 rounding of a bf16 tie may round the other way in the kernel: 2^-9 of ONE term)
    close(st.sum(0)[0], dy0.sum(0), 1e-3, "sum dy0")
    close(st.sum(0)[1], (dy0 * xhat).sum(0), 1e-3, "sum dy0 * xhat0")
    c1, c2, c3 = f["coef0"][:C0], f["coef0"][C0:2 * C0], f["coef0"][2 * C0:]
    exp = c1 * dy0 + c2 * f["z0"] + c3
    close(dz0[:, :C0], exp, 2e-3, "dz0", rtol=2.0 ** -8)
    assert float((dz0[:, C0:].float() - 7.0).abs().max()) == 0
    x1 = torch.clamp(y0, min=0).float().to(BF).double()
    dW1, db1 = chunks_to_matrix(ws, parts, C0, C1)
    close(dW1, x1.t() @ xr, 1e-3, "dW1")
    close(db1, xr.sum(0), 1e-3, "db1")
    # determinism
    dz0b, wsb = torch.zeros_like(dz0), torch.zeros_like(ws)
    call("clsr_att_l1_bwd_x1_h", d["z1"], C1, d["ds"], d["sc1"], d["sh1"], d["wo"], d["coef1"], Wt, Kp, d["z0"], C0, d["sc0"],
         d["sh0"], None, None, d["coef0"], dz0b, ld0, wsb, None, M, C1, C0)
    torch.cuda.synchronize()
    assert torch.equal(dz0b[:, :C0], dz0[:, :C0])


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 40, 80), (64, 1, 50, 40, 80), (9, 8, 7, 44, 40), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (130, 2, 33, 48, 80), (2100, 2, 5, 40, 80), (5, 5, 50, 8, 16)])
def test_layer0_backward_one_piece_bf16(Hn, G, T, Q, A0):
    g = torch.Generator().manual_seed(Hn * 7 + T)
    R, M = Hn * G, Hn * G * T
    dz0, Wp = rnd(g, M, A0, scale=0.5), rnd(g, Q, A0, scale=0.2)
    a, q = rnd(g, Hn * T, Q), rnd(g, R, Q)
    Wt, Kp = ops.pack_weight(dev(Wp), Q, A0, transposed=True)
    da, dq = torch.full((Hn * T, Q), 7.0, device="cuda"), torch.full((R, Q), 7.0, device="cuda")
    dU, dV = torch.full((Hn * T, A0), 7.0, device="cuda"), torch.full((R, A0), 7.0, device="cuda")
    parts = query("clsr_att_l0_bwd_x1_h_parts", Hn)
    C = query("clsr_dw_chunk_floats")
    ws = torch.full((parts * C,), 7.0, device="cuda")
    ddz0, da_, dq_ = dev(dz0).to(BF), dev(a), dev(q)
    call("clsr_att_l0_bwd_x1_h", ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0, ws)
    torch.cuda.synchronize()
    d = ddz0.double().cpu().view(Hn, G, T, A0)
    a4, q4 = da_.double().cpu().view(Hn, 1, T, Q), dq_.double().cpu().view(Hn, G, 1, Q)
    daq = d @ r16(dev(Wp)).t()                                        # dz0 is bf16 already: exact in one piece
    close(da, (daq * q4).sum(1).reshape(Hn * T, Q), 1e-5, "da")
    close(dq, (daq * a4).sum(2).reshape(R, Q), 1e-5, "dq")
    close(dU, d.sum(1).reshape(Hn * T, A0), 1e-5, "dU")
    close(dV, d.sum(2).reshape(R, A0), 1e-5, "dV")
    dWp, _ = chunks_to_matrix(ws, parts, Q, A0)
    aq = r16(da_.view(Hn, 1, T, Q) * dq_.view(Hn, G, 1, Q)).reshape(M, Q)
    close(dWp, aq.t() @ d.reshape(M, A0), 1e-5, "dWp")
    da3, dq3, dV3 = torch.zeros_like(da), torch.zeros_like(dq), torch.zeros_like(dV)
    ws3 = torch.zeros_like(ws)
    call("clsr_att_l0_bwd_x1_h", ddz0, A0, Wt, Kp, da_, Q, dq_, Q, Hn, G, T, Q, A0, da3, Q, dq3, Q, None, 0, dV3, A0, ws3)
    torch.cuda.synchronize()
    assert torch.equal(da3, da) and torch.equal(dq3, dq) and torch.equal(dV3, dV)
