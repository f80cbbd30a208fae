"""GPU tests of the bf16 SPEED mode of the attention block (clsr_amd/csrc/hgemm.hip, CLSRNet(precision="bf16")).

Kernel level: every hgemm variant against a float64 torch restatement computed on the SAME bf16-rounded operands
(so the only differences are fp32 accumulation order and the final bf16 rounding of the output: 2^-8 relative).
Step level: the whole training step in bf16 mode against the float64 oracle at the north_star bar
(logits within 1e-3) with the looser gradient tolerances bf16 activations allow, and against the fp32 mode."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.net import CLSRNet  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _pack_h(W, out_f, in_f, transposed=False):
    """bf16 image of a [in, out] weight block (``transposed``: the block is given as [out, in])."""
    Kp = ops.query("clsr_hgemm_kp", in_f)
    buf = torch.zeros(32 * ((out_f + 31) // 32) * Kp, dtype=BF, device=DEV)
    d = ops.pack_desc(W, out_f, in_f, buf, Kp, transposed=transposed)
    tbl, n, mx = ops.pack_table([d], torch.device(DEV))
    ops.call("clsr_pack_batch_bf16", tbl, n, mx)
    torch.cuda.synchronize()
    return buf, Kp, (d, tbl)


def _r(t):
    """round to bf16, back to float64"""
    return t.to(BF).double()


def _close(got, exp, rtol, atol, name):
    got, exp = got.double().cpu().reshape(-1), exp.double().cpu().reshape(-1)
    err = (got - exp).abs()
    excess = float((err - (atol + rtol * exp.abs())).max())
    assert excess <= 0, "%s: max abs err %.3e (max |exp| %.3e)" % (name, float(err.max()), float(exp.abs().max()))


def _stats(st, parts, N):
    s = st[: parts * 2 * N].view(parts, 2, N).sum(0)
    return s[0], s[1]


@pytest.mark.parametrize("Hn,G,T,Q,N", [(37, 5, 10, 80, 80), (64, 1, 50, 40, 80), (9, 3, 7, 256, 136), (5, 5, 50, 80, 40)])
def test_hgemm_mul_uv(Hn, G, T, Q, N):
    g = torch.Generator().manual_seed(1)
    R, M = Hn * G, Hn * G * T
    a = torch.randn(Hn * T, Q, generator=g).to(DEV)
    q = torch.randn(R, Q, generator=g).to(DEV)
    U = torch.randn(Hn * T, N, generator=g).to(DEV)
    V = torch.randn(R, N, generator=g).to(DEV)
    W = (torch.randn(Q, N, generator=g) * 0.2).to(DEV)
    Wt, Kp, keep = _pack_h(W, N, Q)
    Y = torch.zeros(M, N, dtype=BF, device=DEV)
    parts = ops.query("clsr_hgemm_stats_parts", M)
    st = torch.zeros(parts * 2 * N, dtype=torch.float64, device=DEV)
    ops.call("clsr_hgemm_mul_uv", a, Q, T, G, q, Q, Wt, Kp, U, N, V, N, Y, N, st, M, Q, N)
    torch.cuda.synchronize()
    r = torch.arange(M, device=DEV) // T
    xr = (r // G) * T + torch.arange(M, device=DEV) % T
    prod = _r(a[xr] * q[r])
    exp = prod @ _r(W) + U[xr].double() + V[r].double()
    _close(Y, exp, 2.0 ** -8, 1e-3, "Y")
    s, sq = _stats(st, parts, N)
    Yd = Y.double()
    _close(s, Yd.sum(0), 1e-5, 1e-3, "column sums of the stored values")
    _close(sq, (Yd * Yd).sum(0), 1e-5, 1e-3, "column sums of squares")


@pytest.mark.parametrize("Hn,G,T,Q,N", [(37, 5, 10, 80, 80), (9, 3, 50, 40, 80), (5, 8, 17, 96, 72), (130, 2, 33, 24, 40),
                                        (41, 5, 50, 80, 80), (6, 8, 18, 80, 80), (7, 4, 16, 40, 40), (3, 2, 5, 16, 24)])
def test_hgemm_layer0_one_wave_per_group(Hn, G, T, Q, N):
    """clsr_hgemm_l0_group == clsr_hgemm_mul_uv (same packed weights, same rounding points): z0 bit for bit, the
    batch-norm column sums to fp32 summation noise."""
    assert ops.query("clsr_hgemm_l0_group_supported", G, Q, N) == 1 and ops.query("clsr_hgemm_l0_group_supported", 1, Q, N) == 0
    g = torch.Generator().manual_seed(Hn + T)
    R, M = Hn * G, Hn * G * T
    a, q = torch.randn(Hn * T, Q, generator=g).to(DEV), torch.randn(R, Q, generator=g).to(DEV)
    U, V = torch.randn(Hn * T, N, generator=g).to(DEV), torch.randn(R, N, generator=g).to(DEV)
    W = (torch.randn(Q, N, generator=g) * 0.2).to(DEV)
    Wt, Kp, keep = _pack_h(W, N, Q)
    z_ref = torch.zeros(M, N, dtype=BF, device=DEV)
    p_ref = ops.query("clsr_hgemm_stats_parts", M)
    st_ref = torch.zeros(p_ref * 2 * N, dtype=torch.float64, device=DEV)
    ops.call("clsr_hgemm_mul_uv", a, Q, T, G, q, Q, Wt, Kp, U, N, V, N, z_ref, N, st_ref, M, Q, N)
    z = torch.full((M, N + 8), 3.0, dtype=BF, device=DEV)
    parts = ops.query("clsr_hgemm_l0_group_stats_parts", Hn)
    st = torch.full((parts * 2 * N,), 7.0, dtype=torch.float64, device=DEV)
    ops.call("clsr_hgemm_l0_group", a, Q, q, Q, Wt, Kp, U, N, V, N, z, N + 8, st, Hn, G, T, Q, N)
    z2 = torch.zeros(M, N, dtype=BF, device=DEV)
    ops.call("clsr_hgemm_l0_group", a, Q, q, Q, Wt, Kp, U, N, V, N, z2, N, None, Hn, G, T, Q, N)
    torch.cuda.synchronize()
    assert torch.equal(z[:, :N], z_ref) and torch.equal(z2, z_ref)
    assert float((z[:, N:].float() - 3.0).abs().max()) == 0
    s1, s2 = _stats(st, parts, N)
    r1, r2 = _stats(st_ref, p_ref, N)
    _close(s1, r1, 1e-6, 1e-3, "column sums")
    _close(s2, r2, 1e-6, 1e-3, "column sums of squares")


@pytest.mark.parametrize("M,K,N,aff", [(1000, 80, 40, True), (333, 80, 80, False), (4097, 40, 80, True), (50, 136, 264, True)])
def test_hgemm_affine_relu_prologue(M, K, N, aff):
    g = torch.Generator().manual_seed(2)
    X = torch.randn(M, K, generator=g).to(DEV).to(BF)
    W = (torch.randn(K, N, generator=g) * 0.2).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(DEV), torch.randn(K, generator=g).to(DEV)
    Wt, Kp, keep = _pack_h(W, N, K)
    Y = torch.zeros(M, N, dtype=BF, device=DEV)
    parts = ops.query("clsr_hgemm_stats_parts", M)
    st = torch.zeros(parts * 2 * N, dtype=torch.float64, device=DEV)
    ops.call("clsr_hgemm", X, K, sc if aff else None, sh if aff else None, 1, Wt, Kp, bias, Y, N, st, M, K, N)
    torch.cuda.synchronize()
    x = X.double()
    if aff:
        x = _r(torch.relu(X.float() * sc + sh))
    exp = x @ _r(W) + bias.double()
    _close(Y, exp, 2.0 ** -8, 1e-3, "Y")
    s, sq = _stats(st, parts, N)
    _close(s, Y.double().sum(0), 1e-5, 1e-3, "sums")
    _close(sq, (Y.double() ** 2).sum(0), 1e-5, 1e-3, "sums of squares")


@pytest.mark.parametrize("M,K,N,ldx,acc", [(3000, 480, 40, 480, 1), (777, 120, 80, 480, 0), (1000, 80, 80, 80, 1),
                                           (130, 264, 136, 264, 0), (50, 40, 40, 40, 1)])
def test_hgemm_fp32_in_out_backward_product(M, K, N, ldx, acc):
    """clsr_hgemm_f32: fp32 X (a column slice of a wider matrix), fp32 Y (=|+=), bf16 MFMA; any K (ring of k-tiles)."""
    g = torch.Generator().manual_seed(M + K)
    Xw = torch.randn(M, ldx, generator=g).to(DEV)
    X = Xw[:, :K] if ldx == K else Xw[:, 8:8 + K]
    W = (torch.randn(K, N, generator=g) * 0.2).to(DEV)
    Wt, Kp, keep = _pack_h(W, N, K)
    Y0 = torch.randn(M, N, generator=g).to(DEV)
    Y = Y0.clone()
    ops.call("clsr_hgemm_f32", X, ldx, Wt, Kp, Y, N, acc, M, K, N)
    torch.cuda.synchronize()
    exp = _r(X) @ _r(W) + (Y0.double() if acc else 0.0)
    _close(Y, exp, 1e-4, 2e-3, "Y")
    # the same product from a bf16 X (gradients the producer already stored as bf16): identical results
    Xh = Xw.to(BF)
    Xs = Xh[:, :K] if ldx == K else Xh[:, 8:8 + K]
    Y2 = Y0.clone()
    ops.call("clsr_hgemm_hf32", Xs, ldx, Wt, Kp, Y2, N, acc, M, K, N)
    torch.cuda.synchronize()
    assert torch.equal(Y2, Y)


@pytest.mark.parametrize("M,C1,C0", [(2000, 40, 80), (515, 40, 80), (300, 80, 136)])
def test_hgemm_attention_layer1_backward(M, C1, C0):
    """dz1 recomputed from (z1, ds) -> dh0 = dz1 . W1^T -> ReLU / batch-norm backward of layer 0: the statistics
    pass and the apply pass against the same formulas in float64."""
    g = torch.Generator().manual_seed(3)
    z1 = torch.randn(M, C1, generator=g).to(DEV).to(BF)
    z0 = torch.randn(M, C0, generator=g).to(DEV).to(BF)
    ds = torch.randn(M, generator=g).to(DEV)
    W1 = (torch.randn(C0, C1, generator=g) * 0.3).to(DEV)          # layer 1 weight [in = C0, out = C1]
    sc1, sh1 = (torch.rand(C1, generator=g) + 0.5).to(DEV), (torch.randn(C1, generator=g) * 0.3).to(DEV)
    wo = torch.randn(C1, generator=g).to(DEV)
    coef1 = torch.randn(3 * C1, generator=g).to(DEV) * 0.5
    sc0, sh0 = (torch.rand(C0, generator=g) + 0.5).to(DEV), (torch.randn(C0, generator=g) * 0.3).to(DEV)
    mean0, inv0 = torch.randn(C0, generator=g).to(DEV) * 0.1, (torch.rand(C0, generator=g) + 0.5).to(DEV)
    coef0 = torch.randn(3 * C0, generator=g).to(DEV) * 0.5
    Wt, Kp, keep = _pack_h(W1, C0, C1, transposed=True)      # dh0 = dz1 . W1^T: out = C0, in = C1; W1 is [out, in]
    parts = ops.query("clsr_hgemm_stats_parts", M)
    st = torch.zeros(parts * 2 * C0, dtype=torch.float64, device=DEV)
    ops.call("clsr_hgemm_att_l1_bwd", z1, C1, ds, sc1, sh1, wo, coef1, Wt, Kp, z0, C0, sc0, sh0, mean0, inv0, None,
             None, 0, None, 0, st, M, C1, C0)
    dz1 = torch.zeros(M, C1, dtype=BF, device=DEV)
    dz0 = torch.zeros(M, C0, dtype=BF, device=DEV)
    ops.call("clsr_hgemm_att_l1_bwd", z1, C1, ds, sc1, sh1, wo, coef1, Wt, Kp, z0, C0, sc0, sh0, None, None, coef0,
             dz1, C1, dz0, C0, None, M, C1, C0)
    torch.cuda.synchronize()
    z1f, z0f = z1.float(), z0.float()
    y1 = z1f * sc1 + sh1
    a1, a2, a3 = coef1[:C1], coef1[C1:2 * C1], coef1[2 * C1:]
    x = torch.where(y1 > 0, (a1 * wo) * ds[:, None], torch.zeros_like(y1)) + a2 * z1f + a3
    _close(dz1, x, 2.0 ** -8, 1e-3, "dz1")
    dh0 = _r(x) @ _r(W1).t()
    y0 = z0f * sc0 + sh0
    dy0 = torch.where(y0 > 0, dh0, torch.zeros_like(dh0))
    xhat = ((z0f - mean0) * inv0).double()
    s, sq = _stats(st, parts, C0)
    scale = float(dy0.abs().sum(0).max())
    _close(s, dy0.sum(0), 1e-4, 1e-5 * scale, "sum dy0")
    _close(sq, (dy0 * xhat).sum(0), 1e-4, 1e-5 * scale, "sum dy0 * xhat0")
    c1, c2, c3 = coef0[:C0].double(), coef0[C0:2 * C0].double(), coef0[2 * C0:].double()
    _close(dz0, c1 * dy0 + c2 * z0.double() + c3, 2.0 ** -8, 2e-3, "dz0")


@pytest.mark.parametrize("Hn,G,T,Q,A0", [(37, 5, 50, 80, 80), (64, 1, 50, 40, 80), (9, 8, 7, 96, 96), (6, 3, 17, 24, 40),
                                         (1, 5, 1, 80, 80), (130, 2, 33, 48, 88)])
def test_fused_layer0_backward_reductions(Hn, G, T, Q, A0):
    """clsr_att_l0_bwd_h: da, dq, dU, dV from ONE pass over dz0 (bf16); daq = dz0 . Wp^T never stored.  Against float64 on
    the bf16-rounded operands, and against the three-kernel path it replaces (which rounds daq to bf16)."""
    assert ops.query("clsr_att_l0_bwd_h_supported", G, Q, A0) == 1
    assert ops.query("clsr_att_l0_bwd_h_supported", 9, Q, A0) == 0 and ops.query("clsr_att_l0_bwd_h_supported", G, 128, A0) == 0
    g = torch.Generator().manual_seed(Hn * 7 + T)
    R, M = Hn * G, Hn * G * T
    dz0 = (torch.randn(M, A0, generator=g) * 0.5).to(DEV).to(BF)
    Wp = (torch.randn(Q, A0, generator=g) * 0.2).to(DEV)
    a = torch.randn(Hn * T, Q, generator=g).to(DEV)
    q = torch.randn(R, Q, generator=g).to(DEV)
    Wt, Kp, keep = _pack_h(Wp, Q, A0, transposed=True)
    Wu = (torch.randn(Q, A0, generator=g) * 0.2).to(DEV)
    Wut, Kpu, keep_u = _pack_h(Wu, Q, A0, transposed=True)
    assert Kpu == Kp
    da = torch.full((Hn * T, Q), 7.0, device=DEV)
    dq = torch.full((R, Q), 7.0, device=DEV)
    dU = torch.full((Hn * T, A0), 7.0, device=DEV)
    dV = torch.full((R, A0), 7.0, device=DEV)
    ops.call("clsr_att_l0_bwd_h", dz0, A0, Wt, None, Kp, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0)
    da_u = torch.full((Hn * T, Q), 7.0, device=DEV)       # the same with the U-path share added: da += dU . Wu^T
    ops.call("clsr_att_l0_bwd_h", dz0, A0, Wt, Wut, Kp, a, Q, q, Q, Hn, G, T, Q, A0, da_u, Q, dq, Q, dU, A0, dV, A0)
    torch.cuda.synchronize()
    d = dz0.double().view(Hn, G, T, A0)
    _close(da_u, da.double().view(Hn, T, Q) + d.sum(1) @ _r(Wu).t(), 1e-4, 2e-4, "da + dU.Wu^T")
    daq = d @ _r(Wp).t()                                              # [Hn, G, T, Q]
    a4, q4 = a.double().view(Hn, 1, T, Q), q.double().view(Hn, G, 1, Q)
    _close(da, (daq * q4).sum(1), 1e-4, 1e-4, "da")
    _close(dq, (daq * a4).sum(2), 1e-4, 2e-4, "dq")
    _close(dU, d.sum(1), 1e-5, 1e-5, "dU")
    _close(dV, d.sum(2), 1e-5, 1e-5, "dV")
    # the path it replaces
    daq_h = torch.zeros(M, Q, dtype=BF, device=DEV)
    ops.call("clsr_hgemm", dz0, A0, None, None, 0, Wt, Kp, None, daq_h, Q, None, M, A0, Q)
    da2, dq2 = torch.zeros_like(da), torch.zeros_like(dq)
    dU2, dV2 = torch.zeros_like(dU), torch.zeros_like(dV)
    ops.call("clsr_att_prod_bwd_h", daq_h, Q, a, Q, q, Q, Hn, G, T, Q, da2, Q, dq2, Q, 0)
    ops.call("clsr_att_z0_bwd_reduce_h", dz0, Hn, G, T, A0, dU2, dV2)
    torch.cuda.synchronize()
    _close(da, da2, 2e-2, 2e-2 * float(da2.abs().max()), "da vs three-kernel path")
    _close(dq, dq2, 2e-2, 2e-2 * float(dq2.abs().max()), "dq vs three-kernel path")
    _close(dU, dU2, 1e-5, 1e-5, "dU vs three-kernel path")
    _close(dV, dV2, 1e-5, 1e-5, "dV vs three-kernel path")


def _dw_full(partial_call, M, K, N, with_bias, parts="clsr_pgemm_dw_parts"):
    """run a deferred weight-gradient launch + the batched reduction; returns (dW, db).  ``parts``: the query that tells
    how many partial chunks the kernel wrote (clsr_hdw_parts for the bf16-MFMA kernel)"""
    ws = torch.zeros(ops.query("clsr_pgemm_dw_workspace_floats", M, K, N), device=DEV)
    partial_call(ws)
    dW, db = torch.zeros(K, N, device=DEV), torch.zeros(N, device=DEV)
    sig = ((ws.data_ptr(), dW.data_ptr(), db.data_ptr() if with_bias else 0, 1.0, ops.query(parts, M), K,
            N, N, 0),)
    tab = ops.dw_table(sig, torch.device(DEV))
    ops.call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    return dW, db


def test_weight_gradient_kernels_with_bf16_operands():
    """The bf16-operand instantiations of pgemm_dw against a float64 product of the SAME (upcast) operands."""
    g = torch.Generator().manual_seed(5)
    Hn, G, T, K, N = 13, 5, 10, 80, 80
    R, M = Hn * G, Hn * G * T
    a, q = torch.randn(Hn * T, K, generator=g).to(DEV), torch.randn(R, K, generator=g).to(DEV)
    dz = torch.randn(M, N, generator=g).to(DEV).to(BF)
    dW, _ = _dw_full(lambda ws: ops.call("clsr_pgemm_dw_partial_h", a, 0, K, T, G, q, K, None, None, 1, dz, 1, N, M, K, N,
                                         ws), M, K, N, False)
    rows = torch.arange(M, device=DEV)
    r, t = rows // T, rows % T
    exp = (a[(r // G) * T + t] * q[r]).double().t() @ dz.double()
    _close(dW, exp, 2e-5, 5e-4, "dW (fp32 a * q, bf16 dz0)")
    z0 = torch.randn(M, 80, generator=g).to(DEV).to(BF)
    dz1 = torch.randn(M, 40, generator=g).to(DEV).to(BF)
    sc, sh = (torch.rand(80, generator=g) + 0.5).to(DEV), torch.randn(80, generator=g).to(DEV)
    dW, db = _dw_full(lambda ws: ops.call("clsr_pgemm_dw_partial_h", z0, 1, 80, 0, 0, None, 0, sc, sh, 1, dz1, 1, 40, M,
                                          80, 40, ws), M, 80, 40, True)
    exp = torch.relu(z0.float() * sc + sh).double().t() @ dz1.double()
    _close(dW, exp, 2e-5, 5e-4, "dW (relu(bn(z0 bf16)), bf16 dz1)")
    _close(db, dz1.double().sum(0), 2e-5, 5e-4, "db")


@pytest.mark.parametrize("M,K,N", [(3200, 80, 80), (1000, 40, 480), (777, 80, 120), (130, 164, 80), (64, 80, 40)])
def test_bf16_mfma_weight_gradient_kernel(M, K, N):
    """clsr_hdw_partial (weight gradients on the bf16 matrix pipe) == float64 product of the bf16-ROUNDED operands:
    plain fp32 operands, the X * Xmul[r] prologue with the (T, G) row map, relu(bn(X)) on bf16 X with bf16 dY."""
    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=g).to(DEV)
    dY = torch.randn(M, N, generator=g).to(DEV)
    dW, db = _dw_full(lambda ws: ops.call("clsr_hdw_partial", X, 0, K, 0, 0, None, 0, None, None, 1, dY, 0, N, M, K, N, ws),
                      M, K, N, True, parts="clsr_hdw_parts")
    _close(dW, _r(X).t() @ _r(dY), 1e-4, 2e-3, "dW plain")
    _close(db, dY.double().sum(0), 1e-5, 1e-3, "db (exact fp32 column sums of the unrounded dY)")
    if K % 4 == 0 and M % 10 == 0:
        T, G = 10, 5 if (M // 10) % 5 == 0 else 1
        R, Hn = M // T, M // T // G
        a_, q_ = torch.randn(Hn * T, K, generator=g).to(DEV), torch.randn(R, K, generator=g).to(DEV)
        dYh = dY.to(BF)
        dW, _ = _dw_full(lambda ws: ops.call("clsr_hdw_partial", a_, 0, K, T, G, q_, K, None, None, 1, dYh, 1, N, M, K, N,
                                             ws), M, K, N, False, parts="clsr_hdw_parts")
        rows = torch.arange(M, device=DEV)
        r, t = rows // T, rows % T
        _close(dW, _r(a_[(r // G) * T + t] * q_[r]).t() @ dYh.double(), 1e-4, 2e-3, "dW (a * q)")
    Xh, dYh = X.to(BF), dY.to(BF)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(DEV), torch.randn(K, generator=g).to(DEV)
    dW, db = _dw_full(lambda ws: ops.call("clsr_hdw_partial", Xh, 1, K, 0, 0, None, 0, sc, sh, 1, dYh, 1, N, M, K, N, ws),
                      M, K, N, True, parts="clsr_hdw_parts")
    _close(dW, _r(torch.relu(Xh.float() * sc + sh)).t() @ dYh.double(), 1e-4, 2e-3, "dW relu(bn(X))")
    _close(db, dYh.double().sum(0), 1e-4, 2e-3, "db (bf16 dY)")


@pytest.mark.parametrize("Hn,G,T", [(19, 5, 10), (33, 1, 50), (7, 5, 50)])
def test_bf16_input_variants_equal_the_fp32_kernels_on_upcast_inputs(Hn, G, T):
    """att_out_fwd / att_dy1_stats / att_z0_bwd_reduce / att_prod_bwd with bf16 tensors == the fp32 kernels fed with
    the same values converted to fp32 (same code, only the loads differ)."""
    g = torch.Generator().manual_seed(6)
    R, M, C1, C0, Dk, Q = Hn * G, Hn * G * T, 40, 80, 40, 80
    z1h = torch.randn(M, C1, generator=g).to(DEV).to(BF)
    z1f = z1h.float()
    sc, sh = (torch.rand(C1, generator=g) + 0.5).to(DEV), torch.randn(C1, generator=g).to(DEV) * 0.3
    mu, inv = torch.randn(C1, generator=g).to(DEV) * 0.1, (torch.rand(C1, generator=g) + 0.5).to(DEV)
    wo, bo = torch.randn(C1, generator=g).to(DEV), torch.randn(1, generator=g).to(DEV)
    keys = torch.randn(Hn, T, Dk, generator=g).to(DEV)
    lens = torch.randint(1, T + 1, (Hn,), generator=g).to(torch.int32).to(DEV)
    outs = []
    for name, z in (("clsr_att_out_fwd_h", z1h), ("clsr_att_out_fwd", z1f)):
        wts, out = torch.zeros(R, T, device=DEV), torch.zeros(R, Dk, device=DEV)
        ops.call(name, z, sc, sh, wo, bo, lens, 1, keys, Hn, G, T, C1, Dk, wts, out)
        outs.append((wts, out))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ds = torch.randn(M, generator=g).to(DEV)
    parts = ops.query("clsr_att_dy1_parts", M, C1)
    res = []
    for name, z in (("clsr_att_dy1_stats_h", z1h), ("clsr_att_dy1_stats", z1f)):
        bnp = torch.zeros(parts * 2 * C1, dtype=torch.float64, device=DEV)
        wp = torch.zeros(parts * C1, device=DEV)
        ops.call(name, z, ds, sc, sh, mu, inv, wo, M, C1, bnp, wp)
        res.append((bnp, wp))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    dzh = torch.randn(M, C0, generator=g).to(DEV).to(BF)
    res = []
    for name, z in (("clsr_att_z0_bwd_reduce_h", dzh), ("clsr_att_z0_bwd_reduce", dzh.float())):
        dU, dV = torch.zeros(Hn * T, C0, device=DEV), torch.zeros(R, C0, device=DEV)
        ops.call(name, z, Hn, G, T, C0, dU, dV)
        res.append((dU, dV))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    a, q = torch.randn(Hn * T, Q, generator=g).to(DEV), torch.randn(R, Q, generator=g).to(DEV)
    dah = torch.randn(M, Q, generator=g).to(DEV).to(BF)
    res = []
    for h, z in ((True, dah), (False, dah.float())):
        da, dq = torch.zeros(Hn * T, Q, device=DEV), torch.zeros(R, Q, device=DEV)
        if h:
            ops.call("clsr_att_prod_bwd_h", z, Q, a, Q, q, Q, Hn, G, T, Q, da, Q, dq, Q, 0)
        else:
            ops.call("clsr_att_prod_bwd_ld", z, Q, a, Q, q, Q, Hn, G, T, Q, da, Q, dq, Q, 0)
        res.append((da, dq))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


# ------------------------------------------------------------------------------------------------ whole step
def _feed(golden_dir, name, b=0):
    g = np.load(os.path.join(golden_dir, name))
    pre = "b%d_" % b
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _net(hp, dims, params32, O, precision, dedup=True):
    net = CLSRNet(hp, dims, device=DEV, seed=0, dedup_histories=dedup, precision=precision)
    sd = dict(params32)
    sd.update(O.init_bn_state(params32))
    net.load_state_dict(sd, strict=True)
    return net


CONFIGS = [dict(), dict(sequential_model="gru", contrastive_loss="bpr"), dict(manual_alpha=True, manual_alpha_value=0.3)]


@pytest.mark.parametrize("chain", ["x1", "old"])
@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("cfg", range(len(CONFIGS)))
def test_bf16_train_step_against_oracle_and_fp32_mode(golden_dir, golden_hparams, cfg, dedup, chain, capsys):
    """chain = "x1": the speed mode on the parity mode's chain kernels with one bf16 piece (the default); "old": the
    position-tiled kernels of rounds 2-4 (csrc/hgemm.hip, hdw.hip, hattbwd.hip: what shapes outside the chain kernels'
    limits fall back to, CLSR_BF16_CHAIN=old) run AS A STEP here so that they cannot rot.
    (1) north_star bars against the EXACT oracle: logits within 1e-3, loss terms within 1e-4 relative;
    (2) everything -- forward values, every gradient, the BN statistics -- against the oracle that rounds the
    attention activations to bf16 at the same places (oracle.BF16_ATTENTION): percent-level agreement per variable.
    The gap between the two oracles (what bf16 storage itself does to the gradients of this small batch) is printed."""
    from oracle import clsr_oracle as O
    from clsr_amd.params import TABLES

    hp = copy.deepcopy(golden_hparams)
    for k, v in CONFIGS[cfg].items():
        setattr(hp, k, v)
    dims = _dims(hp)
    params32 = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    params = type(params32)((k, v.double()) for k, v in params32.items())
    feed = _feed(golden_dir, "iterator_train_sa.npz", b=cfg % 3)
    tf = O.to_torch_feed(feed, dtype=torch.float64)
    args = (params, O.init_bn_state(params), O.init_adam(params), 1, tf, hp)
    _, _, _, ls, _, _, out = O.train_step(*args)
    O.BF16_ATTENTION = True
    try:
        _, new_bn_e, _, ls_e, _, _, out_e = O.train_step(*args)
    finally:
        O.BF16_ATTENTION = False
    net = _net(hp, dims, params32, O, "bf16", dedup)
    net.bf16_chain = chain == "x1"
    net.capture_grads = True
    got = net.train_step(net.upload(feed, True))
    torch.cuda.synchronize()
    gl, cap = net.read_losses(), net.captured
    B, T, G, Hn = net.last_shape
    rep = (lambda t: t) if G == 1 else (lambda t: t[::G])
    # ---- (1) the exact oracle
    _close(got["logit"], out["logit"], 0.0, 1e-3, "logit (bf16 mode vs exact oracle)")
    for k in ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss"):
        _close(torch.tensor([gl[k]]), torch.tensor([float(ls[k])]), 1e-4, 1e-6, k)
    # ---- (2) the bf16-emulating oracle
    _close(got["logit"], out_e["logit"], 0.0, 1e-4, "logit (vs emulation)")
    _close(got["att_fea_short"], out_e["att_fea_short"], 2e-3, 2e-5, "att_fea_short")
    _close(got["att_fea_long"], rep(out_e["att_fea_long"]), 2e-3, 2e-5, "att_fea_long")
    _close(got["w_short"], out_e["w_short"], 5e-3, 1e-5, "short attention weights")
    for k in ("loss", "data_loss", "contrastive_loss"):
        _close(torch.tensor([gl[k]]), torch.tensor([float(ls_e[k])]), 1e-5, 1e-7, k + " (vs emulation)")
    raw, raw_e = out["raw_grads"], out_e["raw_grads"]
    floor = 1e-3 * max(float(raw[n].abs().max()) for n in net.dense_names)   # exactly-zero gradients (biases under a BN)
    rows = []
    for name in list(net.dense_names) + list(TABLES.values()):
        key = [k for k, v in TABLES.items() if v == name]
        g_ = (cap["tables"][key[0]] if key else cap["dense"][name]).double().cpu()
        scale = float(raw_e[name].abs().max()) + floor
        if float(raw[name].abs().max()) < floor:
            # analytically zero gradient (a bias under a batch-norm): the bf16 mode returns the sum of ~1e3..1e6 rounded
            # terms that cancel -- noise of a few 1e-4 of the step's largest gradient; held to 2e-3 of that scale
            scale = 100.0 * floor     # 3e-2 * scale == 3e-3 of the largest dense gradient
        rows.append((float((g_ - raw_e[name]).abs().max()) / scale, float((raw[name] - raw_e[name]).abs().max()) / scale,
                     name))
    rows.sort(reverse=True)
    with capsys.disabled():
        print("\n[bf16 step, cfg %d dedup %s] gradient error vs the bf16-emulating oracle | gap between the two "
              "oracles (fractions of the variable's largest gradient)" % (cfg, dedup))
        for e, gap, name in rows[:6]:
            print("   %.2e | %.2e  %s" % (e, gap, name))
    # 3e-2: the noise-sensitive variables (large 'gap' column: sums of cancelling terms such as the GRU candidate bias)
    # sit at 1.5-2.2e-2 once weight gradients and back-propagating products run on the bf16 matrix pipe as well
    bad = [(e, n) for e, _, n in rows if e > 3e-2]
    assert not bad, bad
    sd = net.state_dict()
    for k, v in new_bn_e.items():
        _close(sd[k], v, 1e-3, 1e-5, k)


def test_bf16_eval_scores_match_oracle(golden_dir, golden_hparams):
    from oracle import clsr_oracle as O

    hp = golden_hparams
    dims = _dims(hp)
    params32 = O.init_params(dims, hp, seed=7, scale_dense=8.0)
    params = type(params32)((k, v.double()) for k, v in params32.items())
    feed = _feed(golden_dir, "iterator_eval_sa.npz")
    exp = O.predict(params, O.init_bn_state(params), O.to_torch_feed(feed, dtype=torch.float64), hp)
    net = _net(hp, dims, params32, O, "bf16")
    got = net.forward(net.upload(feed, False), False)
    torch.cuda.synchronize()
    _close(torch.sigmoid(got["logit"]), exp["pred"], 0.0, 1e-3, "pred")


def test_bf16_chain_kernels_agree_with_the_position_tiled_kernels(golden_dir, golden_hparams, monkeypatch):
    """The speed mode runs its (row, step)-level attention layers on the parity mode's chain kernels with one bf16 piece per
    operand (net.bf16_chain, csrc/attbwdx3.hip ...); CLSR_BF16_CHAIN=old keeps the position-tiled csrc/hgemm.hip kernels +
    separate weight-gradient launches.  Same roundings at the same places: logits and every dense gradient agree at the
    level of bf16 noise."""
    from oracle import clsr_oracle as O

    hp = copy.deepcopy(golden_hparams)
    dims = _dims(hp)
    params32 = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    feed = _feed(golden_dir, "iterator_train_sa.npz", b=0)
    res = {}
    for chain in ("x1", "old"):
        monkeypatch.setenv("CLSR_BF16_CHAIN", chain)
        net = _net(hp, dims, params32, O, "bf16")
        assert net.bf16_chain == (chain == "x1")
        net.capture_grads = True
        got = net.train_step(net.upload(feed, True))
        torch.cuda.synchronize()
        res[chain] = (got["logit"].double().cpu(), {k: v.double().cpu() for k, v in net.captured["dense"].items()})
    _close(res["x1"][0], res["old"][0], 0.0, 2e-4, "logit (chain kernels vs position-tiled kernels)")
    top = max(float(v.abs().max()) for v in res["old"][1].values())
    for k, v in res["old"][1].items():
        scale = max(float(v.abs().max()), 1e-3 * top)
        err = float((res["x1"][1][k] - v).abs().max())
        assert err <= 3e-2 * scale, (k, err, scale)
