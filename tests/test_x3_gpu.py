"""GPU parity tests of the split-bf16 ("fp32x3") kernels: fp32 operands in HBM, every value split into bf16 hi + bf16 lo
in registers, products taken as hi*hi + lo*hi + hi*lo on the bf16 matrix pipe with fp32 accumulation (csrc/gemm3.hip, csrc/encbwd.hip ...).
They must hold the EXACT-mode tolerances of the fp32-MFMA kernels they replace (reference arithmetic: fp32,
models/base_model.py:627-708, models/sequential/clsr.py:343-381 through tf.gradients), i.e. they are compared with
float64 products of the UNROUNDED operands -- unlike the speed-mode kernels of tests/test_bf16_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402

DEV = "cuda"


def _close(got, exp, rtol, atol, name):
    got, exp = got.double().cpu().reshape(-1), exp.double().cpu().reshape(-1)
    err = (got - exp).abs()
    excess = float((err - (atol + rtol * exp.abs())).max())
    assert excess <= 0, "%s: max abs err %.3e (max |exp| %.3e)" % (name, float(err.max()), float(exp.abs().max()))


def _within(got, exp, budget, name):
    err = (got.double().cpu() - exp.double().cpu()).abs()
    over = float((err - budget.double().cpu()).max())
    assert over <= 0, "%s: max abs err %.3e (max |exp| %.3e), %.3e over its budget" % (
        name, float(err.max()), float(exp.abs().max()), over)


def _rnd(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def _f(t):
    return None if t is None else t.to(torch.float32).to(DEV).contiguous()


def _pgemm3(X, W, bias=None, T=0, G=0, Xmul=None, in_scale=None, in_shift=None, relu=0, addU=None, addV=None, Y=None,
            accumulate=0, stats=False, M=None):
    K, N = W.shape
    M = X.shape[0] if M is None else M
    Wt, Kp = ops.pack_weight(_f(W), N, K)
    if Y is None:
        Y = torch.zeros(M, N, device=DEV)
    st = torch.zeros(query("clsr_pgemm_stats_parts", M), 2, N, dtype=torch.float64, device=DEV) if stats else None
    assert query("clsr_pgemm3_supported", int(Xmul is not None), int(in_scale is not None), int(addU is not None),
                 int(addV is not None), accumulate, int(stats), M, K, N) == 1
    call("clsr_pgemm3", _f(X), X.shape[1], T, G, _f(Xmul), 0 if Xmul is None else Xmul.shape[1], _f(in_scale), _f(in_shift),
         relu, Wt, Kp, _f(bias), _f(addU), 0 if addU is None else addU.shape[1], _f(addV),
         0 if addV is None else addV.shape[1], Y, Y.shape[1], accumulate, st, M, K, N)
    torch.cuda.synchronize()
    return Y, st


@pytest.mark.parametrize("M,K,N", [(1000, 80, 80), (333, 40, 40), (77, 164, 80), (515, 100, 64), (2304, 256, 256),
                                   (260, 40, 240), (129, 80, 100), (50, 64, 4), (2000, 120, 120), (4100, 40, 480),
                                   (40000, 160, 80), (20480, 200, 80), (90, 320, 40), (65, 32, 16), (33, 8, 12)])
def test_split_bf16_pgemm_plain_bias_stats(M, K, N):
    """clsr_pgemm3 == float64 X . W + b inside the split product's error budget -- 2^-16 of sum |x| |w| per output (three
    rounding sources of <= 2^-18 |x w| each + fp32 accumulation); typical errors are ~sqrt(K) below it -- with the column
    statistics of the stored values; accumulate form."""
    g = torch.Generator().manual_seed(M + K + N)
    X, W, b = _rnd(g, M, K), _rnd(g, K, N, scale=0.3), _rnd(g, N)
    Y, st = _pgemm3(X, W, b, stats=True)
    exp = X @ W + b
    budget = (X.abs() @ W.abs() + b.abs()) * 2.0 ** -16
    _within(Y, exp, budget, "Y")
    # typical size of the error: a few 2^-18 * 0.3 * sqrt(K) (random signs)
    rms = float((Y.double().cpu() - exp).pow(2).mean().sqrt())
    assert rms <= 2.0 ** -18 * 0.3 * K ** 0.5 * 2, "rms error %.3e" % rms
    tot = st.sum(0).cpu()
    got = Y.double().cpu()
    _close(tot[0], got.sum(0), 1e-6, 1e-4 * M ** 0.5, "colsum of the stored values")
    _close(tot[1], (got ** 2).sum(0), 1e-6, 1e-4 * M ** 0.5 * max(1.0, K / 100.0), "colsumsq")
    Y2, _ = _pgemm3(X, W, None, Y=Y.clone(), accumulate=1)
    _within(Y2, 2 * exp - b, 2 * budget, "accumulate")


@pytest.mark.parametrize("K,N", [(480, 40), (516, 80), (1536, 128)])
def test_split_bf16_pgemm_wide_k_chain(K, N):
    """inputs wider than one launch's weight images: a chain of accumulating launches over K ranges"""
    g = torch.Generator().manual_seed(K + N)
    M = 3000
    X, W, b = _rnd(g, M, K), _rnd(g, K, N, scale=0.2), _rnd(g, N)
    Y, _ = _pgemm3(X, W, b)
    budget = (X.abs() @ W.abs() + b.abs()) * 2.0 ** -16
    _within(Y, X @ W + b, budget, "Y")
    Y0 = _rnd(g, M, N)
    Y2, _ = _pgemm3(X, W, None, Y=_f(Y0), accumulate=1)
    _within(Y2, Y0 + X @ W, budget + Y0.abs() * 2.0 ** -22, "accumulate")


def test_split_bf16_pgemm_rowmap_mul_affine_adds():
    g = torch.Generator().manual_seed(5)
    Hn, G, T, K, N = 13, 5, 10, 80, 80
    R = Hn * G
    a, q = _rnd(g, Hn * T, K), _rnd(g, R, K)
    W = _rnd(g, K, N, scale=0.2)
    U, V = _rnd(g, Hn * T, N), _rnd(g, R, N)
    Y, st = _pgemm3(a, W, None, T=T, G=G, Xmul=q, addU=U, addV=V, stats=True, M=R * T)
    rows = torch.arange(R * T)
    r, t = rows // T, rows % T
    xrow = (r // G) * T + t
    exp = (a[xrow] * q[r]) @ W + U[xrow] + V[r]
    bud = lambda x, w, extra=0.0: (x.abs() @ w.abs() + extra) * 2.0 ** -16
    _within(Y, exp, bud(a[xrow] * q[r], W, (U[xrow] + V[r]).abs()), "att z0")
    _close(st.sum(0)[0].cpu(), Y.double().cpu().sum(0), 1e-6, 1e-3, "stats")
    for Kq in (40, 160):      # (K = 40: one k-tile with 24 empty slots; K = 160: the ring form)
        a2, q2, W2 = _rnd(g, Hn * T, Kq), _rnd(g, R, Kq), _rnd(g, Kq, N, scale=0.2)
        Y, _ = _pgemm3(a2, W2, None, T=T, G=G, Xmul=q2, addU=U, addV=V, M=R * T)
        _within(Y, (a2[xrow] * q2[r]) @ W2 + U[xrow] + V[r], bud(a2[xrow] * q2[r], W2, (U[xrow] + V[r]).abs()),
                "att z0, K = %d" % Kq)
    sc, sh = _rnd(g, K), _rnd(g, K)
    X = _rnd(g, 301, K)
    Y, st = _pgemm3(X, W, None, in_scale=sc, in_shift=sh, relu=1, stats=True)
    exp = torch.relu(X * sc + sh) @ W
    _within(Y, exp, bud(torch.relu(X * sc + sh), W, 1e-5), "bn-relu prologue")
    _close(st.sum(0)[1].cpu(), (Y.double().cpu() ** 2).sum(0), 1e-6, 1e-3, "stats (affine)")
    # transposed pack (dX = dY . W^T) and strided output into a wider buffer
    dY = _rnd(g, 200, N)
    Wt, Kp = ops.pack_weight(_f(W), K, N, transposed=True)
    out = torch.zeros(200, K + 8, device=DEV)
    call("clsr_pgemm3", _f(dY), N, 0, 0, None, 0, None, None, 0, Wt, Kp, None, None, 0, None, 0, out[:, 4:], K + 8, 0, None,
         200, N, K)
    torch.cuda.synchronize()
    _within(out[:, 4:4 + K], dY @ W.T, bud(dY, W.T), "dX")
    assert float(out[:, :4].abs().max()) == 0 and float(out[:, 4 + K:].abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(1000, 40, 80), (20480, 64, 100), (333, 100, 64), (515, 80, 161 // 4 * 4)])
def test_split_bf16_pgemm_bn_backward_epilogue(M, K, N):
    """clsr_pgemm3_bnbwd == relu-mask(dY . W^T) with the batch-norm backward sums (sum dy, sum dy * xhat)"""
    g = torch.Generator().manual_seed(M + K)
    dY, W = _rnd(g, M, K), _rnd(g, N, K, scale=0.3)          # W: [N (layer input), K (layer output)]
    z = _rnd(g, M, N)
    sc, sh = _rnd(g, N).abs() + 0.5, _rnd(g, N) * 0.3
    mean, inv = _rnd(g, N) * 0.1, _rnd(g, N).abs() + 0.5
    Wt, Kp = ops.pack_weight(_f(W), N, K, transposed=True)
    assert query("clsr_pgemm3_bnbwd_supported", M, K, N) == 1
    out = torch.zeros(M, N, device=DEV)
    parts = query("clsr_pgemm_stats_parts", M)
    st = torch.zeros(parts, 2, N, dtype=torch.float64, device=DEV)
    call("clsr_pgemm3_bnbwd", _f(dY), K, Wt, Kp, out, N, _f(z), N, _f(sc), _f(sh), _f(mean), _f(inv), st, M, K, N)
    torch.cuda.synchronize()
    zf = _f(z).double().cpu()
    mask = (zf * _f(sc).double().cpu() + _f(sh).double().cpu()) > 0
    exp = torch.where(mask, dY @ W.T, torch.zeros((), dtype=torch.float64))
    # (entries whose pre-activation sits within rounding of the ReLU kink may take either side)
    near = (zf * _f(sc).double().cpu() + _f(sh).double().cpu()).abs() < 1e-6
    got = out.double().cpu()
    err = ((got - exp).abs() - (dY.abs() @ W.T.abs()) * 2.0 ** -16).clamp(min=0)
    assert float(err[~near].max()) == 0.0, "dy: max excess %.3e" % float(err[~near].max())
    xhat = (zf - _f(mean).double().cpu()) * _f(inv).double().cpu()
    _close(st.sum(0)[0].cpu(), got.sum(0), 1e-5, 1e-3, "sum dy")
    _close(st.sum(0)[1].cpu(), (got * xhat).sum(0), 1e-5, 1e-3, "sum dy * xhat")


# ------------------------------------------------------------------------------- fused encoder tail (csrc/encbwd.hip)
@pytest.mark.parametrize("M", [4800, 4117, 37, 32, 204800])
def test_split_bf16_fused_encoder_backward(M):
    """clsr_enc_bwd_fused_x3: the seven encoder-side weight gradients + bias sums from one pass over the fp32 dPin as
    split-bf16 products == float64 products of the UNROUNDED operands at the tolerance of the fp32-MFMA kernel
    (tests/test_kernels_gpu.py::test_fused_encoder_backward_one_pass_over_dpin), through the partial layout +
    clsr_dw_reduce_batch that the step uses."""
    g = torch.Generator().manual_seed(M)
    n = 40
    dPin, hist = _f(_rnd(g, M, 480)), _f(_rnd(g, M, n))
    hp1, hp2, mp, TT = _f(_rnd(g, M, n)), _f(_rnd(g, M, n)), _f(_rnd(g, M, n)), _f(torch.tanh(_rnd(g, M, 2 * n)))
    g1 = _f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64))
    g2 = _f(torch.rand(M, 3 * n, generator=g, dtype=torch.float64))
    parts = query("clsr_enc_bwd_fused_x3_parts", M)
    shapes = [(n, 480), (n, 80), (n, 40), (n, 160), (2 * n, 120), (n, 80), (n, 40)]
    wss = [torch.full((query("clsr_enc_bwd_fused_x3_workspace_floats", M, i),), 9.0, device=DEV) for i in range(7)]
    outs = [torch.zeros(K, N, device=DEV) for K, N in shapes]
    db = torch.zeros(480, device=DEV)
    call("clsr_enc_bwd_fused_x3", dPin, hist, hp1, g1, mp, TT, hp2, g2, *wss, M)
    sig = tuple((ws.data_ptr(), o.data_ptr(), db.data_ptr() if i == 0 else 0, 1.0, parts, K, N, N, 0)
                for i, (ws, o, (K, N)) in enumerate(zip(wss, outs, shapes)))
    tab = ops.dw_table(sig, torch.device(DEV))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    P = d(dPin)
    exp = [d(hist).T @ P, d(hp1).T @ P[:, 0:80], (d(hp1) * d(g1)[:, :n]).T @ P[:, 80:120], d(mp).T @ P[:, 240:400],
           d(TT).T @ P[:, 360:480], d(hp2).T @ P[:, 120:200], (d(hp2) * d(g2)[:, :n]).T @ P[:, 200:240]]
    tol = 2e-5 * M ** 0.5
    for i, (o, e) in enumerate(zip(outs, exp)):
        _close(o, e, 2e-4, tol, "product %d" % i)
    _close(db, P.sum(0), 2e-4, tol, "bias sums")
