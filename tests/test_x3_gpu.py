"""GPU parity tests of the split-bf16 ("fp32x3") kernels: fp32 operands in HBM, every value split into bf16 hi + bf16 lo
in registers, products taken as hi*hi + lo*hi + hi*lo on the bf16 matrix pipe with fp32 accumulation (csrc/encbwd.hip; the chain kernels of csrc/attbwdx3.hip / atthist.hip / projx3.hip have their own files).
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


# ------------------------------------------------------------------------------- fused encoder tail (csrc/encbwd.hip)
@pytest.mark.parametrize("entry,rt,at", [("clsr_enc_bwd_fused_x3", 2e-4, 2e-5), ("clsr_enc_bwd_fused_x6", 2e-6, 4e-6)])
@pytest.mark.parametrize("M", [4800, 4117, 37, 32, 204800])
def test_split_bf16_fused_encoder_backward(M, entry, rt, at):
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
    call(entry, dPin, hist, hp1, g1, mp, TT, hp2, g2, *wss, M)      # (x6: three pieces per operand, fp32 accuracy)
    sig = tuple((ws.data_ptr(), o.data_ptr(), db.data_ptr() if i == 0 else 0, 1.0, parts, K, N, N, 0)
                for i, (ws, o, (K, N)) in enumerate(zip(wss, outs, shapes)))
    tab = ops.dw_table(sig, torch.device(DEV))
    call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    P = d(dPin)
    exp = [d(hist).T @ P, d(hp1).T @ P[:, 0:80], (d(hp1) * d(g1)[:, :n]).T @ P[:, 80:120], d(mp).T @ P[:, 240:400],
           d(TT).T @ P[:, 360:480], d(hp2).T @ P[:, 120:200], (d(hp2) * d(g2)[:, :n]).T @ P[:, 200:240]]
    tol = at * M ** 0.5      # (x6: what is left is the fp32 ACCUMULATION over M positions, not the products)
    for i, (o, e) in enumerate(zip(outs, exp)):
        _close(o, e, rt, tol, "product %d" % i)
    _close(db, P.sum(0), rt, tol, "bias sums")
