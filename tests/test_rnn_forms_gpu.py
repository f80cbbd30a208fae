"""The two forms of the recurrences' hidden-to-hidden products (csrc/rnn.hip, clsr_gru_desc.products): fp32-input MFMA
(bit-exact fp32) and split-bf16 (W.h ~ Whi.hhi + Whi.hlo + Wlo.hhi on v_mfma_f32_16x16x32_bf16, fp32 accumulation), and
the private tile-major image of the Time4LSTM's saved activations (clsr_t4_desc.act_tiled).

Reference: tf.nn.rnn_cell.GRUCell / Time4LSTMCell under dynamic_rnn, clsr.py:160-237, rnn_cell_implement.py:129-298.
The float64 comparisons of both forms live in tests/test_kernels_gpu.py (they run in the process default form, split-bf16);
here the two forms are compared with each other on the same inputs, and layouts that must not change a bit are checked
bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import query  # noqa: E402


def _rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _run(Hn, T, n, products, tiled=False, fused=False, seed=5, scale_w=0.15, D=40):
    """GRU (h0, sequence output) + Time4LSTM forward then backward in the fused launches; returns every output.
    ``fused``: the input projections of the GRU and of the Time4LSTM blocks i | j | f run inside the recurrence from the
    embeddings X (clsr_gru_desc.X / clsr_t4_desc.X); otherwise they are one fp32 GEMM in front of it."""
    g = torch.Generator().manual_seed(seed)
    NX = 9 * n
    lens = torch.randint(1, T + 1, (Hn,), generator=g).int()
    lens[0] = T
    lens = lens.cuda()
    X = _rnd(g, Hn * T, D, scale=0.7)
    Wgf, Wcf, Wkf = _rnd(g, D + n, 2 * n, scale=scale_w), _rnd(g, D + n, n, scale=scale_w), _rnd(g, D + n, 4 * n, scale=scale_w)
    bg, bc, bk = _rnd(g, 2 * n, scale=0.3), _rnd(g, n, scale=0.3), _rnd(g, 4 * n, scale=0.3)
    tgates = _rnd(g, Hn * T, 2 * n, scale=0.5)                      # tns | tls pre-activations
    Pin = torch.cat([X @ Wgf[:D] + bg, X @ Wcf[:D] + bc, X @ Wkf[:D] + bk, tgates], 1).contiguous()
    Wg, Wc, Wm = Wgf[D:], Wcf[D:], Wkf[D:]
    h0 = _rnd(g, Hn, n, scale=0.5)
    dseq_g, dseq_t, dhT = _rnd(g, Hn, T, n, scale=0.3), _rnd(g, Hn, T, n, scale=0.3), _rnd(g, Hn, n, scale=0.3)
    z = lambda *s: torch.zeros(*s, device="cuda")
    hT, gout, hprev, gates = z(Hn, n), z(Hn, T, n), z(Hn, T, n), z(Hn, T, 3 * n)
    tout, mprev = z(Hn, T, n), z(Hn, T, n)
    if tiled:
        act, cst = z(query("clsr_t4_act_tiled_floats", Hn, T, n)), None
    else:
        act, cst = z(Hn, T, 6 * n), z(Hn, T, n)
    if fused:
        P3 = Pin[:, 6 * n:].contiguous()                            # o | tns | tls
        gd = ops.gru_desc(n, X=X, ldx=D, Dx=D, Wgx=Wgf, Wcx=Wcf, bg=bg, bc=bc, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, h0=h0,
                          h0_stride=n, hT=hT, out_seq=gout, hprev=hprev, gates=gates, products=products)
        td = ops.t4_desc(n, Pin=P3, ldp=3 * n, X=X, ldx=D, Dx=D, Wkx=Wkf, bk=bk, Wm=Wm, ldm=4 * n, out_seq=tout, act=act,
                         cst=cst, mprev=mprev, products=products, act_tiled=tiled)
    else:
        gd = ops.gru_desc(n, Pin=Pin, ldp=NX, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, h0=h0, h0_stride=n, hT=hT, out_seq=gout,
                          hprev=hprev, gates=gates, products=products)
        td = ops.t4_desc(n, Pin=Pin[:, 3 * n:], ldp=NX, Wm=Wm, ldm=4 * n, out_seq=tout, act=act, cst=cst, mprev=mprev,
                         products=products, act_tiled=tiled)
    ops.rnn_multi("clsr_rnn_fwd_multi", [gd], td, lens, 1, Hn, T)
    dP = torch.full((Hn * T, NX), 3.0, device="cuda")
    dh0 = z(Hn, n)
    gb = ops.gru_desc(n, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, hprev=hprev, gates=gates, dhT=dhT, dout_seq=dseq_g, dPin=dP,
                      lddp=NX, dh0=dh0, products=products)
    tb = ops.t4_desc(n, Wm=Wm, ldm=4 * n, act=act, cst=cst, dout_seq=dseq_t, dPin=dP[:, 3 * n:], lddp=NX,
                     products=products, act_tiled=tiled)
    ops.rnn_multi("clsr_rnn_bwd_multi", [gb], tb, lens, 1, Hn, T)
    torch.cuda.synchronize()
    return dict(hT=hT, gru_seq=gout, hprev=hprev, gates=gates, t4_seq=tout, mprev=mprev, dPin=dP, dh0=dh0)


@pytest.mark.parametrize("form,tol", [("x3", 2e-4), ("x6", 3e-6)])
@pytest.mark.parametrize("Hn,T,n", [(37, 10, 40), (16, 50, 40), (19, 8, 128), (5, 7, 24)])
def test_split_bf16_products_follow_the_fp32_form(Hn, T, n, form, tol):
    """Same inputs, both forms: every output of the forward and the backward launch within 2e-4 of the tensor's scale
    (x3: 16 significand bits per operand, the dropped lo.lo term 2^-18; errors compound over the T steps) / 3e-6 (x6: three
    pieces per operand, the level of fp32 rounding itself; hidden sizes above 48 take the fp32-input MFMAs in the launch)."""
    a, b = _run(Hn, T, n, "fp32"), _run(Hn, T, n, form)
    for k in a:
        scale = float(a[k].abs().max())
        assert scale > 0, k
        err = float((a[k] - b[k]).abs().max())
        assert err <= tol * max(scale, 1.0), "%s: max abs difference %.3e at scale %.3e" % (k, err, scale)
        # not the same arithmetic: a silent fall back to the fp32 form would make them identical
    if not (form == "x6" and n > 48):
        assert not torch.equal(a["hT"], b["hT"])


@pytest.mark.parametrize("Hn,T,n,form", [(37, 10, 40, "x3"), (16, 50, 40, "x3"), (19, 8, 128, "x3"), (16, 50, 40, "x6")])
def test_tile_major_activation_image_changes_no_bit(Hn, T, n, form):
    a, b = _run(Hn, T, n, form, tiled=False), _run(Hn, T, n, form, tiled=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("form", ["x3", "x6"])
def test_runs_are_bit_identical(form):
    a, b = _run(33, 12, 40, form, tiled=True), _run(33, 12, 40, form, tiled=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("form,tol", [("x3", 2e-4), ("x6", 3e-6)])
@pytest.mark.parametrize("Hn,T,n,D", [(37, 10, 40, 40), (16, 50, 40, 40), (21, 9, 48, 24), (5, 3, 16, 56)])
def test_fused_input_projection_follows_the_gemm_in_front(Hn, T, n, D, form, tol):
    """x_t . W_x + b inside the recurrence (split-bf16, bias as an extra row of the product) against the fp32 GEMM that
    used to write the projection tensor: same bar as the split-bf16 products themselves."""
    a = _run(Hn, T, n, "fp32", D=D)
    b = _run(Hn, T, n, form, fused=True, tiled=True, D=D)
    for k in a:
        scale = float(a[k].abs().max())
        err = float((a[k] - b[k]).abs().max())
        assert err <= tol * max(scale, 1.0), "%s: max abs difference %.3e at scale %.3e" % (k, err, scale)
