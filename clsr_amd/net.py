"""Device-side CLSR network: parameters, optimiser state and the training / scoring step,
orchestrated from Python over the C ABI (every numeric operation is a HIP kernel in
libclsr_hip.so; torch only owns the device memory and the stream).

Mirrors what one ``sess.run`` of the reference executes
(``CLSRModel.train`` models/sequential/clsr.py:383-408, ``eval_with_user``
models/sequential/sequential_base_model.py:294-308):
embedding gathers -> long-term attention, GRU / Time4LSTM / GRU encoders -> short-term
attention -> alpha gate -> logit MLP -> losses -> gradients -> per-tensor clip -> Adam.

History de-duplication.  In training the reference replicates every history
``G = 1 + train_num_ngs`` times (io/sequential_iterator.py:588-610).  Everything that does not
depend on the target item (gathers, long-term attention, the three RNNs, hist_mean/recent) is
computed once per history group (``Hn = B / G`` rows) and shared by the G rows -- exact for the
forward pass and for the summed gradients (BN statistics are identical because every history
is replicated equally; dropout is 0).  The one observable difference: ``tf.clip_by_norm`` of
the embedding IndexedSlices uses the norm of the G un-summed replica slices, here the norm of
their sum is used; ``dedup_histories=False`` runs the reference's replicated computation.
"""
import math
import contextlib
import os
from collections import OrderedDict

import numpy as np
import torch

from clsr_amd import ops
from clsr_amd.ops import call, query
from clsr_amd.params import CL, TABLES, UNUSED_TABLE, init_tensor, param_specs

BN_MOMENTUM = 0.95
BN_EPS = 1e-4
F32 = torch.float32


def _pad4(n):
    return (n + 3) // 4 * 4


class _BN(object):
    """One tf.layers.batch_normalization layer: trainables are views into the flat dense buffer."""

    def __init__(self, net, scope, C):
        dev = net.device
        self.C = C
        self.gamma, self.beta = net.P[scope + "gamma"], net.P[scope + "beta"]
        self.dgamma, self.dbeta = net.Gd[scope + "gamma"], net.Gd[scope + "beta"]
        self.moving_mean = torch.zeros(C, dtype=F32, device=dev)
        self.moving_var = torch.ones(C, dtype=F32, device=dev)
        self.scale = torch.empty(C, dtype=F32, device=dev)
        self.shift = torch.empty(C, dtype=F32, device=dev)
        self.mean = torch.empty(C, dtype=F32, device=dev)
        self.invstd = torch.empty(C, dtype=F32, device=dev)
        self.coef = torch.empty(3 * C, dtype=F32, device=dev)
        self.scope = scope


_SIDE_STREAMS = {}     # device -> {tag: HIP stream}, see CLSRNet.__init__


class StepAborted(RuntimeError):
    """A bounded wait inside a training step gave up (see CLSRNet.check_abort): the step's results are invalid and no
    optimiser update has been applied since."""


class CLSRNet(object):
    def __init__(self, hp, dims, device="cuda:0", seed=None, dedup_histories=True, precision="fp32", table_dtype="fp32"):
        self.hp = hp
        self.dims = dict(dims)
        self.device = torch.device(device)
        self.dedup = bool(dedup_histories)
        if precision not in ("fp32", "fp32x3", "bf16"):
            raise ValueError("precision must be 'fp32', 'fp32x3' or 'bf16'")
        self.precision = precision
        # embedding tables stored as bf16 (SURVEY 8d configs 2, 3, 5 "bf16 tables"): the lookups widen the 2-byte rows, the
        # gradients / Adam moments stay fp32, the (lazy-)Adam update widens a row, updates it in fp32 and writes it back
        # rounded to nearest-even (no fp32 master copy: an update below half a bf16 ulp of the weight is lost)
        if table_dtype not in ("fp32", "bf16"):
            raise ValueError("table_dtype must be 'fp32' or 'bf16'")
        self.table_bf16 = table_dtype == "bf16"
        if self.table_bf16 and type(self) is not CLSRNet:
            raise NotImplementedError("%s: bf16 embedding tables are available for CLSRNet only" % type(self).__name__)
        if self.table_bf16 and (hp.item_embedding_dim % 8 or hp.cate_embedding_dim % 8 or hp.user_embedding_dim % 4):
            raise NotImplementedError("table_dtype='bf16' needs item / cate embedding dims that are multiples of 8")
        self._th = "_h" if self.table_bf16 else ""          # suffix of the table kernels' entry points
        self._check_supported()
        self.Di, self.Dc = hp.item_embedding_dim, hp.cate_embedding_dim
        self.D = self.Di + self.Dc
        self.enc_in = self.D       # leading columns of the history rows the recurrent encoders read
        self.Du, self.H = hp.user_embedding_dim, hp.hidden_size
        self.A0, self.A1 = hp.att_fcn_layer_sizes or (0, 0)    # (models without an attention MLP have none)
        self.L0, self.L1 = hp.layer_sizes
        self.G_train = hp.train_num_ngs + 1
        self.lazy = 1 if hp.optimizer == "lazyadam" else 0
        self._build_params(seed)
        self._bufs = {}
        self._zero_specs = OrderedDict()
        self.packed = {}
        self.packed_h = {}         # bf16 images of the weights the speed-mode attention kernels read (csrc/hgemm.hip)
        self.bf16 = self.precision == "bf16"
        # Product arithmetic of the two fp32-storage modes ("Precision policy"):
        #   "fp32"   -- the reference's arithmetic: EVERY product at fp32 accuracy, either on v_mfma_f32_16x16x4_f32 (bit-exact
        #               fp32 fma chains) or as THREE bf16 pieces per operand on v_mfma_f32_16x16x32_bf16 (x = p0 + p1 + p2, the
        #               six piece products whose indices sum to <= 2, fp32 accumulation: 2^-23 relative -- the level of an
        #               fp32 product).  This is the parity mode and the benchmark's headline.
        #   "fp32x3" -- fp32 storage, statistics, losses and optimiser as above; the recurrences' hidden products, the
        #               attention-MLP backward with its folded weight gradients and the fused encoder tail as TWO-piece
        #               split-bf16 sums hi*hi + hi*lo + lo*hi (2^-16 relative per product term: narrower than fp32; the step
        #               tests hold their fp32 tolerances on the golden batches, but it is NOT the reference's arithmetic).
        # (round 5 ran the two-piece forms under the name "fp32"; round 6 moved them back here.)
        self.x3 = self.precision == "fp32x3"
        self.exact_products = self.precision == "fp32"      # no two-piece product anywhere
        x3d = self.x3                                        # two-piece forms
        x6d = self.precision in ("fp32", "fp32x3")           # three-piece forms (fp32 accuracy): both fp32-storage modes
        self.x3_dw = False      # (the generic split-bf16 weight-gradient kernel, csrc/dw3.hip, measured level or slower in
                                # round 4 and was removed in round 5: the attention weight gradients are folded into the
                                # backward kernels instead, csrc/attbwdx3.hip)
        self.x3_enc = x3d and True        # A/B: fused encoder tail (csrc/encbwd.hip)
        # precision="fp32": CLSR_ENC_BWD=x6 runs the same kernel with three pieces per operand (weight gradients on the
        # weight-gradient stream, beside d(hist) as a three-piece product and the time-feature chain).  Measured and NOT the
        # default: its three 54 KB piece images fill the CU's LDS, the launch takes 394 us and starves what runs beside it
        # (3.09 against 3.03 ms per step with the fp32-MFMA kernel that also computes d(hist), alone on the compute stream)
        self.enc_x6 = self.exact_products and os.environ.get("CLSR_ENC_BWD", "fp32") == "x6"
        self.enc_back_x3 = x3d and True   # A/B: d(hist) and d TT from one pass over dPin (csrc/projx3.hip)
        self.att_l1_fwd_x6 = x6d and True   # A/B: second attention layer, forward, three pieces (csrc/attl1fwd.hip)
        # history-level attention backward in one launch (csrc/atthist.hip): two pieces in "fp32x3" / one-piece-compatible in
        # the speed mode, THREE pieces in "fp32"
        self.att_hist_bwd_x3 = True
        self.att_hist_bwd_pieces = 3 if self.exact_products else 2
        self.att_hist_x3 = True  # A/B: history-level attention prologue in one launch (csrc/atthist.hip)
        # (forward products that feed a batch-norm + ReLU stay at fp32 accuracy in both fp32-storage modes -- three pieces in
        # the history-level kernel, the fp32-MFMA layer-0 kernel -- see csrc/atthist.hip; the two-piece forms are opt-in
        # under "fp32x3": CLSR_ATT_FWD_X3=1, CLSR_ATT_HIST_PIECES=2)
        self.att_fwd_x3 = x3d and False
        # the per-(row, step) layer-0 product over three bf16 pieces per operand (fp32 accuracy, 60 bf16 MFMAs instead of 100
        # fp32 ones per tile: csrc/attl0fwd.hip): 102 -> 90 us alone, nothing in the step (three interleaved A/B runs) --
        # opt-in (CLSR_ATT_FWD_X6=1), the default stays the bit-exact fp32-MFMA form
        self.att_fwd_x6 = x6d and False
        self.att_l0_fwd_entry = ("clsr_att_l0_fwd_x3" if self.att_fwd_x3 else
                                 "clsr_att_l0_fwd_x6" if self.att_fwd_x6 else "clsr_att_l0_fwd")
        self.att_hist_pieces = 3 if x3d else 3  # A/B: first attention layer, forward (csrc/attl0fwd.hip)
        # speed mode, round 5b: the (row, step)-level attention layers on the SAME chain kernels as the parity mode
        # (csrc/attl0fwd.hip, attl1fwd.hip, attbwdx3.hip) with ONE bf16 piece per operand and bf16 storage of z0 / z1 / dz0:
        # the weight gradients dW1 / db1 / dWp ride inside the backward kernels (no clsr_hdw launches over z0 / dz1 / dz0,
        # no stored dz1).  CLSR_BF16_CHAIN=old: the position-tiled csrc/hgemm.hip kernels of rounds 2-4.
        self.bf16_chain = self.precision == "bf16" and os.environ.get("CLSR_BF16_CHAIN", "x1") == "x1"
        self.bf16_dw = True        # A/B switch: weight gradients on the bf16 matrix pipe
        self.bf16_bwd = True      # A/B switch: back-propagating products likewise
        self._cur_descs_h = []
        self._plans, self._plan_keep, self._cur_descs = {}, [], []
        self._sort_bytes = {}
        self._ws_tag = ""          # suffix of shared scratch buffers while a side-stream branch is recording
        # side streams are shared by every net of the process on this device (one net steps at a time): a second net
        # with four streams of its own puts eight hardware queues in play and its step takes 5.7 instead of 3.4 ms
        self._side = _SIDE_STREAMS.setdefault(str(torch.device(device)), {})
        self.rnn_first = True   # A/B switch (see forward)
        self.dw_batch_late = True   # A/B: merged launches of the attention / head weight gradients
        self.bn_bwd_fused = True   # A/B: coefficient + apply of the row-level batch-norm backward in one launch
        self.hist_grad_two = True   # A/B: dhist + dhist_lt summed inside the segmented sums
        self.l0_bwd_halves = True   # A/B: wide layer-0 backward as two launches of the x3 kernel over column halves
        self.dense_upd_dw = True   # A/B: dense regulariser + Adam on the weight-gradient stream
        self._dense_fork = None
        self.tick_early = True   # A/B: Adam clock in the first launch of the update phase
        self.dw_stream = True   # A/B switch (see _dw)
        self._dw_async = False
        # A/B switch (see _att_qh); the bf16 speed mode keeps the whole query in the per-(row, step) GEMM (K is cheap there)
        self.split_query = True
        # (speed mode, round 5: with the history-level share of the product term inside the fused history-level kernels
        # -- csrc/atthist.hip -- the split costs no launch any more and halves K of the per-(row, step) kernels;
        # CLSR_BF16_NO_SPLIT_QUERY=1: the whole query in those kernels, as before)
        self.bf16_split_query = True
        # history-level half of the short-term query folded into U (see _att_qh): pays off at every width once the per-row
        # half runs on the one-wave-per-history kernels (round 3; the position-tiled kernels needed Du >= 64)
        self.split_query_min = 16
        self.split_emb_grad = True   # A/B switch (embedding gradient sites)
        self.use_plans = not os.environ.get("CLSR_NO_PLAN")                # replay recorded launch sequences
        self.split_g2 = True             # A/B switch (causal GRU off the main launch)
        self.g2_stream = "@lt"           # A/B: which side stream runs the causal GRU's forward
        self._step_plans = {}
        self.dw_streams = 1
        self.fused_l0_bwd = True   # A/B switch (see _att_bwd)
        self.dw_batching = True          # A/B switch (see _dw_batched)
        self.flush_side = True          # A/B switch (dense path off the main stream)
        self.l1_bwd_2pass = True      # A/B switch (exact mode, see _att_bwd)
        self.dpin_h = (self.bf16 and self.bf16_dw and self.bf16_bwd and type(self) is CLSRNet
                       and True)          # bf16 dPin (speed mode, CLSR graph only)
        # where the long-term attention backward forks: beside the short-term one (exact mode: -40 us) or underneath the
        # backward-through-time launch (speed mode: the short-term backward is bandwidth bound there); CLSR_LT_BWD_EARLY=0|1
        # (round 2: off in speed mode, +30 us there; with the fused encoder tail the long-term chain had become the LAST thing
        #  to finish in that mode: early wins by 0.06 ms since round 3)
        self.lt_bwd_early = True
        self._dw_batch = None
        self._buf_allocs = 0
        # HIP stream priorities of the side streams (0 = normal; the callers' compute stream can be created with -1 = high)
        # branch tag -> stream tag.  The @aux branches share the @lt stream: compute + @lt + @dw0 = three HIP streams, so
        # that RCCL's stream is the FOURTH under data parallelism -- a fifth active queue (or a fourth next to a
        # high-priority compute stream) cost 4.0 -> 6.6 ms per step (r03, CLSR_FORCE_DP=1); on a single GPU the fold
        # measured 3.768 against 3.779 ms.  CLSR_FOLD_AUX=0: the round-2 layout with a stream of its own.
        self.stream_alias = {"@aux": "@lt"}
        self.side_priority = 0
        self.dw_priority = 0
        # Recurrences: hidden-to-hidden products as split-bf16 sums (csrc/rnn.hip, "x3") or fp32-input MFMAs ("fp32", bit-exact
        # fp32); with x3 the input projections of the GRUs and of the Time4LSTM blocks i | j | f run INSIDE the recurrence
        # launch from the history embeddings (no projection tensor, no GEMM in front of the T-serial chain), and the
        # Time4LSTM keeps its saved activations in a private tile-major image.  CLSR_RNN_PRODUCTS=fp32 restores the exact form.
        # precision="fp32": "x6" = the same kernels with THREE pieces per operand (fp32 accuracy; wide encoders fall back to the
        # fp32-input MFMAs inside the launch), "fp32" = fp32-input MFMAs everywhere
        self.rnn_products = os.environ.get("CLSR_RNN_PRODUCTS", "x6" if self.exact_products else "x3")
        if self.rnn_products not in ("x6", "fp32") + (() if self.exact_products else ("x3",)):
            raise ValueError("CLSR_RNN_PRODUCTS must be 'x6' or 'fp32' (precision='fp32'), also 'x3' otherwise")
        if self.rnn_products == "x6" and max(self.H, self.Du) > 48:
            self.rnn_products = "fp32"      # (wide encoders: no three-piece instance -- fp32-input MFMAs, same accuracy)
        self.rnn_fused_proj = self.rnn_products in ("x3", "x6") and not os.environ.get("CLSR_NO_RNN_FUSED_PROJ")
        # the Time4LSTM's K-fused time-gate projection as split products too (csrc/projx3.hip): three pieces in "fp32", two
        # in the other modes (it feeds the same sigmoid gates as the recurrences' two-piece hidden products there)
        self.proj_x3 = True
        self.proj_gate_pieces = 3 if self.exact_products else 2
        # ... and the whole input projection when it is NOT fused into the recurrence launch (hidden sizes > 48: configs[4])
        self.proj_tt = self.proj_x3 and True      # A/B: tanh time features in that kernel's prologue
        self.proj_x3_wide = self.proj_x3 and True
        self.gemm_wide_x3 = self.proj_x3_wide and True   # A/B: every plain wide product
        self.proj_bwd_pieces = 3 if self.exact_products else 2
        self.proj_wide_pieces = (2 if self.precision == "bf16" else 3)
        self.rnn_act_tiled = self.rnn_products in ("x3", "x6")
        # Attention-MLP backward (exact mode): "x3" = the two-pass layer-1 kernel and the one-pass layer-0 kernel as split-bf16
        # products with the weight gradients dW1 / db1 / dWp accumulated inside them (csrc/attbwdx3.hip: no separate
        # weight-gradient launches, no stored dz1); "fp32" = the fp32-MFMA kernels + clsr_pgemm_dw_partial beside them
        # precision="fp32": "x6" = the same kernels with THREE pieces per operand (fp32 accuracy; CLSR_ATT_BWD=fp32: the fp32-MFMA
        # kernels + separate weight-gradient launches)
        # ("x6l1": three pieces in the layer-1 kernel only, the layer-0 backward on fp32 MFMAs + its weight-gradient launch)
        self.set_att_bwd(os.environ.get("CLSR_ATT_BWD", "x6l1" if self.exact_products else "x3"))   # (before the three-piece recurrences: x6l1 3.19-3.22 ms, fp32 3.32-3.34; x6 with the per-iteration K = 16 weight-gradient products is level with x6l1: 3.05-3.08 both)
        self.fuse_tt = True   # A/B: time-gate blocks of the input projection as one product over [hist | TT]
        # the row-level heads (alpha gate, alpha / logit MLPs, loss, their backward) as two persistent launches with grid
        # barriers for the batch-norm statistics (csrc/headsfused.hip) instead of a chain of 22 dependent launches
        self.heads_fused = not os.environ.get("CLSR_NO_HEADS_FUSED")
        # weight gradients of wide layers (K, N >= 96: BASELINE configs[4]) by the 128 x 128-tile kernel (csrc/dwwide.hip)
        self.dw_wide = True
        # ... as two-piece split-bf16 products like the other weight gradients outside "fp32"; precision="fp32": three pieces
        # per operand (fp32 accuracy; CLSR_DW_WIDE=fp32: fp32-input MFMAs)
        self.dw_wide_entry = (("clsr_pgemm_dw_wide" if os.environ.get("CLSR_DW_WIDE", "x6") == "fp32" else "clsr_pgemm_dw_wide_x6")
                              if self.exact_products else "clsr_pgemm_dw_wide_x3")
        self._dw_batch_wide = None
        self._heads_defer = False
        self._early_lists = None
        self.heads_comm = None      # data-parallel runs: communicator of the fused heads (clsr_amd/p2p.py: HeadsComm)
        self.fused_logit_tail = True   # A/B: output layer + softmax loss + their backward in one launch
        self._defer_logit_out = False
        self.early_scatter = True   # A/B: row scatters of the user / target lookups beside the encoder-backward tail instead of behind it
        self.dhist_side = True      # A/B (speed mode): d(hist) product on the long-term stream (2.87 -> 2.84 ms)
        self.enc_bwd_fused_h = True   # A/B: the speed-mode (bf16 dPin) form of the fused encoder tail
        self.enc_bwd_fused = True   # A/B: one pass over dPin for the seven encoder-side weight gradients + d(hist) (csrc/encbwd.hip)
        # ... which also adds the long-term d(hist) and the history prologue's shares, so that the segmented sums run lean
        self.fold_hist_shares = not os.environ.get("CLSR_NO_FOLD_SHARES")
        self._fold_args, self._folded = None, False
        self.early_user_update = not os.environ.get("CLSR_NO_EARLY_USER_UPDATE")   # user tables: regulariser + Adam behind their row scatters
        self._updated_early = set()
        self.l0_fwd_wave = True      # A/B switch (exact mode, see _att_fwd)
        self.fused_l0_wu = True   # A/B: dU . Wu^T inside that kernel as well (time-neutral, two launches fewer)
        self._joins = []
        self._dw_pending, self._dw_tables, self._dw_after, self._rp_pending = {}, {}, {}, {}
        self.defer_dw = True       # one batched reduction of the weight-gradient partials per stream and step
        self.overlap = not os.environ.get("CLSR_NO_OVERLAP")   # run the long-term attention chain etc. on side streams (fork / join); off: one stream, every kernel alone (diagnosis: contention-free kernel times)
        self.sorted_hist_grad = True   # history-row gradients by sort + segmented sums (False: float atomics)
        # deterministic embedding gradients (csrc/segsum.hip): STABLE radix sort of every lookup site's ids + segmented sums
        # that write each row once, in a fixed order -- no float atomics; two runs of a step give bit-identical gradient
        # tables and clip norms.  CLSR_NO_DET_GRADS=1: the counting sort + atomics of rounds 2-3.
        self.det_grads = not os.environ.get("CLSR_NO_DET_GRADS")
        # step | beta1^t | beta2^t | lr_t | abort flag (raised on the device when a bounded wait of the step gives up: every
        # optimiser kernel then returns without touching anything; reported by check_abort)
        self.adam_state = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=self.device)
        # squared norms of the IndexedSlices pieces (16) + loss numerators (8): ONE buffer, so that the data-parallel
        # exchange sums them with one collective and no staging copies
        self.stats24 = torch.zeros(24, dtype=torch.float64, device=self.device)
        self.sumsq_tab, self.losses = self.stats24[:16], self.stats24[16:]
        self.ucount = torch.zeros(1, dtype=F32, device=self.device)
        self.last_shape = None
        self._counts_zeroed = self._ucount_zeroed = self._ticked = False
        self.dp_world = 1          # data-parallel world size (loss normalisers are global)
        self.dp_stats_hook = None  # optional callable(tensor): sum BN partial statistics across ranks
        self.dp_comm = None        # optional communicator handle (clsr_amd/p2p.py): the same sum as ONE kernel on this stream
        self.bn_sync_fused = True  # ... fused with the fold of the partial sums and the layer's finalisation (round 6; False: three launches)
        # data-parallel exchange hooks (clsr_amd/dp.py): called (and recorded into launch plans) at the points of the
        # step where a piece of the gradient state becomes final, so that its collective overlaps the rest of the
        # backward pass: flags_ready() | dense_ready() | table_ready(name)
        self.dp_hooks = None
        self.capture_grads = False
        self.captured = None
        self._aborted = None       # sticky reason of an aborted step (check_abort)

    def set_att_bwd(self, mode):
        """Form of the attention-MLP backward: 'x3' (two bf16 pieces, not with precision='fp32'), 'x6' (three pieces, both
        layers), 'x6l1' (three pieces in the layer-1 kernel, fp32-input MFMAs + a weight-gradient launch for layer 0) or
        'fp32' (fp32-input MFMAs + separate weight-gradient launches)."""
        if mode not in ("x6", "x6l1", "fp32") + (() if self.exact_products else ("x3",)):
            raise ValueError("CLSR_ATT_BWD must be 'x6', 'x6l1' or 'fp32' (precision='fp32'), also 'x3' otherwise")
        self.att_bwd_l0 = "fp32" if mode == "x6l1" else mode
        self.att_bwd = "x6" if mode == "x6l1" else mode
        self._l1x = "clsr_att_l1_bwd_x6" if self.att_bwd == "x6" else "clsr_att_l1_bwd_x3"
        self._l0x = "clsr_att_l0_bwd_x6" if self.att_bwd_l0 == "x6" else "clsr_att_l0_bwd_x3"

    # ------------------------------------------------------------------ configuration guard
    def _check_supported(self):
        hp = self.hp
        bad = []
        if hp.sequential_model not in ("time4lstm", "gru", "lstm"):
            bad.append("sequential_model=%r (time4lstm | gru | lstm)" % hp.sequential_model)
        if hp.enable_BN is not True:
            bad.append("enable_BN must be True")
        if list(hp.activation) != ["relu"] * len(hp.activation):
            bad.append("activation must be relu")
        if any(float(d) != 0.0 for d in hp.dropout) or float(hp.embedding_dropout) != 0.0 or hp.user_dropout:
            bad.append("dropout must be 0")
        if any(float(getattr(hp, k)) != 0.0 for k in ("cross_l1", "cross_l2")):
            bad.append("cross regularisers must be 0 (no model of this family has cross-layer parameters)")
        if hp.loss != "softmax" or hp.method != "classification":
            bad.append("loss must be softmax / method classification")
        if hp.optimizer not in ("adam", "lazyadam"):
            bad.append("optimizer must be adam or lazyadam")
        if len(hp.att_fcn_layer_sizes) != 2 or len(hp.layer_sizes) != 2:
            bad.append("att_fcn_layer_sizes / layer_sizes must have two layers")
        if hp.contrastive_loss not in ("bpr", "triplet"):
            bad.append("contrastive_loss must be bpr or triplet")
        D = hp.item_embedding_dim + hp.cate_embedding_dim
        if hp.hidden_size != D or hp.user_embedding_dim != D:
            bad.append("hidden_size and user_embedding_dim must equal item+cate dims (alpha fusion, clsr.py:265)")
        bad += self._shape_limits(hp, rnn=True)
        if self.precision == "bf16":
            # the speed-mode kernels (csrc/hgemm.hip, hdw.hip, hattbwd.hip) move 8 bf16 values per 16-byte access: K, N
            # and the leading dimensions of the attention block must be multiples of 8 -- said here, by name, instead of
            # as an 'unsupported shape' code of some launch in the middle of a step
            att = [int(w) for w in hp.att_fcn_layer_sizes]
            qs = (int(hp.user_embedding_dim), int(hp.user_embedding_dim) + D)
            if any(w % 8 for w in att) or any(q % 8 for q in qs) or int(hp.hidden_size) % 8:
                bad.append("precision='bf16' needs att_fcn_layer_sizes %r, the attention query widths %r and hidden_size "
                           "%d to be multiples of 8 (use precision='fp32')" % (att, qs, int(hp.hidden_size)))
        if bad:
            raise NotImplementedError("CLSR HIP path does not support: " + "; ".join(bad))

    @staticmethod
    def _shape_limits(hp, rnn):
        """Shape limits of the kernels (16-byte vector accesses, register-resident recurrent weights, 64-step
        chunks of the attention softmax), reported here by name instead of as an 'unsupported shape' error code
        of some launch in the middle of a step."""
        bad = []
        Di, Dc = int(hp.item_embedding_dim), int(hp.cate_embedding_dim)
        if Di % 4 or Dc % 4 or Di <= 0 or Dc <= 0 or Di + Dc > 256:
            bad.append("item / cate embedding dims must be positive multiples of 4, at most 256 together")
        if rnn and (int(hp.hidden_size) % 4 or not 4 <= int(hp.hidden_size) <= 128):
            bad.append("hidden_size must be a multiple of 4 in 4..128")
        if int(hp.max_seq_length) > 256:
            bad.append("max_seq_length must be at most 256")
        widths = list(hp.layer_sizes) + list(getattr(hp, "att_fcn_layer_sizes", None) or [])
        if any(int(w) % 4 or not 4 <= int(w) <= 1024 for w in widths):
            bad.append("MLP layer widths must be multiples of 4 in 4..1024")
        if any(int(w) > 256 for w in (getattr(hp, "att_fcn_layer_sizes", None) or [])):
            bad.append("att_fcn_layer_sizes must be at most 256 wide")
        return bad

    # ------------------------------------------------------------------ parameters
    # The variable inventory, the trained embedding tables and the recurrent encoders are hooks so that the sibling
    # models of the reference that share these kernels (clsr_amd/seqnet.py) reuse everything below.
    # ------------------------------------------------------------------ launch plans
    #: every mode switch that changes the recorded launch sequence: ONE tuple, so that flipping any of them on a live net (A/B
    #: runs on the same feed, tests) records a new plan instead of silently replaying the old one
    _SWITCHES = ("precision", "table_bf16", "exact_products", "overlap", "defer_dw", "sorted_hist_grad", "det_grads", "lazy",
                 "rnn_first", "tick_early", "bn_sync_fused", "hist_grad_two", "dw_batch_late", "bn_bwd_fused", "dw_stream", "dw_streams",
                 "split_query", "split_query_min", "bf16_split_query", "split_emb_grad", "bf16_chain", "bf16_dw", "bf16_bwd",
                 "fused_l0_bwd", "fused_l0_wu", "l0_fwd_wave", "l0_bwd_halves", "dw_batching", "lt_bwd_early", "dpin_h", "flush_side",
                 "l1_bwd_2pass", "split_g2", "g2_stream", "rnn_products", "rnn_fused_proj", "rnn_act_tiled", "att_bwd", "att_bwd_l0", "_l1x", "_l0x", "att_hist_x3",
                 "att_hist_bwd_x3", "att_hist_bwd_pieces", "att_hist_pieces", "att_l1_fwd_x6", "att_fwd_x3", "att_fwd_x6",
                 "att_l0_fwd_entry", "x3_enc", "enc_x6", "enc_back_x3", "enc_bwd_fused", "enc_bwd_fused_h", "fold_hist_shares", "early_user_update", "proj_x3", "proj_tt", "proj_x3_wide",
                 "gemm_wide_x3", "proj_gate_pieces", "proj_bwd_pieces", "proj_wide_pieces", "dhist_side", "early_scatter",
                 "fused_logit_tail", "fuse_tt", "heads_fused", "dense_upd_dw", "dw_wide", "dw_wide_entry", "rowlist_min_elems")

    def _plan_key(self, what, f):
        hp = self.hp
        g = lambda k: getattr(hp, k, None)
        return (what, id(f), ops.stream_ptr(), self.dp_world, id(self.dp_hooks), id(self.dp_stats_hook), self.dp_comm, self.heads_comm,
                getattr(self, "dp_generation", 0)) + tuple(getattr(self, n, None) for n in self._SWITCHES) + (
                g("learning_rate"), g("embed_l2"), g("layer_l2"), g("embed_l1"), g("layer_l1"), g("max_grad_norm"), g("is_clip_norm"),
                g("discrepancy_loss_weight"), g("contrastive_loss_weight"), g("triplet_margin"),
                g("contrastive_length_threshold"), g("manual_alpha_value"))

    def _planned(self, what, f, run):
        """Run a step eagerly, or replay its recorded launch sequence (``ops.LaunchPlan``) when this exact step --
        same uploaded feed object (its device arena is persistent), stream, scalars -- has run before.  The second
        occurrence is the one recorded: the first one allocates workspaces and builds descriptor tables."""
        if (not self.use_plans) or self.capture_grads or ops.recording():
            return run()
        key = self._plan_key(what, f)
        ent = self._step_plans.get(key)
        if ent is None:
            self._step_plans[key] = [f, None]          # keeps f alive: id(f) cannot be recycled
            return run()
        if ent[1] is None:
            plan = ops.record_begin()
            try:
                plan.result = run()
            finally:
                ops.record_end()
            ent[1] = plan
            ent.append(self.last_shape)
            return plan.result
        ops.replay(ent[1])
        self.last_shape = ent[2]
        return ent[1].result

    def _param_specs(self):
        return param_specs(self.dims, self.hp)

    def _table_map(self):
        return TABLES

    def _unused_tables(self):
        return (UNUSED_TABLE,)

    def _build_params(self, seed):
        hp, dev = self.hp, self.device
        gen = torch.Generator()
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(seed))
        specs = self._param_specs()
        self.specs = specs
        TABLES = self._table_map()
        table_names = set(TABLES.values()) | set(self._unused_tables())
        dense = [(n, s, k) for n, s, k in specs if n not in table_names]
        sizes = [_pad4(int(np.prod(s))) for _, s, _ in dense]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n_dense = int(off[-1])
        self.dense = torch.zeros(self.n_dense, dtype=F32, device=dev)
        self.dense_grad = None     # a view of grad_flat (below): dense and table gradients travel in ONE collective
        self.dense_m = torch.zeros_like(self.dense)
        self.dense_v = torch.zeros_like(self.dense)
        self.seg_off = torch.tensor(off, dtype=torch.int32, device=dev)
        # (dense regulariser: one workgroup per tensor -- 1 024 threads when a tensor is large: 128-wide encoders)
        self.dense_reg_threads = 1024 if max(sizes) >= 65536 else 0
        self.seg_of = torch.tensor(np.repeat(np.arange(len(sizes)), sizes), dtype=torch.int32, device=dev)
        self.dense_sumsq = torch.zeros(len(sizes), dtype=torch.float64, device=dev)
        self.dense_names = [n for n, _, _ in dense]
        self.P, self.Gd = OrderedDict(), OrderedDict()
        host = torch.zeros(self.n_dense, dtype=F32)
        self.tables, self.tab_grad, self.tab_m, self.tab_v, self.tab_flags = {}, {}, {}, {}, {}
        # gradient tables and involved-row flags live in ONE flat buffer each (one collective per kind)
        tshape = {k: [tuple(sh) for n, sh, _ in specs if n == v][0] for k, v in TABLES.items()}
        goff, foff = {}, {}
        gtot = ftot = 0
        for k, sh in tshape.items():
            goff[k], foff[k] = gtot, ftot
            # every gradient table starts on a 256-byte boundary: the 384-byte rows of a 96-wide item table are then three
            # whole 128-byte lines each, not four with two partial ones (the item-site segmented sums of configs[4] on a
            # table that began 16-byte aligned inside this buffer: 59 us; on an aligned one: 43.5)
            gtot += (sh[0] * sh[1] + 63) // 64 * 64
            ftot += (sh[0] + 15) // 16 * 16
        self.tab_goff, self.tab_foff, self.tab_shape = goff, foff, tshape   # layout of the two flat buffers
        # [dense gradients | table gradients | BN moving statistics]: with per-rank batch-norm the moving statistics
        # are averaged over the replicas by the SAME all-reduce that sums the gradients (clsr_amd/dp.py)
        n_bn = sum(2 * _pad4(int(np.prod(sh))) for n, sh, _ in dense if n.endswith("/gamma"))
        t0 = (self.n_dense + 63) // 64 * 64
        self.grad_flat = torch.zeros(t0 + gtot + n_bn, dtype=F32, device=dev)
        self.dense_grad = self.grad_flat[:self.n_dense]
        self.tab_grad_flat = self.grad_flat[t0:t0 + gtot]
        self.bn_moving = self.grad_flat[t0 + gtot:]
        self.tab_flags_flat = torch.zeros(ftot, dtype=torch.uint8, device=dev)
        it_dense = iter(zip(dense, off[:-1]))
        for name, shape, kind in specs:
            big = name in table_names and int(np.prod(shape)) > (1 << 27)
            val = None if big else init_tensor(kind, tuple(shape), hp, gen)
            if name in table_names:
                if big:   # 100M-item catalogues: initialise in place on the device (no 38 GB host copy)
                    t = torch.empty(tuple(shape), dtype=F32, device=dev)
                    v0 = float(hp.init_value)
                    torch.nn.init.trunc_normal_(t, mean=0.0, std=v0, a=-2 * v0, b=2 * v0)
                else:
                    t = val.to(dev)
                self.P[name] = t
                if name not in self._unused_tables():
                    key = [k for k, v in TABLES.items() if v == name][0]
                    if self.table_bf16:
                        t = t.to(torch.bfloat16)
                        self.P[name] = t
                    self.tables[key] = t
                    self.tab_grad[key] = self.tab_grad_flat[goff[key]:goff[key] + t.numel()].view(t.shape)
                    self.tab_m[key] = torch.zeros(t.shape, dtype=F32, device=dev)
                    self.tab_v[key] = torch.zeros(t.shape, dtype=F32, device=dev)
                    self.tab_flags[key] = self.tab_flags_flat[foff[key]:foff[key] + shape[0]]
            else:
                (_, _, _), o = next(it_dense)
                n = int(np.prod(shape))
                host[o:o + n] = val.reshape(-1)
                self.P[name] = self.dense[o:o + n].view(*shape)
                self.Gd[name] = self.dense_grad[o:o + n].view(*shape)
        self.dense.copy_(host)
        # batch-norm layers
        self.bn = {}
        for name in self.dense_names:
            if name.endswith("/gamma"):
                scope = name[:-len("gamma")]
                self.bn[scope] = _BN(self, scope, self.P[name].numel())
        # moving statistics of all layers as views of ONE buffer (a single collective averages them across ranks)
        o = 0
        for bn in self.bn.values():
            for attr in ("moving_mean", "moving_var"):
                view = self.bn_moving[o:o + bn.C]
                view.copy_(getattr(bn, attr))
                setattr(bn, attr, view)
                o += _pad4(bn.C)
        assert o == self.bn_moving.numel()

    def state_dict(self):
        """All variables under their TF names + BN moving stats + Adam slots (checkpoint payload).  Raises ``StepAborted``
        instead of handing out the state an aborted step left behind (check_abort)."""
        self.check_abort()
        sd = OrderedDict()
        for name, t in self.P.items():
            sd[name] = t.detach().float().cpu().clone()      # (bf16 tables: widened -- the payload is always fp32)
        for scope, bn in self.bn.items():
            sd[scope + "moving_mean"] = bn.moving_mean.cpu().clone()
            sd[scope + "moving_variance"] = bn.moving_var.cpu().clone()
        sd["__adam__/dense_m"] = self.dense_m.cpu().clone()
        sd["__adam__/dense_v"] = self.dense_v.cpu().clone()
        for k in self.tables:
            sd["__adam__/%s_m" % k] = self.tab_m[k].cpu().clone()
            sd["__adam__/%s_v" % k] = self.tab_v[k].cpu().clone()
        sd["__adam__/state"] = self.adam_state[:4].cpu().clone()
        return sd

    @staticmethod
    def _alias_old_names(sd):
        """Checkpoints written before round 3 named the Time4LSTM variables ``.../time4lstm/<var>``; TF puts the
        variables of a plain ``RNNCell`` under the cell's own layer scope (``.../time4lstm/time4lstm_cell/<var>``,
        reference cell rnn_cell_implement.py:46-128, call sites clsr.py:194-200 / sli_rec.py:43-57)."""
        out = None
        for k in list(sd.keys()):
            if "/time4lstm/" in k and "/time4lstm/time4lstm_cell/" not in k:
                if out is None:
                    out = dict(sd)
                new = k.replace("/time4lstm/", "/time4lstm/time4lstm_cell/")
                if new in out:
                    raise ValueError("checkpoint holds both spellings of one variable: %s and %s" % (k, new))
                out[new] = out.pop(k)
        return sd if out is None else out

    def load_state_dict(self, sd, strict=True):
        sd = self._alias_old_names(sd)
        for name, t in self.P.items():
            if name in sd:
                src = torch.as_tensor(np.asarray(sd[name]), dtype=F32)
                if tuple(src.shape) != tuple(t.shape):
                    raise ValueError("shape mismatch for %s: %s vs %s" % (name, tuple(src.shape), tuple(t.shape)))
                t.copy_(src)
            elif strict:
                raise KeyError(name)
        for scope, bn in self.bn.items():
            if scope + "moving_mean" in sd:
                bn.moving_mean.copy_(torch.as_tensor(np.asarray(sd[scope + "moving_mean"]), dtype=F32))
                bn.moving_var.copy_(torch.as_tensor(np.asarray(sd[scope + "moving_variance"]), dtype=F32))
            elif strict:
                raise KeyError(scope + "moving_mean")
        if "__adam__/dense_m" in sd:
            self.dense_m.copy_(torch.as_tensor(sd["__adam__/dense_m"]))
            self.dense_v.copy_(torch.as_tensor(sd["__adam__/dense_v"]))
            for k in self.tables:
                self.tab_m[k].copy_(torch.as_tensor(sd["__adam__/%s_m" % k]))
                self.tab_v[k].copy_(torch.as_tensor(sd["__adam__/%s_v" % k]))
            self.adam_state[:4].copy_(torch.as_tensor(sd["__adam__/state"])[:4])
            if self._aborted or float(self.adam_state[4].item()) != 0.0:
                self._reset_after_abort()       # a full checkpoint is the recovery path of an aborted step

    # ------------------------------------------------------------------ stream fork / join
    class _Branch(object):
        """``with net._branch("@tag"):`` -- launches inside go to a side HIP stream that first waits for
        everything enqueued so far on the current stream; scratch buffers get their own copies (tag).
        ``net._join()`` makes the current stream wait for every finished branch.  Works eagerly and under
        hipGraph capture (the event record / wait pairs become graph dependencies)."""

        def __init__(self, net, tag, after=None, name=None):
            self.net, self.tag, self.after, self.name = net, tag, after, name or tag

        def __enter__(self):
            net = self.net
            self.inline = (not net.overlap) or self.tag == "@main"     # "@main": stay on the current stream
            if self.inline:
                return self
            stag = net.stream_alias.get(self.tag, self.tag)      # (scratch buffers keep the branch's own tag)
            side = net._side.get(stag)
            if side is None:
                side = net._side_stream(stag)
            if side.cuda_stream == ops.current_stream().cuda_stream:
                # a branch onto the stream it is opened from (an @aux branch inside an @lt one, now that they share a
                # stream): plain stream order -- a stream waiting for its own event crashes hipGraph capture
                self.inline = True
                return self
            self.side = side
            ev = self.after
            if ev is None:
                ev = ops.event_record(ops.current_stream())
            ops.stream_wait(side, ev)
            self.old_tag, net._ws_tag = net._ws_tag, self.tag
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            self.scope = ops.stream_scope(side)
            self.scope.__enter__()
            return self

        def __exit__(self, *exc):
            net = self.net
            if self.inline:
                return False
            ev = ops.event_record(self.side)
            self.scope.__exit__(*exc)
            self.ctx.__exit__(*exc)
            net._ws_tag = self.old_tag
            net._joins.append((self.name, ev, self.side.cuda_stream))
            return False

    def _branch(self, tag, after=None, name=None):
        """``after``: an event already recorded on the current stream -- the branch then depends on the work up to
        that point only, so the caller can enqueue main-stream work FIRST and the branch second (launch order
        is what the device sees: a branch enqueued ahead of a long main-stream kernel delays that kernel)."""
        return CLSRNet._Branch(self, tag, after, name)     # name: what ``_join(only=...)`` refers to (default: tag)

    def _dp_hook(self, name, *args):
        """Tell the data-parallel exchange that a piece of the gradient state is final on the CURRENT stream (host
        call, recorded into the launch plan like a kernel launch)."""
        h = self.dp_hooks
        if h is not None:
            # the stream travels as an argument: a replayed plan does not re-enter the torch stream contexts
            ops.host_call(getattr(h, name), ops.current_stream(), *args)

    def _fork_point(self):
        if not self.overlap:
            return None
        return ops.event_record(ops.current_stream())

    def _join(self, only=None, but=None, keep=False):
        """The current stream waits for the finished branches (all of them, those named ``only``, or all ``but`` one).
        ``keep``: the branches stay on the list -- another stream will join them again."""
        main = ops.current_stream()
        rest, last = [], {}
        but = but if isinstance(but, tuple) else (but,)
        for tag, ev, sid in self._joins:
            if (only is None or tag == only) and tag not in but:
                if sid != main.cuda_stream:      # (work of this very stream is ordered already)
                    last[sid] = ev               # branches that shared a stream: its LAST event covers the earlier ones
                if keep:
                    rest.append((tag, ev, sid))
            else:
                rest.append((tag, ev, sid))
        for ev in last.values():                 # (every wait is a barrier packet of ~5 us in front of the next kernel)
            ops.stream_wait(main, ev)
        self._joins = rest

    def _side_stream(self, tag):
        """The side stream ``tag`` of this device, created on first use by whichever call site gets there first: the
        weight-gradient streams (@dw*) with CLSR_DW_PRIORITY, every other one with CLSR_SIDE_PRIORITY.  A new stream
        orders itself behind everything enqueued on the current stream so far -- the zero fills of workspaces allocated
        before it existed included (``_buf`` makes the streams that exist at allocation time wait; a stream created later
        in the same first step could still have written such a buffer ahead of its pending fill)."""
        side = self._side.get(tag)
        if side is None:
            prio = self.dw_priority if tag.startswith("@dw") else self.side_priority
            side = self._side[tag] = torch.cuda.Stream(device=self.device, priority=prio)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            side.wait_event(ev)
        return side

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, *shape, dtype=F32):
        """Named persistent workspace tensor (allocated zeroed on first use, then reused)."""
        key = (name,) + tuple(int(s) for s in shape) + (dtype,)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(*[int(s) for s in shape], dtype=dtype, device=self.device)
            self._bufs[key] = t
            self._buf_allocs += 1     # (the zero fill is a kernel on the CURRENT stream: see _dw_batched)
            # ... and every side stream orders itself behind it: a branch forked BEFORE this point but enqueued after
            # it could otherwise write the new buffer while the fill is still pending (first step of a net only; seen
            # as a run-to-run difference of the time-input gradients, tests/test_fullsize_gpu.py)
            if self._side:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                for side in self._side.values():
                    side.wait_event(ev)
        return t

    def _pack(self, key, W, out_f, in_f, transposed=False, W2=None, s2=1.0, in_pad=None, o0=0, i0=0,
              total=None):
        """Plan one packed weight block: dst rows = out features, cols = in features (MFMA A operand
        layout, row stride Kp).  ``W`` is an [in, out] view (``transposed``: an [out, in] view).  With
        ``total=(out_total, in_total)`` the block is placed at (o0, i0) of an assembled matrix.  Nothing is
        launched here: the descriptors are executed by ONE clsr_pack_batch launch per step."""
        out_t, in_t = total if total is not None else (out_f, in_f)
        Kp = ops.kp_for(in_pad or in_t)
        opad = 16 * ((out_t + 15) // 16)
        buf = self._buf("pack:" + key, opad * Kp)   # zero-initialised; padding is never written
        self.packed[key] = (buf, Kp)
        self._cur_descs.append(ops.pack_desc(W, out_f, in_f, buf, Kp, transposed=transposed, o0=o0, i0=i0,
                                             src2=W2, s2=s2))
        if self.bf16 and self.bf16_bwd and key.endswith("^T") and out_t % 8 == 0 and in_t % 8 == 0:
            # speed mode: the back-propagating products run on the bf16 matrix pipe too (csrc/hgemm.hip: clsr_hgemm_f32)
            hKp = query("clsr_hgemm_kp", in_t)
            hbuf = self._buf("packh:" + key, 32 * ((out_t + 31) // 32) * hKp, dtype=torch.bfloat16)
            self.packed_h[key] = (hbuf, hKp)
            self._cur_descs_h.append(ops.pack_desc(W, out_f, in_f, hbuf, hKp, transposed=transposed, o0=o0, i0=i0,
                                                   src2=W2, s2=s2))

    def _pack_h(self, key, W, out_f, in_f, transposed=False):
        """bf16 image of one weight block for the speed-mode kernels (row-permuted pairs of MFMA tiles, row stride
        ``clsr_hgemm_kp(in)``; executed by ONE clsr_pack_batch_bf16 launch per step)."""
        Kp = query("clsr_hgemm_kp", in_f)
        opad = 32 * ((out_f + 31) // 32)
        buf = self._buf("packh:" + key, opad * Kp, dtype=torch.bfloat16)   # zero-initialised; padding is never written
        self.packed_h[key] = (buf, Kp)
        self._cur_descs_h.append(ops.pack_desc(W, out_f, in_f, buf, Kp, transposed=transposed))

    @staticmethod
    def _hgemm_fits(K, N):
        """does the bf16 weight image of a K -> N product fit the LDS of csrc/hgemm.hip's position-tiled kernel?  (Not at
        K = 1 536 -> 128, d(hist) of the 128-wide encoders: the speed mode then takes the fp32 route -- clsr_proj_x3_wide.)"""
        np_ = 2 if N <= 64 else (3 if (N <= 96 or 128 < N <= 192) else 4)      # (hgemm_np, csrc/hgemm.hip)
        return 32 * np_ * (32 * ((K + 31) // 32) + 8) * 2 <= 150 * 1024

    def _pack_pair(self, key, W, K, N, K_pad=None):
        """forward pack (K->N) and transposed pack (N->K) of the same [K, N] block."""
        self._pack(key, W, N, K, in_pad=K_pad)
        self._pack(key + "^T", W, K, N, transposed=True)

    def _gemm(self, X, ldx, wkey, M, K, N, Y, ldy, bias=None, T=0, G=0, Xmul=None, ldmul=0, aff=None,
              addU=None, ldu=0, addV=None, ldv=0, acc=0, stats=None):
        if (self.bf16 and wkey in self.packed_h and wkey.endswith("^T") and bias is None and Xmul is None and aff is None
                and addU is None and addV is None and stats is None and T == 0 and K % 8 == 0 and N % 8 == 0
                and self._hgemm_fits(K, N)):
            Wt, Kp = self.packed_h[wkey]
            call("clsr_hgemm_hf32" if X.dtype == torch.bfloat16 else "clsr_hgemm_f32", X, ldx, Wt, Kp, Y, ldy, acc, M, K, N)
            return
        Wt, Kp = self.packed[wkey]
        if (self.gemm_wide_x3 and Xmul is None and aff is None and addU is None and addV is None and stats is None and T == 0
                and X.dtype == F32 and Y.dtype == F32 and M >= 32768 and (K >= 128 or N >= 128) and K <= 4096 and K % 8 == 0 and ldx % 4 == 0
                and query("clsr_proj_x3_wide_supported", M, K, N)):
            # plain position-level products of WIDE layers (BASELINE configs[4]): operands in registers, 128 output columns
            # per workgroup column, K in slabs of 128 (csrc/projx3.hip) -- the position-tiled fp32 kernel ran these at
            # 0.3-0.4 of the fp32 matrix peak (profiles/r05_catalogue_pmc.md)
            # (back-propagating products -- transposed weights -- take two pieces like every other backward product of the
            # default modes: half the MFMAs, two workgroups per CU instead of one; CLSR_PROJ_BWD_PIECES=3: as the forward ones)
            pieces = self.proj_bwd_pieces if wkey.endswith("^T") else self.proj_wide_pieces
            call("clsr_proj_x3_wide", X, ldx, Wt, Kp, bias, Y, ldy, M, K, N, pieces, int(acc))
            return
        sc, sh = (aff.scale, aff.shift) if aff is not None else (None, None)
        call("clsr_pgemm", X, ldx, T, G, Xmul, ldmul, sc, sh, 1, Wt, Kp, bias, addU, ldu, addV, ldv, Y, ldy,
             acc, stats, M, K, N)

    def _dw(self, X, ldx, dY, ldy, M, K, N, dW, ldw, db=None, T=0, G=0, Xmul=None, ldmul=0, aff=None, acc=0,
            x_bf16=0, dy_bf16=0):
        """Weight gradient dW = f(X)^T dY.  Deferred: this launches only the kernel that writes the per-block
        partial chunks (into a workspace of its own); ``_dw_flush`` reduces every pending gradient of the current
        stream in ONE launch."""
        if (self.dw_wide and not x_bf16 and not dy_bf16 and aff is None and T == 0 and G == 0
                and not (self.bf16 and self.bf16_dw) and not self.x3_dw and query("clsr_pgemm_dw_wide_supported", M, K, N)):
            return self._dw_wide(X, ldx, Xmul, ldmul, dY, ldy, M, K, N, dW, ldw, db, acc)
        pend = self._dw_pending.setdefault(self._ws_tag, [])
        need = query("clsr_pgemm_dw_workspace_floats", M, K, N)
        ws = self._buf("dw_ws%s.%d" % (self._ws_tag, len(pend)), max(need, 1))
        sc, sh = (aff.scale, aff.shift) if aff is not None else (None, None)
        if self._dw_batch is not None and not x_bf16 and not (dy_bf16 and not (self.bf16 and self.bf16_dw)):
            # inside _dw_batched(): the product joins the ONE multi-job launch issued when the block ends
            ptr = lambda t: 0 if t is None else t.data_ptr()
            self._dw_batch.append((ptr(X), ptr(Xmul), ptr(sc), ptr(sh), ptr(dY), ptr(ws), 0, ldx, T, G, ldmul, 1,
                                   int(dy_bf16), ldy, M, K, N, 0))
        elif self.dw_stream and self.overlap and self._ws_tag == "":
            # nothing on the main chain needs a weight gradient before the flush: the partial-sum kernels of the
            # main stream go to a stream of their own (inputs are final at this point and stay untouched until the
            # flush joins that stream), so they run beside the back-propagating GEMMs instead of between them.
            # ONE such stream: main + @lt + @aux + @dw0 = 4 concurrently active streams; a fifth one was measured at
            # 7.2 ms/step with GPU_MAX_HW_QUEUES=8 (hardware-queue oversubscription) -- do not add streams
            name = "@dw%d" % (len(pend) % self.dw_streams)
            side = self._side.get(name)
            if side is None:
                side = self._side_stream(name)
            ops.stream_wait(side, self._fork_point())
            self._dw_launch(X, x_bf16, ldx, T, G, Xmul, ldmul, sc, sh, dY, dy_bf16, ldy, M, K, N, ws, side.cuda_stream)
            self._dw_async = True
        else:
            self._dw_launch(X, x_bf16, ldx, T, G, Xmul, ldmul, sc, sh, dY, dy_bf16, ldy, M, K, N, ws, None)
        pend.append((ws.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, 1.0,
                     query(self._dw_parts_query(x_bf16 or dy_bf16), M), K, N, ldw, acc))
        if not self.defer_dw:
            self._dw_flush()

    def _dw_wide(self, X, ldx, Xmul, ldmul, dY, ldy, M, K, N, dW, ldw, db, acc):
        """Wide layer (K, N >= 96): partial tiles + their sum as two launches of csrc/dwwide.hip, on the weight-gradient
        stream like the other products (inside ``_dw_batched``: behind the block's multi-job launch); the gradient is
        complete when ``_dw_flush`` has joined that stream -- nothing is left for the batched reduction."""
        key = "dww_ws%s.%d.%d.%x" % (self._ws_tag, K, N, dW.data_ptr())      # (one workspace per gradient: jobs of different streams never share)
        ws = self._buf(key, query("clsr_pgemm_dw_wide_workspace_floats", M, K, N))
        job = (X, ldx, Xmul, ldmul, dY, ldy, M, K, N, ws, dW, ldw, db, acc)
        if self._dw_batch is not None:
            self._dw_batch_wide.append(job)
            return
        side = None
        if self.dw_stream and self.overlap and self._ws_tag == "":
            side = self._side_stream("@dw0")
            ops.stream_wait(side, self._fork_point())
            self._dw_async = True
        call(self.dw_wide_entry, *job, stream=side.cuda_stream if side is not None else None)

    @contextlib.contextmanager
    def _dw_batched(self, late=False):
        """Every ``_dw`` issued inside the block becomes a job of ONE launch (clsr_*_dw_partial_multi) that waits only
        for what had been enqueued on the current stream when the block was ENTERED: the encoders' hidden-to-hidden /
        time-feature / input-side weight gradients all read the finished dPin and forward activations.  One after the
        other on the weight-gradient stream those ~7 products were the last 350 us of the backward pass."""
        if not self.dw_batching or not self.defer_dw or self._dw_batch is not None or (late and not self.dw_batch_late):
            yield
            return
        fork = self._fork_point()
        allocs = self._buf_allocs
        self._dw_batch, self._dw_batch_wide = [], []
        try:
            yield
        finally:
            jobs, self._dw_batch = self._dw_batch, None
            wide, self._dw_batch_wide = self._dw_batch_wide, None
        if not jobs and not wide:
            return
        if late or self._buf_allocs != allocs:
            # ``late``: the jobs' operands are produced INSIDE the block (small products of one backward stage whose
            # launches are merged for the launch count, not for overlap): the launch waits for the block's end.
            # Otherwise, first occurrence of this shape: workspaces were allocated inside the block, and their zero
            # fills sit on the current stream BEHIND the entry point -- this once the launch waits for everything
            # enqueued so far
            fork = self._fork_point()
        name = ("clsr_hdw_partial_multi" if (self.bf16 and self.bf16_dw) else
                "clsr_pgemm_dw_partial_multi")
        if self.dw_stream and self.overlap and self._ws_tag == "":
            side = self._side.get("@dw0")
            if side is None:
                side = self._side_stream("@dw0")
            ops.stream_wait(side, fork)
            if jobs:
                ops.dw_multi(name, jobs, stream=side.cuda_stream)
            for job in wide:
                call(self.dw_wide_entry, *job, stream=side.cuda_stream)
            self._dw_async = True
        else:
            if jobs:
                ops.dw_multi(name, jobs)
            for job in wide:
                call(self.dw_wide_entry, *job)

    def _dw_parts_query(self, any_bf16=False):
        """which query tells how many partial chunks the weight-gradient kernel of this mode writes"""
        if self.bf16 and self.bf16_dw:
            return "clsr_hdw_parts"
        return "clsr_pgemm_dw_parts"

    def _dw_launch(self, X, x_bf16, ldx, T, G, Xmul, ldmul, sc, sh, dY, dy_bf16, ldy, M, K, N, ws, stream):
        """Partial-sum kernel of one weight gradient: the exact fp32-MFMA kernel, or -- speed mode -- the bf16-MFMA
        one (csrc/hdw.hip: operands rounded to bf16 when staged, fp32 accumulation), same partial layout."""
        if self.bf16 and self.bf16_dw:
            call("clsr_hdw_partial", X, x_bf16, ldx, T, G, Xmul, ldmul, sc, sh, 1, dY, dy_bf16, ldy, M, K, N, ws,
                 stream=stream)
        elif x_bf16 or dy_bf16:
            call("clsr_pgemm_dw_partial_h", X, x_bf16, ldx, T, G, Xmul, ldmul, sc, sh, 1, dY, dy_bf16, ldy, M, K, N, ws,
                 stream=stream)
        else:
            call("clsr_pgemm_dw_partial", X, ldx, T, G, Xmul, ldmul, sc, sh, 1, dY, ldy, M, K, N, ws, stream=stream)

    def _dw_fused(self, ws, parts, K, N, dW, ldw, db=None, acc=0):
        """A weight gradient whose ``parts`` partial chunks were written by the back-propagating kernel itself
        (csrc/attbwdx3.hip) on the CURRENT stream: only the entry of the batched reduction is left."""
        self._dw_pending.setdefault(self._ws_tag, []).append(
            (ws.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, 1.0, parts, K, N, ldw, acc))
        if not self.defer_dw:
            self._dw_flush()

    def _rp(self, partial, parts, stride, n, out):
        """out[0:n] = sum over ``parts`` per-block partial rows (deferred to ``_dw_flush`` like the dW reductions;
        the partial buffer must stay untouched until then)."""
        self._rp_pending.setdefault(self._ws_tag, []).append(
            (partial.data_ptr(), out.data_ptr(), 1.0, parts, stride, n, 0, 0))

    def _dw_flush(self, tag=None):
        """Reduce the partial chunks of every ``_dw`` issued under workspace tag ``tag`` (default: the current
        stream's) since the last flush, then run the operations that were waiting for those gradients."""
        tag = self._ws_tag if tag is None else tag
        pend = self._dw_pending.pop(tag, [])
        if tag == "" and self._dw_async:
            cur = ops.current_stream()
            for i in range(self.dw_streams):
                side = self._side.get("@dw%d" % i)
                if side is not None and side.cuda_stream != cur.cuda_stream:   # (a stream does not wait for itself:
                    ops.stream_wait(cur, ops.event_record(side))              #  under hipGraph capture that crashed)
            self._dw_async = False
        if pend:
            sig = tuple(pend)
            tab = self._dw_tables.get(sig)
            if tab is None:   # descriptor table: built and uploaded once per shape signature
                tab = self._dw_tables[sig] = ops.dw_table(sig, self.device)
            call("clsr_dw_reduce_batch", tab[0], tab[1], tab[2])
        rp = self._rp_pending.pop(tag, [])
        if rp:
            ops.multi("clsr_reduce_parts_multi", ops.RpDesc, rp)
        for fn in self._dw_after.pop(tag, []):
            fn()

    def _stats_buf(self, M, N):
        parts = query("clsr_pgemm_stats_parts", M)
        return self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * N], parts

    def _bn_fwd(self, bn, stats, parts, count, training):
        if training and self.dp_stats_hook is not None and self.dp_comm is not None and 2 * bn.C <= 256 and self.bn_sync_fused:
            # SyncBN in ONE launch: partial sums folded, exchanged with the peers and finalised inside the kernel (csrc/p2p.hip)
            call("clsr_bn_finalize_sync", self.dp_comm, stats, parts, bn.C, float(count * self.dp_world), bn.gamma, bn.beta,
                 bn.moving_mean, bn.moving_var, BN_MOMENTUM, BN_EPS, bn.scale, bn.shift, bn.mean, bn.invstd)
            return
        if training and self.dp_stats_hook is not None:   # SyncBN: global batch statistics
            stats, parts = self._dp_sum_stats(stats, parts, bn.C)
            count = count * self.dp_world
        call("clsr_bn_finalize", stats, parts, bn.C, float(count), bn.gamma, bn.beta, bn.moving_mean,
             bn.moving_var, BN_MOMENTUM, BN_EPS, 1 if training else 0, bn.scale, bn.shift, bn.mean, bn.invstd)

    def _dp_sum_stats(self, stats, parts, C):
        """Sync-BN: fold the per-block partial sums into one row of 2*C doubles and sum THAT across the ranks."""
        one = self._buf("dp.stats" + self._ws_tag, 2 * 1024, dtype=torch.float64)[: 2 * C]
        call("clsr_sum_parts_d", stats, parts, 2 * C, one)
        if self.dp_comm is not None and 2 * C <= 256:
            # peer-to-peer sum over the mapped exchange buffers: a plain C-ABI launch on this stream (part of the launch
            # plan like any other), no host callback, no hop into RCCL's stream
            call("clsr_allreduce_small", self.dp_comm, one, 2 * C)
        else:
            ops.host_call(self.dp_stats_hook, one, ops.current_stream())
        return one, 1

    def _bn_bwd_from_partial(self, bn, part, parts, dy, z, M):
        if self.dp_stats_hook is None and M * bn.C <= (1 << 23) and bn.C % 4 == 0 and self.bn_bwd_fused:
            # row-level layers (alpha / logit MLPs): coefficients + apply in one launch
            call("clsr_bn_bwd_coef_apply", part, parts, bn.C, float(M), bn.gamma, bn.mean, bn.invstd, bn.coef,
                 bn.dgamma, bn.dbeta, dy, z, M)
            return
        self._bn_bwd_coef(bn, part, parts, M)
        call("clsr_bn_bwd_apply", dy, z, bn.coef, M, bn.C)

    def _bn_bwd_coef(self, bn, part, parts, M):
        if self.dp_stats_hook is not None and self.dp_comm is not None and 2 * bn.C <= 256 and self.bn_sync_fused:
            call("clsr_bn_bwd_coef_sync", self.dp_comm, part, parts, bn.C, float(M * self.dp_world), bn.gamma, bn.mean, bn.invstd,
                 bn.coef, bn.dgamma, bn.dbeta, 1.0 / self.dp_world)
            return
        count = M
        if self.dp_stats_hook is not None:
            part, parts = self._dp_sum_stats(part, parts, bn.C)
            count = M * self.dp_world
        # (synchronised statistics: the partial sums are already global -- dgamma / dbeta leave pre-divided by the world
        # size so that the gradient all-reduce (SUM) restores them)
        call("clsr_bn_bwd_coef_scaled", part, parts, bn.C, float(count), bn.gamma, bn.mean, bn.invstd, bn.coef,
             bn.dgamma, bn.dbeta, 0, 1.0 / self.dp_world if self.dp_stats_hook is not None else 1.0)

    def _gemm_bnbwd(self, dY, ldy_in, wkey, M, K, N, out, bn, z):
        """out = relu-mask(dY . W^T) fused with the BN backward sums of layer ``bn`` (pre-BN activation z),
        then the coefficient + apply pass: ``out`` ends up holding dz of that layer."""
        Wt, Kp = self.packed[wkey]
        parts = query("clsr_pgemm_stats_parts", M)
        st = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * N]
        call("clsr_pgemm_bnbwd", dY, ldy_in, Wt, Kp, out, N, z, N,
             bn.scale, bn.shift, bn.mean, bn.invstd, st, M, K, N)
        self._bn_bwd_from_partial(bn, st, parts, out, z, M)

    def _bn_relu_bwd(self, bn, dh, z, M):
        parts = query("clsr_colred_parts", M, bn.C)
        part = self._buf("colred" + self._ws_tag, 2048 * 2 * 256, dtype=torch.float64)[: parts * 2 * bn.C]
        call("clsr_bn_relu_bwd_reduce", dh, z, bn.scale, bn.shift, bn.mean, bn.invstd, M, bn.C, part)
        self._bn_bwd_from_partial(bn, part, parts, dh, z, M)

    # ------------------------------------------------------------------ weight packing (per step)
    def _xw_blocks(self):
        """Column layout of the fused history projection  hist . [all input-side weight blocks]:
        list of (encoder key, kind, weight view [D, w], bias view, column offset, width)."""
        P, D, H = self.P, self.enc_in, self.H
        out, off = [], 0
        for key, scope, n in self._gru_list():
            for kind, W, b, w in (("gx", P[scope + "gates/kernel"][0:D], P[scope + "gates/bias"], 2 * n),
                                  ("cx", P[scope + "candidate/kernel"][0:D], P[scope + "candidate/bias"], n)):
                out.append((key, kind, W, b, off, w))
                off += w
        if self._t4_kind == "time4lstm":
            t = self._t4_scope
            for kind, W, b, w in (("kx", P[t + "kernel"][0:D], P[t + "bias"], 4 * H),
                                  ("w1", P[t + "_time_kernel_w1"], P[t + "_time_bias1"], H),
                                  ("w2", P[t + "_time_kernel_w2"], P[t + "_time_bias2"], H)):
                out.append(("t4", kind, W, b, off, w))
                off += w
        elif self._t4_kind == "lstm":
            # tf.nn.rnn_cell.LSTMCell (clsr.py:209-216) == the Time4LSTM recurrence with both time gates stuck
            # open: constant pre-activations of +30 (sigmoid(30) == 1.0f) instead of learned time projections
            g = self._t4_scope
            zero_w = self._buf("lstm.const_w", D, H)
            open_b = self._bufs.get("lstm.const_b")
            if open_b is None:
                open_b = self._bufs["lstm.const_b"] = torch.full((H,), 30.0, dtype=F32, device=self.device)
            for kind, W, b, w in (("kx", P[g + "kernel"][0:D], P[g + "bias"], 4 * H),
                                  ("const", zero_w, open_b, H), ("const", zero_w, open_b, H)):
                out.append(("t4", kind, W, b, off, w))
                off += w
        return out, off

    @property
    def _t4_kind(self):
        """"time4lstm" | "lstm" | None: the LSTM-type encoder of the model, if any."""
        sm = self.hp.sequential_model
        return sm if sm in ("time4lstm", "lstm") else None

    @property
    def _t4_scope(self):
        """Variable scope of the LSTM-type short-term encoder (None for the GRU encoder)."""
        kind = self._t4_kind
        if kind == "time4lstm":
            return CL + "short_term/time4lstm/time4lstm_cell/"
        if kind == "lstm":
            return CL + "short_term/simple_lstm/lstm_cell/"
        return None

    def _enc_off(self, key):
        blocks, _ = self._xw_blocks()
        return min(off for k, _, _, _, off, _ in blocks if k == key)

    def _pack_all(self, training):
        """Fill every packed weight buffer of the step with one batched launch (the plan -- buffers +
        device descriptor table -- is built on first use per mode)."""
        plan = self._plans.get(bool(training))
        if plan is None:
            self._cur_descs, self._cur_descs_h = [], []
            self._plan_weights(training)
            plan = ops.pack_table(self._cur_descs, self.device)
            self._plans[bool(training)] = plan
            self._plan_keep.append(self._cur_descs)   # descriptors hold raw pointers of live tensors
            if self._cur_descs_h:
                self._plans[(bool(training), "bf16")] = ops.pack_table(self._cur_descs_h, self.device)
                self._plan_keep.append(self._cur_descs_h)
        tbl, n, max_elems = plan
        call("clsr_pack_batch", tbl, n, max_elems)
        plan_h = self._plans.get((bool(training), "bf16"))
        if plan_h is not None:
            call("clsr_pack_batch_bf16", plan_h[0], plan_h[1], plan_h[2])

    def _att_qh(self, key):
        """Leading query columns of attention ``key`` that are history-level.  The short-term query is
        ``[short_term_intention[h] | target[r]]`` (clsr.py:219): its first Du columns do not depend on the candidate,
        so their share of the product term ``(a*q).Wp`` is computed once per history (Hn*T positions) and folded
        into U[h,t]; only the target half stays in the per-(row, step) GEMM (K = D instead of Du + D).  Pays off
        when the layers are wide (BASELINE configs[4], Du = 128: 15.15 -> 14.36 ms/step); at Du = 40 the big GEMMs
        are bound by their stores, not by K, and the five extra launches cost 1 % -- hence the width threshold."""
        if not (key == "st" and self.split_query and self.Du >= self.split_query_min):
            return 0
        if self.precision == "bf16":
            # speed mode: only with the fused history-level kernels and the one-pass layer-0 backward (default widths)
            ok = (self.bf16_split_query and self.att_hist_x3 and self.att_hist_bwd_x3 and self.fused_l0_bwd
                  and query("clsr_att_hist_fwd_x3_supported", self.H, self.Du + self.D, self.A0, self.Du)
                  and query("clsr_att_hist_bwd_x3_supported", self.H, self.Du + self.D, self.A0, self.Du)
                  and query("clsr_att_l0_bwd_h_supported", self.G_train, self.D, self.A0))
            return self.Du if ok else 0
        return self.Du

    def _plan_weights(self, training):
        hp, P = self.hp, self.P
        D, Du, H, A0, A1 = self.D, self.Du, self.H, self.A0, self.A1
        self._plan_att("lt", CL + "long_term/attention_fcn/", D, Du, training)
        self._plan_att("st", CL + "short_term/attention_fcn/", H, Du + D, training)
        self._plan_encoders(training)
        pair = self._pack_pair if training else (lambda k, W, K, N, K_pad=None: self._pack(k, W, N, K, in_pad=K_pad))
        if not hp.manual_alpha:
            a = CL + "fcn_alpha/nn_part/"
            pair("al.W0", P[a + "w_nn_layer0"], self.a_in, A0, K_pad=_pad4(self.a_in))
            pair("al.W1", P[a + "w_nn_layer1"], A0, A1)
        lg = "sequential/logit_fcn/nn_part/"
        pair("lg.W0", P[lg + "w_nn_layer0"], 2 * D, self.L0)
        pair("lg.W1", P[lg + "w_nn_layer1"], self.L0, self.L1)

    def _plan_att(self, key, scope, Dk, Q, training):
        """Packed weights of one ``_attention_fcn`` block (projection, re-associated first layer, second layer)."""
        P, A0, A1 = self.P, self.A0, self.A1
        W0 = P[scope + "att_fcn/nn_part/w_nn_layer0"]
        W1 = P[scope + "att_fcn/nn_part/w_nn_layer1"]
        self._pack(key + ".A", P[scope + "attention_mat"], Q, Dk)
        self._pack(key + ".Wu", W0[0:Q], A0, Q, W2=W0[2 * Q:3 * Q], s2=1.0)
        self._pack(key + ".Wv", W0[Q:2 * Q], A0, Q, W2=W0[2 * Q:3 * Q], s2=-1.0)
        self._pack(key + ".Wp", W0[3 * Q:4 * Q], A0, Q)
        qh = self._att_qh(key)
        if qh:   # product-term weights split into the history-level and the per-row query columns
            self._pack(key + ".Wp1", W0[3 * Q:3 * Q + qh], A0, qh)
            self._pack(key + ".Wp2", W0[3 * Q + qh:4 * Q], A0, Q - qh)
        self._pack(key + ".W1", W1, A1, A0)
        if self.bf16:
            self._pack_h(key + ".Wp", W0[3 * Q:4 * Q], A0, Q)
            if qh:
                self._pack_h(key + ".Wp2", W0[3 * Q + qh:4 * Q], A0, Q - qh)
                if training:
                    self._pack_h(key + ".Wp2^T", W0[3 * Q + qh:4 * Q], Q - qh, A0, transposed=True)
            self._pack_h(key + ".W1", W1, A1, A0)
            if training:
                self._pack_h(key + ".Wp^T", W0[3 * Q:4 * Q], Q, A0, transposed=True)
                self._pack_h(key + ".W1^T", W1, A0, A1, transposed=True)
        if training:
            self._pack(key + ".A^T", P[scope + "attention_mat"], Dk, Q, transposed=True)
            self._pack(key + ".Wu^T", W0[0:Q], Q, A0, transposed=True, W2=W0[2 * Q:3 * Q], s2=1.0)
            self._pack(key + ".Wv^T", W0[Q:2 * Q], Q, A0, transposed=True, W2=W0[2 * Q:3 * Q], s2=-1.0)
            self._pack(key + ".Wp^T", W0[3 * Q:4 * Q], Q, A0, transposed=True)
            if qh:
                self._pack(key + ".Wp1^T", W0[3 * Q:3 * Q + qh], qh, A0, transposed=True)
                self._pack(key + ".Wp2^T", W0[3 * Q + qh:4 * Q], Q - qh, A0, transposed=True)
            self._pack(key + ".W1^T", W1, A0, A1, transposed=True)

    def _plan_encoders(self, training):
        """Fused history projection: every input-side weight block of every encoder side by side."""
        hp, P, D, H = self.hp, self.P, self.enc_in, self.H
        blocks, NX = self._xw_blocks()
        self.NX = NX
        bias = self._buf("xw.bias", NX)
        for _, _, W, b, off, w in blocks:
            self._pack("xw", W, w, D, o0=off, total=(NX, D))
            self._cur_descs.append(ops.pack_desc(b, w, 1, bias, 1, ld1=1, o0=off))
            if training:
                self._pack("xw^T", W, D, w, transposed=True, i0=off, total=(D, NX))
        if self._t4_kind == "time4lstm":
            t = self._t4_scope
            # [Tn | Tl] (2H) -> [o | tns | tls] (3H) block matrix of the four time kernels
            for nm, o0, i0 in (("_o_kernel_t1", 0, 0), ("_o_kernel_t2", 0, H), ("_time_kernel_t1", H, 0),
                               ("_time_kernel_t2", 2 * H, H)):
                self._pack("t4.tw", P[t + nm], H, H, o0=o0, i0=i0, total=(3 * H, 2 * H))
                if training:
                    self._pack("t4.tw^T", P[t + nm], H, H, transposed=True, o0=i0, i0=o0, total=(2 * H, 3 * H))
            if self._fuse_tt_ok():
                # K-fused projection of the three time-gate blocks [o | tns | tls]:  [hist | pad | TT] . [W_x ; 0 ; W_t]
                Dp = 16 * ((D + 15) // 16)
                for nm, o0, i0 in (("_o_kernel_t1", 0, 0), ("_o_kernel_t2", 0, H), ("_time_kernel_t1", H, 0),
                                   ("_time_kernel_t2", 2 * H, H)):
                    self._pack("xw.t", P[t + nm], H, H, o0=o0, i0=Dp + i0, total=(3 * H, Dp + 2 * H))
                for o0, W in ((0, P[t + "kernel"][0:D][:, 3 * H:4 * H]), (H, P[t + "_time_kernel_w1"]),
                              (2 * H, P[t + "_time_kernel_w2"])):
                    self._pack("xw.t", W, H, D, o0=o0, i0=0, total=(3 * H, Dp + 2 * H))

    def _fuse_tt_ok(self):
        """One product [hist | TT] . [W_x ; W_t] for the three time-gate blocks of the Time4LSTM projection (the last 3H
        columns of the fused input projection) instead of hist . W_x followed by an accumulating TT . W_t pass over them."""
        return (self.fuse_tt and self._t4_kind == "time4lstm" and self.enc_in % 4 == 0
                and self._enc_off("t4") + 6 * self.H == self._xw_blocks()[1])

    def _unpack_grads(self):
        """Scatter the assembled gradient blocks (fused projection, time kernels) into the variables."""
        plan = self._plans.get("unpack")
        if plan is None:
            Gd, D, H = self.Gd, self.enc_in, self.H
            blocks, NX = self._xw_blocks()
            dXW, dxb = self._buf("xw.dW", D, NX), self._buf("xw.db", NX)
            descs = []
            gname = {"gx": ("gates/kernel", "gates/bias"), "cx": ("candidate/kernel", "candidate/bias")}
            scopes = {k: sc for k, sc, _ in self._gru_list()}
            t = self._t4_scope
            tname = {"kx": ("kernel", "bias"), "w1": ("_time_kernel_w1", "_time_bias1"),
                     "w2": ("_time_kernel_w2", "_time_bias2")}
            for key, kind, W, b, off, w in blocks:
                if kind == "const":
                    continue              # the always-open time gates of the plain LSTM have no variables
                if key == "t4":
                    gw, gb = Gd[t + tname[kind][0]], Gd[t + tname[kind][1]]
                else:
                    gw, gb = Gd[scopes[key] + gname[kind][0]], Gd[scopes[key] + gname[kind][1]]
                descs.append(ops.pack_desc(dXW[:, off:], D, w, gw, gw.stride(0), ld1=NX, transposed=True))
                descs.append(ops.pack_desc(dxb[off:], 1, w, gb, w, ld1=NX, transposed=True))
            if self._t4_kind == "time4lstm":
                dTW = self._buf("t4.dTW", 2 * H, 3 * H)
                for nm, r0, c0 in (("_o_kernel_t1", 0, 0), ("_o_kernel_t2", H, 0), ("_time_kernel_t1", 0, H),
                                   ("_time_kernel_t2", H, 2 * H)):
                    descs.append(ops.pack_desc(dTW[r0:, c0:], H, H, Gd[t + nm], H, ld1=3 * H, transposed=True))
            plan = ops.pack_table(descs, self.device)
            self._plans["unpack"] = plan
            self._plan_keep.append(descs)
        tbl, n, max_elems = plan
        call("clsr_pack_batch", tbl, n, max_elems)

    def _gru_list(self):
        hp = self.hp
        out = []
        if hp.interest_evolve:
            out.append(("g1", CL + "short_term/short_term_intention/gru_cell/", self.Du))
        if hp.sequential_model == "gru":
            out.append(("gs", CL + "short_term/simple_gru/gru_cell/", self.H))
        if (not hp.manual_alpha) and hp.predict_long_short:
            out.append(("g2", CL + "causal2/causal2/gru_cell/", self.H))
        return out

    @property
    def a_in(self):
        hp = self.hp
        return (self.H if hp.predict_long_short else 0) + 3 * self.D + 1

    # ------------------------------------------------------------------ feed upload
    def host_arrays(self, feed, training=True):
        """numpy feed (iterator layout) -> dict of contiguous numpy arrays with the device dtypes
        (+ the per-batch scalars the kernels read from device memory).

        A COMPACT training feed (``hist_group`` = 1 + train_num_ngs present: users / histories / mask /
        time arrays hold one row per positive line, see sequential_iterator.LazyFeed) is uploaded as is
        when the step de-duplicates histories anyway; otherwise it is expanded to the row layout here."""
        hg = int(feed.get("hist_group", 0) or 0)
        hist_keys = ("users", "item_history", "item_cate_history", "mask", "time_from_first_action", "time_to_now")
        # training: the group is the iterator's 1 + train_num_ngs; scoring: whatever run of consecutive rows with one
        # history the caller found (an evaluation file holds 1 + num_ngs lines per positive, CLSRModel._to_arrays)
        compact = bool(hg) and self.dedup and (hg == self.G_train if training else hg > 1)
        if hg and not compact:
            feed = dict(feed, **{k: np.repeat(np.asarray(feed[k]), hg, axis=0) for k in hist_keys})
            hg = 0
        mask = np.asarray(feed["mask"])
        seq_len = mask.sum(1).astype(np.int32)
        h = {"users": np.ascontiguousarray(np.asarray(feed["users"]), dtype=np.int32)}
        for k in ("items", "cates", "item_history", "item_cate_history"):
            h[k] = np.ascontiguousarray(feed[k], dtype=np.int32)
        for k in ("time_from_first_action", "time_to_now"):
            h[k] = np.ascontiguousarray(feed[k], dtype=np.float32)
        h["labels"] = np.ascontiguousarray(np.asarray(feed["labels"]).reshape(-1), dtype=np.float32)
        if "users_rows" in feed:      # scoring feeds with shared histories: the user of every LINE (device metrics)
            h["users_rows"] = np.ascontiguousarray(np.asarray(feed["users_rows"]).reshape(-1), dtype=np.int32)
        h["seq_len"] = seq_len
        rep = hg if compact else 1
        thr = getattr(self.hp, "contrastive_length_threshold", None) or 0     # (unset for the sibling models)
        h["denom"] = np.asarray([float((seq_len > thr).sum() * rep)], dtype=np.float32)
        self._last_group = hg if compact else 0
        return h, int(mask.shape[0]) * rep, int(mask.shape[1]), compact

    _STAGE_SLOTS = 3

    def upload(self, feed, training, into=None, copy_stream=None):
        """numpy feed -> device tensors (views into ONE device arena).  The arrays are packed into a
        pinned host arena and pulled across PCIe by ``clsr_stage_feed`` -- an ordinary kernel launch on
        the current stream, so the upload never blocks the host behind queued device work.  ``into``
        (a dict returned by an earlier call for the same shape) re-uses the device arena and a ring of
        pinned arenas: no allocation, and a pinned slot is only rewritten once the kernel that read it
        has completed (event per slot).

        ``copy_stream``: the transfer is issued there instead (one asynchronous 4 MB copy) and ``into["_ready"]``
        holds its completion event: with TWO alternating device arenas and a caller that stages batch N+1 before it
        launches step N (``SequentialBaseModel.batch_train``), the batch crosses PCIe while the previous step
        computes.  The caller records ``into["_free"]`` (an event on the compute stream) after the last launch that
        reads this arena; the copy waits for it."""
        h, B, T, compact = self.host_arrays(feed, training)
        if into is None:
            lay, off = {}, 0
            for k, arr in h.items():
                lay[k] = (off, arr.nbytes)
                off += (arr.nbytes + 15) // 16 * 16
            dev = torch.empty(off, dtype=torch.uint8, device=self.device)
            into = {"B": B, "T": T, "compact": compact, "G": self._last_group, "_lay": lay, "_nbytes": off,
                    "_dev": dev, "_slot": 0,
                    "_stage": [None] * self._STAGE_SLOTS}
            for k, arr in h.items():
                o, n = lay[k]
                into[k] = dev[o:o + n].view(torch.from_numpy(arr).dtype).view(arr.shape)
        assert into["B"] == B and into["T"] == T and into["compact"] == compact and into["G"] == self._last_group
        slot = into["_slot"]
        into["_slot"] = (slot + 1) % self._STAGE_SLOTS
        if into["_stage"][slot] is None:
            pin = torch.empty(into["_nbytes"], dtype=torch.uint8, pin_memory=True)
            into["_stage"][slot] = [pin, None, pin.numpy()]
        pin, ev, pin_np = into["_stage"][slot]
        if ev is not None:
            ev.synchronize()
        for k, arr in h.items():
            o, n = into["_lay"][k]
            assert arr.nbytes == n
            # plain single-threaded memcpy on purpose: a torch CPU copy_ wakes the whole OpenMP pool,
            # whose spinning workers were measured to stall this thread's kernel launches for ~7 ms/step
            np.copyto(pin_np[o:o + n], arr.reshape(-1).view(np.uint8))
        if copy_stream is None:
            call("clsr_stage_feed", into["_dev"], pin.data_ptr(), into["_nbytes"])
            ev = torch.cuda.Event()
            ev.record()
        else:
            if into.get("_free") is not None:
                copy_stream.wait_event(into["_free"])
            with torch.cuda.stream(copy_stream):
                into["_dev"].copy_(pin, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            into["_ready"] = ev      # the caller makes the compute stream wait for it right before the step
        into["_stage"][slot][1] = ev
        return into

    # ------------------------------------------------------------------ attention block
    def _att_fwd(self, key, scope, keys, q, Hn, G, T, Dk, Q, seq_len, len_stride, training, q_hist=None):
        P, A0, A1 = self.P, self.A0, self.A1
        qh = self._att_qh(key) if (q_hist is not None and G > 1) else 0
        R = Hn * G
        nn = scope + "att_fcn/nn_part/"
        bn0, bn1 = self.bn[nn + "batch_normalization/"], self.bn[nn + "batch_normalization_1/"]
        a = self._buf(key + ".a", Hn * T, Q)
        U = self._buf(key + ".U", Hn * T, A0)
        V = self._buf(key + ".V", R, A0)
        BF = torch.bfloat16
        z0 = self._buf(key + ".z0", R * T, A0, dtype=BF if self.bf16 else F32)
        z1 = self._buf(key + ".z1", R * T, A1, dtype=BF if self.bf16 else F32)
        wts = self._buf(key + ".wts", R, T)
        out = self._buf(key + ".out", R, Dk)
        hist_x3 = self.att_hist_x3 and bool(query("clsr_att_hist_fwd_x3_supported", Dk, Q, A0, qh))
        if hist_x3:
            # a = keys . A and U = a . Wu (+ the history-level share of the product term) in ONE pass over the keys
            # (csrc/atthist.hip) instead of three position-tiled GEMM launches on the dependent chain
            At, Kpa = self.packed[key + ".A"]
            Wut, Kpu = self.packed[key + ".Wu"]
            Wpt, Kpp = self.packed[key + ".Wp1"] if qh else (None, 0)
            call("clsr_att_hist_fwd_x3", keys, Dk, At, Kpa, Wut, Kpu, Wpt, Kpp, q_hist if qh else None, qh, Hn, T, Dk, Q,
                 A0, qh, self.att_hist_pieces, a, Q, U, A0)
        else:
            self._gemm(keys, Dk, key + ".A", Hn * T, Dk, Q, a, Q)
            self._gemm(a, Q, key + ".Wu", Hn * T, Q, A0, U, A0)
        self._gemm(q, Q, key + ".Wv", R, Q, A0, V, A0, bias=P[nn + "b_nn_layer0"])
        if self.bf16:
            # speed mode: the two (row, step)-level layers on bf16 MFMA with bf16 storage (csrc/hgemm.hip)
            M = R * T
            parts = query("clsr_hgemm_stats_parts", M) if training else 0
            sbuf = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)
            st = sbuf[: parts * 2 * A0] if training else None
            if qh and not hist_x3:
                self._gemm(a, Q, key + ".Wp1", Hn * T, qh, A0, U, A0, T=T, G=1, Xmul=q_hist, ldmul=qh, addU=U, ldu=A0,
                           addV=self._buf("att.zeroV", Hn, A0), ldv=A0)
            # (split query: only the target columns stay in the per-(row, step) product, K = Q - qh)
            Qe, ae, qe = Q - qh, (a[:, qh:] if qh else a), (q[:, qh:] if qh else q)
            if self._bf16_chain_ok(G, Qe):
                # the parity mode's chain kernels with one bf16 piece per operand, z0 / z1 stored as bf16
                Wt, Kp = self.packed[key + (".Wp2" if qh else ".Wp")]
                p0 = query("clsr_att_l0_fwd_stats_parts", Hn) if training else 0
                call("clsr_att_l0_fwd_x1_h", ae, Q, qe, Q, Wt, Kp, U, A0, V, A0, z0, A0,
                     sbuf[: p0 * 2 * A0] if training else None, Hn, G, T, Qe, A0)
                self._bn_fwd(bn0, sbuf[: p0 * 2 * A0] if training else None, p0, M, training)
                p1 = query("clsr_att_l1_fwd_stats_parts", M) if training else 0
                st = sbuf[: p1 * 2 * A1] if training else None
                Wt, Kp = self.packed[key + ".W1"]
                call("clsr_att_l1_fwd_x1_h", z0, A0, bn0.scale, bn0.shift, Wt, Kp, P[nn + "b_nn_layer1"], z1, A1, st, M, A0, A1)
                self._bn_fwd(bn1, st, p1, M, training)
                call("clsr_att_out_fwd_h", z1, bn1.scale, bn1.shift, P[nn + "w_nn_output"], P[nn + "b_nn_output"],
                     seq_len, len_stride, keys, Hn, G, T, A1, Dk, wts, out)
                return out
            Wt, Kp = self.packed_h[key + (".Wp2" if qh else ".Wp")]
            if self.l0_fwd_wave and query("clsr_hgemm_l0_group_supported", G, Qe, A0):
                # one wave per history group: a / U loaded once per 16 steps and re-used for the G rows
                p0 = query("clsr_hgemm_l0_group_stats_parts", Hn) if training else 0
                call("clsr_hgemm_l0_group", ae, Q, qe, Q, Wt, Kp, U, A0, V, A0, z0, A0,
                     sbuf[: p0 * 2 * A0] if training else None, Hn, G, T, Qe, A0)
                self._bn_fwd(bn0, sbuf[: p0 * 2 * A0] if training else None, p0, M, training)
            else:
                call("clsr_hgemm_mul_uv", ae, Q, T, G, qe, Q, Wt, Kp, U, A0, V, A0, z0, A0, st, M, Qe, A0)
                self._bn_fwd(bn0, st, parts, M, training)
            st = sbuf[: parts * 2 * A1] if training else None
            Wt, Kp = self.packed_h[key + ".W1"]
            call("clsr_hgemm", z0, A0, bn0.scale, bn0.shift, 1, Wt, Kp, P[nn + "b_nn_layer1"], z1, A1, st, M, A0, A1)
            self._bn_fwd(bn1, st, parts, M, training)
            call("clsr_att_out_fwd_h", z1, bn1.scale, bn1.shift, P[nn + "w_nn_output"], P[nn + "b_nn_output"],
                 seq_len, len_stride, keys, Hn, G, T, A1, Dk, wts, out)
            return out
        st, parts = self._stats_buf(R * T, A0) if training else (None, 0)
        if qh:
            # U[h,t] += (a[h,t,:qh] * q_hist[h]) . Wp[:qh]   (in place: every tile reads its own U before storing)
            if not hist_x3:
                self._gemm(a, Q, key + ".Wp1", Hn * T, qh, A0, U, A0, T=T, G=1, Xmul=q_hist, ldmul=qh, addU=U, ldu=A0,
                           addV=self._buf("att.zeroV", Hn, A0), ldv=A0)
            if self.l0_fwd_wave and query("clsr_att_l0_fwd_supported", G, Q - qh, A0):
                # the per-row half (target columns of the query) on the one-wave-per-history kernel: K = Q - qh
                parts = query("clsr_att_l0_fwd_stats_parts", Hn) if training else 0
                st = (self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
                      if training else None)
                Wt, Kp = self.packed[key + ".Wp2"]
                call(self._att_l0_fwd_entry_for(Q - qh), a[:, qh:], Q, q[:, qh:], Q, Wt, Kp, U, A0, V, A0, z0, A0, st, Hn, G, T, Q - qh, A0)
            else:
                self._gemm(a[:, qh:], Q, key + ".Wp2", R * T, Q - qh, A0, z0, A0, T=T, G=G, Xmul=q[:, qh:], ldmul=Q,
                           addU=U, ldu=A0, addV=V, ldv=A0, stats=st)
        elif self.l0_fwd_wave and query("clsr_att_l0_fwd_supported", G, Q, A0):
            # one wave per history, a / U loaded once per group of rows (csrc/attl0fwd.hip)
            parts = query("clsr_att_l0_fwd_stats_parts", Hn) if training else 0
            st = (self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
                  if training else None)
            Wt, Kp = self.packed[key + ".Wp"]
            call(self._att_l0_fwd_entry_for(Q), a, Q, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, st, Hn, G, T, Q, A0)
        else:
            self._gemm(a, Q, key + ".Wp", R * T, Q, A0, z0, A0, T=T, G=G, Xmul=q, ldmul=Q, addU=U, ldu=A0,
                       addV=V, ldv=A0, stats=st)
        self._bn_fwd(bn0, st, parts, R * T, training)
        if self.att_l1_fwd_x6 and query("clsr_att_l1_fwd_supported", A0, A1) and R * T * A1 * 4 < (1 << 30):
            # z1 = relu(bn0(z0)) . W1 + b1 on the bf16 matrix pipe with three pieces per operand (csrc/attl1fwd.hip)
            parts = query("clsr_att_l1_fwd_stats_parts", R * T) if training else 0
            st = (self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A1]
                  if training else None)
            Wt, Kp = self.packed[key + ".W1"]
            call("clsr_att_l1_fwd", z0, A0, bn0.scale, bn0.shift, Wt, Kp, P[nn + "b_nn_layer1"], z1, A1, st, R * T, A0, A1)
        else:
            st, parts = self._stats_buf(R * T, A1) if training else (None, 0)
            self._gemm(z0, A0, key + ".W1", R * T, A0, A1, z1, A1, bias=P[nn + "b_nn_layer1"], aff=bn0, stats=st)
        self._bn_fwd(bn1, st, parts, R * T, training)
        call("clsr_att_out_fwd", z1, bn1.scale, bn1.shift, P[nn + "w_nn_output"], P[nn + "b_nn_output"],
             seq_len, len_stride, keys, Hn, G, T, A1, Dk, wts, out)
        return out

    def _att_bwd(self, key, scope, dout, keys, q, dkeys, Hn, G, T, Dk, Q, seq_len, len_stride, q_hist=None,
                 dq_hist=None, dw_in=None):
        """Returns dq [R, Q]; accumulates into dkeys [Hn, T, Dk]; writes every dense gradient.  With ``q_hist``
        (see ``_att_qh``) the product-term gradient of the history-level query columns is ACCUMULATED into
        ``dq_hist`` [Hn, qh] and dq[:, :qh] only holds the V-path share."""
        P, Gd, A0, A1 = self.P, self.Gd, self.A0, self.A1
        qh = self._att_qh(key) if (q_hist is not None and G > 1) else 0
        R = Hn * G
        nn = scope + "att_fcn/nn_part/"
        bn0, bn1 = self.bn[nn + "batch_normalization/"], self.bn[nn + "batch_normalization_1/"]
        BF = torch.bfloat16
        AD = BF if self.bf16 else F32
        a = self._buf(key + ".a", Hn * T, Q)
        z0, z1 = self._buf(key + ".z0", R * T, A0, dtype=AD), self._buf(key + ".z1", R * T, A1, dtype=AD)
        wts = self._buf(key + ".wts", R, T)
        dz1 = self._buf(key + ".dz1", R * T, A1, dtype=AD)
        dz0 = self._buf(key + ".dz0", R * T, A0, dtype=AD)
        # score / softmax backward per history group: d score, dkeys, d b_out
        ds = self._buf(key + ".ds", R * T)
        if dw_in is not None:
            # the caller consumed the attention WEIGHTS (not their weighted sum): the gradient arrives as d w[r, t]
            parts = query("clsr_softmax_weights_bwd_parts", R)
            bp = self._buf("att.bp." + key, 4096)[:parts]
            call("clsr_softmax_weights_bwd", dw_in, wts, seq_len, len_stride, Hn, G, T, ds, bp)
        else:
            parts = query("clsr_att_score_bwd_parts", Hn)
            bp = self._buf("att.bp." + key, 4096)[:parts]             # own buffers per attention: reduced at the flush
            call("clsr_att_score_bwd", dout, wts, seq_len, len_stride, keys, Hn, G, T, Dk, ds, dkeys, bp)
        self._rp(bp, parts, 1, 1, Gd[nn + "b_nn_output"])
        # dy1 = ds * w_out * relu'(bn1(z1)) is never materialised: one streaming pass for the BN-1 backward sums
        # and d w_out, the coefficient kernel, one streaming pass that writes dz1
        parts = query("clsr_att_dy1_parts", R * T, A1)
        bnp = self._buf("att.bnp" + self._ws_tag, 2048 * 2 * 256, dtype=torch.float64)[: parts * 2 * A1]
        wp = self._buf("att.wp." + key, 2048 * 256)[: parts * A1]
        call("clsr_att_dy1_stats_h" if self.bf16 else "clsr_att_dy1_stats", z1, ds, bn1.scale, bn1.shift, bn1.mean,
             bn1.invstd, P[nn + "w_nn_output"], R * T, A1, bnp, wp)
        self._rp(wp, parts, A1, A1, Gd[nn + "w_nn_output"])
        self._bn_bwd_coef(bn1, bnp, parts, R * T)
        dW0 = Gd[nn + "w_nn_layer0"]
        da = self._buf(key + ".da", Hn * T, Q)
        dq = self._buf(key + ".dq", R, Q)
        dV = self._buf(key + ".dV", R, A0)
        if self.bf16:
            # speed mode: dz1 is recomputed from (z1, ds) in the prologue of the GEMM that back-propagates through the
            # second layer; TWO passes over (z1, z0): the batch-norm sums of layer 0, then the finished dz0 (+ dz1 for the
            # weight gradient) -- no separate dy1-apply / bn-apply sweeps (csrc/hgemm.hip: clsr_hgemm_att_l1_bwd)
            M = R * T
            if self._bf16_chain_ok(G, Q - qh):
                return self._att_bwd_chain_h(key, scope, nn, bn0, bn1, a, q, keys, dkeys, z0, z1, dz0, ds, da, dq, dV, dW0,
                                             Hn, G, R, T, Dk, Q, qh, q_hist, dq_hist)
            Wt, Kp = self.packed_h[key + ".W1^T"]
            parts = query("clsr_hgemm_stats_parts", M)
            st = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
            wo = P[nn + "w_nn_output"]
            call("clsr_hgemm_att_l1_bwd", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, bn0.mean, bn0.invstd, None, None, 0, None, 0, st, M, A1, A0)
            self._bn_bwd_coef(bn0, st, parts, M)
            call("clsr_hgemm_att_l1_bwd", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, None, None, bn0.coef, dz1, A1, dz0, A0, None, M, A1, A0)
            self._dw(z0, A0, dz1, A1, M, A0, A1, Gd[nn + "w_nn_layer1"], A1, db=Gd[nn + "b_nn_layer1"], aff=bn0,
                     x_bf16=1, dy_bf16=1)
            Qe, ae, qe = Q - qh, (a[:, qh:] if qh else a), (q[:, qh:] if qh else q)
            self._dw(ae, Q, dz0, A0, M, Qe, A0, dW0[3 * Q + qh:4 * Q], A0, T=T, G=G, Xmul=qe, ldmul=Q, dy_bf16=1)
            Wt, Kp = self.packed_h[key + (".Wp2^T" if qh else ".Wp^T")]
            dU = self._buf(key + ".dU", Hn * T, A0)
            if self.fused_l0_bwd and query("clsr_att_l0_bwd_h_supported", G, Qe, A0):
                # da, dq, dU, dV in one pass over dz0; daq = dz0 . Wp^T is never written (csrc/hattbwd.hip)
                # (dU . Wu^T inside the kernel only while the history-level tail is not the fused launch of csrc/atthist.hip)
                wu_in = self.bf16_bwd and self.fused_l0_wu and not qh and not self._hist_bwd_x3(Dk, Q, 0)
                Wu, Kpu = self.packed_h[key + ".Wu^T"] if wu_in else (None, Kp)
                assert Kpu == Kp
                call("clsr_att_l0_bwd_h", dz0, A0, Wt, Wu, Kp, ae, Q, qe, Q, Hn, G, T, Qe, A0,
                     da[:, qh:] if qh else da, Q, dq[:, qh:] if qh else dq, Q, dU, A0, dV, A0)
                if qh:
                    # the V path over ALL query columns (dq[:, :qh] was cleared with the step's accumulators), the weight
                    # gradient of the history-level share of the product term; the rest of that share: _att_bwd_hist
                    self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
                    self._dw(a, Q, dU, A0, Hn * T, qh, A0, dW0[3 * Q:3 * Q + qh], A0, T=T, G=1, Xmul=q_hist, ldmul=qh)
                return self._att_bwd_hist(key, scope, nn, a, q, keys, dkeys, dU, dV, da, dq, dW0, Hn, R, T, Dk, Q, qh,
                                          da_has_u=Wu is not None, q_hist=q_hist, dq_hist=dq_hist)
            else:
                daq = self._buf(key + ".daq", M, Q, dtype=BF)
                call("clsr_hgemm", dz0, A0, None, None, 0, Wt, Kp, None, daq, Q, None, M, A0, Q)
                call("clsr_att_prod_bwd_h", daq, Q, a, Q, q, Q, Hn, G, T, Q, da, Q, dq, Q, 0)
                call("clsr_att_z0_bwd_reduce_h", dz0, Hn, G, T, A0, dU, dV)
            return self._att_bwd_hist(key, scope, nn, a, q, keys, dkeys, dU, dV, da, dq, dW0, Hn, R, T, Dk, Q, 0)
        x3b = self.att_bwd in ("x3", "x6")
        if x3b and self.l1_bwd_2pass and query("clsr_att_l1_bwd_x3_supported", A1, A0) and (R * T * A0 + 80) * 4 < (1 << 31) - 1:
            # the same two passes as split-bf16 products; pass 2 also accumulates dW1 / db1 (dz1 is never stored)
            M = R * T
            Wt, Kp = self.packed[key + ".W1^T"]
            parts = query("clsr_att_l1_bwd_x3_parts", M)
            st = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
            wo = P[nn + "w_nn_output"]
            call(self._l1x, z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, bn0.mean, bn0.invstd, None, None, 0, None, st, M, A1, A0)
            self._bn_bwd_coef(bn0, st, parts, M)
            ws = self._buf(key + ".dw1x_ws", parts * query("clsr_dw_chunk_floats"))
            call(self._l1x, z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, None, None, bn0.coef, dz0, A0, ws, None, M, A1, A0)
            self._dw_fused(ws, parts, A0, A1, Gd[nn + "w_nn_layer1"], A1, db=Gd[nn + "b_nn_layer1"])
        elif self.l1_bwd_2pass and query("clsr_att_l1_bwd_supported", A1, A0):
            # two passes over (z1, z0) with dz1 recomputed in the GEMM prologue: the batch-norm sums of layer 0, then the
            # finished dz0 (+ dz1 for the weight gradient) -- no dy1-apply / bn-apply sweeps (csrc/attl1bwd.hip)
            M = R * T
            Wt, Kp = self.packed[key + ".W1^T"]
            parts = query("clsr_att_l1_bwd_stats_parts", M)
            st = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
            wo = P[nn + "w_nn_output"]
            call("clsr_att_l1_bwd", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, bn0.mean, bn0.invstd, None, None, 0, None, 0, st, M, A1, A0)
            self._bn_bwd_coef(bn0, st, parts, M)
            call("clsr_att_l1_bwd", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
                 bn0.shift, None, None, bn0.coef, dz1, A1, dz0, A0, None, M, A1, A0)
            self._dw(z0, A0, dz1, A1, M, A0, A1, Gd[nn + "w_nn_layer1"], A1, db=Gd[nn + "b_nn_layer1"], aff=bn0)
        else:
            call("clsr_att_dy1_apply", z1, ds, bn1.scale, bn1.shift, P[nn + "w_nn_output"], bn1.coef, R * T, A1, dz1)
            # layer 1: z1 = relu(bn0(z0)) . W1 + b1
            self._dw(z0, A0, dz1, A1, R * T, A0, A1, Gd[nn + "w_nn_layer1"], A1, db=Gd[nn + "b_nn_layer1"], aff=bn0)
            self._gemm_bnbwd(dz1, A1, key + ".W1^T", R * T, A1, A0, dz0, bn0, z0)
        # layer 0 (re-associated): z0 = U[h,t] + V[r] + (a[h,t]*q[r]) . Wp
        x3b = self.att_bwd_l0 in ("x3", "x6")
        if qh:
            Q2 = Q - qh
            l0x3 = x3b and self.fused_l0_bwd and query("clsr_att_l0_bwd_x3_supported", G, Q2, A0)
            halves = (not l0x3) and x3b and self._l0_bwd_halves_ok(G, Q2)
            if not (l0x3 or halves):
                self._dw(a[:, qh:], Q, dz0, A0, R * T, Q2, A0, dW0[3 * Q + qh:4 * Q], A0, T=T, G=G, Xmul=q[:, qh:],
                         ldmul=Q)
            dU = self._buf(key + ".dU", Hn * T, A0)
            if halves:
                self._att_l0_bwd_x3_halves(key, ".Wp2^T", dz0, a, q, da, dq, dU, dV, dW0, Hn, G, R, T, Q, qh, Q2)
                self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
            elif l0x3:
                # the same pass as split-bf16 products, with the partial sums of dWp[qh:] (csrc/attbwdx3.hip)
                Wt, Kp = self.packed[key + ".Wp2^T"]
                parts = query("clsr_att_l0_bwd_x3_parts", Hn)
                ws = self._buf(key + ".dwpx_ws", parts * query("clsr_dw_chunk_floats"))
                call(self._l0x, dz0, A0, Wt, Kp, a[:, qh:], Q, q[:, qh:], Q, Hn, G, T, Q2, A0, da[:, qh:], Q,
                     dq[:, qh:], Q, dU, A0, dV, A0, ws)
                self._dw_fused(ws, parts, Q2, A0, dW0[3 * Q + qh:4 * Q], A0)
                self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
            elif self.fused_l0_bwd and query("clsr_att_l0_bwd_supported", G, Q2, A0):
                # per-row half in one pass over dz0 (csrc/hattbwd.hip): da / dq of the target columns, dU, dV; then the V
                # path over ALL query columns is added (dq[:, :qh] was cleared with the step's accumulators)
                Wt, Kp = self.packed[key + ".Wp2^T"]
                call("clsr_att_l0_bwd", dz0, A0, Wt, Kp, a[:, qh:], Q, q[:, qh:], Q, Hn, G, T, Q2, A0, da[:, qh:], Q,
                     dq[:, qh:], Q, dU, A0, dV, A0)
                self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
            else:
                daq = self._buf(key + ".daq2", R * T, Q2)
                self._gemm(dz0, A0, key + ".Wp2^T", R * T, A0, Q2, daq, Q2)
                call("clsr_att_z0_bwd_reduce", dz0, Hn, G, T, A0, dU, dV)
                # dq = dV . Wv^T first (all Q columns), then the per-row product term is added to its target columns
                self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q)
                call("clsr_att_prod_bwd_ld", daq, Q2, a[:, qh:], Q, q[:, qh:], Q, Hn, G, T, Q2, da[:, qh:], Q,
                     dq[:, qh:], Q, 1)
            # history-level share of the product term: through dU = sum over the group's rows of dz0
            self._dw(a, Q, dU, A0, Hn * T, qh, A0, dW0[3 * Q:3 * Q + qh], A0, T=T, G=1, Xmul=q_hist, ldmul=qh)
            if not self._hist_bwd_x3(Dk, Q, qh):      # (else: inside the fused history-level kernel, _att_bwd_hist)
                daq1 = self._buf(key + ".daq1", Hn * T, qh)
                self._gemm(dU, A0, key + ".Wp1^T", Hn * T, A0, qh, daq1, qh)
                call("clsr_att_prod_bwd_ld", daq1, qh, a, Q, q_hist, qh, Hn, 1, T, qh, da, Q, dq_hist, qh, 1)
        else:
            l0x3 = x3b and self.fused_l0_bwd and query("clsr_att_l0_bwd_x3_supported", G, Q, A0)
            halves = (not l0x3) and x3b and self._l0_bwd_halves_ok(G, Q)
            if not (l0x3 or halves):
                self._dw(a, Q, dz0, A0, R * T, Q, A0, dW0[3 * Q:4 * Q], A0, T=T, G=G, Xmul=q, ldmul=Q)
            dU = dz0 if G == 1 else self._buf(key + ".dU", Hn * T, A0)
            if halves:
                self._att_l0_bwd_x3_halves(key, ".Wp^T", dz0, a, q, da, dq, None if G == 1 else dU, dV, dW0, Hn, G, R, T, Q, 0, Q)
            elif l0x3:
                Wt, Kp = self.packed[key + ".Wp^T"]
                parts = query("clsr_att_l0_bwd_x3_parts", Hn)
                ws = self._buf(key + ".dwpx_ws", parts * query("clsr_dw_chunk_floats"))
                call(self._l0x, dz0, A0, Wt, Kp, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q,
                     None if G == 1 else dU, A0, dV, A0, ws)
                self._dw_fused(ws, parts, Q, A0, dW0[3 * Q:4 * Q], A0)
            elif self.fused_l0_bwd and query("clsr_att_l0_bwd_supported", G, Q, A0):
                # da, dq, dU, dV in one pass over dz0 on the fp32 matrix pipe; daq = dz0 . Wp^T is never written
                Wt, Kp = self.packed[key + ".Wp^T"]
                call("clsr_att_l0_bwd", dz0, A0, Wt, Kp, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q,
                     None if G == 1 else dU, A0, dV, A0)
            else:
                daq = self._buf(key + ".daq", R * T, Q)
                self._gemm(dz0, A0, key + ".Wp^T", R * T, A0, Q, daq, Q)
                call("clsr_att_prod_bwd", daq, a, q, Hn, G, T, Q, da, dq)
                call("clsr_att_z0_bwd_reduce", dz0, Hn, G, T, A0, None if G == 1 else dU, dV)
        return self._att_bwd_hist(key, scope, nn, a, q, keys, dkeys, dU, dV, da, dq, dW0, Hn, R, T, Dk, Q, qh,
                                  q_hist=q_hist, dq_hist=dq_hist)

    def _l0_bwd_halves_ok(self, G, Qw):
        """wide product term (80 < Qw <= 160 query columns: BASELINE configs[4]): the one-pass layer-0 backward with folded dWp
        (csrc/attbwdx3.hip covers <= 80 columns) as TWO launches over the column halves?"""
        return bool(self.l0_bwd_halves and self.fused_l0_bwd and Qw > 80 and Qw % 8 == 0 and Qw // 2 <= 80
                    and query("clsr_att_l0_bwd_x3_supported", G, Qw // 2, self.A0))

    def _att_l0_bwd_x3_halves(self, key, wkey, dz0, a, q, da, dq, dU, dV, dW0, Hn, G, R, T, Q, c_lo, Qw):
        """Layer-0 backward of a product term of Qw query columns (a / q / da / dq columns [c_lo, c_lo + Qw)) as two launches of
        clsr_att_l0_bwd_x3 over the column halves: each reads dz0 once and leaves da / dq and the partial sums of dWp of ITS
        columns; dU / dV come from the first one.  Replaces, at 128 columns, the position-tiled chain daq = dz0 . Wp^T (written:
        0.5 GB) -> clsr_att_prod_bwd -> clsr_att_z0_bwd_reduce + the separate weight-gradient launch over (a, q, dz0)."""
        A0 = self.A0
        Wt, Kp = self.packed[key + wkey]
        parts = query("clsr_att_l0_bwd_x3_parts", Hn)
        C = query("clsr_dw_chunk_floats")
        Qh = Qw // 2
        for i in (0, 1):
            c0 = c_lo + i * Qh
            ws = self._buf(key + ".dwpx_ws%d" % i, parts * C)
            call(self._l0x, dz0, A0, Wt[i * Qh * Kp:], Kp, a[:, c0:], Q, q[:, c0:], Q, Hn, G, T, Qh, A0, da[:, c0:], Q,
                 dq[:, c0:], Q, dU if i == 0 else None, A0, dV if i == 0 else self._buf(key + ".dV_scratch", R, A0), A0, ws)
            self._dw_fused(ws, parts, Qh, A0, dW0[3 * Q + c0:3 * Q + c0 + Qh], A0)

    def _att_l0_fwd_entry_for(self, K):
        """Layer-0 forward entry for a product term of K query columns: wide layers (K > 80: BASELINE configs[4]) take the
        three-piece form (fp32 accuracy on the bf16 pipe: 120 bf16 MFMAs per 16 positions where the fp32-input form issues
        160 of twice the cost) unless bit-exact fp32 products are asked for."""
        if K > 80 and not self.exact_products and self.att_l0_fwd_entry == "clsr_att_l0_fwd":
            return "clsr_att_l0_fwd_x6"
        return self.att_l0_fwd_entry

    def _bf16_chain_ok(self, G, Qe):
        """speed mode on the parity mode's chain kernels (one bf16 piece per operand, bf16 z0 / z1 / dz0)?"""
        A0, A1 = self.A0, self.A1
        return bool(self.bf16_chain and A0 % 8 == 0 and A1 % 8 == 0 and self.l0_fwd_wave and self.fused_l0_bwd
                    and query("clsr_att_l0_fwd_supported", G, Qe, A0) and query("clsr_att_l1_fwd_supported", A0, A1)
                    and query("clsr_att_l1_bwd_x3_supported", A1, A0) and query("clsr_att_l0_bwd_x3_supported", G, Qe, A0))

    def _att_bwd_chain_h(self, key, scope, nn, bn0, bn1, a, q, keys, dkeys, z0, z1, dz0, ds, da, dq, dV, dW0, Hn, G, R, T,
                         Dk, Q, qh, q_hist, dq_hist):
        """Speed-mode tail of ``_att_bwd`` behind the BN-1 coefficients: two passes over (z1, z0) -- the second one writes
        dz0 (bf16) and the partial chunks of dW1 / db1 --, then ONE pass over dz0 for da, dq, dU, dV and dWp
        (csrc/attbwdx3.hip with NP = 1, ST = bf16)."""
        P, Gd, A0, A1 = self.P, self.Gd, self.A0, self.A1
        M = R * T
        Wt, Kp = self.packed[key + ".W1^T"]
        parts = query("clsr_att_l1_bwd_x3_parts", M)
        st = self._buf("stats" + self._ws_tag, 1024 * 2 * 256, dtype=torch.float64)[: parts * 2 * A0]
        wo = P[nn + "w_nn_output"]
        call("clsr_att_l1_bwd_x1_h", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
             bn0.shift, bn0.mean, bn0.invstd, None, None, 0, None, st, M, A1, A0)
        self._bn_bwd_coef(bn0, st, parts, M)
        ws = self._buf(key + ".dw1x_ws", parts * query("clsr_dw_chunk_floats"))
        call("clsr_att_l1_bwd_x1_h", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0, bn0.scale,
             bn0.shift, None, None, bn0.coef, dz0, A0, ws, None, M, A1, A0)
        self._dw_fused(ws, parts, A0, A1, Gd[nn + "w_nn_layer1"], A1, db=Gd[nn + "b_nn_layer1"])
        Qe, ae, qe = Q - qh, (a[:, qh:] if qh else a), (q[:, qh:] if qh else q)
        dU = self._buf(key + ".dU", Hn * T, A0)
        Wt, Kp = self.packed[key + (".Wp2^T" if qh else ".Wp^T")]
        parts = query("clsr_att_l0_bwd_x1_h_parts", Hn)
        ws = self._buf(key + ".dwpx_ws", parts * query("clsr_dw_chunk_floats"))
        call("clsr_att_l0_bwd_x1_h", dz0, A0, Wt, Kp, ae, Q, qe, Q, Hn, G, T, Qe, A0, da[:, qh:] if qh else da, Q,
             dq[:, qh:] if qh else dq, Q, dU, A0, dV, A0, ws)
        self._dw_fused(ws, parts, Qe, A0, dW0[3 * Q + qh:4 * Q], A0)
        if qh:
            # the V path over ALL query columns (dq[:, :qh] was cleared with the step's accumulators), the weight
            # gradient of the history-level share of the product term; the rest of that share: _att_bwd_hist
            self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
            self._dw(a, Q, dU, A0, Hn * T, qh, A0, dW0[3 * Q:3 * Q + qh], A0, T=T, G=1, Xmul=q_hist, ldmul=qh)
        return self._att_bwd_hist(key, scope, nn, a, q, keys, dkeys, dU, dV, da, dq, dW0, Hn, R, T, Dk, Q, qh,
                                  q_hist=q_hist, dq_hist=dq_hist)

    def _hist_bwd_x3(self, Dk, Q, qh):
        """history-level tail of the attention backward as ONE launch of split-bf16 products (csrc/atthist.hip)?"""
        return self.att_hist_bwd_x3 and bool(query("clsr_att_hist_bwd_x3_supported", Dk, Q, self.A0, qh))

    def _att_bwd_hist(self, key, scope, nn, a, q, keys, dkeys, dU, dV, da, dq, dW0, Hn, R, T, Dk, Q, qh,
                      da_has_u=False, q_hist=None, dq_hist=None):
        """History-level / row-level tail of the attention backward (fp32 in both precision modes): gradients of
        the U / V projections, of the attention matrix, and d keys."""
        Gd, A0 = self.Gd, self.A0
        # the two weight gradients that read the finished dU / dV leave as ONE two-job launch (operands final on entry:
        # no waiting; merging the attention_mat product behind d(a) as well was measured at +0.05 ms per step)
        with self._dw_batched():
            self._dw(a, Q, dU, A0, Hn * T, Q, A0, dW0[0:Q], A0)                        # d(W0a + W0d)
            self._dw(q, Q, dV, A0, R, Q, A0, dW0[Q:2 * Q], A0, db=Gd[nn + "b_nn_layer0"])  # d(W0q - W0d)
        # d(W0d) = d(W0a+W0d) - d(W0q-W0d): a third reduction descriptor over the partial sums of the two products above
        # (the batched reduction writes it in the same launch; it used to be an axpby launch behind the flush)
        if self.defer_dw:
            pend = self._dw_pending[self._ws_tag]
            pa, pq = pend[-2], pend[-1]
            pend.append((pa[0], dW0[2 * Q:3 * Q].data_ptr(), 0, 1.0, pa[4], Q, A0, A0, 0, pq[0], -1.0, pq[4]))
        else:
            call("clsr_axpby", dW0[2 * Q:3 * Q], dW0[0:Q], 1.0, dW0[Q:2 * Q], -1.0, Q * A0)
        fused = not da_has_u and self._hist_bwd_x3(Dk, Q, qh)
        if fused:
            # da += dU . Wu^T (+ the history-level product term and its dq_hist), dkeys += da . A^T from one pass over dU
            WuT, Kpu = self.packed[key + ".Wu^T"]
            WpT, Kpp = self.packed[key + ".Wp1^T"] if qh else (None, 0)
            AT, Kpa = self.packed[key + ".A^T"]
            call("clsr_att_hist_bwd_x3", dU, A0, WuT, Kpu, WpT, Kpp, AT, Kpa, a, Q, q_hist if qh else None, qh, Hn, T, Dk,
                 Q, A0, qh, self.att_hist_bwd_pieces, da, Q, dq_hist if qh else None, qh, dkeys, Dk)
        elif not da_has_u:      # (speed mode: clsr_att_l0_bwd_h has already added dU . Wu^T)
            self._gemm(dU, A0, key + ".Wu^T", Hn * T, A0, Q, da, Q, acc=1)
        if not qh:
            self._gemm(dV, A0, key + ".Wv^T", R, A0, Q, dq, Q, acc=1)
        self._dw(keys, Dk, da, Q, Hn * T, Dk, Q, Gd[scope + "attention_mat"], Q)
        if not fused:
            self._gemm(da, Q, key + ".A^T", Hn * T, Q, Dk, dkeys, Dk, acc=1)
        return dq

    # ------------------------------------------------------------------ MLP (alpha / logit)
    def _mlp_fwd(self, key, nn, X, ldx, K0, sizes, B, training):
        P = self.P
        C0, C1 = sizes
        bn0, bn1 = self.bn[nn + "batch_normalization/"], self.bn[nn + "batch_normalization_1/"]
        z0, z1 = self._buf(key + ".z0", B, C0), self._buf(key + ".z1", B, C1)
        logit = self._buf(key + ".logit", B)
        st, parts = self._stats_buf(B, C0) if training else (None, 0)
        self._gemm(X, ldx, key + ".W0", B, K0, C0, z0, C0, bias=P[nn + "b_nn_layer0"], stats=st)
        self._bn_fwd(bn0, st, parts, B, training)
        st, parts = self._stats_buf(B, C1) if training else (None, 0)
        self._gemm(z0, C0, key + ".W1", B, C0, C1, z1, C1, bias=P[nn + "b_nn_layer1"], aff=bn0, stats=st)
        self._bn_fwd(bn1, st, parts, B, training)
        if not (key == "lg" and self._defer_logit_out):   # (training step: the logits come out of clsr_mlp_tail_softmax)
            call("clsr_mlp_out_fwd", z1, bn1.scale, bn1.shift, P[nn + "w_nn_output"], P[nn + "b_nn_output"], B, C1, logit)
        return logit

    def _mlp_bwd(self, key, nn, dlogit, X, ldx, K0, K0_real, sizes, B, tail=None):
        """Returns dX [B, K0] (K0 = padded input width).  ``tail = (labels, groups, rows per group, loss scale)``: the
        output layer, the group-softmax loss and their backward run as ONE launch (the logits were not computed by the
        forward: ``_defer_logit_out``)."""
        P, Gd = self.P, self.Gd
        C0, C1 = sizes
        bn0, bn1 = self.bn[nn + "batch_normalization/"], self.bn[nn + "batch_normalization_1/"]
        z0, z1 = self._buf(key + ".z0", B, C0), self._buf(key + ".z1", B, C1)
        dz1, dz0 = self._buf(key + ".dz1", B, C1), self._buf(key + ".dz0", B, C0)
        parts = query("clsr_mlp_tail_softmax_parts", tail[1]) if tail else query("clsr_mlp_out_bwd_parts", B, C1)
        bnp = self._buf("mlp.bnp", 512 * 2 * 256, dtype=torch.float64)[: parts * 2 * C1]
        wp = self._buf("mlp.wp." + key, 512 * (256 + 4))[: parts * (C1 + 4)]   # own buffer: reduced at the flush
        if tail:
            labels, ngroups, Gl, lscale = tail
            call("clsr_mlp_tail_softmax", z1, bn1.scale, bn1.shift, bn1.mean, bn1.invstd, P[nn + "w_nn_output"],
                 P[nn + "b_nn_output"], labels, ngroups, Gl, C1, lscale, self.losses[0:], self._buf(key + ".logit", B),
                 dlogit, dz1, bnp, wp)
        else:
            call("clsr_mlp_out_bwd", dlogit, z1, bn1.scale, bn1.shift, bn1.mean, bn1.invstd, P[nn + "w_nn_output"],
                 B, C1, dz1, bnp, wp)
        self._rp(wp, parts, C1 + 4, C1, Gd[nn + "w_nn_output"])
        self._rp(wp[C1:], parts, C1 + 4, 1, Gd[nn + "b_nn_output"])
        self._bn_bwd_from_partial(bn1, bnp, parts, dz1, z1, B)
        self._dw(z0, C0, dz1, C1, B, C0, C1, Gd[nn + "w_nn_layer1"], C1, db=Gd[nn + "b_nn_layer1"], aff=bn0)
        self._gemm_bnbwd(dz1, C1, key + ".W1^T", B, C1, C0, dz0, bn0, z0)
        self._dw(X, ldx, dz0, C0, B, K0_real, C0, Gd[nn + "w_nn_layer0"], C0, db=Gd[nn + "b_nn_layer0"])
        dX = self._buf(key + ".dX", B, K0)
        self._gemm(dz0, C0, key + ".W0^T", B, C0, K0, dX, K0)
        return dX

    # ------------------------------------------------------------------ recurrent encoders
    def _gru_fwd_desc(self, key, scope, n, PinAll, Hn, T, h0, training, want_seq=False):
        P, D = self.P, self.enc_in
        hT = self._buf(key + ".hT", Hn, n)
        seq = self._buf(key + ".seq", Hn, T, n) if want_seq else None
        hprev = self._buf(key + ".hprev", Hn, T, n) if training else None
        gates = self._buf(key + ".gates", Hn, T, 3 * n) if training else None
        Wg, Wc = P[scope + "gates/kernel"], P[scope + "candidate/kernel"]
        fp = {}
        if self._rnn_fp_on():      # input projection inside the recurrence: x = the history embeddings (Pin is not read)
            fp = dict(X=self._buf("hist", Hn, T, D), ldx=D, Dx=D, Wgx=Wg, Wcx=Wc, bg=P[scope + "gates/bias"],
                      bc=P[scope + "candidate/bias"])
        d = ops.gru_desc(n, Pin=PinAll[:, self._enc_off(key):], ldp=self.NX, Wgh=Wg[D:], ldg=2 * n, Wch=Wc[D:],
                         ldc=n, h0=h0, h0_stride=n if h0 is not None else 0, hT=hT, out_seq=seq, hprev=hprev,
                         gates=gates, products=self.rnn_products, **fp)
        return d, hT, seq

    def _gru_bwd_desc(self, key, scope, n, dPinAll, Hn, T, dhT, dseq, dh0):
        P, D = self.P, self.enc_in
        Wg, Wc = P[scope + "gates/kernel"], P[scope + "candidate/kernel"]
        return ops.gru_desc(n, Wgh=Wg[D:], ldg=2 * n, Wch=Wc[D:], ldc=n,
                            hprev=self._buf(key + ".hprev", Hn, T, n), gates=self._buf(key + ".gates", Hn, T, 3 * n),
                            dhT=dhT, dout_seq=dseq, dPin=dPinAll[:, self._enc_off(key):], lddp=self.NX, dh0=dh0,
                            products=self.rnn_products)

    def _rnn_fp_on(self):
        """Input projections of the encoders inside the recurrence launch (csrc/rnn.hip, fused projection)?  Needs the
        split-bf16 products, hidden sizes <= 48, an embedding width that leaves a spare k slot for the bias row, and --
        with a Time4LSTM -- the K-fused time-gate product (it writes exactly the three blocks the kernel still reads)."""
        D = self.enc_in
        ok = (self.rnn_fused_proj and max(self.H, self.Du) <= 48 and D % 8 == 0 and D < 64
              and self._t4_kind in (None, "time4lstm"))
        if ok and self._t4_kind == "time4lstm":
            ok = self._fuse_tt_ok()
        return bool(ok)

    def _t4_act(self, Hn, T, training):
        """(act, cst, tiled) buffers of the LSTM-type encoder's saved activations."""
        H = self.H
        if not training:
            return None, None, False
        if self.rnn_act_tiled:
            return self._buf("t4.act_tiled", query("clsr_t4_act_tiled_floats", Hn, T, H)), None, True
        return self._buf("t4.act", Hn, T, 6 * H), self._buf("t4.cst", Hn, T, H), False

    def _t4_bwd_weights(self, f, dPinAll, Hn, T, hs):
        """Hidden-to-hidden and time-feature weight gradients of the LSTM-type encoder from its slice of dPin."""
        Gd, H, NX, E = self.Gd, self.H, self.NX, self.enc_in
        t, M = self._t4_scope, Hn * T
        dPt = dPinAll[:, self._enc_off("t4"):]
        hb = int(dPinAll.dtype == torch.bfloat16)
        self._dw(self._buf("t4.mprev", Hn, T, H), H, dPt, NX, M, H, 4 * H, Gd[t + "kernel"][E:], 4 * H, dy_bf16=hb)
        if self._t4_kind != "time4lstm":
            return
        TT = self._buf("t4.TT", M, 2 * H)
        self._dw(TT, 2 * H, dPt[:, 3 * H:], NX, M, 2 * H, 3 * H, self._buf("t4.dTW", 2 * H, 3 * H), 3 * H, dy_bf16=hb)
        self._t4_time_chain_bwd(f, dPinAll, Hn, T, hs)

    def _t4_time_chain_bwd(self, f, dPinAll, Hn, T, hs, have_dtt=False):
        """d TT = dPin[:, o | tns | tls] . tw^T, its tanh backward and the sums for the four time-input vectors."""
        Gd, H, NX = self.Gd, self.H, self.NX
        t, M = self._t4_scope, Hn * T
        dPt = dPinAll[:, self._enc_off("t4"):]
        TT = self._buf("t4.TT", M, 2 * H)
        dTT = self._buf("t4.dTT", M, 2 * H)
        parts = query("clsr_t4_time_inputs_bwd_parts", Hn, T, H)
        tp = self._buf("t4.tpart", 512 * 4 * 128)[: parts * 4 * H]
        if not have_dtt:
            self._gemm(dPt[:, 3 * H:], NX, "t4.tw^T", M, 3 * H, 2 * H, dTT, 2 * H)
        call("clsr_t4_time_inputs_bwd", dTT, TT, f["time_to_now"], f["time_from_first_action"], hs * T, Hn, T, H, tp)
        for off_, nm in ((0, "_time_input_w1"), (H, "_time_input_w2"), (2 * H, "_time_input_bias1"),
                         (3 * H, "_time_input_bias2")):
            self._rp(tp[off_:], parts, 4 * H, H, Gd[t + nm])

    def _gru_bwd_hidden(self, key, scope, n, dPinAll, Hn, T):
        """Hidden-to-hidden weight gradients of one GRU from its slice of dPin."""
        Gd, D, NX = self.Gd, self.enc_in, self.NX
        hprev, gates = self._buf(key + ".hprev", Hn, T, n), self._buf(key + ".gates", Hn, T, 3 * n)
        dP = dPinAll[:, self._enc_off(key):]
        M = Hn * T
        hb = int(dPinAll.dtype == torch.bfloat16)
        self._dw(hprev, n, dP, NX, M, n, 2 * n, Gd[scope + "gates/kernel"][D:], 2 * n, dy_bf16=hb)
        self._dw(hprev, n, dP[:, 2 * n:], NX, M, n, n, Gd[scope + "candidate/kernel"][D:], n, Xmul=gates,
                 ldmul=3 * n, dy_bf16=hb)

    def _enc_bwd_fused_ok(self, dpin_h):
        """The default graph at the default widths: short_term_intention GRU + Time4LSTM + causal GRU, D = Du = H = 40,
        fp32 dPin with the column layout csrc/encbwd.hip is written for."""
        hp = self.hp
        if self.bf16 and not (dpin_h and self.enc_bwd_fused_h):     # (speed mode: the bf16-dPin form of the kernel)
            return False
        return (self.enc_bwd_fused and (dpin_h if self.bf16 else not dpin_h) and type(self) is CLSRNet and self.defer_dw
                and self._t4_kind == "time4lstm" and bool(hp.interest_evolve)
                and (not hp.manual_alpha) and bool(hp.predict_long_short) and self.enc_in == self.D
                and self.D == self.Du == self.H and bool(query("clsr_enc_bwd_fused_supported", self.D, self.H, self.NX))
                and self._enc_off("g1") == 0 and self._enc_off("g2") == 3 * self.H and self._enc_off("t4") == 6 * self.H)

    def _enc_bwd_fused(self, f, hist, dPinAll, dhist, Hn, T, hs):
        """Seven encoder-side weight gradients + d(hist) from ONE pass over dPin (csrc/encbwd.hip) on the compute stream,
        behind the small time-feature chain (d TT, its tanh backward and parameter sums)."""
        Gd, D, H, NX, E = self.Gd, self.D, self.H, self.NX, self.enc_in
        M = Hn * T
        st, t = CL + "short_term/", self._t4_scope
        g1, g2 = st + "short_term_intention/gru_cell/", CL + "causal2/causal2/gru_cell/"
        parts = query("clsr_enc_bwd_fused_parts", M)
        prods = [(self._buf("xw.dW", D, NX), NX, self._buf("xw.db", NX), D, NX),
                 (Gd[g1 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g1 + "candidate/kernel"][E:], H, None, H, H),
                 (Gd[t + "kernel"][E:], 4 * H, None, H, 4 * H), (self._buf("t4.dTW", 2 * H, 3 * H), 3 * H, None, 2 * H, 3 * H),
                 (Gd[g2 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g2 + "candidate/kernel"][E:], H, None, H, H)]
        pend = self._dw_pending.setdefault("", [])
        wss = []
        for i, (dW, ldw, db, K, N) in enumerate(prods):
            ws = self._buf("encb.ws%d" % i, query("clsr_enc_bwd_fused_workspace_floats", M, i))
            wss.append(ws)
            pend.append((ws.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, 1.0, parts, K, N, ldw, 0))
        # d TT = dPin[:, o | tns | tls] . tw^T FIRST, on the compute stream (~35 us alone; beside the fused kernel, whose
        # workgroups fill every CU's LDS, it found no room and took 500 us); its tanh backward + parameter sums then
        # run beside the fused kernel on the weight-gradient stream, which the dense reduction follows in stream order
        dPt = dPinAll[:, self._enc_off("t4"):]
        dTT = self._buf("t4.dTT", M, 2 * H)
        TT = self._buf("t4.TT", M, 2 * H)
        # (every workspace of the side branch exists BEFORE the fork: a first-use allocation zero-fills on the compute
        #  stream, and a fill enqueued behind the fork raced with the branch's writes in the first step of a net)
        parts_t = query("clsr_t4_time_inputs_bwd_parts", Hn, T, H)
        tp = self._buf("t4.tpart", 512 * 4 * 128)[: parts_t * 4 * H]
        self._gemm(dPt[:, 3 * H:], NX, "t4.tw^T", M, 3 * H, 2 * H, dTT, 2 * H)
        # ... and its tanh backward + parameter sums (27 us): beside the MFMA-saturated fused kernel they took 400 us and
        # the batched reduction of ALL dense gradients waited for them
        call("clsr_t4_time_inputs_bwd", dTT, TT, f["time_to_now"], f["time_from_first_action"], hs * T, Hn, T, H, tp)
        for off_, nm in ((0, "_time_input_w1"), (H, "_time_input_w2"), (2 * H, "_time_input_bias1"),
                         (3 * H, "_time_input_bias2")):
            self._rp(tp[off_:], parts_t, 4 * H, H, Gd[t + nm])
        Wt, Kp = self.packed["xw^T"]
        fold = self._fold_args if self.fold_hist_shares else None
        if fold is not None:
            # d(hist) += the long-term branch's d(hist) + the mean / recent-k shares of the history prologue INSIDE this launch
            # (it reads and writes every d(hist) row anyway): the segmented sums of the item / category sites then run in
            # their lean, software-pipelined form (csrc/segsum.hip) -- they are what the table update at the end of the step
            # waits for
            dhist_lt, dM, dR, seq_len, ls = fold
            self._join(only="@lt")          # (the long-term attention backward wrote dhist_lt on its branch, long ago)
            call("clsr_enc_bwd_fused_fold", dPinAll, hist, self._buf("g1.hprev", Hn, T, H), self._buf("g1.gates", Hn, T, 3 * H),
                 self._buf("t4.mprev", Hn, T, H), TT, self._buf("g2.hprev", Hn, T, H), self._buf("g2.gates", Hn, T, 3 * H),
                 Wt, Kp, dhist, dhist_lt, dM, dR, seq_len, ls, T, int(self.hp.contrastive_recent_k), *wss, M)
            self._folded = True
            return
        call("clsr_enc_bwd_fused", dPinAll, hist, self._buf("g1.hprev", Hn, T, H), self._buf("g1.gates", Hn, T, 3 * H),
             self._buf("t4.mprev", Hn, T, H), TT, self._buf("g2.hprev", Hn, T, H), self._buf("g2.gates", Hn, T, 3 * H),
             Wt, Kp, dhist, *wss, M)

    def _enc_bwd_fused_x3(self, f, hist, dPinAll, dhist, Hn, T, hs):
        """fp32x3 mode: the seven encoder-side weight gradients from ONE pass over the fp32 dPin as split-bf16 products
        (clsr_enc_bwd_fused_x3) on the weight-gradient stream, beside d(hist) = dPin . W_x^T (long-term stream) and the
        time-feature chain on the compute stream -- the launch structure of the speed mode's tail."""
        Gd, D, H, NX, E = self.Gd, self.D, self.H, self.NX, self.enc_in
        M = Hn * T
        st, t = CL + "short_term/", self._t4_scope
        g1, g2 = st + "short_term_intention/gru_cell/", CL + "causal2/causal2/gru_cell/"
        parts = query("clsr_enc_bwd_fused_x3_parts", M)
        prods = [(self._buf("xw.dW", D, NX), NX, self._buf("xw.db", NX), D, NX),
                 (Gd[g1 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g1 + "candidate/kernel"][E:], H, None, H, H),
                 (Gd[t + "kernel"][E:], 4 * H, None, H, 4 * H), (self._buf("t4.dTW", 2 * H, 3 * H), 3 * H, None, 2 * H, 3 * H),
                 (Gd[g2 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g2 + "candidate/kernel"][E:], H, None, H, H)]
        pend = self._dw_pending.setdefault("", [])
        wss = []
        for i, (dW, ldw, db, K, N) in enumerate(prods):
            ws = self._buf("encbx.ws%d" % i, query("clsr_enc_bwd_fused_x3_workspace_floats", M, i))
            wss.append(ws)
            pend.append((ws.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, 1.0, parts, K, N, ldw, 0))
        self._buf("t4.dTT", M, 2 * H)      # (workspaces of the branches exist before the fork: see _enc_bwd_fused)
        side = self.dw_stream and self.overlap
        with self._branch("@dw0" if side else "@main", after=self._fork_point(), name="@encw"):
            call("clsr_enc_bwd_fused_x6" if self.enc_x6 else "clsr_enc_bwd_fused_x3", dPinAll, hist, self._buf("g1.hprev", Hn, T, H), self._buf("g1.gates", Hn, T, 3 * H),
                 self._buf("t4.mprev", Hn, T, H), self._buf("t4.TT", M, 2 * H), self._buf("g2.hprev", Hn, T, H),
                 self._buf("g2.gates", Hn, T, 3 * H), *wss, M)
        if side:
            self._dw_async = True       # (the flush waits for the weight-gradient stream)
        tcol0 = self._enc_off("t4") + 3 * H
        if (self.enc_back_x3 and "t4.tw^T" in self.packed
                and query("clsr_enc_back_x3_supported", M, NX, D, 2 * H, 3 * H, tcol0)):
            # d(hist) += dPin . W_x^T and d TT = dPin[:, o | tns | tls] . tw^T from ONE pass over dPin (csrc/projx3.hip)
            Wx, Kpx = self.packed["xw^T"]
            Wt, Kpt = self.packed["t4.tw^T"]
            call("clsr_enc_back_x3", dPinAll, NX, Wx, Kpx, Wt, Kpt, tcol0, dhist, D, self._buf("t4.dTT", M, 2 * H), 2 * H,
                 M, NX, D, 2 * H, 3 * H)
            self._t4_time_chain_bwd(f, dPinAll, Hn, T, hs, have_dtt=True)
            return
        if self.dhist_side and self.overlap:
            with self._branch("@lt", after=self._fork_point(), name="@dhist"):
                self._gemm(dPinAll, NX, "xw^T", M, NX, D, dhist, D, acc=1)
        else:
            self._gemm(dPinAll, NX, "xw^T", M, NX, D, dhist, D, acc=1)
        self._t4_time_chain_bwd(f, dPinAll, Hn, T, hs)

    def _enc_bwd_fused_h(self, f, hist, dPinAll, dhist, Hn, T, hs):
        """Speed mode: the seven encoder-side weight gradients from ONE pass over the bf16 dPin (clsr_enc_bwd_fused_h) on
        the weight-gradient stream, beside d(hist) = dPin . W_x^T and the time-feature chain on the compute stream."""
        Gd, D, H, NX, E = self.Gd, self.D, self.H, self.NX, self.enc_in
        M = Hn * T
        st, t = CL + "short_term/", self._t4_scope
        g1, g2 = st + "short_term_intention/gru_cell/", CL + "causal2/causal2/gru_cell/"
        parts = query("clsr_enc_bwd_fused_h_parts", M)
        prods = [(self._buf("xw.dW", D, NX), NX, self._buf("xw.db", NX), D, NX),
                 (Gd[g1 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g1 + "candidate/kernel"][E:], H, None, H, H),
                 (Gd[t + "kernel"][E:], 4 * H, None, H, 4 * H), (self._buf("t4.dTW", 2 * H, 3 * H), 3 * H, None, 2 * H, 3 * H),
                 (Gd[g2 + "gates/kernel"][E:], 2 * H, None, H, 2 * H), (Gd[g2 + "candidate/kernel"][E:], H, None, H, H)]
        pend = self._dw_pending.setdefault("", [])
        wss = []
        for i, (dW, ldw, db, K, N) in enumerate(prods):
            ws = self._buf("encbh.ws%d" % i, query("clsr_enc_bwd_fused_h_workspace_floats", M, i))
            wss.append(ws)
            pend.append((ws.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, 1.0, parts, K, N, ldw, 0))
        # (the kernel can also accumulate d(hist) = dPin . W_x^T -- Wt_bf16 / dhist arguments, tested -- but that product
        #  contracts over the COLUMNS: its operands come from global memory / L1, 100-150 us more in the kernel against the
        #  230 us of the separate clsr_hgemm_hf32 launch that otherwise runs BESIDE it: 2.93 against 2.90 ms per step)
        side = self.dw_stream and self.overlap
        with self._branch("@dw0" if side else "@main", after=self._fork_point(), name="@encw"):
            call("clsr_enc_bwd_fused_h", dPinAll, hist, self._buf("g1.hprev", Hn, T, H), self._buf("g1.gates", Hn, T, 3 * H),
                 self._buf("t4.mprev", Hn, T, H), self._buf("t4.TT", M, 2 * H), self._buf("g2.hprev", Hn, T, H),
                 self._buf("g2.gates", Hn, T, 3 * H), *wss, None, 0, None, M)
        if side:
            self._dw_async = True       # (the flush waits for the weight-gradient stream)
        if self.dhist_side and self.overlap:
            # three ways: weight gradients on @dw0, d(hist) on the (by now idle) long-term stream, the time-feature chain here
            with self._branch("@lt", after=self._fork_point(), name="@dhist"):
                self._gemm(dPinAll, NX, "xw^T", M, NX, D, dhist, D, acc=1)
        else:
            self._gemm(dPinAll, NX, "xw^T", M, NX, D, dhist, D, acc=1)
        self._t4_time_chain_bwd(f, dPinAll, Hn, T, hs)

    # ------------------------------------------------------------------ forward
    def forward(self, f, training, after_attention=None, early_aux=None):
        """Run the forward pass on an uploaded feed; returns dict of device tensors.  ``after_attention(out)``
        (training) is launched on a side stream as soon as both attention outputs exist, beside the alpha / logit
        MLPs (the contrastive loss: it needs the interest vectors, not the logits)."""
        with ops.stream_scope():
            if training or after_attention is not None or early_aux is not None:
                return self._forward(f, training, after_attention, early_aux)
            return self._planned(("fwd",), f, lambda: self._forward(f, False, None, None))

    def _forward(self, f, training, after_attention, early_aux):
        hp, P = self.hp, self.P
        B, T = f["B"], f["T"]
        G = self.G_train if (training and self.dedup) else ((f.get("G") or 1) if not training else 1)
        if B % G:
            raise ValueError("training feed rows (%d) must be a multiple of 1+train_num_ngs (%d)" % (B, G))
        Hn = B // G
        D, Du, H, Di, Dc = self.D, self.Du, self.H, self.Di, self.Dc
        self.last_shape = (B, T, G, Hn)
        self._pack_all(training)
        # rows between consecutive history groups in the uploaded history-level arrays
        hs = 1 if f.get("compact") else G
        seq_len, ls = f["seq_len"], hs
        step_start = self._fork_point()
        # ---- gathers
        hist = self._buf("hist", Hn, T, D)
        hmean, hrec = self._buf("hist_mean", Hn, D), self._buf("hist_recent", Hn, D)
        if self.table_bf16:
            call("clsr_gather_hist_fwd_h", self.tables["item"], self.tables["cate"], f["item_history"], f["item_cate_history"],
                 hs * T, seq_len, ls, Hn, T, Di, Dc, hp.contrastive_recent_k, hist, 0, hmean, hrec)
        else:
            call("clsr_gather_hist_fwd", self.tables["item"], self.tables["cate"], f["item_history"],
                 f["item_cate_history"], hs * T, seq_len, ls, Hn, T, Di, Dc, hp.contrastive_recent_k, hist, hmean, hrec)
        target = self._buf("target", B, D)
        ulong, ushort = self._buf("u_long", Hn, Du), self._buf("u_short", Hn, Du)
        tb = self.tables
        ops.multi("clsr_gather_rows_multi" + self._th, ops.GatherDesc, [     # target = [item | cate], user_long, user_short
            (tb["item"].data_ptr(), f["items"].data_ptr(), target.data_ptr(), 1, B, Di, D, 0),
            (tb["cate"].data_ptr(), f["cates"].data_ptr(), target.data_ptr(), 1, B, Dc, D, Di),
            (tb["user_long"].data_ptr(), f["users"].data_ptr(), ulong.data_ptr(), hs, Hn, Du, Du, 0),
            (tb["user_short"].data_ptr(), f["users"].data_ptr(), ushort.data_ptr(), hs, Hn, Du, Du, 0)])
        # ---- sequence encoders: ONE fused input projection, then ONE fused launch for all recurrences
        st = CL + "short_term/"
        M = Hn * T
        NX = self.NX
        PinAll = self._buf("xw.Pin", M, NX)
        if self._t4_scope is not None and hp.sequential_model == "time4lstm":
            # tanh time features of the Time4LSTM gates depend on the feed only: on the (still idle) @lt stream
            # beside the input projection instead of after it
            t = self._t4_scope
            fuse_tt = self._fuse_tt_ok()
            Dp = 16 * ((D + 15) // 16)
            XT = self._buf("t4.XT", M, Dp + 2 * H) if fuse_tt else None
            TTb = self._buf("t4.TT", M, 2 * H)
            with self._branch("@lt"):
                call("clsr_t4_time_inputs_fwd2", f["time_to_now"], f["time_from_first_action"], hs * T,
                     P[t + "_time_input_w1"], P[t + "_time_input_bias1"], P[t + "_time_input_w2"],
                     P[t + "_time_input_bias2"], Hn, T, H, TTb, hist if fuse_tt else None, D if fuse_tt else 0, XT,
                     Dp + 2 * H if fuse_tt else 0, Dp if fuse_tt else 0)
        else:
            fuse_tt = False
        # (with the K-fused time-gate product the first launch stops in front of those 3H columns)
        rnn_fp = self._rnn_fp_on()
        if not rnn_fp:      # (fused projection: the recurrences read the embeddings themselves)
            Nw = NX - 3 * H if fuse_tt else NX
            if self.proj_x3_wide and D % 8 == 0 and query("clsr_proj_x3_supported", M, D, Nw):
                # wide encoders (BASELINE configs[4]: 128 -> 1 536): the projection in front of the recurrences as split
                # products with the operands in registers, 128 output columns per workgroup column (csrc/projx3.hip)
                Wt, Kp = self.packed["xw"]
                # (three pieces per operand in the parity mode: at K = 128 the 2^-16 perturbation of two pieces, carried through
                # the recurrences into the short-term query, moved one attention gradient of tests/test_fullsize_gpu.py past
                # its fp32 tolerance -- 6.0e-7 against 4.8e-7)
                call("clsr_proj_x3", hist, D, Wt, Kp, self._buf("xw.bias", NX), PinAll, NX, M, D, Nw, self.proj_wide_pieces)
            else:
                self._gemm(hist, D, "xw", M, D, Nw, PinAll, NX, bias=self._buf("xw.bias", NX))
        if early_aux is not None or (training and self.sorted_hist_grad):
            # work that depends on the feed only -- accumulator zeroing / row marks of the training step
            # (``early_aux``) and the ~35 tiny launches that sort the history ids by row id for the backward's
            # segmented sums -- on the @aux stream (NOT a stream of its own: see _dw, four streams in all), ordered
            # after the START of the step but ENQUEUED here, behind the first big main-stream launches: when the
            # host is the slower side (tracer, fit loop sharing the GIL with the iterator thread) the device
            # executes in enqueue order and must not find 40 tiny launches ahead of the main chain
            with self._branch("@aux", after=step_start):
                if early_aux is not None:
                    early_aux()
                if training and self.sorted_hist_grad:
                    self._sort_hist_ids(f, Hn, T, hs)
        grus, t4d = [], None
        short_int, rnn_out, fs = ushort, None, None
        if hp.interest_evolve:
            d, short_int, _ = self._gru_fwd_desc("g1", st + "short_term_intention/gru_cell/", Du, PinAll, Hn, T,
                                                 ushort, training)
            grus.append(d)
        if self._t4_scope is not None:
            t = self._t4_scope
            t4off = self._enc_off("t4")
            if hp.sequential_model == "time4lstm":
                TT = self._buf("t4.TT", M, 2 * H)
                # the tanh time features INSIDE the projection's prologue (csrc/projx3.hip, clsr_proj_x3_tt): the launch no
                # longer waits for the side kernel that writes TT / the [hist | TT] image (those are for the backward pass
                # only now: joined with the long-term branch in front of the heads) -- one kernel + two stream hops off the
                # chain that starts the recurrences
                tt_fused = bool(fuse_tt and self.proj_tt and query("clsr_proj_x3_tt_supported", M, D, Dp, H, 3 * H))
                if not tt_fused:
                    self._join("@lt")     # the time features were computed beside the fused input projection
                if tt_fused:
                    Wt, Kp = self.packed["xw.t"]
                    call("clsr_proj_x3_tt", hist, D, f["time_to_now"], f["time_from_first_action"], hs * T, T,
                         P[t + "_time_input_w1"], P[t + "_time_input_bias1"], P[t + "_time_input_w2"],
                         P[t + "_time_input_bias2"], H, Dp, Wt, Kp, self._buf("xw.bias", NX)[t4off + 3 * H:],
                         PinAll[:, t4off + 3 * H:], NX, M, 3 * H, self.proj_gate_pieces)
                elif fuse_tt:
                    # ONE product over [hist | TT] (K = 48 + 80) writes the time-gate columns: the separate pass that
                    # re-read and re-wrote them (hist . W_x first, += TT . W_t behind it: 112 us alone) is gone
                    if self.proj_x3 and query("clsr_proj_x3_supported", M, Dp + 2 * H, 3 * H):
                        # (split-bf16 products, result lanes = output features: csrc/projx3.hip)
                        Wt, Kp = self.packed["xw.t"]
                        call("clsr_proj_x3", XT, Dp + 2 * H, Wt, Kp, self._buf("xw.bias", NX)[t4off + 3 * H:],
                             PinAll[:, t4off + 3 * H:], NX, M, Dp + 2 * H, 3 * H, self.proj_gate_pieces)
                    elif self.proj_x3_wide and query("clsr_proj_x3_wide_supported", M, Dp + 2 * H, 3 * H):
                        # (hidden 128: K = 384 as three slabs of 128, the later ones accumulating -- 3 x ~130 us against the
                        # 0.99 ms of the position-tiled product, on the chain in front of the recurrences)
                        Wt, Kp = self.packed["xw.t"]
                        call("clsr_proj_x3_wide", XT, Dp + 2 * H, Wt, Kp, self._buf("xw.bias", NX)[t4off + 3 * H:],
                             PinAll[:, t4off + 3 * H:], NX, M, Dp + 2 * H, 3 * H, self.proj_wide_pieces, 0)
                    else:
                        self._gemm(XT, Dp + 2 * H, "xw.t", M, Dp + 2 * H, 3 * H, PinAll[:, t4off + 3 * H:], NX,
                                   bias=self._buf("xw.bias", NX)[t4off + 3 * H:])
                else:
                    self._gemm(TT, 2 * H, "t4.tw", M, 2 * H, 3 * H, PinAll[:, t4off + 3 * H:], NX, acc=1)
            rnn_out = self._buf("rnn_out", Hn, T, H)
            act, cst, tiled = self._t4_act(Hn, T, training)
            fp = dict(X=hist, ldx=D, Dx=D, Wkx=P[t + "kernel"], bk=P[t + "bias"]) if rnn_fp else {}
            t4d = ops.t4_desc(H, Pin=PinAll[:, t4off + (3 * H if rnn_fp else 0):], ldp=NX, Wm=P[t + "kernel"][D:], ldm=4 * H,
                              out_seq=rnn_out, act=act, cst=cst, act_tiled=tiled,
                              mprev=self._buf("t4.mprev", Hn, T, H) if training else None, products=self.rnn_products, **fp)
        else:
            d, _, rnn_out = self._gru_fwd_desc("gs", st + "simple_gru/gru_cell/", H, PinAll, Hn, T, None, training,
                                               want_seq=True)
            grus.append(d)
        g2_side = None
        if (not hp.manual_alpha) and hp.predict_long_short:
            d, fs, _ = self._gru_fwd_desc("g2", CL + "causal2/causal2/gru_cell/", H, PinAll, Hn, T, None, training)
            if self.split_g2 and self.overlap and (grus or self._t4_scope is not None):
                # only the alpha gate needs this state: off the critical recurrence launch
                g2_side = d
            else:
                grus.append(d)
        # ---- long term attention (independent of the encoders and of the short-term attention) on the side
        #      stream, forked HERE: the T-serial recurrences occupy only ~3 waves per CU, so the long-term chain
        #      runs underneath them instead of beside the big GEMMs.  The recurrence is ENQUEUED FIRST: packets
        #      reach the device in launch order, and with the branch's ~10 launches ahead of it the recurrence
        #      started ~350 us late (profiles/r01_step_timeline_graph_before_after.txt)
        lt = CL + "long_term/attention_fcn/"
        fork = self._fork_point()
        if self.rnn_first:
            ops.rnn_multi("clsr_rnn_fwd_multi", grus, t4d, seq_len, ls, Hn, T)
        g2_tag = self.g2_stream if g2_side is not None else None
        if g2_tag and g2_tag != "@lt":
            # the causal GRU on the weight-gradient stream, idle in the forward pass: behind it on @lt the long-term attention
            # finished ~70 us after the short-term one and the heads waited for it (profiles/r05_step_timeline_fp32.txt)
            with self._branch(g2_tag, after=fork, name="@g2"):
                ops.rnn_multi("clsr_rnn_fwd_multi", [g2_side], None, seq_len, ls, Hn, T)
        with self._branch("@lt", after=fork):
            if g2_side is not None and g2_tag == "@lt":
                ops.rnn_multi("clsr_rnn_fwd_multi", [g2_side], None, seq_len, ls, Hn, T)
            att_long = self._att_fwd("lt", lt, hist, ulong, Hn, 1, T, D, Du, seq_len, ls, training)
        if not self.rnn_first:
            ops.rnn_multi("clsr_rnn_fwd_multi", grus, t4d, seq_len, ls, Hn, T)
        # ---- short term attention: query = [short_term_intention | target]
        Qs = Du + D
        q = self._buf("st.q", B, Qs)
        call("clsr_query_concat", short_int, Du, G, target, D, B, Du, D, q, Qs)
        att_short = self._att_fwd("st", st + "attention_fcn/", rnn_out, q, Hn, G, T, H, Qs, seq_len, ls, training,
                                  q_hist=short_int)
        # ---- alpha gate
        self._join()
        if after_attention is not None:
            with self._branch("@aux"):
                after_attention(dict(att_fea_long=att_long, att_fea_short=att_short, hist_mean=hmean,
                                     hist_recent=hrec))
        alpha = self._buf("alpha", B)
        mo = self._buf("model_output", B, 2 * D)
        if self._heads_defer:
            # training step: everything from here to d(model_output) is the first clsr_heads_fused launch (_train_step)
            logit = self._buf("lg.logit", B)
        elif not hp.manual_alpha:
            nfs = H if hp.predict_long_short else 0
            ld = _pad4(self.a_in)
            ain = self._buf("al.in", B, ld)
            call("clsr_alpha_concat", fs, nfs, target, att_long, att_short, f["time_to_now"], T, T - 1,
                 G if f.get("compact") else 1, B, G, D, ain, ld)
            al_logit = self._mlp_fwd("al", CL + "fcn_alpha/nn_part/", ain, ld, ld, (self.A0, self.A1), B, training)
            call("clsr_alpha_fuse_fwd", al_logit, 0.0, att_long, att_short, target, B, G, D, alpha, mo)
        else:
            call("clsr_alpha_fuse_fwd", None, float(hp.manual_alpha_value), att_long, att_short, target, B, G, D,
                 alpha, mo)
        if not self._heads_defer:
            logit = self._mlp_fwd("lg", "sequential/logit_fcn/nn_part/", mo, 2 * D, 2 * D, (self.L0, self.L1), B, training)
        return dict(logit=logit, alpha=alpha, att_fea_long=att_long, att_fea_short=att_short, hist_input=hist,
                    hist_mean=hmean, hist_recent=hrec, target=target, u_long=ulong, u_short=ushort,
                    short_intention=short_int, rnn_out=rnn_out, model_output=mo, causal_state=fs,
                    w_long=self._buf("lt.wts", Hn, T), w_short=self._buf("st.wts", B, T), q_short=q)

    def _heads_bwd_launches(self, f, out, B, G, Hn, Gl, lscale, fused_tail, dlogit, dtarget, dL, dS, dfs):
        """Backward of the row-level heads as the chain of launches (any widths; synchronised batch-norm statistics)."""
        hp, D, H = self.hp, self.D, self.H
        with self._dw_batched(late=True):
            dmo = self._mlp_bwd("lg", "sequential/logit_fcn/nn_part/", dlogit, out["model_output"], 2 * D, 2 * D, 2 * D,
                                (self.L0, self.L1), B, tail=(f["labels"], B // Gl, Gl, lscale) if fused_tail else None)
            self._join()              # the contrastive branch: its dL / dS are accumulated into from here on
            if not hp.manual_alpha:
                dal = self._buf("dalpha_logit", B)
                call("clsr_alpha_fuse_bwd", dmo, out["alpha"], 0.0, out["att_fea_long"], out["att_fea_short"], Hn, G, D,
                     dal, dL, dS, dtarget)
                ld = _pad4(self.a_in)
                dain = self._mlp_bwd("al", CL + "fcn_alpha/nn_part/", dal, self._buf("al.in", B, ld), ld, ld, self.a_in,
                                     (self.A0, self.A1), B)
        if not hp.manual_alpha:
            nfs = H if hp.predict_long_short else 0
            call("clsr_alpha_concat_bwd", dain, ld, nfs, Hn, G, D, dfs if nfs else None, dtarget, dL, dS)
        else:
            call("clsr_alpha_fuse_bwd", dmo, None, float(hp.manual_alpha_value), out["att_fea_long"],
                 out["att_fea_short"], Hn, G, D, None, dL, dS, dtarget)

    def _heads_fused_ok(self, B, G):
        """The two-launch form of the heads: the reference's default widths, local batch-norm statistics, the fused
        output-layer / softmax tail's conditions."""
        hp = self.hp
        return bool(self.heads_fused and type(self) is CLSRNet and not hp.manual_alpha and hp.predict_long_short
                    and (self.dp_stats_hook is None or self.heads_comm is not None) and self.fused_logit_tail
                    and G == hp.train_num_ngs + 1
                    and query("clsr_heads_fused_supported", B, G, self.D, self.H, self.a_in, self.A0, self.A1,
                              self.L0, self.L1))

    def _heads_ws(self):
        return self._buf("heads.ws", (int(query("clsr_heads_fused_workspace_bytes")) + 3) // 4)

    def _heads_fused_step(self, f, out, B, G, Hn, T, lscale, dlogit, dtarget, dL, dS, dfs):
        """alpha gate -> alpha MLP -> fusion -> logit MLP -> softmax loss -> their backward: two persistent launches
        (csrc/headsfused.hip) around the join with the contrastive branch, then the four dense layers' weight gradients in
        the step's batched dW launch (they read the activations / dz tensors the two launches wrote)."""
        P, Gd, D = self.P, self.Gd, self.D
        al, lg = CL + "fcn_alpha/nn_part/", "sequential/logit_fcn/nn_part/"
        A0, A1, L0, L1 = self.A0, self.A1, self.L0, self.L1
        ld = _pad4(self.a_in)
        parts = query("clsr_heads_fused_parts", B, G)
        ws = self._heads_ws()
        buf = self._buf
        bns = [self.bn[al + "batch_normalization/"], self.bn[al + "batch_normalization_1/"],
               self.bn[lg + "batch_normalization/"], self.bn[lg + "batch_normalization_1/"]]
        wp_al = buf("mlp.wp.al", 512 * (256 + 4))[: parts * (A1 + 4)]
        wp_lg = buf("mlp.wp.lg", 512 * (256 + 4))[: parts * (L1 + 4)]
        ain, mo = buf("al.in", B, ld), out["model_output"]
        z = {k: buf(k, B, n) for k, n in (("al.z0", A0), ("al.z1", A1), ("lg.z0", L0), ("lg.z1", L1), ("al.dz0", A0),
                                          ("al.dz1", A1), ("lg.dz0", L0), ("lg.dz1", L1))}
        pk = self.packed
        d = ops.heads_desc(
            fs=out["causal_state"], target=out["target"], att_long=out["att_fea_long"], att_short=out["att_fea_short"],
            tnow=f["time_to_now"], labels=f["labels"], tnow_stride=T, tnow_col=T - 1,
            tnow_group=G if f.get("compact") else 1, B=B, G=G, D=D, nfs=self.H, a_in=self.a_in, ld=ld, A0=A0, A1=A1,
            L0=L0, L1=L1,
            al_w0=pk["al.W0"][0], al_w1=pk["al.W1"][0], lg_w0=pk["lg.W0"][0], lg_w1=pk["lg.W1"][0],
            al_w0T=pk["al.W0^T"][0], al_w1T=pk["al.W1^T"][0], lg_w0T=pk["lg.W0^T"][0], lg_w1T=pk["lg.W1^T"][0],
            kp_al_w0=pk["al.W0"][1], kp_al_w1=pk["al.W1"][1], kp_lg_w0=pk["lg.W0"][1], kp_lg_w1=pk["lg.W1"][1],
            kp_al_w0T=pk["al.W0^T"][1], kp_al_w1T=pk["al.W1^T"][1], kp_lg_w0T=pk["lg.W0^T"][1],
            kp_lg_w1T=pk["lg.W1^T"][1],
            al_b0=P[al + "b_nn_layer0"], al_b1=P[al + "b_nn_layer1"], al_wout=P[al + "w_nn_output"],
            al_bout=P[al + "b_nn_output"], lg_b0=P[lg + "b_nn_layer0"], lg_b1=P[lg + "b_nn_layer1"],
            lg_wout=P[lg + "w_nn_output"], lg_bout=P[lg + "b_nn_output"],
            bn=[dict(gamma=b.gamma, beta=b.beta, moving_mean=b.moving_mean, moving_var=b.moving_var, scale=b.scale,
                     shift=b.shift, mean=b.mean, invstd=b.invstd, coef=b.coef, dgamma=b.dgamma, dbeta=b.dbeta) for b in bns],
            momentum=BN_MOMENTUM, eps=BN_EPS, lscale=lscale,
            ain=ain, al_z0=z["al.z0"], al_z1=z["al.z1"], alpha=out["alpha"], mo=mo, lg_z0=z["lg.z0"], lg_z1=z["lg.z1"],
            logit=out["logit"], dlogit=dlogit, loss=self.losses[0:],
            lg_dz1=z["lg.dz1"], lg_dz0=z["lg.dz0"], dmo=buf("lg.dX", B, 2 * D), al_dz1=z["al.dz1"], al_dz0=z["al.dz0"],
            lg_wp=wp_lg, al_wp=wp_al, dL=dL, dS=dS, dtarget=dtarget, dfs=dfs, workspace=ws, workspace_bytes=ws.numel() * 4,
            comm=(self.heads_comm or 0) if self.dp_stats_hook is not None else 0, abort_flag=self.adam_state[4:])
        with self._dw_batched(late=True):
            ops.heads_fused(1, d)
            self._rp(wp_lg, parts, L1 + 4, L1, Gd[lg + "w_nn_output"])
            self._rp(wp_lg[L1:], parts, L1 + 4, 1, Gd[lg + "b_nn_output"])
            self._dw(z["lg.z0"], L0, z["lg.dz1"], L1, B, L0, L1, Gd[lg + "w_nn_layer1"], L1, db=Gd[lg + "b_nn_layer1"],
                     aff=bns[2])
            self._dw(mo, 2 * D, z["lg.dz0"], L0, B, 2 * D, L0, Gd[lg + "w_nn_layer0"], L0, db=Gd[lg + "b_nn_layer0"])
            self._join()              # the contrastive branch: the second launch adds to its dL / dS
            ops.heads_fused(2, d)
            self._rp(wp_al, parts, A1 + 4, A1, Gd[al + "w_nn_output"])
            self._rp(wp_al[A1:], parts, A1 + 4, 1, Gd[al + "b_nn_output"])
            self._dw(z["al.z0"], A0, z["al.dz1"], A1, B, A0, A1, Gd[al + "w_nn_layer1"], A1, db=Gd[al + "b_nn_layer1"],
                     aff=bns[0])
            self._dw(ain, ld, z["al.dz0"], A0, B, self.a_in, A0, Gd[al + "w_nn_layer0"], A0, db=Gd[al + "b_nn_layer0"])

    # ------------------------------------------------------------------ training step
    def train_step(self, f, apply=True):
        """forward + backward (+ clip + Adam when ``apply``) on an uploaded feed.  Losses land in
        self.losses (device doubles: data, regular, contrastive, discrepancy).  Data-parallel runs call
        with apply=False, all-reduce the gradient buffers, then call :meth:`_apply_updates`."""
        if self._aborted:
            self.check_abort()      # sticky: an aborted net is restored from a checkpoint first
        with ops.stream_scope():
            return self._planned(("train", bool(apply)), f, lambda: self._train_step(f, apply))

    def _train_step(self, f, apply):
        hp, P, Gd = self.hp, self.P, self.Gd
        self._cur_feed = f
        B, T = f["B"], f["T"]
        G = self.G_train if self.dedup else 1
        if B % G:
            raise ValueError("training feed rows (%d) must be a multiple of 1+train_num_ngs (%d)" % (B, G))
        Hn = B // G
        D, Du, H, Di, Dc = self.D, self.Du, self.H, self.Di, self.Dc
        hs = 1 if f.get("compact") else G
        seq_len, ls = f["seq_len"], hs
        heads_fused = self._heads_fused_ok(B, G)
        # gradient accumulators (zeroed every step) and the involved-row flags depend on the feed only: zeroed /
        # marked on a side stream underneath the forward's first kernels
        zpool = self._buf("zero_pool", Hn * T * (2 * D + H) + B * D * 2 + Hn * (3 * D + H + Du))
        fl = self.tab_flags

        def zero_and_mark():
            # ONE zero-fill launch: squared norms (16) + loss terms (8), the gradient pool, the counters of the
            # history-id sort and the distinct-user counter of the discrepancy loss
            zr = [(self.stats24.data_ptr(), 24 * 8), (zpool.data_ptr(), zpool.numel() * 4)]
            if self.sorted_hist_grad:
                counts = self._sort_counts()[0]
                zr.append((counts.data_ptr(), counts.numel() * 4))
                self._counts_zeroed = True
            if "user_long" in self.tables:
                zr.append((self.ucount.data_ptr(), 4))
                self._ucount_zeroed = True
            if heads_fused:
                zr.append((self._heads_ws().data_ptr(), int(query("clsr_heads_fused_counter_bytes"))))
            if self._att_qh("st") and G > 1:
                # split short-term query: the history-level columns of d(query) only receive the V path (an
                # accumulating product): cleared here with the other accumulators
                dq_st = self._buf("st.dq", B, Du + D)
                zr.append((dq_st.data_ptr(), dq_st.numel() * 4))
            ops.multi("clsr_zero_multi", ops.ZeroDesc, zr)
            # involved-row flags (tf.unique id sets)
            ops.multi("clsr_mark_rows_multi", ops.MarkDesc, [
                (f["item_history"].data_ptr(), fl["item"].data_ptr(), Hn, hs * T, T, 0),
                (f["items"].data_ptr(), fl["item"].data_ptr(), B, 1, 1, 0),
                (f["item_cate_history"].data_ptr(), fl["cate"].data_ptr(), Hn, hs * T, T, 0),
                (f["cates"].data_ptr(), fl["cate"].data_ptr(), B, 1, 1, 0),
                (f["users"].data_ptr(), fl["user_long"].data_ptr(), Hn, hs, 1, 0),
                (f["users"].data_ptr(), fl["user_short"].data_ptr(), Hn, hs, 1, 0)])
            self._dp_hook("flags_ready")      # involved-row byte maps are final: their exchange hides under the forward
            if (apply and self.tick_early and not self.capture_grads and self.dp_hooks is None
                    and "user_long" in self.tables):
                # the distinct-user count of the discrepancy loss and the Adam clock need the marks only: HERE, under the
                # forward, instead of as the first launch of the update phase (9 us + a stream hop in front of the table
                # regulariser).  Not under data parallelism: the count is over the EXCHANGED byte map.
                call("clsr_count_flags_tick", fl["user_long"], self.dims["Vu"], self.ucount, self.adam_state,
                     float(hp.learning_rate), 0.9, 0.999)
                self._ticked = True
            if apply and self.dp_hooks is None:
                # ... and so do the ascending lists of the touched rows of the big tables (lazy Adam): one single-workgroup
                # scan over the 100M-row byte map of BASELINE configs[4] is 0.33 ms + 0.15 ms of list writing -- under the
                # forward here, they were the first half millisecond of the update phase
                self._early_lists = {k: self._involved_list(k) for k, t in self.tables.items()
                                     if t.numel() > self.rowlist_min_elems}
        o = [0]

        def take(*shape):
            n = int(np.prod(shape))
            t = zpool[o[0]:o[0] + n].view(*shape)
            o[0] += n
            return t
        dhist, drnn, dhist_lt = take(Hn, T, D), take(Hn, T, H), take(Hn, T, D)
        dtarget, dS = take(B, D), take(B, D)
        dL, dM, dR = take(Hn, D), take(Hn, D), take(Hn, D)
        dfs, dsi = take(Hn, H), take(Hn, Du)

        def contrastive(o_):
            call("clsr_contrastive", o_["att_fea_long"], o_["att_fea_short"], o_["hist_mean"], o_["hist_recent"],
                 seq_len, ls, Hn, G, D, int(hp.contrastive_length_threshold),
                 1 if hp.contrastive_loss == "triplet" else 0, float(hp.triplet_margin),
                 float(hp.contrastive_loss_weight), f["denom"], self.losses[2:], dL, dS, dM, dR)

        Gl = hp.train_num_ngs + 1
        lscale = 1.0 / ((B // Gl) * self.dp_world)
        # output layer of the logit MLP + data loss + their backward as ONE launch at the turn of the step
        fused_tail = (self.fused_logit_tail and B % Gl == 0
                      and bool(query("clsr_mlp_tail_softmax_supported", Gl, self.L1)))
        self._defer_logit_out = fused_tail
        self._heads_defer = heads_fused
        try:
            out = self._forward(f, True, contrastive, zero_and_mark)
        finally:
            self._defer_logit_out = False
            self._heads_defer = False
        assert self.last_shape == (B, T, G, Hn)
        # ---- losses on the forward outputs
        dlogit = self._buf("dlogit", B)
        if not fused_tail:
            call("clsr_softmax_loss", out["logit"], f["labels"], B // Gl, Gl, lscale, self.losses[0:], dlogit)
        # ---- logit MLP, fusion, alpha MLP (their four small weight gradients: ONE multi-job launch at the end)
        if heads_fused:
            self._heads_fused_step(f, out, B, G, Hn, T, lscale, dlogit, dtarget, dL, dS, dfs)
        else:
            self._heads_bwd_launches(f, out, B, G, Hn, Gl, lscale, fused_tail, dlogit, dtarget, dL, dS, dfs)
        dul = None
        if self.lt_bwd_early:
            # long-term attention backward (dL is final here) on the side stream from NOW, beside the short-term one
            with self._branch("@lt", after=self._fork_point()):
                dul = self._att_bwd("lt", CL + "long_term/attention_fcn/", dL, out["hist_input"], out["u_long"],
                                    dhist_lt, Hn, 1, T, D, Du, seq_len, ls)
                self._dw_flush()
        # ---- short-term attention
        st = CL + "short_term/"
        Qs = Du + D
        dq = self._att_bwd("st", st + "attention_fcn/", dS, out["rnn_out"], out["q_short"], drnn, Hn, G, T, H, Qs,
                           seq_len, ls, q_hist=out["short_intention"], dq_hist=dsi)
        call("clsr_query_split_bwd", dq, Qs, G, Hn, Du, D, dsi, Du, dtarget, D)
        # ---- sequence encoders: ONE fused backward-through-time launch, then the batched weight grads
        M = Hn * T
        NX = self.NX
        hist = out["hist_input"]
        scat_early = False
        # speed mode: the gradients of the input projections leave the backward-through-time kernel as bf16 -- their
        # consumers (input-side / hidden-side weight gradients, d(hist) = dPin . W^T) run on the bf16 matrix pipe anyway
        dpin_h = (self.dpin_h and "xw^T" in self.packed_h and H % 8 == 0 and Du % 8 == 0 and D % 8 == 0
                  and self._hgemm_fits(NX, D)
                  and (self._t4_kind != "time4lstm" or "t4.tw^T" in self.packed_h))
        dPinAll = self._buf("xw.dPin", M, NX, dtype=torch.bfloat16 if dpin_h else F32)
        grus, t4d = [], None
        dushort = dsi
        if hp.interest_evolve:
            dushort = self._buf("d_u_short", Hn, Du)
            grus.append(self._gru_bwd_desc("g1", st + "short_term_intention/gru_cell/", Du, dPinAll, Hn, T, dsi, None,
                                           dushort))
        if self._t4_scope is not None:
            t = self._t4_scope
            t4off = self._enc_off("t4")
            act, cst, tiled = self._t4_act(Hn, T, True)
            t4d = ops.t4_desc(H, Wm=P[t + "kernel"][D:], ldm=4 * H, act=act, cst=cst, act_tiled=tiled, dout_seq=drnn,
                              dPin=dPinAll[:, t4off:], lddp=NX, products=self.rnn_products)
        else:
            grus.append(self._gru_bwd_desc("gs", st + "simple_gru/gru_cell/", H, dPinAll, Hn, T, None, drnn, None))
        if (not hp.manual_alpha) and hp.predict_long_short:
            grus.append(self._gru_bwd_desc("g2", CL + "causal2/causal2/gru_cell/", H, dPinAll, Hn, T, dfs, None, None))
        # ---- long-term attention backward (dL has been final since the alpha gate): forked here so that it runs
        #      on the side stream underneath the T-serial backward-through-time (own scratch + own d(hist))
        fork = self._fork_point()
        if self.rnn_first:
            ops.rnn_multi("clsr_rnn_bwd_multi", grus, t4d, seq_len, ls, Hn, T)
        if dul is None:
            with self._branch("@lt", after=fork):
                dul = self._att_bwd("lt", CL + "long_term/attention_fcn/", dL, out["hist_input"], out["u_long"],
                                    dhist_lt, Hn, 1, T, D, Du, seq_len, ls)
                self._dw_flush()
        if not self.rnn_first:
            ops.rnn_multi("clsr_rnn_bwd_multi", grus, t4d, seq_len, ls, Hn, T)
        # the row scatters of the user / target-item / target-category lookups need nothing from the encoder tail
        # below: they go to the side streams early instead of queueing behind it -- long-term user rows and target
        # rows UNDER the backward-through-time launch (latency bound, leaves issue slots; beside the MFMA-saturated
        # fused tail they ran 4x slower and the history-row sums waited for them), the short-term user rows (final
        # only now) beside the small d TT product
        scat_early = self.early_scatter and self.sorted_hist_grad and self.split_emb_grad and self.overlap
        if scat_early:
            self._scatter_rows_early(f, dul, None, dtarget, Hn, B, hs, fork)
            self._scatter_rows_early(f, None, dushort, None, Hn, B, hs, self._fork_point())
            if (self.early_user_update and apply and self._ticked and self.dp_hooks is None and not self.capture_grads
                    and "user_long" in self.tables):
                # (same stream as the two scatters: ordered behind them; a branch name of its own -- only the end of the update
                #  phase waits for it)
                with self._branch("@aux", after=self._fork_point(), name="@uupd"):
                    self._update_user_tables_early()
        self._folded = False
        self._fold_args = ((dhist_lt, dM, dR, seq_len, ls) if (self.sorted_hist_grad and self.det_grads and self.hist_grad_two
                                                              and dhist.dtype == F32) else None)
        if self._enc_bwd_fused_ok(dpin_h):
            (self._enc_bwd_fused_h if self.bf16 else self._enc_bwd_fused_x3 if (self.x3_enc or self.enc_x6) else
             self._enc_bwd_fused)(f, hist, dPinAll, dhist, Hn, T, hs)
        else:
          # input-side weights of every encoder in one reduction; d(hist) in one product; the hidden-side / time-feature
          # weight gradients ride in the same multi-job launch as the input-side one (they all depend on dPin only)
          with self._dw_batched():
              self._dw(hist, D, dPinAll, NX, M, D, NX, self._buf("xw.dW", D, NX), NX, db=self._buf("xw.db", NX),
                       dy_bf16=int(dpin_h))
              self._gemm(dPinAll, NX, "xw^T", M, NX, D, dhist, D, acc=1)
              if self._t4_kind is not None:
                  self._t4_bwd_weights(f, dPinAll, Hn, T, hs)
              else:
                  self._gru_bwd_hidden("gs", st + "simple_gru/gru_cell/", H, dPinAll, Hn, T)
              if hp.interest_evolve:
                  self._gru_bwd_hidden("g1", st + "short_term_intention/gru_cell/", Du, dPinAll, Hn, T)
              if (not hp.manual_alpha) and hp.predict_long_short:
                  self._gru_bwd_hidden("g2", CL + "causal2/causal2/gru_cell/", H, dPinAll, Hn, T)
        # ---- join the long-term attention branch; its d(hist) contribution was accumulated separately (NOT the early row
        #      scatters: nothing reads their tables before the update phase -- every wait is a barrier packet in front of
        #      the history-row sums)
        side_dense = self.flush_side and self.overlap and self.dw_stream
        # (deterministic segmented sums update a row with a plain read-modify-write: the target / user row sums of the
        #  same tables -- the @scat branches -- must have finished, which they have long since: the wait costs a packet)
        keep = () if (self.det_grads and self.sorted_hist_grad) else ("@scat",)
        self._join(but=keep + ("@encw", "@uupd") if side_dense else keep + ("@uupd",))
        if side_dense:
            # the dense path from here on (batched reduction of every weight gradient, unpacking, later the dense
            # regulariser + Adam) does not meet the embedding path (gradient tables, table regulariser / Adam) again:
            # it stays on the weight-gradient stream, where the last partial-sum kernel has just finished, and the main
            # stream goes straight to the embedding gradients (the reduction used to sit between them: ~140 us)
            fork_dense = self._fork_point()
            # (only the step that applies its own update right away keeps the dense update on this stream: under data
            # parallelism the gradients are exchanged first, and the update must be ordered behind THAT, not behind this event)
            self._dense_fork = fork_dense if (apply and self.dp_hooks is None) else None
            with self._branch("@dw0", after=fork_dense, name="@dense"):
                self._dense_grads_final()
        else:
            self._dense_grads_final()
        # d(hist) = dhist + dhist_lt (the long-term branch's share): the segmented sums add the two on the fly; the
        # float-atomics path gets one tensor
        if self._folded:       # (the fused encoder tail added the long-term share and the prologue's shares to d(hist))
            dhist_lt = dM = dR = None
        elif not (self.sorted_hist_grad and self.hist_grad_two):
            call("clsr_axpby", dhist, dhist, 1.0, dhist_lt, 1.0, dhist.numel())
            dhist_lt = None
        # ---- embedding gradients (IndexedSlices values -> dense grad tables + squared norms)
        ss = self.sumsq_tab
        if self.sorted_hist_grad:
            # segmented sums over the ids sorted during the forward (119 us vs 350 us for float atomics).  The
            # lookup sites are independent (distinct norm slots; the two sites of a table meet in fp32 atomics):
            # category history + category targets, item / user row scatters and the item history run on three
            # streams that are idle by now instead of one after the other
            fork = self._fork_point()
            with self._branch("@lt" if self.split_emb_grad else "@main", after=fork):
                self._hist_grad_sorted(dhist, dM, dR, Hn, T, seq_len, ls, ss, only="cate", dhist2=dhist_lt, dtarget=dtarget)
                if not scat_early:
                    self._scatter_cate_targets(f, dtarget, B)
                self._dp_hook("table_ready", "cate")
            if not scat_early:
                with self._branch("@aux" if self.split_emb_grad else "@main", after=fork):
                    self._scatter_user_item_rows(f, dul, dushort, dtarget, Hn, B, hs)
            self._hist_grad_sorted(dhist, dM, dR, Hn, T, seq_len, ls, ss, only="item", dhist2=dhist_lt, dtarget=dtarget)
            self._join(but=("@dense", "@uupd"))
            self._dp_hook("table_ready", "item")
        else:
            call("clsr_gather_hist_bwd", dhist, dM, dR, f["item_history"], f["item_cate_history"], hs * T, seq_len,
                 ls, Hn, T, Di, Dc, hp.contrastive_recent_k, self.tab_grad["item"], self.tab_grad["cate"], ss[0:])
            call("clsr_scatter_add_rows", dtarget, D, 0, f["items"], 1, B, Di, self.tab_grad["item"], ss[2:])
            call("clsr_scatter_add_rows", dtarget, D, Di, f["cates"], 1, B, Dc, self.tab_grad["cate"], ss[3:])
            call("clsr_scatter_add_rows", dul, Du, 0, f["users"], hs, Hn, Du, self.tab_grad["user_long"], ss[6:])
            call("clsr_scatter_add_rows", dushort, Du, 0, f["users"], hs, Hn, Du, self.tab_grad["user_short"], ss[7:])
            for k in ("cate", "user_long", "user_short", "item"):
                self._dp_hook("table_ready", k)
        if apply:
            self._apply_updates()
            self._dense_fork = None
        elif self.dp_hooks is None:
            self._join()              # callers that read the gradients themselves: everything back on this stream
        return out

    def _dense_grads_final(self):
        """One batched reduction of every weight gradient of the main stream (joins the weight-gradient stream),
        the assembled blocks go back to their variables, and the data-parallel exchange is told."""
        self._dw_flush(tag="")
        self._unpack_grads()
        self._dp_hook("dense_ready")          # every dense gradient is final: all-reduce under the embedding kernels

    def _sort_tables(self):
        return (("item", "item_history", self.dims["Vi"], 0, self.Di, 0),
                ("cate", "item_cate_history", self.dims["Vc"], self.Di, self.Dc, 1))

    def _sort_counts(self):
        bits = [query("clsr_sort_ids_bits", V) for _, _, V, _, _, _ in self._sort_tables()]
        return self._buf("sort.counts", sum(1 << b for b in bits), dtype=torch.int32), bits

    def _sort_hist_ids(self, f, Hn, T, hs):
        """(row id, position) pairs of both history lookups grouped by row id: hand-written counting sort, one
        zeroing launch + three launches for both tables together (csrc/sparse.hip: clsr_sort_ids_multi)."""
        n = Hn * T
        tabs = self._sort_tables()
        if self.det_grads:
            return self._sort_ids_stable(f, Hn, T, hs)
        counts, bits = self._sort_counts()
        if not self._counts_zeroed:      # (the training step clears them with its other accumulators)
            call("clsr_zero_floats", counts.view(F32), counts.numel())
        self._counts_zeroed = False
        rows, o = [], 0
        for (name, fkey, V, _, _, _), b in zip(tabs, bits):
            keys = self._buf("sort.keys." + name, n, dtype=torch.int32)
            perm = self._buf("sort.perm." + name, n, dtype=torch.int32)
            rows.append((f[fkey].data_ptr(), keys.data_ptr(), perm.data_ptr(), counts[o:].data_ptr(), Hn, hs * T, T, b))
            o += 1 << b
        ops.sort_ids_multi(rows)

    def _det_merged(self, f):
        """CLSR graph: the target rows' ids ride behind the history lookup's ids in ONE sorted list per table (item,
        category), so that the segmented sums write every gradient row exactly once -- stored, not added: no row read, no
        second launch for the target site.  -> {table: (feed key of the target ids, rows)} or {}"""
        if type(self) is not CLSRNet or "user_long" not in self.tables:
            return {}
        return {"item": ("items", f["B"]), "cate": ("cates", f["B"])}

    def _det_sites(self, f):
        """row-level lookup sites with a sorted list of their own: (list name, feed key, vocabulary, rows, row stride)"""
        if type(self) is not CLSRNet or "user_long" not in self.tables:
            return ()
        G = self.G_train if self.dedup else 1
        hs = 1 if f.get("compact") else G
        return (("user", "users", self.dims["Vu"], f["B"] // G, hs),)

    def _sort_ids_stable(self, f, Hn, T, hs):
        """Stable radix sort (ascending ids, equal ids in slice order) of the ids of every lookup site of the step: the
        two history lookups (CLSR graph: each followed by the target rows' ids of the same table) and the user rows: one
        launch set."""
        n = Hn * T
        rows, total = [], 0
        merged = self._det_merged(f)
        for name, fkey, V, _, _, _ in self._sort_tables():
            extra = merged.get(name)
            ne = n + (extra[1] if extra else 0)
            keys = self._buf("sort.keys." + name, ne, dtype=torch.int32)
            perm = self._buf("sort.perm." + name, ne, dtype=torch.int32)
            row = (f[fkey].data_ptr(), keys.data_ptr(), perm.data_ptr(), Hn, hs * T, T, max(1, (V - 1).bit_length()))
            if extra:
                row += (f[extra[0]].data_ptr(), extra[1], 1)
            rows.append(row)
            total += ne
        for name, fkey, V, nr, stride in self._det_sites(f):
            keys = self._buf("sort.keys." + name, nr, dtype=torch.int32)
            perm = self._buf("sort.perm." + name, nr, dtype=torch.int32)
            rows.append((f[fkey].data_ptr(), keys.data_ptr(), perm.data_ptr(), nr, stride, 1, max(1, (V - 1).bit_length())))
            total += nr
        ws = self._buf("sort.ws", query("clsr_sort_ids_stable_workspace_bytes", total, len(rows)), dtype=torch.uint8)
        ops.sort_ids_stable_multi(rows, ws)

    def _segsum(self, tag, rows):
        ws = self._buf("segsum.ws." + tag, ops.segsum_workspace_bytes(rows), dtype=torch.uint8)
        ops.segsum_multi(rows, ws)

    def _site_row(self, name, src, ld, col0, C, n, grad, sumsq):
        """segmented-sum descriptor of a row-level lookup site (users): slice e = row perm[e] of ``src``; the table has no
        other lookup site, its gradient rows are zero: the totals are stored"""
        keys = self._buf("sort.keys." + name, n, dtype=torch.int32)
        perm = self._buf("sort.perm." + name, n, dtype=torch.int32)
        return (src.data_ptr(), 0, 0, 0, keys.data_ptr(), perm.data_ptr(), 0, grad.data_ptr(), sumsq.data_ptr(), n, 0, 0, 1,
                ld, col0, C, 1, grad.shape[1], 0, 1)

    def _scatter_user_item_rows(self, f, dul, dushort, dtarget, Hn, B, hs):
        """User rows (long / short table) and the target items' rows: ONE launch, blockIdx.y = lookup site (``None``:
        that site is not part of this launch)."""
        tg, ss, dp_ = self.tab_grad, self.sumsq_tab, lambda t: t.data_ptr()
        D, Du, Di = self.D, self.Du, self.Di
        if self.det_grads and self._det_sites(f):
            rows = []
            if dul is not None:
                rows.append(self._site_row("user", dul, Du, 0, Du, Hn, tg["user_long"], ss[6:]))
            if dushort is not None:
                rows.append(self._site_row("user", dushort, Du, 0, Du, Hn, tg["user_short"], ss[7:]))
            # (the target rows' sums are part of the history lookups' lists: _hist_grad_sorted)
            if rows:
                self._segsum("rows%d%d" % (dul is not None, dushort is not None), rows)
            if dul is not None:
                self._dp_hook("table_ready", "user_long")
            if dushort is not None:
                self._dp_hook("table_ready", "user_short")
            return
        sites = []
        if dul is not None:
            sites.append((dp_(dul), dp_(f["users"]), dp_(tg["user_long"]), dp_(ss[6:]), hs, Du, 0, Hn, Du))
        if dushort is not None:
            sites.append((dp_(dushort), dp_(f["users"]), dp_(tg["user_short"]), dp_(ss[7:]), hs, Du, 0, Hn, Du))
        if dtarget is not None:
            sites.append((dp_(dtarget), dp_(f["items"]), dp_(tg["item"]), dp_(ss[2:]), 1, D, 0, B, Di))
        ops.multi("clsr_scatter_add_rows_multi", ops.ScatterDesc, sites)
        if dul is not None:
            self._dp_hook("table_ready", "user_long")
        if dushort is not None:
            self._dp_hook("table_ready", "user_short")

    def _scatter_rows_early(self, f, dul, dushort, dtarget, Hn, B, hs, fork):
        """Scatters that do not read d(hist), on the side streams behind ``fork`` (an event of the compute stream).  The
        long-term user gradient is final on @lt, which the user launch joins and the category launch follows in stream
        order."""
        with self._branch("@aux", after=fork, name="@scat"):
            if dul is not None:
                self._join(only="@lt", keep=True)
            self._scatter_user_item_rows(f, dul, dushort, dtarget, Hn, B, hs)
        if dtarget is not None:
            with self._branch("@lt", after=fork, name="@scat"):
                self._scatter_cate_targets(f, dtarget, B)

    def _scatter_cate_targets(self, f, dtarget, B):
        if self.det_grads and self._det_merged(f):
            return          # (summed with the category history lookup: _hist_grad_sorted)
        else:
            call("clsr_scatter_add_rows", dtarget, self.D, self.Di, f["cates"], 1, B, self.Dc, self.tab_grad["cate"],
                 self.sumsq_tab[3:])

    def _hist_grad_sorted(self, dhist, dM, dR, Hn, T, seq_len, ls, ss, only=None, dhist2=None, dtarget=None):
        """IndexedSlices of the history lookups -> dense gradient tables via segmented sums over the
        sorted ids (no float atomics on hot rows).  ``only``: one table ("item" / "cate").  Not bit-reproducible from run
        to run: the counting sort (csrc/sparse.hip) leaves the order INSIDE a run of equal ids to its cursor claims, and
        the run is summed in that order -- the fp32 sums (and the squared norms that feed the clip factor) can differ in
        the last bits, like the order of the per-run atomics always could."""
        n = Hn * T
        k = self.hp.contrastive_recent_k
        if self.det_grads:
            rows = []
            merged = self._det_merged(self._cur_feed) if dtarget is not None else {}
            for name, _, V, col0, C, slot in self._sort_tables():
                if only is not None and name != only:
                    continue
                ne = n + (merged[name][1] if name in merged else 0)
                keys = self._buf("sort.keys." + name, ne, dtype=torch.int32)
                perm = self._buf("sort.perm." + name, ne, dtype=torch.int32)
                ptr = lambda t: 0 if t is None else t.data_ptr()
                row = (dhist.data_ptr(), ptr(dhist2), ptr(dM), ptr(dR), keys.data_ptr(), perm.data_ptr(),
                       seq_len.data_ptr(), self.tab_grad[name].data_ptr(), ss[slot:].data_ptr(), ne,
                       int(dhist.dtype == torch.bfloat16), ls, T, self.D, col0, C, k, C, 0)
                if name in merged:
                    # second source = the target rows' gradients [B, D] (same column slice), their squared norms in the
                    # target site's slot (2 item, 3 category); every row is written once: stored
                    row += (1, dtarget.data_ptr(), ss[slot + 2:].data_ptr(), n, self.D, col0)
                else:
                    row += (0, 0, 0, 0, 0, 0)
                rows.append(row + (0, 0))
            self._segsum("hist." + (only or "all"), rows)
            return
        for name, _, V, col0, C, slot in self._sort_tables():
            if only is not None and name != only:
                continue
            keys = self._buf("sort.keys." + name, n, dtype=torch.int32)
            perm = self._buf("sort.perm." + name, n, dtype=torch.int32)
            blk = query("clsr_gather_bwd_sorted_max_cols", self.D, col0, C, C, 0)   # 256 with 16-byte accesses, else 64
            for c0 in range(0, C, blk):   # column blocks; squared norms accumulate in the slot
                call("clsr_gather_bwd_sorted2", dhist, dhist2, dM, dR, keys, perm, seq_len, ls, n, T, self.D, col0 + c0,
                     min(blk, C - c0), k, self.tab_grad[name], C, c0, ss[slot:])

    #: tables with more elements than this are regularised / lazily updated through the compacted list of
    #: their involved rows instead of a sweep over all V*C elements (100M-item catalogues)
    rowlist_min_elems = 1 << 26

    def rows_bound(self, key):
        """Upper bound on the rows of table ``key`` touched in the current step (all data-parallel ranks)."""
        B, T, G, Hn = self.last_shape
        V = self.tables[key].shape[0]
        per_rank = Hn if key.startswith("user") else Hn * T + B
        return int(min(V, per_rank * self.dp_world))

    def _involved_list(self, key):
        """(ids, count, cap): ascending ids of the flagged rows of table ``key``, count on the device."""
        V = self.tables[key].shape[0]
        cap = self.rows_bound(key)
        nws = query("clsr_flags_compact_workspace_bytes", V)
        ws = self._buf("rows.ws." + key, nws, dtype=torch.uint8)
        ids = self._buf("rows.ids." + key, cap, dtype=torch.int32)
        count = self._buf("rows.count." + key, 2, dtype=torch.int32)
        call("clsr_flags_compact", self.tab_flags[key], V, ids, cap, count, ws, nws)
        return ids, count, cap

    def _update_spec(self):
        """(table, partner, reg-norm slot, discrepancy grad scale, discrepancy loss scale, loss slot, Adam clip-norm
        base slot, number of norm slots) per trained embedding table."""
        wd = float(self.hp.discrepancy_loss_weight)
        return (("item", None, 4, 0.0, 0.0, None, 0, 3), ("cate", None, 5, 0.0, 0.0, None, 1, 3),
                ("user_long", "user_short", 8, -2.0 * wd, -wd, self.losses[3:], 6, 2),
                ("user_short", "user_long", 9, -2.0 * wd, 0.0, None, 7, 2))

    def _sweep_row(self, key, partner, slot, dscale, dloss_scale, dloss, base, nsum):
        tb, ss = self.tables, self.sumsq_tab
        V, C = tb[key].shape
        pt = tb[partner] if partner else None
        return (tb[key].data_ptr(), ops._ptr(pt), self.tab_grad[key].data_ptr(), self.tab_m[key].data_ptr(),
                self.tab_v[key].data_ptr(), self.tab_flags[key].data_ptr(), ss[slot:].data_ptr(), ops._ptr(dloss),
                ss[base:].data_ptr(), V, C, nsum, 2, dscale, dloss_scale, 0)

    def _update_user_tables_early(self):
        """Regulariser (+ discrepancy term) and Adam of the two user tables as soon as their gradients are final -- the row
        scatters of the long- and short-term user gradients, long before the end of the backward pass -- instead of in the sweep
        over all four tables that the step ends with (tf.clip_by_norm is per variable, base_model.py:290-296: a table's update
        needs its own norms only).  60 % of the values of that sweep at BASELINE configs[1]; what stays at the end of the step
        is the item and category tables, whose gradients the history-row sums finish last."""
        hp, tb = self.hp, self.tables
        spec = [sp for sp in self._update_spec() if sp[0] in ("user_long", "user_short")]
        if len(spec) != 2 or any(tb[sp[0]].numel() > self.rowlist_min_elems for sp in spec):
            return
        rows = [self._sweep_row(*sp) for sp in spec]
        clip = float(hp.max_grad_norm) if hp.is_clip_norm else 0.0
        ops.multi("clsr_tables_reg_multi" + self._th, ops.TableDesc, rows, float(hp.embed_l2), float(hp.embed_l1), self.ucount,
                  self.losses[1:])
        ops.multi("clsr_tables_adam_multi" + self._th, ops.TableDesc, rows, clip, self.adam_state, 0.9, 0.999, 1e-8, self.lazy)
        self._updated_early = {sp[0] for sp in spec}

    def _apply_updates(self):
        hp = self.hp
        ss = self.sumsq_tab
        Vu, Vi, Vc = self.dims["Vu"], self.dims["Vi"], self.dims["Vc"]
        l2e, l1e = float(hp.embed_l2), float(hp.embed_l1)
        tb, tg, fl = self.tables, self.tab_grad, self.tab_flags
        if "user_long" in tb:      # number of distinct users of the batch: the discrepancy loss is a mean over them
            if not self._ucount_zeroed:      # (the training step clears it with its other accumulators)
                call("clsr_zero_floats", self.ucount, 1)
            self._ucount_zeroed = False
        # the Adam clock ticks in the FIRST launch of the update phase (with the user count when there is one): the table
        # path and the dense path both only read it afterwards, so neither waits for the other (the clock used to tick
        # on the dense path, and the table Adam launches waited for the whole weight-gradient reduction: 50 us)
        tick_early = self.tick_early and not self.capture_grads
        lr = float(hp.learning_rate)
        if self._ticked:        # (count + clock went out with the row marks at the start of the step)
            self._ticked = False
        elif "user_long" in tb:
            call("clsr_count_flags_tick", fl["user_long"], Vu, self.ucount, self.adam_state if tick_early else None, lr,
                 0.9, 0.999)
        elif tick_early:
            call("clsr_adam_tick", self.adam_state, lr, 0.9, 0.999)
        clip = float(hp.max_grad_norm) if hp.is_clip_norm else 0.0
        # dense variables (regulariser + norms, Adam clock, Adam) on the @aux stream beside the table regulariser
        # (single-GPU step with the dense path on the weight-gradient stream: the dense update STAYS on that stream, behind the
        # batched reduction -- on @aux it started two stream hops, ~25 us, after the reduction had finished, and the dense
        # Adam launch was the last kernel of the step; CLSR_DENSE_UPD_AUX=1: as before)
        fork = getattr(self, "_dense_fork", None)
        on_dw = (fork is not None and not self.capture_grads and self.split_emb_grad and self.dense_upd_dw)
        with (self._branch("@dw0", after=fork, name="@denseupd") if on_dw else
              self._branch("@main" if (self.capture_grads or not self.split_emb_grad) else "@aux")):
            self._join(only="@dense")     # (the batched weight-gradient reduction, see _train_step)
            # (the Adam clock of the step ticks in the same launch: every reader is ordered after it)
            call("clsr_dense_reg_norm_tick_t", self.dense, self.dense_grad, self.seg_off, len(self.dense_names),
                 float(hp.layer_l2), float(hp.layer_l1), self.dense_sumsq, self.losses[1:],
                 None if (self.capture_grads or tick_early) else self.adam_state, lr, 0.9, 0.999, self.dense_reg_threads)
            if not self.capture_grads:
                call("clsr_dense_adam", self.dense, self.dense_grad, self.dense_m, self.dense_v, self.seg_of,
                     self.dense_sumsq, clip, self.adam_state, 0.9, 0.999, 1e-8, self.n_dense)
        lists, self._early_lists = self._early_lists, None
        if lists is None:
            lists = {k: self._involved_list(k) for k, t in tb.items() if t.numel() > self.rowlist_min_elems}
        done_early, self._updated_early = self._updated_early, set()
        spec = [sp for sp in self._update_spec() if sp[0] not in done_early]
        sweep = []
        for key, partner, slot, dscale, dloss_scale, dloss, base, nsum in spec:
            V, C = tb[key].shape
            pt = tb[partner] if partner else None
            if key in lists:
                ids, count, cap = lists[key]
                call("clsr_table_reg_rows" + self._th, tb[key], pt, ids, count, cap, C, l2e, l1e, dscale, dloss_scale,
                     self.ucount if partner else None, tg[key], ss[slot:], self.losses[1:], dloss)
            else:   # small tables: one launch sweeps all of them (blockIdx.y = table)
                sweep.append((tb[key].data_ptr(), ops._ptr(pt), tg[key].data_ptr(), self.tab_m[key].data_ptr(),
                              self.tab_v[key].data_ptr(), fl[key].data_ptr(), ss[slot:].data_ptr(), ops._ptr(dloss),
                              ss[base:].data_ptr(), V, C, nsum, 2, dscale, dloss_scale, 0))
        if sweep:
            ops.multi("clsr_tables_reg_multi" + self._th, ops.TableDesc, sweep, l2e, l1e, self.ucount, self.losses[1:])
        if self.capture_grads:  # test hook: pre-clip gradients (regularisers included) + squared norms
            self.captured = dict(dense={n: g.detach().clone() for n, g in self.Gd.items()},
                                 tables={k: g.detach().clone() for k, g in tg.items()},
                                 dense_sumsq=self.dense_sumsq.clone(), table_sumsq=ss.clone())
            call("clsr_adam_tick", self.adam_state, float(hp.learning_rate), 0.9, 0.999)
            call("clsr_dense_adam", self.dense, self.dense_grad, self.dense_m, self.dense_v, self.seg_of,
                 self.dense_sumsq, clip, self.adam_state, 0.9, 0.999, 1e-8, self.n_dense)
        if not tick_early:
            self._join()          # the Adam clock ticked on @aux: the table updates below read it
        rest = []
        for key, partner, slot, dscale, dloss_scale, dloss, base, nsum in spec:
            V, C = tb[key].shape
            if key in lists and self.lazy:
                ids, count, cap = lists[key]
                call("clsr_table_adam_rows" + self._th, tb[key], tg[key], self.tab_m[key], self.tab_v[key], fl[key], ids, count,
                     cap, C, ss[base:], 2, nsum, clip, self.adam_state, 0.9, 0.999, 1e-8)
            elif key in lists:   # huge table with the reference's dense Adam: the O(vocabulary) sweep is inherent
                call("clsr_table_adam" + self._th, tb[key], tg[key], self.tab_m[key], self.tab_v[key], fl[key], V, C, ss[base:],
                     2, nsum, clip, self.adam_state, 0.9, 0.999, 1e-8, self.lazy)
            else:
                rest.append(key)
        if rest:
            ops.multi("clsr_tables_adam_multi" + self._th, ops.TableDesc, [r for r in sweep if r[0] in
                                                                {tb[k].data_ptr() for k in rest}],
                      clip, self.adam_state, 0.9, 0.999, 1e-8, self.lazy)
        if tick_early:
            self._join()          # the dense path (@aux): the next step reads the updated variables on this stream

    # ------------------------------------------------------------------ measurement hooks (bench.py)
    def precision_note(self):
        if self.precision == "fp32":
            return ("all tensors fp32; EVERY product at fp32 accuracy: v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chains: recurrences, "
                    "layer-0 forward, heads, encoder-side weight gradients) or three bf16 pieces per operand on "
                    "v_mfma_f32_16x16x32_bf16 with fp32 accumulation (2^-23 relative: history-level attention forward / backward, "
                    "layer-1 forward, the attention-MLP backward with its folded weight gradients%s, the Time4LSTM time-gate "
                    "projection); no two-piece (2^-16) product anywhere" % ("" if self.att_bwd == "x6" else " [CLSR_ATT_BWD=fp32: fp32 MFMAs]"))
        if self.precision == "fp32x3":
            return ("all tensors fp32 (storage, statistics, losses, optimiser exactly as in the fp32 mode); forward products in front "
                    "of a batch-norm + ReLU at fp32 accuracy; the recurrences' hidden products, the attention-MLP backward with its "
                    "folded weight gradients, the history-level attention backward and the fused encoder tail as TWO-piece split-bf16 "
                    "sums hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (2^-16 relative per product "
                    "term: narrower than the reference's fp32 products)")
        return ("attention-block activations at (row, step) level (z0, z1, dz0) stored as bf16, their products with ONE bf16 piece per "
                "operand on v_mfma_f32_16x16x32_bf16 with fp32 accumulation"
                + (" (the parity mode's chain kernels with the weight gradients folded in, csrc/attbwdx3.hip)" if self.bf16_chain
                   else " (position-tiled csrc/hgemm.hip kernels, CLSR_BF16_CHAIN=old)")
                + "; history- and row-level tensors, batch-norm statistics, softmax, recurrences, losses, gradients of the "
                "parameters and the optimiser stay fp32")

    def bench_att_layer0(self, f, time_kernel):
        """HIP-event timing of the most expensive GEMM of the step in isolation: the short-term attention's first
        layer  z0 = U[h,t] + V[r] + (a[h,t] * q[r]) . Wp  over B*T positions (after at least one training step on
        ``f``: buffers and packed weights exist)."""
        B, T, G, Hn = self.last_shape
        Qs, A0 = self.Du + self.D, self.A0
        run = self._att_layer0_launcher("st", Hn, G, T, Qs)
        t_mm = time_kernel(run)
        flops = 2.0 * B * T * Qs * A0
        bf = self.precision == "bf16"
        peak = 2500.0 if bf else 157.3
        if bf and self._bf16_chain_ok(G, Qs):
            nbytes = float(B) * T * A0 * 2 + float(Hn) * T * (Qs + A0) * 4 + float(B) * (Qs + A0) * 4
            return dict(bound="hbm", kernel="att_l0_fwd_kernel<5,5,1,bf16> (short-term attention layer 0, one wave per history, one "
                                            "bf16 piece per operand, bf16 z0)",
                        achieved=round(nbytes / t_mm / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(nbytes / t_mm / 8e12, 4),
                        bytes_per_launch=nbytes, us_per_launch=round(t_mm * 1e6, 2),
                        formula="B*T*A0*2 (bf16 z0 written) + Hn*T*(Q + A0)*4 (a, U read) + B*(Q + A0)*4 (q, V read)",
                        mfma_tflops=round(flops / t_mm / 1e12, 2))
        return dict(bound="mfma", kernel=(("hgemm_l0g_kernel (short-term attention layer 0, bf16 MFMA, one wave per history group)"
                                           if self.l0_fwd_wave and query("clsr_hgemm_l0_group_supported", G, Qs, A0) else
                                           "hgemm_kernel<MUL,UV> (short-term attention layer 0, bf16 MFMA)") if bf else
                                          "att_l0_fwd_kernel<5,5> (short-term attention layer 0, one wave per history)"
                                          if self._att_layer0_wave(G, Qs) else
                                          "pgemm_fast_kernel<5,MUL,UV,false> (short-term attention layer 0)"),
                    achieved=round(flops / t_mm / 1e12, 2), peak=peak, unit="TFLOP/s",
                    frac=round(flops / t_mm / (peak * 1e12), 4), us_per_launch=round(t_mm * 1e6, 2),
                    note=("bf16-input MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate; K = N = 80: the kernel is "
                          "bound by its 164 MB bf16 output + operand reads, not by the matrix pipe" if bf else
                          "fp32-input MFMA (v_mfma_f32_16x16x4_f32); peak = dense fp32 matrix rate"))

    def bench_att_l1_bwd(self, f, time_kernel):
        """HIP-event timing of the HBM-heaviest kernel of the backward pass in isolation: pass 2 of the second attention
        layer's backward (short-term attention, B*T positions): reads z1 and z0, writes dz0, accumulates dW1 / db1."""
        B, T, G, Hn = self.last_shape
        A0, A1, M = self.A0, self.A1, B * T
        key, nn = "st", CL + "short_term/attention_fcn/att_fcn/nn_part/"
        bn0, bn1 = self.bn[nn + "batch_normalization/"], self.bn[nn + "batch_normalization_1/"]
        if self.bf16:
            if not self._bf16_chain_ok(G, self.D):
                return None
            BF = torch.bfloat16
            z0, z1 = self._buf(key + ".z0", M, A0, dtype=BF), self._buf(key + ".z1", M, A1, dtype=BF)
            dz0, ds = self._buf(key + ".dz0", M, A0, dtype=BF), self._buf(key + ".ds", M)
            Wt, Kp = self.packed[key + ".W1^T"]
            parts = query("clsr_att_l1_bwd_x3_parts", M)
            ws = self._buf(key + ".dw1x_ws", parts * query("clsr_dw_chunk_floats"))
            wo = self.P[nn + "w_nn_output"]
            t = time_kernel(lambda: call("clsr_att_l1_bwd_x1_h", z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0,
                                         bn0.scale, bn0.shift, None, None, bn0.coef, dz0, A0, ws, None, M, A1, A0))
            nbytes = float(M) * ((A1 + 2 * A0) * 2 + 4)
            return dict(bound="hbm", kernel="att_l1_bwd_x3_kernel<5,3,true,1,bf16> (short-term attention, layer-1 backward pass 2: dz0 + "
                                            "the partial sums of dW1 / db1 from one pass over the bf16 z1, z0; one bf16 piece per operand)",
                        achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(nbytes / t / 8e12, 4),
                        bytes_per_launch=nbytes, us_per_launch=round(t * 1e6, 2),
                        formula="M * (A1 + A0) * 2 read + M * A0 * 2 written + M * 4 (score gradient), M = B*T positions",
                        note="one 512-register wave per SIMD: bound by instruction issue, not by HBM")
        if self.att_bwd not in ("x3", "x6") or not query("clsr_att_l1_bwd_x3_supported", self.A1, self.A0):
            return None
        z0, z1 = self._buf(key + ".z0", M, A0), self._buf(key + ".z1", M, A1)
        dz0, ds = self._buf(key + ".dz0", M, A0), self._buf(key + ".ds", M)
        Wt, Kp = self.packed[key + ".W1^T"]
        parts = query("clsr_att_l1_bwd_x3_parts", M)
        ws = self._buf(key + ".dw1x_ws", parts * query("clsr_dw_chunk_floats"))
        wo = self.P[nn + "w_nn_output"]
        t = time_kernel(lambda: call(self._l1x, z1, A1, ds, bn1.scale, bn1.shift, wo, bn1.coef, Wt, Kp, z0, A0,
                                     bn0.scale, bn0.shift, None, None, bn0.coef, dz0, A0, ws, None, M, A1, A0))
        nbytes = float(M) * (A1 + 2 * A0 + 1) * 4
        return dict(bound="hbm", kernel="att_l1_bwd_x3_kernel<5,3,true,%d> (short-term attention, layer-1 backward pass 2: dz0 + the "
                                        "partial sums of dW1 / db1 from one pass over z1, z0; %s bf16 pieces per operand)"
                                        % ((3, "three") if self.att_bwd == "x6" else (2, "two")),
                    achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(nbytes / t / 8e12, 4),
                    bytes_per_launch=nbytes, us_per_launch=round(t * 1e6, 2),
                    formula="M * (A1 + A0) * 4 read + M * A0 * 4 written + M * 4 (score gradient), M = B*T positions")

    def _att_layer0_launcher(self, key, Hn, G, T, Q):
        R, A0 = Hn * G, self.A0
        a, q = self._buf(key + ".a", Hn * T, Q), self._buf(key + ".q", R, Q)
        U, V = self._buf(key + ".U", Hn * T, A0), self._buf(key + ".V", R, A0)
        if self.bf16:
            z0 = self._buf(key + ".z0", R * T, A0, dtype=torch.bfloat16)
            if self._bf16_chain_ok(G, Q):
                Wt, Kp = self.packed[key + ".Wp"]
                return lambda: call("clsr_att_l0_fwd_x1_h", a, Q, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, None, Hn, G, T, Q, A0)
            Wt, Kp = self.packed_h[key + ".Wp"]
            if self.l0_fwd_wave and query("clsr_hgemm_l0_group_supported", G, Q, A0):
                return lambda: call("clsr_hgemm_l0_group", a, Q, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, None, Hn, G, T, Q, A0)
            return lambda: call("clsr_hgemm_mul_uv", a, Q, T, G, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, None, R * T, Q, A0)
        z0 = self._buf(key + ".z0", R * T, A0)
        Wt, Kp = self.packed[key + ".Wp"]
        if self._att_layer0_wave(G, Q):
            return lambda: call(self.att_l0_fwd_entry, a, Q, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, None, Hn, G, T, Q, A0)
        return lambda: call("clsr_pgemm", a, Q, T, G, q, Q, None, None, 1, Wt, Kp, None, U, A0, V, A0, z0, A0, 0,
                            None, R * T, Q, A0)

    def _att_layer0_wave(self, G, Q):
        return (not self.bf16) and self.l0_fwd_wave and bool(query("clsr_att_l0_fwd_supported", G, Q, self.A0))

    def check_abort(self):
        """Raise ``StepAborted`` if a bounded wait of some step gave up -- a grid barrier of the fused heads launches (a
        launch wider than the device can hold at once, a peer rank that never pushed its statistics) or a small all-reduce
        of the data-parallel step (csrc/p2p.hip).  The device raised adam_state[4] when it happened; every optimiser kernel
        since has returned without touching a parameter or a moment and the Adam clock has stopped.  That is NOT a state
        to continue from: the batch-norm moving statistics of the aborted step were updated from invalid sums (the
        peer-to-peer all-reduce returns NaN on purpose) and the gradient accumulators / involved-row marks were not cleared.
        The condition is therefore STICKY: every later ``check_abort`` / ``state_dict`` / ``train_step`` raises again until
        the net is restored with ``load_state_dict`` of a full checkpoint (variables + moving statistics + Adam slots),
        which clears the flag and the accumulators.  Synchronises the device.  Callers: every loss read, ``state_dict``
        (so: every checkpoint), the end of an epoch in ``fit``; the data-parallel stepper looks at an asynchronous copy
        of the flag one step late, in ``train_step`` and in the ``run()`` of ``capture()``."""
        flag = float(self.adam_state[4].item())
        if flag == 0.0 and not self._aborted:
            return
        if not self._aborted:
            why = []
            ws = self._bufs.get(("heads.ws", (int(query("clsr_heads_fused_workspace_bytes")) + 3) // 4, F32))
            if ws is not None and query("clsr_heads_fused_error", ws.data_ptr()) != 0:
                why.append("a grid barrier of the fused heads launches timed out (the device could not hold every workgroup "
                           "of the launch at once, or a peer rank never pushed its statistics; CLSR_NO_HEADS_FUSED=1 runs "
                           "the launch chain instead)")
            comm = getattr(self, "dp_comm", None)
            if comm and query("clsr_comm_error", comm) != 0:
                why.append("a small all-reduce gave up waiting for a peer rank (%.1f s: CLSR_P2P_TIMEOUT_S)"
                           % (query("clsr_p2p_timeout_ms") * 1e-3))
            self._aborted = "; ".join(why) or "abort flag %g" % flag
        raise StepAborted("the training step was aborted on the device: %s; no update has been applied since, the moving "
                          "statistics and gradient accumulators are invalid -- restore the last checkpoint "
                          "(load_state_dict / load_model)" % self._aborted)

    def _reset_after_abort(self):
        """Part of ``load_state_dict`` of a full checkpoint: clear the device abort flag, the sticky error words and every
        accumulator an aborted step left behind (gradient tables, involved-row marks, dense gradients, statistics)."""
        ws = self._bufs.get(("heads.ws", (int(query("clsr_heads_fused_workspace_bytes")) + 3) // 4, F32))
        if ws is not None:
            torch.cuda.synchronize(self.device)
            query("clsr_heads_fused_clear_error", ws.data_ptr())      # (synchronous host-side entry point: no stream argument)
        self.adam_state[4] = 0.0
        for k in self.tables:
            self.tab_grad[k].zero_()
            self.tab_flags[k].zero_()
        self.dense_grad.zero_()
        self.stats24.zero_()
        self.ucount.zero_()
        self._counts_zeroed = self._ucount_zeroed = self._ticked = False
        self._aborted = None

    def read_losses(self):
        """Synchronising read of the step's loss terms -> dict of python floats."""
        v = self.losses.cpu().tolist()
        self.check_abort()
        return dict(data_loss=v[0], regular_loss=v[1], contrastive_loss=v[2], discrepancy_loss=v[3],
                    loss=v[0] + v[1] + v[2] + v[3])
