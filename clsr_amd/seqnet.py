"""GRU4Rec, DIN, DIEN, SLi-Rec and A2SVD on the kernels of the CLSR step.

The reference's quick-start trains these from the same script as CLSR (examples/00_quick_start/sequential.py:94-205);
they sit on the same ``SequentialBaseModel`` trunk (embeddings + involved-row regulariser + logit MLP + softmax loss +
clip + Adam) and differ in ``_build_seq_graph``:

  GRU4RecModel   models/sequential/gru4rec.py:21-76     dynamic_rnn(GRUCell) final state ++ target
  DINModel       models/sequential/din.py:13-34          target ++ masked history sum ++ _attention_fcn(target, history)
  A2SVDModel     models/sequential/asvd.py:13-45         A2SVD attention ++ target
  DIENModel      models/sequential/dien.py:13-64         GRU -> _attention_fcn WEIGHTS -> attentional GRU (one sequence
                                                         per candidate row) final state, ++ target, history sum, product
  SLI_RECModel   models/sequential/sli_rec.py:25-147     A2SVD attention (unmasked), Time4LSTM over the item embedding,
                                                         _attention_fcn(target, rnn_outputs), alpha fusion

``SeqNet`` re-uses CLSRNet wholesale -- parameter store, packed weights, the position-tiled GEMMs, the fused recurrence
launch, the attention block (forward and backward), the MLP heads, sorted segmented embedding gradients, regularisers,
clip, Adam, the stream structure, data parallelism -- and only re-states the three graphs and their backward passes.
New device code: the A2SVD attention and the length scaling of DIN's history sum (csrc/sibling.hip).
"""
import numpy as np
import torch

from clsr_amd import ops
from clsr_amd.net import CLSRNet, _pad4
from clsr_amd.ops import call, query
from clsr_amd.params import SIB_TABLES, UNUSED_TABLE, sibling_kind, sibling_scopes, sibling_specs

LG = "sequential/logit_fcn/nn_part/"


class SeqNet(CLSRNet):
    def __init__(self, hp, dims, kind=None, **kw):
        self.kind = sibling_kind(kind if kind is not None else hp.model_type)
        if self.kind is None:
            raise ValueError("SeqNet builds gru4rec / din / dien / sli_rec / a2svd, got %r" % (kind or hp.model_type,))
        self.sc = sibling_scopes(self.kind)
        super(SeqNet, self).__init__(hp, dims, **kw)
        if self.kind == "sli_rec":
            self.enc_in = self.Di          # Time4LSTM reads the item embedding only (sli_rec.py:43-57)
        self.split_query = False

    # ------------------------------------------------------------------ configuration
    def _check_supported(self):
        hp, bad = self.hp, []
        if hp.enable_BN is not True:
            bad.append("enable_BN must be True")
        if list(hp.activation) != ["relu"] * len(hp.activation):
            bad.append("activation must be relu")
        if any(float(d) != 0.0 for d in hp.dropout) or float(hp.embedding_dropout) != 0.0 or hp.user_dropout:
            bad.append("dropout must be 0")
        if any(float(getattr(hp, k)) != 0.0 for k in ("embed_l1", "layer_l1", "cross_l1", "cross_l2")):
            bad.append("l1 / cross regularisers must be 0")
        if hp.loss != "softmax" or hp.method != "classification":
            bad.append("loss must be softmax / method classification")
        if hp.optimizer not in ("adam", "lazyadam"):
            bad.append("optimizer must be adam or lazyadam")
        if len(hp.layer_sizes) != 2:
            bad.append("layer_sizes must have two layers")
        D = hp.item_embedding_dim + hp.cate_embedding_dim
        if self.kind in ("din", "sli_rec", "dien") and len(hp.att_fcn_layer_sizes or ()) != 2:
            bad.append("att_fcn_layer_sizes must have two layers")
        if self.kind == "a2svd" and hp.attention_size != D:
            bad.append("attention_size must equal item+cate dims (tensordot with query, base_model.py:622)")
        if self.kind == "sli_rec":
            if hp.hidden_size != D:
                bad.append("hidden_size must equal item+cate dims (alpha fusion of att_fea1 and att_fea2, sli_rec.py:92)")
            if hp.attention_size != D:
                bad.append("attention_size must equal item+cate dims (tensordot with query, base_model.py:622)")
        bad += self._shape_limits(hp, rnn=self.kind in ("gru4rec", "sli_rec", "dien"))
        if self.kind in ("a2svd", "sli_rec") and int(hp.attention_size) % 4:
            bad.append("attention_size must be a multiple of 4")
        if bad:
            raise NotImplementedError("%s HIP path does not support: %s" % (self.kind, "; ".join(bad)))

    def _param_specs(self):
        return sibling_specs(self.dims, self.hp, self.kind)

    def _table_map(self):
        return SIB_TABLES

    def _unused_tables(self):
        return (UNUSED_TABLE,)

    def _update_spec(self):
        return (("item", None, 4, 0.0, 0.0, None, 0, 3), ("cate", None, 5, 0.0, 0.0, None, 1, 3))

    def _att_qh(self, key):
        return 0

    # ------------------------------------------------------------------ encoders
    def _gru_list(self):
        if self.kind == "gru4rec":
            return [("gs", self.sc["gru"], self.H)]
        if self.kind == "dien":
            return [("gs", self.sc["gru1"], self.H)]
        return []

    @property
    def _t4_kind(self):
        return "time4lstm" if self.kind == "sli_rec" else None

    @property
    def _t4_scope(self):
        return self.sc["t4"] if self.kind == "sli_rec" else None

    @property
    def a_in(self):
        return 3 * self.D + 1

    @property
    def out_dim(self):
        """Width of ``model_output`` (the logit MLP's input)."""
        return {"gru4rec": self.H + self.D, "din": 3 * self.D, "sli_rec": 2 * self.D, "a2svd": 2 * self.D,
                "dien": 3 * self.D + self.H}[self.kind]

    def _plan_weights(self, training):
        hp, P, D, H = self.hp, self.P, self.D, self.H
        pair = self._pack_pair if training else (lambda k, W, K, N, K_pad=None: self._pack(k, W, N, K, in_pad=K_pad))
        if self.kind == "din":
            self._plan_att("st", self.sc["att"], D, D, training)
        elif self.kind == "sli_rec":
            self._plan_att("st", self.sc["att"], H, D, training)
            pair("asvd.A", P[self.sc["asvd"] + "attention_mat"], D, D)
            if not hp.manual_alpha:
                a = self.sc["alpha"] + "nn_part/"
                pair("al.W0", P[a + "w_nn_layer0"], self.a_in, self.A0, K_pad=_pad4(self.a_in))
                pair("al.W1", P[a + "w_nn_layer1"], self.A0, self.A1)
        elif self.kind == "a2svd":
            pair("asvd.A", P[self.sc["asvd"] + "attention_mat"], D, D)
        elif self.kind == "dien":
            self._plan_att("st", self.sc["att"], H, D, training)
            # input side of the attentional GRU: [gates (2H) | candidate (H)] over the first GRU's outputs
            g2 = self.sc["gru2"]
            bias = self._buf("a2.bias", 3 * H)
            for W_, b_, off, w in ((P[g2 + "gates/kernel"][0:H], P[g2 + "gates/bias"], 0, 2 * H),
                                   (P[g2 + "candidate/kernel"][0:H], P[g2 + "candidate/bias"], 2 * H, H)):
                self._pack("a2.xw", W_, w, H, o0=off, total=(3 * H, H))
                self._cur_descs.append(ops.pack_desc(b_, w, 1, bias, 1, ld1=1, o0=off))
                if training:
                    self._pack("a2.xw^T", W_, H, w, transposed=True, i0=off, total=(H, 3 * H))
        if self.kind in ("gru4rec", "sli_rec", "dien"):
            self._plan_encoders(training)
        pair("lg.W0", P[LG + "w_nn_layer0"], self.out_dim, self.L0)
        pair("lg.W1", P[LG + "w_nn_layer1"], self.L0, self.L1)

    # ------------------------------------------------------------------ forward
    def _forward(self, f, training, after_attention=None, early_aux=None):
        hp, P, kind, sc = self.hp, self.P, self.kind, self.sc
        B, T = f["B"], f["T"]
        G = self.G_train if (training and self.dedup) else ((f.get("G") or 1) if not training else 1)
        if B % G:
            raise ValueError("training feed rows (%d) must be a multiple of 1+train_num_ngs (%d)" % (B, G))
        Hn = B // G
        D, H, Di, Dc, E = self.D, self.H, self.Di, self.Dc, self.enc_in
        self.last_shape = (B, T, G, Hn)
        self._pack_all(training)
        hs = 1 if f.get("compact") else G
        seq_len, ls = f["seq_len"], hs
        step_start = self._fork_point()
        # ---- gathers (padded steps hold embedding row 0, as in the reference)
        hist = self._buf("hist", Hn, T, D)
        hmean, hrec = self._buf("hist_mean", Hn, D), self._buf("hist_recent", Hn, D)
        call("clsr_gather_hist_fwd", self.tables["item"], self.tables["cate"], f["item_history"],
             f["item_cate_history"], hs * T, seq_len, ls, Hn, T, Di, Dc, 1, hist, hmean, hrec)
        target = self._buf("target", B, D)
        tb = self.tables
        ops.multi("clsr_gather_rows_multi", ops.GatherDesc, [
            (tb["item"].data_ptr(), f["items"].data_ptr(), target.data_ptr(), 1, B, Di, D, 0),
            (tb["cate"].data_ptr(), f["cates"].data_ptr(), target.data_ptr(), 1, B, Dc, D, Di)])
        M = Hn * T
        W = self.out_dim
        mo = self._buf("model_output", B, W)
        out = dict(hist_input=hist, target=target, model_output=mo)

        def aux():
            if early_aux is not None or (training and self.sorted_hist_grad):
                with self._branch("@aux", after=step_start):
                    if early_aux is not None:
                        early_aux()
                    if training and self.sorted_hist_grad:
                        self._sort_hist_ids(f, Hn, T, hs)

        if kind == "gru4rec":
            NX = self.NX
            PinAll = self._buf("xw.Pin", M, NX)
            self._gemm(hist, D, "xw", M, E, NX, PinAll, NX, bias=self._buf("xw.bias", NX))
            aux()
            d, hT, _ = self._gru_fwd_desc("gs", sc["gru"], H, PinAll, Hn, T, None, training)
            ops.rnn_multi("clsr_rnn_fwd_multi", [d], None, seq_len, ls, Hn, T)
            call("clsr_copy_cols", hT, H, 0, G, B, H, mo, W, 0, 0)
            call("clsr_copy_cols", target, D, 0, 1, B, D, mo, W, H, 0)
            out["final_state"] = hT
        elif kind == "dien":
            NX = self.NX
            PinAll = self._buf("xw.Pin", M, NX)
            self._gemm(hist, D, "xw", M, E, NX, PinAll, NX, bias=self._buf("xw.bias", NX))
            aux()
            d1, _, rnn1 = self._gru_fwd_desc("gs", sc["gru1"], H, PinAll, Hn, T, None, training, want_seq=True)
            ops.rnn_multi("clsr_rnn_fwd_multi", [d1], None, seq_len, ls, Hn, T)
            hsum = self._buf("hist_sum", Hn, D)
            call("clsr_scale_rows_by_len", hmean, seq_len, ls, Hn, D, hsum, 0)
            # attention over the first GRU's outputs: only its WEIGHTS are used (return_alpha=True)
            self._att_fwd("st", sc["att"], rnn1, target, Hn, G, T, H, D, seq_len, ls, training)
            wts = self._buf("st.wts", B, T)
            # attentional GRU: one sequence per candidate row, inputs shared by the rows of a history group
            g2 = sc["gru2"]
            Pin2 = self._buf("a2.Pin", M, 3 * H)
            self._gemm(rnn1, H, "a2.xw", M, H, 3 * H, Pin2, 3 * H, bias=self._buf("a2.bias", 3 * H))
            hT2 = self._buf("a2.hT", B, H)
            d2 = ops.gru_desc(H, Pin=Pin2, ldp=3 * H, Wgh=P[g2 + "gates/kernel"][H:], ldg=2 * H,
                              Wch=P[g2 + "candidate/kernel"][H:], ldc=H, hT=hT2,
                              hprev=self._buf("a2.hprev", B, T, H) if training else None,
                              gates=self._buf("a2.gates", B, T, 3 * H) if training else None, att=wts, in_div=G)
            ops.rnn_multi("clsr_rnn_fwd_multi", [d2], None, seq_len, ls, B, T)
            call("clsr_copy_cols", target, D, 0, 1, B, D, mo, W, 0, 0)
            call("clsr_copy_cols", hT2, H, 0, 1, B, H, mo, W, D, 0)
            call("clsr_copy_cols", hsum, D, 0, G, B, D, mo, W, D + H, 0)
            call("clsr_mul_rows", target, D, hsum, D, G, B, D, mo[:, 2 * D + H:], W)
            out.update(hist_sum=hsum, rnn_out=rnn1, w_att=wts, final_state=hT2)
        elif kind == "a2svd":
            ai = self._buf("asvd.ai", M, D)
            self._gemm(hist, D, "asvd.A", M, D, D, ai, D)
            aux()
            w1, att1 = self._buf("asvd.wts", Hn, T), self._buf("asvd.out", Hn, D)
            call("clsr_asvd_att_fwd", ai, P[sc["asvd"] + "query"], hist, Hn, T, D, w1, att1)
            call("clsr_copy_cols", att1, D, 0, G, B, D, mo, W, 0, 0)
            call("clsr_copy_cols", target, D, 0, 1, B, D, mo, W, D, 0)
            out.update(asvd_output=att1, w_asvd=w1)
        elif kind == "din":
            hsum = self._buf("hist_sum", Hn, D)
            call("clsr_scale_rows_by_len", hmean, seq_len, ls, Hn, D, hsum, 0)
            aux()
            att = self._att_fwd("st", sc["att"], hist, target, Hn, G, T, D, D, seq_len, ls, training)
            call("clsr_copy_cols", target, D, 0, 1, B, D, mo, W, 0, 0)
            call("clsr_copy_cols", hsum, D, 0, G, B, D, mo, W, D, 0)
            call("clsr_copy_cols", att, D, 0, 1, B, D, mo, W, 2 * D, 0)
            out.update(hist_sum=hsum, att_fea=att, w_att=self._buf("st.wts", B, T))
        else:
            NX = self.NX
            t = sc["t4"]
            # long-term: A2SVD attention over the whole (padded) history
            ai = self._buf("asvd.ai", M, D)
            self._gemm(hist, D, "asvd.A", M, D, D, ai, D)
            w1, att1 = self._buf("asvd.wts", Hn, T), self._buf("asvd.out", Hn, D)
            call("clsr_asvd_att_fwd", ai, P[sc["asvd"] + "query"], hist, Hn, T, D, w1, att1)
            # short-term: Time4LSTM over [item embedding | t_first | t_now], then attention_fcn(target, outputs)
            TT = self._buf("t4.TT", M, 2 * H)
            call("clsr_t4_time_inputs_fwd", f["time_to_now"], f["time_from_first_action"], hs * T,
                 P[t + "_time_input_w1"], P[t + "_time_input_bias1"], P[t + "_time_input_w2"],
                 P[t + "_time_input_bias2"], Hn, T, H, TT)
            PinAll = self._buf("xw.Pin", M, NX)
            self._gemm(hist, D, "xw", M, E, NX, PinAll, NX, bias=self._buf("xw.bias", NX))
            aux()
            t4off = self._enc_off("t4")
            self._gemm(TT, 2 * H, "t4.tw", M, 2 * H, 3 * H, PinAll[:, t4off + 3 * H:], NX, acc=1)
            rnn_out = self._buf("rnn_out", Hn, T, H)
            t4d = ops.t4_desc(H, Pin=PinAll[:, t4off:], ldp=NX, Wm=P[t + "kernel"][E:], ldm=4 * H, out_seq=rnn_out,
                              act=self._buf("t4.act", Hn, T, 6 * H) if training else None,
                              cst=self._buf("t4.cst", Hn, T, H) if training else None,
                              mprev=self._buf("t4.mprev", Hn, T, H) if training else None)
            ops.rnn_multi("clsr_rnn_fwd_multi", [], t4d, seq_len, ls, Hn, T)
            att2 = self._att_fwd("st", sc["att"], rnn_out, target, Hn, G, T, H, D, seq_len, ls, training)
            alpha = self._buf("alpha", B)
            if not hp.manual_alpha:
                ld = _pad4(self.a_in)
                ain = self._buf("al.in", B, ld)
                call("clsr_alpha_concat", None, 0, target, att1, att2, f["time_to_now"], T, T - 1,
                     G if f.get("compact") else 1, B, G, D, ain, ld)
                al_logit = self._mlp_fwd("al", sc["alpha"] + "nn_part/", ain, ld, ld, (self.A0, self.A1), B, training)
                call("clsr_alpha_fuse_fwd", al_logit, 0.0, att1, att2, target, B, G, D, alpha, mo)
            else:
                call("clsr_alpha_fuse_fwd", None, float(hp.manual_alpha_value), att1, att2, target, B, G, D, alpha, mo)
            out.update(att_fea1=att1, att_fea2=att2, rnn_out=rnn_out, alpha=alpha, w_asvd=w1,
                       w_att=self._buf("st.wts", B, T))
        self._join()
        out["logit"] = self._mlp_fwd("lg", LG, mo, W, W, (self.L0, self.L1), B, training)
        return out

    # ------------------------------------------------------------------ training step
    def _train_step(self, f, apply):
        hp, P, Gd, kind, sc = self.hp, self.P, self.Gd, self.kind, self.sc
        B, T = f["B"], f["T"]
        G = self.G_train if self.dedup else 1
        Hn = B // G
        D, H, Di, Dc, E = self.D, self.H, self.Di, self.Dc, self.enc_in
        hs = 1 if f.get("compact") else G
        seq_len, ls = f["seq_len"], hs
        M, W = Hn * T, self.out_dim
        zpool = self._buf("zero_pool", M * (D + H) + B * D * 3 + Hn * (3 * D + H) + B * T)
        fl = self.tab_flags

        def zero_and_mark():
            call("clsr_zero_doubles", self.losses, 8)
            call("clsr_zero_doubles", self.sumsq_tab, 16)
            call("clsr_zero_floats", zpool, zpool.numel())
            ops.multi("clsr_mark_rows_multi", ops.MarkDesc, [
                (f["item_history"].data_ptr(), fl["item"].data_ptr(), Hn, hs * T, T, 0),
                (f["items"].data_ptr(), fl["item"].data_ptr(), B, 1, 1, 0),
                (f["item_cate_history"].data_ptr(), fl["cate"].data_ptr(), Hn, hs * T, T, 0),
                (f["cates"].data_ptr(), fl["cate"].data_ptr(), B, 1, 1, 0)])
        o = [0]

        def take(*shape):
            n = int(np.prod(shape))
            t_ = zpool[o[0]:o[0] + n].view(*shape)
            o[0] += n
            return t_
        dhist, drnn = take(Hn, T, D), take(Hn, T, H)
        dtarget, dS, datt = take(B, D), take(B, D), take(B, D)
        dL, dM, dR, dhT = take(Hn, D), take(Hn, D), take(Hn, D), take(Hn, H)
        dwts = take(B, T)      # DIEN: gradient w.r.t. the attention WEIGHTS (datt is d att_fea, [B, D])
        out = self._forward(f, True, None, zero_and_mark)
        dlogit = self._buf("dlogit", B)
        Gl = hp.train_num_ngs + 1
        call("clsr_softmax_loss", out["logit"], f["labels"], B // Gl, Gl, 1.0 / ((B // Gl) * self.dp_world),
             self.losses[0:], dlogit)
        dmo = self._mlp_bwd("lg", LG, dlogit, out["model_output"], W, W, W, (self.L0, self.L1), B)
        hist = out["hist_input"]
        if kind == "gru4rec":
            NX = self.NX
            call("clsr_group_sum_cols", dmo, W, 0, G, Hn, H, dhT, H, 0, 1)
            call("clsr_copy_cols", dmo, W, H, 1, B, D, dtarget, D, 0, 1)
            dPinAll = self._buf("xw.dPin", M, NX)
            ops.rnn_multi("clsr_rnn_bwd_multi", [self._gru_bwd_desc("gs", sc["gru"], H, dPinAll, Hn, T, dhT, None,
                                                                     None)], None, seq_len, ls, Hn, T)
            self._dw(hist, D, dPinAll, NX, M, E, NX, self._buf("xw.dW", E, NX), NX, db=self._buf("xw.db", NX))
            self._gemm(dPinAll, NX, "xw^T", M, NX, E, dhist, D, acc=1)
            self._gru_bwd_hidden("gs", sc["gru"], H, dPinAll, Hn, T)
            self._dw_flush()
            self._unpack_grads()
        elif kind == "dien":
            NX = self.NX
            g2 = sc["gru2"]
            rnn1, wts, target, hsum = out["rnn_out"], out["w_att"], out["target"], out["hist_sum"]
            call("clsr_copy_cols", dmo, W, 0, 1, B, D, dtarget, D, 0, 1)
            dhT2 = self._buf("a2.dhT", B, H)
            call("clsr_copy_cols", dmo, W, D, 1, B, H, dhT2, H, 0, 0)
            dsum = self._buf("d_hist_sum", Hn, D)
            call("clsr_group_sum_cols", dmo, W, D + H, G, Hn, D, dsum, D, 0, 0)
            # product feature target * hist_sum: d target += dp * hist_sum[h], d hist_sum[h] += sum_g dp * target
            dsum2 = self._buf("d_hist_sum2", Hn, D)
            call("clsr_att_prod_bwd_ld", dmo[:, 2 * D + H:], W, hsum, D, target, D, Hn, G, 1, D, dsum2, D, dtarget, D, 1)
            call("clsr_axpby", dsum, dsum, 1.0, dsum2, 1.0, Hn * D)
            call("clsr_scale_rows_by_len", dsum, seq_len, ls, Hn, D, dM, 0)
            # attentional GRU backward (row level): d att scores, d input projections per row
            dPin2 = self._buf("a2.dPin", B * T, 3 * H)
            hprev2, gates2 = self._buf("a2.hprev", B, T, H), self._buf("a2.gates", B, T, 3 * H)
            d2 = ops.gru_desc(H, Wgh=P[g2 + "gates/kernel"][H:], ldg=2 * H, Wch=P[g2 + "candidate/kernel"][H:], ldc=H,
                              hprev=hprev2, gates=gates2, dhT=dhT2, dPin=dPin2, lddp=3 * H, att=wts, datt=dwts,
                              in_div=G)
            ops.rnn_multi("clsr_rnn_bwd_multi", [d2], None, seq_len, ls, B, T)
            self._dw(hprev2, H, dPin2, 3 * H, B * T, H, 2 * H, Gd[g2 + "gates/kernel"][H:], 2 * H)
            self._dw(hprev2, H, dPin2[:, 2 * H:], 3 * H, B * T, H, H, Gd[g2 + "candidate/kernel"][H:], H, Xmul=gates2,
                     ldmul=3 * H)
            if G > 1:       # the rows of a group share the inputs: sum their d(input projections)
                dPin2h = self._buf("a2.dPinH", M, 3 * H)
                call("clsr_att_z0_bwd_reduce", dPin2, Hn, G, T, 3 * H, dPin2h, self._buf("a2.dV", B, 3 * H))
            else:
                dPin2h = dPin2
            self._dw(rnn1, H, dPin2h, 3 * H, M, H, 2 * H, Gd[g2 + "gates/kernel"][0:H], 2 * H,
                     db=Gd[g2 + "gates/bias"])
            self._dw(rnn1, H, dPin2h[:, 2 * H:], 3 * H, M, H, H, Gd[g2 + "candidate/kernel"][0:H], H,
                     db=Gd[g2 + "candidate/bias"])
            self._gemm(dPin2h, 3 * H, "a2.xw^T", M, 3 * H, H, drnn, H, acc=1)
            # attention backward through its weights, then the first GRU
            dq = self._att_bwd("st", sc["att"], None, rnn1, target, drnn, Hn, G, T, H, D, seq_len, ls, dw_in=dwts)
            call("clsr_copy_cols", dq, D, 0, 1, B, D, dtarget, D, 0, 1)
            dPinAll = self._buf("xw.dPin", M, NX)
            ops.rnn_multi("clsr_rnn_bwd_multi", [self._gru_bwd_desc("gs", sc["gru1"], H, dPinAll, Hn, T, None, drnn,
                                                                     None)], None, seq_len, ls, Hn, T)
            self._dw(hist, D, dPinAll, NX, M, E, NX, self._buf("xw.dW", E, NX), NX, db=self._buf("xw.db", NX))
            self._gemm(dPinAll, NX, "xw^T", M, NX, E, dhist, D, acc=1)
            self._gru_bwd_hidden("gs", sc["gru1"], H, dPinAll, Hn, T)
            self._dw_flush()
            self._unpack_grads()
        elif kind == "a2svd":
            call("clsr_group_sum_cols", dmo, W, 0, G, Hn, D, dL, D, 0, 1)
            call("clsr_copy_cols", dmo, W, D, 1, B, D, dtarget, D, 0, 1)
            self._asvd_bwd(dL, hist, dhist, Hn, T)
            self._dw_flush()
        elif kind == "din":
            call("clsr_copy_cols", dmo, W, 0, 1, B, D, dtarget, D, 0, 1)
            dsum = self._buf("d_hist_sum", Hn, D)
            call("clsr_group_sum_cols", dmo, W, D, G, Hn, D, dsum, D, 0, 0)
            call("clsr_copy_cols", dmo, W, 2 * D, 1, B, D, datt, D, 0, 0)
            dq = self._att_bwd("st", sc["att"], datt, hist, out["target"], dhist, Hn, G, T, D, D, seq_len, ls)
            call("clsr_copy_cols", dq, D, 0, 1, B, D, dtarget, D, 0, 1)
            # d(masked sum) reaches every valid step: hand it to the embedding backward as len * d(mean)
            call("clsr_scale_rows_by_len", dsum, seq_len, ls, Hn, D, dM, 0)
            self._dw_flush()
        else:
            NX = self.NX
            t = sc["t4"]
            att1, att2 = out["att_fea1"], out["att_fea2"]
            if not hp.manual_alpha:
                dal = self._buf("dalpha_logit", B)
                call("clsr_alpha_fuse_bwd", dmo, out["alpha"], 0.0, att1, att2, Hn, G, D, dal, dL, dS, dtarget)
                ld = _pad4(self.a_in)
                dain = self._mlp_bwd("al", sc["alpha"] + "nn_part/", dal, self._buf("al.in", B, ld), ld, ld, self.a_in,
                                     (self.A0, self.A1), B)
                call("clsr_alpha_concat_bwd", dain, ld, 0, Hn, G, D, None, dtarget, dL, dS)
            else:
                call("clsr_alpha_fuse_bwd", dmo, None, float(hp.manual_alpha_value), att1, att2, Hn, G, D, None, dL,
                     dS, dtarget)
            dq = self._att_bwd("st", sc["att"], dS, out["rnn_out"], out["target"], drnn, Hn, G, T, H, D, seq_len, ls)
            call("clsr_copy_cols", dq, D, 0, 1, B, D, dtarget, D, 0, 1)
            dPinAll = self._buf("xw.dPin", M, NX)
            t4off = self._enc_off("t4")
            t4d = ops.t4_desc(H, Wm=P[t + "kernel"][E:], ldm=4 * H, act=self._buf("t4.act", Hn, T, 6 * H),
                              cst=self._buf("t4.cst", Hn, T, H), dout_seq=drnn, dPin=dPinAll[:, t4off:], lddp=NX)
            ops.rnn_multi("clsr_rnn_bwd_multi", [], t4d, seq_len, ls, Hn, T)
            self._dw(hist, D, dPinAll, NX, M, E, NX, self._buf("xw.dW", E, NX), NX, db=self._buf("xw.db", NX))
            self._gemm(dPinAll, NX, "xw^T", M, NX, E, dhist, D, acc=1)
            self._t4_bwd_weights(f, dPinAll, Hn, T, hs)
            self._asvd_bwd(dL, hist, dhist, Hn, T)
            self._dw_flush()
            self._unpack_grads()
        self._join()
        # ---- embedding gradients
        ss = self.sumsq_tab
        if self.sorted_hist_grad:
            self._hist_grad_sorted(dhist, dM, dR, Hn, T, seq_len, ls, ss)
        else:
            call("clsr_gather_hist_bwd", dhist, dM, dR, f["item_history"], f["item_cate_history"], hs * T, seq_len,
                 ls, Hn, T, Di, Dc, 1, self.tab_grad["item"], self.tab_grad["cate"], ss[0:])
        call("clsr_scatter_add_rows", dtarget, D, 0, f["items"], 1, B, Di, self.tab_grad["item"], ss[2:])
        call("clsr_scatter_add_rows", dtarget, D, Di, f["cates"], 1, B, Dc, self.tab_grad["cate"], ss[3:])
        if apply:
            self._apply_updates()
        return out

    def _asvd_bwd(self, dout, hist, dhist, Hn, T):
        """Backward of the A2SVD attention: d query, d attention_mat, and its two contributions to d(hist)."""
        P, Gd, D, M, sc = self.P, self.Gd, self.D, Hn * T, self.sc
        ai, w1 = self._buf("asvd.ai", M, D), self._buf("asvd.wts", Hn, T)
        dai = self._buf("asvd.dai", M, D)
        parts = query("clsr_asvd_att_bwd_parts", Hn)
        qp = self._buf("asvd.qpart", 1024 * 256)[: parts * D]
        qname = sc["asvd"] + "query"
        call("clsr_asvd_att_bwd", dout, w1, ai, P[qname], hist, Hn, T, D, dai, dhist, qp)
        self._rp(qp, parts, D, D, Gd[qname])
        self._dw(hist, D, dai, D, M, D, D, Gd[sc["asvd"] + "attention_mat"], D)
        self._gemm(dai, D, "asvd.A^T", M, D, D, dhist, D, acc=1)

    def read_losses(self):
        v = self.losses.cpu().tolist()
        return dict(data_loss=v[0], regular_loss=v[1], loss=v[0] + v[1])
