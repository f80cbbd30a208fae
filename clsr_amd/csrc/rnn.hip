// Recurrent encoders of CLSR as persistent per-wave kernels (gfx950): GRU and Time4LSTM,
// forward and backward-through-time.
//
// Reference: tf.nn.rnn_cell.GRUCell under tensorflow.nn.dynamic_rnn at
//   models/sequential/clsr.py:160-168 (short_term_intention, h0 = user_short_embedding),
//   clsr.py:229-237 (causal2, zero h0), clsr.py:201-208 (sequential_model == 'gru'),
// and Time4LSTMCell.call (models/sequential/rnn_cell_implement.py:129-298) at clsr.py:179-200.
// dynamic_rnn semantics: steps t >= sequence_length copy the state through and emit zeros.
//
// Split of work: every input-side product (x.W_x, time-gate terms, biases) is hoisted out of the
// time loop into one batched pgemm that produces Pin[h, t, :]; these kernels only run the
// recurrence  state -> gates  with the hidden-to-hidden weights held in VGPRs as MFMA A
// operands for the whole sequence.  One WORKGROUP of RNT waves owns 16 histories; MFMA tile =
// D[16 hidden features][16 histories]; wave w owns feature tile w of every gate (its slice of the
// weights, of the state and of every elementwise update).  The D layout of a step (lane (j,g):
// features 16*tile+4g+{0..3} of history j) is exactly the B-operand layout of the next matvec, so
// the only cross-wave traffic is one 16-byte LDS write + barrier + RNT 16-byte LDS reads per lane
// whenever a full state vector is needed (GRU: r.h and h', Time4LSTM: m).  Compared with one wave
// per 16 histories this divides the T-serial MFMA chain and the register footprint by RNT.
// Hidden size n <= 48 runs with RNT = 3 feature tiles (3 waves), n <= 128 with 8 tiles (8 waves); n % 4 == 0.
#include "common.h"
#include "clsr_hip.h"
#include "rnn_args.h"
#include "hmma.h"


__device__ __forceinline__ f32x4 sig4(f32x4 v) {
  return (f32x4){sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
}
__device__ __forceinline__ f32x4 tanh4(f32x4 v) {
  return (f32x4){tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)};
}
__device__ __forceinline__ f32x4 sel4(bool c, f32x4 a, f32x4 b) { return c ? a : b; }
#define Z4 ((f32x4){0.f, 0.f, 0.f, 0.f})

// Compact last k-tile.  The recurrent products run over k-tiles of 16 state features; lane (i, g) of a tile holds
// the four features 4g..4g+3 of sequence i (the MFMA output layout, so a wave's result is the next step's operand
// without a shuffle) and MFMA number c of the tile reduces over the features {4g + c}.  With n = 40 the last tile
// holds 8 features (32..39) in the lanes g = 0, 1 and every one of its four MFMAs is half empty.  When the last tile
// has <= 8 features it is therefore read back from the exchange buffer feature-major instead -- component x =
// feature g, y = feature 4 + g of the tile, one scalar LDS read each -- and reduced by TWO full MFMAs: 10 instead of
// 12 MFMAs per gate and step at n = 40 on the T-serial chain.
template <int RNT>
__device__ __forceinline__ bool last_tile_compact(int n) { return n - 16 * (RNT - 1) <= 8; }

// forward use: out[o] = sum_in W[in*ld + colbase + o] * state[in]
__device__ __forceinline__ f32x4 load_w_fwd(const float* W, int ld, int colbase, int n, int ot, int kt,
                                            int i, int g, bool compact = false) {
  f32x4 v = Z4;
  const int o = 16 * ot + i;
  if (o < n) {
    const int in0 = compact ? 16 * kt + g : 16 * kt + 4 * g;
    const int st = compact ? 4 : 1;
    if (in0 + 0 * st < n) v.x = W[(long)(in0 + 0 * st) * ld + colbase + o];
    if (in0 + 1 * st < n) v.y = W[(long)(in0 + 1 * st) * ld + colbase + o];
    if (!compact) {
      if (in0 + 2 < n) v.z = W[(long)(in0 + 2) * ld + colbase + o];
      if (in0 + 3 < n) v.w = W[(long)(in0 + 3) * ld + colbase + o];
    }
  }
  return v;
}
// backward use: dstate[in] = sum_o W[in*ld + colbase + o] * dgate[o]
__device__ __forceinline__ f32x4 load_w_bwd(const float* W, int ld, int colbase, int n, int ot, int kt,
                                            int i, int g, bool compact = false) {
  const int in = 16 * ot + i;
  if (compact) {
    f32x4 v = Z4;
    const int o0 = 16 * kt + g;
    if (in < n && o0 < n) v.x = W[(long)in * ld + colbase + o0];
    if (in < n && o0 + 4 < n) v.y = W[(long)in * ld + colbase + o0 + 4];
    return v;
  }
  const int o0 = 16 * kt + 4 * g;
  if (in < n && o0 < n) return ld4(W + (long)in * ld + colbase + o0);
  return Z4;
}

// one k-tile of an exchange buffer as MFMA operand (tile = buf + kt * 64)
__device__ __forceinline__ f32x4 read_tile(const f32x4* tile, int lane, bool compact) {
  if (!compact) return tile[lane];
  const float* f = (const float*)tile;
  const int j = lane & 15, g = lane >> 4;
  return (f32x4){f[4 * j + g], f[4 * (16 + j) + g], 0.f, 0.f};
}

// acc (one feature tile) += W-tile . state over one k-tile (two MFMAs for the compact last tile)
__device__ __forceinline__ void mvt(f32x4& acc, const f32x4& w, const f32x4& b, bool two) {
  MFMA4(acc, w.x, b.x);
  MFMA4(acc, w.y, b.y);
  if (!two) {
    MFMA4(acc, w.z, b.z);
    MFMA4(acc, w.w, b.w);
  }
}

// acc (one feature tile) += sum over all RNT k-tiles of W-tile . state
template <int RNT>
__device__ __forceinline__ void mv1(f32x4& acc, const f32x4 (&w)[RNT], const f32x4 (&b)[RNT], bool cmp) {
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) mvt(acc, w[kt], b[kt], kt == RNT - 1 && cmp);
}

// collect the full vector (RNT k-tiles) of the workgroup from an exchange buffer
template <int RNT>
__device__ __forceinline__ void collect(const f32x4* buf, int lane, bool cmp, f32x4 (&full)[RNT]) {
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) full[kt] = read_tile(buf + kt * 64, lane, kt == RNT - 1 && cmp);
}

// publish this wave's tile of a vector and collect the full vector (RNT tiles) of the workgroup
template <int RNT>
__device__ __forceinline__ void xchg(f32x4* buf, int w, int lane, f32x4 own, bool cmp, f32x4 (&full)[RNT]) {
  buf[w * 64 + lane] = own;
  __syncthreads();
  collect<RNT>(buf, lane, cmp, full);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// ===================================================================================== GRU
// gradient-of-input-projection stores: fp32, or bf16 when the consumers (weight gradients, d(hist) product) run on the
// bf16 matrix pipe anyway (speed mode: halves the 393 MB that the backward-through-time kernel writes and two kernels read)
typedef __bf16 rnn_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_dpin(float* base, long off, f32x4 v, int h) {
  if (h) *reinterpret_cast<rnn_bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = __builtin_convertvector(v, rnn_bf16x4);
  else st4(base + off, v);
}

// The same store WITHOUT a branch around it (backward-through-time loops): a raw buffer store whose per-lane byte offset
// is out of range for the lanes that must not write.  A store under `if (ok)` makes the number of outstanding memory
// operations unknown at the top of the next step; the compiler then drains the queue (s_waitcnt vmcnt(0)) before the
// first use of the operands PREFETCHED for that step -- and waits for the stores' acknowledgements every step.
// rs: resource over the block's 16 histories (base advanced per block: element offsets stay far below 2^31).
typedef __amdgpu_buffer_rsrc_t rnn_rsrc_t;
__device__ __forceinline__ rnn_rsrc_t rnn_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000);
}
// resource over the 16 histories of block bx of an optional [Hn, T, row] tensor (null: every store is dropped)
__device__ __forceinline__ rnn_rsrc_t rnn_rsrc_blk(const float* p, int bx, int T, int row) {
  return __builtin_amdgcn_make_buffer_rsrc(p ? const_cast<float*>(p + (long)bx * 16 * T * row) : nullptr, 0,
                                           p ? 0x80000000u : 0u, 0x00020000);
}
__device__ __forceinline__ void st4_b(rnn_rsrc_t rs, bool ok, unsigned off, f32x4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ok ? off * 4u : 0x80000000u, 0, 0);
}
__device__ __forceinline__ void st_dpin_b(rnn_rsrc_t rs, bool ok, unsigned off, f32x4 v, int h) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if (h) {
    const rnn_bf16x4 hv = __builtin_convertvector(v, rnn_bf16x4);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rs, ok ? off * 2u : 0x80000000u, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ok ? off * 4u : 0x80000000u, 0, 0);
  }
}

// xb: LDS exchange area of the workgroup (f32x4 units): fwd uses [0, 2*RNT*64), bwd [0, 3*RNT*64)
// ATT: the attentional-update-gate variant (own kernels below, so the plain GRU's code and registers are untouched)
template <int RNT, bool ATT = false>
__device__ __forceinline__ void gru_fwd_body(const GruArgs& a, const int bx, f32x4* xb) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  const bool cmp = last_tile_compact<RNT>(n);
  f32x4 wr[RNT], wu[RNT], wc[RNT];
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) {
    const bool ck = kt == RNT - 1 && cmp;
    wr[kt] = load_w_fwd(a.Wgh, a.ldg, 0, n, w, kt, j, g, ck);
    wu[kt] = load_w_fwd(a.Wgh, a.ldg, n, n, w, kt, j, g, ck);
    wc[kt] = load_w_fwd(a.Wch, a.ldc, 0, n, w, kt, j, g, ck);
  }
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 hs[RNT];
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) {
    hs[kt] = Z4;
    if (!(hvalid && a.h0)) continue;
    const float* hp0 = a.h0 + h * a.h0_stride + 16 * kt;
    if (kt == RNT - 1 && cmp) {
      if (16 * kt + g < n) hs[kt].x = hp0[g];
      if (16 * kt + 4 + g < n) hs[kt].y = hp0[4 + g];
    } else if (16 * kt + 4 * g < n) {
      hs[kt] = ld4(hp0 + 4 * g);
    }
  }
  f32x4 hown = (cval && a.h0) ? ld4(a.h0 + h * a.h0_stride + col) : Z4;
  const long hin = ATT ? (hvalid ? h : 0) / a.in_div : (hvalid ? h : 0);   // history of this sequence
  const int len = hvalid ? min(a.seq_len[hin * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  // loads are UNCONDITIONAL (clamped addresses, select afterwards): a load under a branch makes the
  // number of outstanding memory operations unknown to the compiler, which then drains the whole queue
  // (s_waitcnt vmcnt(0)) right after issuing the prefetch -- one exposed HBM latency per time step
  const float* pin = a.Pin + hin * (long)T * a.ldp + (cval ? col : 0);
  const float* attp = ATT ? a.att + (hvalid ? h : 0) * (long)T : nullptr;
  const int t0 = a.t0, tend = min(Tmax, a.t1);
  f32x4 pn[3];
#pragma unroll
  for (int gb = 0; gb < 3; ++gb) pn[gb] = sel4(cval && t0 < len, ld4(pin + (long)t0 * a.ldp + gb * n), Z4);
  float an = ATT ? attp[t0] : 0.f;
  f32x4* bufA = xb;
  f32x4* bufB = xb + RNT * 64;
  const rnn_rsrc_t rhp = rnn_rsrc_blk(a.hprev, bx, T, n), rga = rnn_rsrc_blk(a.gates, bx, T, 3 * n);
  const rnn_rsrc_t ros = rnn_rsrc_blk(a.out_seq, bx, T, n);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 accr = pn[0], accu = pn[1], accc = pn[2];
    const float keep = 1.0f - an;       // (1 - att_score) of this step; exactly 1 without attention
    {  // prefetch next step's input projections
      const bool nl = (t + 1) < len;
      const long tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int gb = 0; gb < 3; ++gb) pn[gb] = sel4(cval && nl, ld4(pin + tn * a.ldp + gb * n), Z4);
      if (ATT) an = attp[tn];
    }
    mv1(accr, wr, hs, cmp);
    mv1(accu, wu, hs, cmp);
    const f32x4 r = sig4(accr), u = sig4(accu);
    f32x4 rh[RNT];
    xchg(bufA, w, lane, r * hown, cmp, rh);
    mv1(accc, wc, rh, cmp);
    const f32x4 c = tanh4(accc);
    const f32x4 ue = u * keep;
    const f32x4 hn = ue * hown + (1.0f - ue) * c;
    {  // branch-free stores (see st_dpin_b): the prefetched projections are waited for with an exact count
      const bool ok = live && cval;
      const unsigned pos = (unsigned)(j * T + t);
      st4_b(rhp, ok, pos * n + col, hown);
      st4_b(rga, ok, pos * 3 * n + col, r); st4_b(rga, ok, pos * 3 * n + n + col, u); st4_b(rga, ok, pos * 3 * n + 2 * n + col, c);
      st4_b(ros, ok, pos * n + col, hn);
    }
    hown = sel4(live, hn, hown);
    xchg(bufB, w, lane, hown, cmp, hs);
  }
  if (cval) {
    if (a.hT) st4(a.hT + h * n + col, hown);
    if (a.out_seq)
      for (int t = max(len, t0); t < a.t1; ++t) st4(a.out_seq + (h * T + t) * n + col, Z4);
  }
}

template <int RNT, bool ATT = false>
__device__ __forceinline__ void gru_bwd_body(const GruArgs& a, const int bx, f32x4* xb) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  // transposed operands: d(state in) = sum_o W[in][o] * dgate[o]; this wave owns in-tile w
  const bool cmp = last_tile_compact<RNT>(n);
  f32x4 wr[RNT], wu[RNT], wc[RNT];
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) {
    const bool ck = kt == RNT - 1 && cmp;
    wr[kt] = load_w_bwd(a.Wgh, a.ldg, 0, n, w, kt, j, g, ck);
    wu[kt] = load_w_bwd(a.Wgh, a.ldg, n, n, w, kt, j, g, ck);
    wc[kt] = load_w_bwd(a.Wch, a.ldc, 0, n, w, kt, j, g, ck);
  }
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 dh = (cval && a.dhT) ? ld4(a.dhT + h * n + col) : Z4;
  const long hin = ATT ? (hvalid ? h : 0) / a.in_div : (hvalid ? h : 0);
  const int len = hvalid ? min(a.seq_len[hin * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  const float* attp = ATT ? a.att + (hvalid ? h : 0) * (long)T : nullptr;
  if (cval)  // zero dPin past len
    for (int t = max(len, a.t0); t < a.t1; ++t) {
      const long dp = (h * T + t) * a.lddp + col;
      st_dpin(a.dPin, dp, Z4, a.dpin_bf16); st_dpin(a.dPin, dp + n, Z4, a.dpin_bf16); st_dpin(a.dPin, dp + 2 * n, Z4, a.dpin_bf16);
    }
  f32x4* bufA = xb;
  f32x4* bufR = xb + RNT * 64;
  f32x4* bufU = xb + 2 * RNT * 64;
  // Saved activations of step t are fetched ONE STEP AHEAD (unconditional loads at clamped, always-valid addresses,
  // masks applied at use); with them at the top of the step every step waited out an HBM round trip (SQ counters of the
  // round-2 kernel: half of all wave cycles parked in s_waitcnt, profiles/r03_rnn_pmc.md)
  const long hc = hvalid ? h : 0;
  const int colc = cval ? col : 0;
  const float* gbase = a.gates + hc * T * 3 * n + colc;
  const float* hbase = a.hprev + hc * T * n + colc;
  const float* dsbase = (a.dout_seq ? a.dout_seq : a.hprev) + hc * T * n + colc;   // (no dout_seq: loaded, not used)
  const bool has_ds = a.dout_seq != nullptr;
  const rnn_rsrc_t rdp = rnn_rsrc(a.dPin + (a.dpin_bf16 ? (long)bx * 16 * T * a.lddp / 2 : (long)bx * 16 * T * a.lddp));
  struct In { f32x4 r, u, c, hp, ds; float at; };
  auto fetch = [&](int t) {
    t = max(t, 0);
    In v;
    const float* gp = gbase + (long)t * 3 * n;
    v.r = ld4(gp); v.u = ld4(gp + n); v.c = ld4(gp + 2 * n);
    v.hp = ld4(hbase + (long)t * n);
    v.ds = ld4(dsbase + (long)t * n);
    v.at = ATT ? attp[t] : 0.f;
    return v;
  };
  const int tstart = min(Tmax, a.t1) - 1;
  In cur = fetch(tstart);
  for (int t = tstart; t >= a.t0; --t) {
    const In nxt = fetch(t - 1);
    const bool live = t < len;
    const bool ok = live && cval;
    const f32x4 r = sel4(ok, cur.r, Z4), u = sel4(ok, cur.u, Z4), c = sel4(ok, cur.c, Z4);
    const f32x4 hp = sel4(ok, cur.hp, Z4);
    f32x4 d = dh;
    if (has_ds) d += cur.ds;
    d = sel4(ok, d, Z4);
    const float keep = ATT ? 1.0f - cur.at : 1.0f;
    const f32x4 ue = u * keep;                       // effective update gate (u itself is what was saved)
    const f32x4 due = d * (hp - c);
    const f32x4 du = due * keep;
    const f32x4 dcp = d * (1.0f - ue) * (1.0f - c * c);
    f32x4 dhn = d * ue;
    if (ATT) {        // d att[s, t] = - sum over the features of u * d(ue): 4 lanes per sequence, RNT waves
      float sa = -(u.x * due.x + u.y * due.y + u.z * due.z + u.w * due.w);
      sa += __shfl_xor(sa, 16);
      sa += __shfl_xor(sa, 32);
      if (g == 0 && live && hvalid) atomicAdd(a.datt + h * (long)T + t, sa);
    }
    f32x4 full[RNT];
    xchg(bufA, w, lane, dcp, cmp, full);
    f32x4 drh = Z4;
    mv1(drh, wc, full, cmp);
    const f32x4 drp = drh * hp * r * (1.0f - r);
    const f32x4 dup = du * u * (1.0f - u);
    dhn += drh * r;
    bufR[w * 64 + lane] = drp;
    xchg(bufU, w, lane, dup, cmp, full);
    mv1(dhn, wu, full, cmp);
    collect<RNT>(bufR, lane, cmp, full);
    mv1(dhn, wr, full, cmp);
    {
      const unsigned dp = (unsigned)((j * T + t) * a.lddp + col);
      st_dpin_b(rdp, ok, dp, drp, a.dpin_bf16); st_dpin_b(rdp, ok, dp + n, dup, a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 2 * n, dcp, a.dpin_bf16);
    }
    dh = sel4(live, dhn, dh);
    cur = nxt;
  }
  if (a.dh0 && cval) st4(a.dh0 + h * n + col, dh);
}

// dynamic LDS: the largest exchange area (Time4LSTM backward) is 2 * 4 * RNT * 64 float4 (three-piece products: 2 buffers x 3
// piece images x 2 RNT chunks x 64 lanes x 16 bytes)
static size_t rnn_lds_bytes(int rnt, int pc = 2) { return (size_t)2 * (pc > 2 ? 6 : 4) * rnt * 64 * sizeof(f32x4); }
static int rnn_tiles(int n) { return n <= 48 ? 3 : 8; }

// launch K<3, X3> or K<8, X3> by the widest hidden size of the launch and the form of its hidden-to-hidden products
#define RNN_LAUNCH1(K, RNT_, X3_, grid, stream, args)                                                    \
  do {                                                                                                   \
    const size_t lds_ = rnn_lds_bytes(RNT_, X3_);                                                        \
    if (lds_ > 48 * 1024)                                                                                \
      CLSR_HIP(hipFuncSetAttribute((const void*)K<RNT_, X3_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
    hipLaunchKernelGGL((K<RNT_, X3_>), grid, dim3(64 * RNT_), lds_, (hipStream_t)(stream), args);        \
  } while (0)
#define RNN_LAUNCH(K, rnt, x3, grid, stream, args)                                                       \
  do {                                                                                                   \
    if ((rnt) == 3) {                                                                                    \
      if ((x3) == 3) RNN_LAUNCH1(K, 3, 3, grid, stream, args);                                           \
      else if ((x3) == 2) RNN_LAUNCH1(K, 3, 2, grid, stream, args);                                      \
      else RNN_LAUNCH1(K, 3, 0, grid, stream, args);                                                     \
    } else { /* (wide encoders: the three-piece form is not instantiated -- fp32-input MFMAs, the same accuracy) */ \
      if ((x3) == 2) RNN_LAUNCH1(K, 8, 2, grid, stream, args); else RNN_LAUNCH1(K, 8, 0, grid, stream, args); \
    }                                                                                                    \
  } while (0)

// Form of the products of a recurrence: 1 = fp32-input MFMA (bit-exact fp32), 2 = two bf16 pieces per operand (2^-16 per
// product term), 3 = three pieces (2^-23: fp32 accuracy; hidden sizes <= 48) -- see the *_x3 bodies.  Descriptors carry it
// (clsr_gru_desc.products / clsr_t4_desc.products; 0 = two pieces).  -> pieces per operand, 0 = fp32-input MFMAs
static int rnn_x3(int products) { return products == 1 ? 0 : products == 3 ? 3 : 2; }

// (the bodies are defined below the fp32 Time4LSTM ones)
template <int RNT, bool ATT, bool FP, int PC> __device__ __forceinline__ void gru_fwd_x3(const GruArgs& a, const int bx, bf16x8* xb);
template <int RNT, bool ATT, int PC> __device__ __forceinline__ void gru_bwd_x3(const GruArgs& a, const int bx, bf16x8* xb);
template <int RNT, bool FP, int PC> __device__ __forceinline__ void t4lstm_fwd_x3(const T4Args& a, const int bx, bf16x8* xb);
template <int RNT, int PC> __device__ __forceinline__ void t4lstm_bwd_x3(const T4Args& a, const int bx, bf16x8* xb);
#define XPC (X3 ? X3 : 2)      // (a valid piece count for the branch that is compiled out)
#define XB8(xb) reinterpret_cast<bf16x8*>(xb)

template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) gru_fwd_kernel(GruArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) gru_fwd_x3<RNT, false, false, XPC>(a, blockIdx.x, XB8(xb));
  else gru_fwd_body<RNT>(a, blockIdx.x, xb);
}
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) gru_bwd_kernel(GruArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) gru_bwd_x3<RNT, false, XPC>(a, blockIdx.x, XB8(xb));
  else gru_bwd_body<RNT>(a, blockIdx.x, xb);
}

static int check_rnn_shape(int Hn, int T, int n, int ld) {
  CLSR_CHECK_ARG(Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(n % 4 == 0 && n >= 4 && n <= 128 && ld % 4 == 0);
  return CLSR_OK;
}

extern "C" int clsr_gru_fwd(const float* Pin, int ldp, const float* Wgh, int ldg, const float* Wch,
                            int ldc, const float* h0, long h0_stride, const int* seq_len,
                            int len_stride, int Hn, int T, int n, float* hT, float* out_seq,
                            float* hprev, float* gates, void* stream) {
  CLSR_CHECK_ARG(Pin && Wgh && Wch && seq_len);
  int rc = check_rnn_shape(Hn, T, n, ldp);
  if (rc) return rc;
  CLSR_CHECK_SUPPORTED(h0_stride % 4 == 0);
  GruArgs a = {};
  a.Pin = Pin; a.ldp = ldp; a.Wgh = Wgh; a.ldg = ldg; a.Wch = Wch; a.ldc = ldc; a.h0 = h0;
  a.h0_stride = h0_stride; a.seq_len = seq_len; a.len_stride = len_stride; a.Hn = Hn; a.T = T; a.n = n;
  a.hT = hT; a.out_seq = out_seq; a.hprev = hprev; a.gates = gates; a.t0 = 0; a.t1 = T;
  RNN_LAUNCH(gru_fwd_kernel, rnn_tiles(n), false /* the plain entry points: always the fp32 form */, dim3(clsr_cdiv(Hn, 16)), stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_gru_bwd(const float* gates, const float* hprev, const float* Wgh, int ldg,
                            const float* Wch, int ldc, const int* seq_len, int len_stride, int Hn,
                            int T, int n, const float* dhT, const float* dout_seq, float* dPin,
                            float* dh0, void* stream) {
  CLSR_CHECK_ARG(gates && hprev && Wgh && Wch && seq_len && dPin);
  int rc = check_rnn_shape(Hn, T, n, ldg);
  if (rc) return rc;
  CLSR_CHECK_SUPPORTED(ldc % 4 == 0 && ((uintptr_t)Wgh % 16) == 0 && ((uintptr_t)Wch % 16) == 0);
  GruArgs a = {};
  a.gates = const_cast<float*>(gates); a.hprev = const_cast<float*>(hprev);
  a.Wgh = Wgh; a.ldg = ldg; a.Wch = Wch; a.ldc = ldc; a.seq_len = seq_len; a.len_stride = len_stride;
  a.Hn = Hn; a.T = T; a.n = n; a.dhT = dhT; a.dout_seq = dout_seq; a.dPin = dPin; a.dh0 = dh0;
  a.lddp = 3 * n; a.t0 = 0; a.t1 = T;
  RNN_LAUNCH(gru_bwd_kernel, rnn_tiles(n), false /* the plain entry points: always the fp32 form */, dim3(clsr_cdiv(Hn, 16)), stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ================================================================================ Time4LSTM
// Pin blocks (each n wide): 0 i | 1 j | 2 f | 3 o (already holds Tn.Wo1 + Tl.Wo2) | 4 tns | 5 tls
//   c' = sig(f + 1) * sig(tls) * c + sig(i) * sig(tns) * tanh(j) ;  m' = sig(o) * tanh(c')
// saved (training): act[Hn,T,6n] = sig i | tanh j | sig(f+1) | sig o | sig tns | sig tls,
//                   cst[Hn,T,n] = c', mprev[Hn,T,n] = m entering the step.
template <int RNT>
__device__ __forceinline__ void t4lstm_fwd_body(const T4Args& a, const int bx, f32x4* xb) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  const bool cmp = last_tile_compact<RNT>(n);
  f32x4 wm[4][RNT];
#pragma unroll
  for (int gb = 0; gb < 4; ++gb)
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt)
      wm[gb][kt] = load_w_fwd(a.Wm, a.ldm, gb * n, n, w, kt, j, g, kt == RNT - 1 && cmp);
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 cs = Z4, mown = Z4, ms[RNT];
#pragma unroll
  for (int kt = 0; kt < RNT; ++kt) {
    ms[kt] = Z4;
    if (!(hvalid && a.st_in)) continue;
    const float* mp0 = a.st_in + h * 2 * n + n + 16 * kt;      // m entering t0: the B operand of the first matvec
    if (kt == RNT - 1 && cmp) {
      if (16 * kt + g < n) ms[kt].x = mp0[g];
      if (16 * kt + 4 + g < n) ms[kt].y = mp0[4 + g];
    } else if (16 * kt + 4 * g < n) {
      ms[kt] = ld4(mp0 + 4 * g);
    }
  }
  if (cval && a.st_in) { cs = ld4(a.st_in + h * 2 * n + col); mown = ld4(a.st_in + h * 2 * n + n + col); }
  const int len = hvalid ? min(a.seq_len[h * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  const int t0 = a.t0, tend = min(Tmax, a.t1);
  const float* pin = a.Pin + (hvalid ? h : 0) * (long)T * a.ldp + (cval ? col : 0);
  f32x4 pn[6];
#pragma unroll
  for (int gb = 0; gb < 6; ++gb) pn[gb] = sel4(cval && t0 < len, ld4(pin + (long)t0 * a.ldp + gb * n), Z4);
  const rnn_rsrc_t ros = rnn_rsrc_blk(a.out_seq, bx, T, n), rac = rnn_rsrc_blk(a.act, bx, T, 6 * n);
  const rnn_rsrc_t rcs = rnn_rsrc_blk(a.act ? a.cst : nullptr, bx, T, n), rmp = rnn_rsrc_blk(a.act ? a.mprev : nullptr, bx, T, n);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 acc[4];
#pragma unroll
    for (int gb = 0; gb < 4; ++gb) acc[gb] = pn[gb];
    const f32x4 tns = pn[4], tls = pn[5];
    {
      const bool nl = (t + 1) < len;
      const long tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int gb = 0; gb < 6; ++gb) pn[gb] = sel4(cval && nl, ld4(pin + tn * a.ldp + gb * n), Z4);
    }
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt) {  // four independent accumulators interleaved
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) MFMA4(acc[gb], wm[gb][kt].x, ms[kt].x);
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) MFMA4(acc[gb], wm[gb][kt].y, ms[kt].y);
      if (kt == RNT - 1 && cmp) continue;
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) MFMA4(acc[gb], wm[gb][kt].z, ms[kt].z);
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) MFMA4(acc[gb], wm[gb][kt].w, ms[kt].w);
    }
    const f32x4 ig = sig4(acc[0]), jg = tanh4(acc[1]), fg = sig4(acc[2] + 1.0f);
    const f32x4 og = sig4(acc[3]), tn = sig4(tns), tlg = sig4(tls);
    const f32x4 cn = fg * tlg * cs + ig * tn * jg;
    const f32x4 mn = og * tanh4(cn);
    {  // branch-free stores (see st_dpin_b); scoring: the three training-only tensors are null resources
      const bool ok = live && cval;
      const unsigned pos = (unsigned)(j * T + t);
      st4_b(ros, ok, pos * n + col, mn);
      const unsigned ap = pos * 6 * n + col;
      st4_b(rac, ok, ap, ig); st4_b(rac, ok, ap + n, jg); st4_b(rac, ok, ap + 2 * n, fg); st4_b(rac, ok, ap + 3 * n, og);
      st4_b(rac, ok, ap + 4 * n, tn); st4_b(rac, ok, ap + 5 * n, tlg);
      st4_b(rcs, ok, pos * n + col, cn);
      st4_b(rmp, ok, pos * n + col, mown);
    }
    cs = sel4(live, cn, cs);
    mown = sel4(live, mn, mown);
    xchg(xb + (t & 1) * RNT * 64, w, lane, mown, cmp, ms);  // double buffered: one barrier per step
  }
  if (cval) {
    for (int t = max(len, t0); t < a.t1; ++t) st4(a.out_seq + (h * T + t) * n + col, Z4);
    if (a.st_out) { st4(a.st_out + h * 2 * n + col, cs); st4(a.st_out + h * 2 * n + n + col, mown); }
  }
}

template <int RNT>
__device__ __forceinline__ void t4lstm_bwd_body(const T4Args& a, const int bx, f32x4* xb) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  const bool cmp = last_tile_compact<RNT>(n);
  f32x4 wm[4][RNT];
#pragma unroll
  for (int gb = 0; gb < 4; ++gb)
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt)
      wm[gb][kt] = load_w_bwd(a.Wm, a.ldm, gb * n, n, w, kt, j, g, kt == RNT - 1 && cmp);
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 dc = Z4, dm = Z4;
  if (cval && a.dst_in) { dc = ld4(a.dst_in + h * 2 * n + col); dm = ld4(a.dst_in + h * 2 * n + n + col); }
  const int len = hvalid ? min(a.seq_len[h * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  if (cval)
    for (int t = max(len, a.t0); t < a.t1; ++t) {
      const long dp = (h * T + t) * a.lddp + col;
#pragma unroll
      for (int gb = 0; gb < 6; ++gb) st_dpin(a.dPin, dp + gb * n, Z4, a.dpin_bf16);
    }
  // saved activations one step ahead + branch-free dPin stores: see gru_bwd_body
  const long hc = hvalid ? h : 0;
  const int colc = cval ? col : 0;
  const float* abase = a.act + hc * T * 6 * n + colc;
  const float* cbase = a.cst + hc * T * n + colc;
  const float* dsbase = a.dout_seq + hc * T * n + colc;
  const rnn_rsrc_t rdp = rnn_rsrc(a.dPin + (a.dpin_bf16 ? (long)bx * 16 * T * a.lddp / 2 : (long)bx * 16 * T * a.lddp));
  struct In { f32x4 ig, jg, fg, og, tn, tlg, cn, cp, ds; };
  auto fetch = [&](int t) {
    t = max(t, 0);
    In v;
    const float* ap = abase + (long)t * 6 * n;
    v.ig = ld4(ap); v.jg = ld4(ap + n); v.fg = ld4(ap + 2 * n); v.og = ld4(ap + 3 * n); v.tn = ld4(ap + 4 * n);
    v.tlg = ld4(ap + 5 * n);
    v.cn = ld4(cbase + (long)t * n);
    v.cp = ld4(cbase + (long)max(t - 1, 0) * n);
    v.ds = ld4(dsbase + (long)t * n);
    return v;
  };
  // (128-wide layers: 32 weight tiles per wave leave no room for a second operand set -- the look-ahead spilled)
  constexpr bool PF = RNT <= 4;
  const int tstart = min(Tmax, a.t1) - 1;
  In cur = fetch(tstart), nxt = cur;
  for (int t = tstart; t >= a.t0; --t) {
    if (PF) nxt = fetch(t - 1);
    else cur = fetch(t);
    const bool live = t < len;
    const bool ok = live && cval;
    const f32x4 ig = sel4(ok, cur.ig, Z4), jg = sel4(ok, cur.jg, Z4), fg = sel4(ok, cur.fg, Z4);
    const f32x4 og = sel4(ok, cur.og, Z4), tn = sel4(ok, cur.tn, Z4);
    const f32x4 tlg = sel4(ok, cur.tlg, Z4);
    const f32x4 cn = sel4(ok, cur.cn, Z4);
    const f32x4 cp = sel4(ok && t > 0, cur.cp, Z4);
    f32x4 d = dm + cur.ds;
    d = sel4(ok, d, Z4);
    const f32x4 tc = tanh4(cn);
    const f32x4 dcc = sel4(ok, dc + d * og * (1.0f - tc * tc), Z4);
    f32x4 dg[4];
    dg[3] = d * tc * og * (1.0f - og);                         // d o_pre
    dg[2] = dcc * tlg * cp * fg * (1.0f - fg);                 // d f_pre
    dg[0] = dcc * tn * jg * ig * (1.0f - ig);                  // d i_pre
    dg[1] = dcc * ig * tn * (1.0f - jg * jg);                  // d j_pre
    const f32x4 dtn = dcc * ig * jg * tn * (1.0f - tn);        // d tns_pre
    const f32x4 dtl = dcc * fg * cp * tlg * (1.0f - tlg);      // d tls_pre
    const f32x4 dcn = dcc * fg * tlg;
    {
      const unsigned dp = (unsigned)((j * T + t) * a.lddp + col);
      st_dpin_b(rdp, ok, dp, dg[0], a.dpin_bf16); st_dpin_b(rdp, ok, dp + n, dg[1], a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 2 * n, dg[2], a.dpin_bf16); st_dpin_b(rdp, ok, dp + 3 * n, dg[3], a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 4 * n, dtn, a.dpin_bf16); st_dpin_b(rdp, ok, dp + 5 * n, dtl, a.dpin_bf16);
    }
    // publish the four gate-gradient tiles (double buffered by step parity), one barrier, then
    // d m_prev[own tile] = sum over gates and k-tiles
    f32x4* buf = xb + (t & 1) * 4 * RNT * 64;
#pragma unroll
    for (int gb = 0; gb < 4; ++gb) buf[(gb * RNT + w) * 64 + lane] = dg[gb];
    __syncthreads();
    f32x4 dma = Z4, dmb = Z4;   // two accumulators: two independent MFMA chains instead of one of 4 * RNT * 4 links
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt) {
      f32x4 bq[4];
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) bq[gb] = read_tile(buf + (gb * RNT + kt) * 64, lane, kt == RNT - 1 && cmp);
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) mvt((gb & 1) ? dmb : dma, wm[gb][kt], bq[gb], kt == RNT - 1 && cmp);
    }
    const f32x4 dmn = dma + dmb;
    dc = sel4(live, dcn, dc);
    dm = sel4(live, dmn, dm);
    if (PF) cur = nxt;
  }
  if (cval && a.dst_out) { st4(a.dst_out + h * 2 * n + col, dc); st4(a.dst_out + h * 2 * n + n + col, dm); }
}

template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) t4lstm_fwd_kernel(T4Args a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) t4lstm_fwd_x3<RNT, false, XPC>(a, blockIdx.x, XB8(xb));
  else t4lstm_fwd_body<RNT>(a, blockIdx.x, xb);
}
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) t4lstm_bwd_kernel(T4Args a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) t4lstm_bwd_x3<RNT, XPC>(a, blockIdx.x, XB8(xb));
  else t4lstm_bwd_body<RNT>(a, blockIdx.x, xb);
}


// ============================================================ split-bf16 forms of the four bodies above
// The hidden-to-hidden products of a step as  W.h ~ Whi.hhi + Whi.hlo + Wlo.hhi  on v_mfma_f32_16x16x32_bf16 (x = hi + lo,
// hi = RNE_bf16(x), lo = RNE_bf16(x - hi): 16 significand bits, the dropped lo.lo term is 2^-18 relative), fp32
// accumulation; everything else of a step (gates, state, saved activations, dPin) is the fp32 code of the bodies above.
// Why: an fp32-input MFMA runs at the fp32 VECTOR rate and blocks its SIMD's issue port for 32 cycles -- the 40 MFMAs
// of a Time4LSTM step (1 280 cycles) and the ~300 VALU instructions of its gates ran in SERIES, and two waves that share a
// SIMD could not hide each other's stalls (profiles/r03_rnn_pmc.md).  The bf16 MFMA takes ~17 cycles, a K = 32 chunk per
// instruction (6 per gate tile and step at n = 40 instead of 10 fp32 ones: 102 against 320 cycles) and co-issues with
// VALU work of the same and of other waves (scripts/mfma_valu_overlap.hip).
// Operand layouts of v_mfma_f32_16x16x32_bf16: A lane (i, g) = row i, k = 8g..8g+7; B lane (j, g) = column j (history),
// k = 8g..8g+7; D lane (j, g) = rows 4g..4g+3 of column j.  A state / gate-gradient VECTOR of the workgroup's 16
// histories lives in LDS in B-operand order -- [k chunk of 32][g][history j] x 8 bf16, hi image then lo image -- so
// that collecting it is one conflict-free 16-byte read per chunk and image, and publishing an own f32x4 (4 consecutive
// k of history j) is one 8-byte write per image.  Backward products concatenate their gate blocks along k
// (Time4LSTM: k = gate * n + feature, 4n = 160 = exactly 5 chunks at n = 40; GRU: [d r_pre | d u_pre], 80 -> 3 chunks)
// instead of padding every block to a multiple of 32.
// Tile-major private image of the Time4LSTM's saved activations (clsr_t4_desc.act_tiled): [16-history tile][t][7 blocks:
// sig i | tanh j | sig(f+1) | sig o | sig tns | sig tls | c'][feature tile w][lane] x f32x4 -- exactly the registers of the
// wave that produced them.  In the [Hn, T, 6n] layout a wave's 16-byte store touches 16 different rows (64 separate
// 64-byte pieces): the training forward of the Time4LSTM alone took 128 us, 71 us without its stores (ablation,
// scripts/abl_rnn.sh).  Only the forward and the backward recurrence read / write these tensors.
#define T4_TILED_ROW(RNT) (7 * (RNT) * 16)
// PC = bf16 pieces per operand: x = p[0] + p[1] (+ p[2]), p[i] = RNE_bf16(x - p[0] - .. - p[i-1]).  A product takes every
// piece pair whose indices sum to < PC, smallest terms first: PC = 2 -> hi.lo + lo.hi + hi.hi (2^-16 relative per term),
// PC = 3 -> six products, 2^-23 (the level of an fp32 product: precision="fp32")
// SPL: the THIRD piece of a weight operand lives in a per-lane LDS slot instead of registers (a lane reads back what it
// wrote: no barrier).  The fused-projection forward with three pieces holds seven weight blocks per wave: 326 registers with
// all pieces resident -- one wave per SIMD, the launch no longer fits the chip at once; 256 with 51 of them in scratch.
template <int KC, int PC, bool SPL = false> struct XA {
  bf16x8 p[SPL ? 2 : PC][KC];   // weights of this wave's 16 output rows, KC chunks of k
  bf16x8* sp;                   // SPL: this lane's third pieces, chunk c at sp[c * 64]
  __device__ __forceinline__ bf16x8 piece(int i, int c) const {
    if constexpr (SPL) { if (i == 2) return sp[c * 64]; }
    return p[i < (SPL ? 2 : PC) ? i : 0][c];
  }
};
template <int KC, int PC> struct XB { bf16x8 p[PC][KC]; };   // a full vector of the 16 histories

template <int KC, int PC>
__device__ __forceinline__ void xsplit8(f32x8 v, bf16x8 (&p)[PC][KC], int c) {
#pragma unroll
  for (int i = 0; i < PC; ++i) {
    p[i][c] = to_h(v);
    if (i + 1 < PC) v -= to_f(p[i][c]);
  }
}
template <int KC, int PC, bool SPL>
__device__ __forceinline__ void xsplit8a(f32x8 v, XA<KC, PC, SPL>& a, int c) {
  if constexpr (SPL) {
    a.p[0][c] = to_h(v);
    v -= to_f(a.p[0][c]);
    a.p[1][c] = to_h(v);
    a.sp[c * 64] = to_h(v - to_f(a.p[1][c]));
  } else {
    xsplit8<KC, PC>(v, a.p, c);
  }
}
// acc[n] += A[n] . B over chunk c for N accumulators side by side (independent MFMA chains, issued interleaved)
template <int N, int KC, int PC, bool SPL>
__device__ __forceinline__ void xmn(f32x4 (&acc)[N], const XA<KC, PC, SPL> (&a)[N], const XB<KC, PC>& b, int c) {
#pragma unroll
  for (int sidx = PC - 1; sidx >= 0; --sidx)
#pragma unroll
    for (int i = 0; i <= sidx; ++i)
#pragma unroll
      for (int nn = 0; nn < N; ++nn) HMFMA(acc[nn], a[nn].piece(i, c), b.p[sidx - i][c]);
}

// forward use: rows = outputs o = 16w + i of the block at colbase, k = input feature (W is [in][out]: strided reads, once)
template <int KC, int PC, bool SPL>
__device__ __forceinline__ void xa_load_fwd(XA<KC, PC, SPL>& a, const float* W, int ld, int colbase, int n, int w, int lane) {
  const int i = lane & 15, g = lane >> 4, o = 16 * w + i;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * c + 8 * g + e;
      v[e] = (o < n && k < n) ? W[(long)k * ld + colbase + o] : 0.f;
    }
    xsplit8a<KC, PC, SPL>(v, a, c);
  }
}
// the same with an input width K != n (input-projection weights: k = embedding feature) and the bias of the block as row
// k = K (the B operand carries a constant 1 there: the bias costs no register and no add)
template <int KC, int PC, bool SPL>
__device__ __forceinline__ void xa_load_fwd_k(XA<KC, PC, SPL>& a, const float* W, int ld, int colbase, int n, int K, int w, int lane,
                                              const float* bias) {
  const int i = lane & 15, g = lane >> 4, o = 16 * w + i;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * c + 8 * g + e;
      v[e] = (o < n && k < K) ? W[(long)k * ld + colbase + o] : (o < n && k == K) ? bias[o] : 0.f;
    }
    xsplit8a<KC, PC, SPL>(v, a, c);
  }
}
// backward use: rows = inputs in = 16w + i, k = the K consecutive columns of W's row from colbase on
template <int KC, int PC, bool SPL>
__device__ __forceinline__ void xa_load_bwd(XA<KC, PC, SPL>& a, const float* W, int ld, int colbase, int K, int n, int w, int lane) {
  const int i = lane & 15, g = lane >> 4, in = 16 * w + i;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k0 = 32 * c + 8 * g;
    const float* p = W + (long)(in < n ? in : 0) * ld + colbase;
    const f32x4 lo4 = (in < n && k0 < K) ? ld4(p + k0) : Z4;
    const f32x4 hi4 = (in < n && k0 + 4 < K) ? ld4(p + k0 + 4) : Z4;
    xsplit8a<KC, PC, SPL>((f32x8){lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w}, a, c);
  }
}

// publish the 4 consecutive k = k0..k0+3 (k0 % 4 == 0) of history j: one 8-byte write per piece image
template <int KC, int PC>
__device__ __forceinline__ void xb_publish(bf16x8* buf, int k0, int j, bool ok, f32x4 v) {
  const int slot = ((((k0 >> 5) * 4 + ((k0 & 31) >> 3)) * 16 + j) << 1) + ((k0 & 7) >> 2);   // 8-byte units
  bf16x4* b4 = reinterpret_cast<bf16x4*>(buf);
#pragma unroll
  for (int i = 0; i < PC; ++i) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    if (ok) b4[i * KC * 128 + slot] = h;
    if (i + 1 < PC) v -= __builtin_convertvector(h, f32x4);
  }
}
template <int KC, int PC>
__device__ __forceinline__ void xb_collect(const bf16x8* buf, int lane, XB<KC, PC>& v) {
#pragma unroll
  for (int i = 0; i < PC; ++i)
#pragma unroll
    for (int c = 0; c < KC; ++c) v.p[i][c] = buf[i * KC * 64 + c * 64 + lane];
}
// acc += A . B over the first kcu chunks, two accumulators (two independent MFMA chains)
template <int KC, int PC, bool SPL>
__device__ __forceinline__ void xmac2(f32x4& acc0, f32x4& acc1, const XA<KC, PC, SPL>& a, const XB<KC, PC>& b, int kcu) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < kcu) {     // (uniform; a `break` here keeps the loop rolled and the operand arrays in scratch)
      f32x4& acc = (c & 1) ? acc1 : acc0;
#pragma unroll
      for (int sidx = PC - 1; sidx >= 0; --sidx)
#pragma unroll
        for (int i = 0; i <= sidx; ++i) HMFMA(acc, a.piece(i, c), b.p[sidx - i][c]);
    }
  }
}
// the same with the vector read from LDS chunk by chunk (wide backward products: the whole vector would take as many
// registers as the weights)
template <int KC, int PC>
__device__ __forceinline__ void xmac2_lds(f32x4& acc0, f32x4& acc1, const XA<KC, PC>& a, const bf16x8* buf, int lane, int kcu) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < kcu) {
      bf16x8 bp[PC];
#pragma unroll
      for (int i = 0; i < PC; ++i) bp[i] = buf[i * KC * 64 + c * 64 + lane];
      f32x4& acc = (c & 1) ? acc1 : acc0;
#pragma unroll
      for (int sidx = PC - 1; sidx >= 0; --sidx)
#pragma unroll
        for (int i = 0; i <= sidx; ++i) HMFMA(acc, a.p[i][c], bp[sidx - i]);
    }
  }
}
__device__ __forceinline__ void xb_zero(bf16x8* xb, int n16) {
  const f32x4 z = Z4;
  for (int i = threadIdx.x; i < n16; i += blockDim.x) reinterpret_cast<f32x4*>(xb)[i] = z;
  __syncthreads();
}

// ---- fused input projection (FP): the input-side pre-activations of a step, x_t . W_x + b, are computed IN the
// recurrence from the history embeddings instead of being read from a [Hn, T, 3n | 6n] tensor that a batched GEMM wrote
// (clsr_gru_desc.X / clsr_t4_desc.X).  The recurrences are HBM-bound (Time4LSTM training forward alone: 492 MB in 120 us;
// neither its MFMAs nor its gate arithmetic show up in an ablation, its loads and stores each cost ~50 us,
// scripts/abl_rnn.sh): the projection tensor was 1 920 of the 4 640 bytes a (history, step) moved through the forward
// launch -- plus the same bytes written by the GEMM on the critical path in front of it.  x_t is read as fp32 in
// B-operand order (lane (j, g): features 8g..8g+7 of chunk c of history j) one step ahead and split in registers; the
// products of step t + 1 are issued between publishing the state of step t and the barrier that waits for the other
// waves, where the wave would idle.  D % 8 == 0, D <= 32 KC.
template <int KC>
__device__ __forceinline__ void xload(f32x8 (&xr)[KC], const float* xrow, int D, int g) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k0 = 32 * c + 8 * g;
    xr[c] = ld8f(xrow + (k0 < D ? k0 : 0));     // (unconditional, clamped: masked in xsplit)
  }
}
template <int KC, int PC>
__device__ __forceinline__ void xsplit(const f32x8 (&xr)[KC], int D, int g, XB<KC, PC>& xb) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k0 = 32 * c + 8 * g;
    f32x8 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    z[0] = k0 == D ? 1.0f : 0.f;               // the constant input that multiplies the bias row (xa_load_fwd_k)
    xsplit8<KC, PC>(k0 < D ? xr[c] : z, xb.p, c);
  }
}

template <int RNT, bool ATT, bool FP, int PC>
__device__ __forceinline__ void gru_fwd_x3(const GruArgs& a, const int bx, bf16x8* xb) {
  constexpr int KC = (RNT * 16 + 31) / 32;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const int kcu = (n + 31) >> 5;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  constexpr bool SPL = FP && PC == 3;
  bf16x8* const spill = xb + 2 * PC * KC * 64 + (size_t)w * 7 * KC * 64 + lane;      // (SPL: per-lane third pieces, 7 blocks per wave)
  XA<KC, PC, SPL> wg[2], wc;       // gates r | u, candidate
  wg[0].sp = spill; wg[1].sp = spill + KC * 64; wc.sp = spill + 2 * KC * 64;
  xa_load_fwd(wg[0], a.Wgh, a.ldg, 0, n, w, lane);
  xa_load_fwd(wg[1], a.Wgh, a.ldg, n, n, w, lane);
  xa_load_fwd(wc, a.Wch, a.ldc, 0, n, w, lane);
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 hown = (cval && a.h0) ? ld4(a.h0 + h * a.h0_stride + col) : Z4;
  const long hin = ATT ? (hvalid ? h : 0) / a.in_div : (hvalid ? h : 0);
  const int len = hvalid ? min(a.seq_len[hin * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  const float* pin = FP ? nullptr : a.Pin + hin * (long)T * a.ldp + (cval ? col : 0);
  const float* attp = ATT ? a.att + (hvalid ? h : 0) * (long)T : nullptr;
  const int t0 = a.t0, tend = min(Tmax, a.t1);
  // input side of a step: FP -- x . [Wgx | Wcx] + bias from the embeddings; else the projection tensor
  const int Dx = a.Dx, kcx = (Dx + 32) >> 5;     // (+ the bias row)
  XA<KC, PC, SPL> px[3];           // input side: r | u | c
  px[0].sp = spill + 3 * KC * 64; px[1].sp = spill + 4 * KC * 64; px[2].sp = spill + 5 * KC * 64;
  f32x8 xr[KC];
  const float* xrow = FP ? a.X + hin * (long)T * a.ldx : nullptr;
  f32x4 pn[3];
  if (FP) {
    xa_load_fwd_k(px[0], a.Wgx, a.ldg, 0, n, Dx, w, lane, a.bg);
    xa_load_fwd_k(px[1], a.Wgx, a.ldg, n, n, Dx, w, lane, a.bg + n);
    xa_load_fwd_k(px[2], a.Wcx, a.ldc, 0, n, Dx, w, lane, a.bc);
    xload<KC>(xr, xrow + (long)t0 * a.ldx, Dx, g);
  } else {
#pragma unroll
    for (int gb = 0; gb < 3; ++gb) pn[gb] = sel4(cval && t0 < len, ld4(pin + (long)t0 * a.ldp + gb * n), Z4);
  }
  auto project = [&](int tnext) {   // pn <- bias + x . W of the step whose embeddings sit in xr; then fetch step tnext
    XB<KC, PC> xv;
    xsplit<KC, PC>(xr, Dx, g, xv);
    const long tn = tnext < T ? tnext : T - 1;
    xload<KC>(xr, xrow + tn * a.ldx, Dx, g);
    f32x4 q[3] = {Z4, Z4, Z4};
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c < kcx) xmn(q, px, xv, c);
    pn[0] = q[0]; pn[1] = q[1]; pn[2] = q[2];
  };
  float an = ATT ? attp[t0] : 0.f;
  bf16x8* bufA = xb;                   // r . h
  bf16x8* bufB = xb + PC * KC * 64;     // h
  xb_zero(xb, 2 * PC * KC * 64);
  XB<KC, PC> hs, rh;
  xb_publish<KC, PC>(bufB, col, j, true, hown);
  if (FP) project(t0 + 1);
  __syncthreads();
  xb_collect<KC, PC>(bufB, lane, hs);
  const rnn_rsrc_t rhp = rnn_rsrc_blk(a.hprev, bx, T, n), rga = rnn_rsrc_blk(a.gates, bx, T, 3 * n);
  const rnn_rsrc_t ros = rnn_rsrc_blk(a.out_seq, bx, T, n);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 accg[2] = {pn[0], pn[1]}, accc = pn[2], accc1 = Z4;
    const float keep = 1.0f - an;
    if (!FP) {
      const bool nl = (t + 1) < len;
      const long tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int gb = 0; gb < 3; ++gb) pn[gb] = sel4(cval && nl, ld4(pin + tn * a.ldp + gb * n), Z4);
    }
    if (ATT) an = attp[t + 1 < T ? t + 1 : t];
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c < kcu) xmn(accg, wg, hs, c);
    // (FP) the input-side products of step t + 1 right behind the recurrent ones: independent MFMAs for the matrix pipe while
    // the VALU works on the gates -- between publish and barrier they sat on the critical path of EVERY wave (335 against
    // 176 us in the step)
    if (FP) project(t + 2);
    const f32x4 r = sig4(accg[0]), u = sig4(accg[1]);
    xb_publish<KC, PC>(bufA, col, j, true, r * hown);
    __syncthreads();
    xb_collect<KC, PC>(bufA, lane, rh);
    xmac2(accc, accc1, wc, rh, kcu);
    const f32x4 c = tanh4(accc + accc1);
    const f32x4 ue = u * keep;
    const f32x4 hn = ue * hown + (1.0f - ue) * c;
    {
      const bool ok = live && cval;
      const unsigned pos = (unsigned)(j * T + t);
      st4_b(rhp, ok, pos * n + col, hown);
      st4_b(rga, ok, pos * 3 * n + col, r); st4_b(rga, ok, pos * 3 * n + n + col, u); st4_b(rga, ok, pos * 3 * n + 2 * n + col, c);
      st4_b(ros, ok, pos * n + col, hn);
    }
    hown = sel4(live, hn, hown);
    xb_publish<KC, PC>(bufB, col, j, true, hown);
    __syncthreads();
    xb_collect<KC, PC>(bufB, lane, hs);
  }
  if (cval) {
    if (a.hT) st4(a.hT + h * n + col, hown);
    if (a.out_seq)
      for (int t = max(len, t0); t < a.t1; ++t) st4(a.out_seq + (h * T + t) * n + col, Z4);
  }
}

template <int RNT, bool ATT, int PC>
__device__ __forceinline__ void gru_bwd_x3(const GruArgs& a, const int bx, bf16x8* xb) {
  constexpr int KC = (RNT * 16 + 31) / 32;     // d c_pre: k = feature
  constexpr int KC2 = RNT;                     // [d r_pre | d u_pre]: k = gate * n + feature, 2n <= 32 RNT
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const int kcu = (n + 31) >> 5, kcu2 = (2 * n + 31) >> 5;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  XA<KC, PC> wc;
  XA<KC2, PC> wg;
  xa_load_bwd(wc, a.Wch, a.ldc, 0, n, n, w, lane);
  xa_load_bwd(wg, a.Wgh, a.ldg, 0, 2 * n, n, w, lane);
  const int col = 16 * w + 4 * g;
  const bool colv = col < n;
  const bool cval = hvalid && colv;
  f32x4 dh = (cval && a.dhT) ? ld4(a.dhT + h * n + col) : Z4;
  const long hin = ATT ? (hvalid ? h : 0) / a.in_div : (hvalid ? h : 0);
  const int len = hvalid ? min(a.seq_len[hin * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  const float* attp = ATT ? a.att + (hvalid ? h : 0) * (long)T : nullptr;
  if (cval)
    for (int t = max(len, a.t0); t < a.t1; ++t) {
      const long dp = (h * T + t) * a.lddp + col;
      st_dpin(a.dPin, dp, Z4, a.dpin_bf16); st_dpin(a.dPin, dp + n, Z4, a.dpin_bf16); st_dpin(a.dPin, dp + 2 * n, Z4, a.dpin_bf16);
    }
  bf16x8* bufA = xb;
  bf16x8* bufG = xb + PC * KC * 64;
  xb_zero(xb, PC * KC * 64 + PC * KC2 * 64);
  const long hc = hvalid ? h : 0;
  const int colc = cval ? col : 0;
  const float* gbase = a.gates + hc * T * 3 * n + colc;
  const float* hbase = a.hprev + hc * T * n + colc;
  const float* dsbase = (a.dout_seq ? a.dout_seq : a.hprev) + hc * T * n + colc;
  const bool has_ds = a.dout_seq != nullptr;
  const rnn_rsrc_t rdp = rnn_rsrc(a.dPin + (a.dpin_bf16 ? (long)bx * 16 * T * a.lddp / 2 : (long)bx * 16 * T * a.lddp));
  struct In { f32x4 r, u, c, hp, ds; float at; };
  auto fetch = [&](int t) {
    t = max(t, 0);
    In v;
    const float* gp = gbase + (long)t * 3 * n;
    v.r = ld4(gp); v.u = ld4(gp + n); v.c = ld4(gp + 2 * n);
    v.hp = ld4(hbase + (long)t * n);
    v.ds = ld4(dsbase + (long)t * n);
    v.at = ATT ? attp[t] : 0.f;
    return v;
  };
  const int tstart = min(Tmax, a.t1) - 1;
  In cur = fetch(tstart);
  for (int t = tstart; t >= a.t0; --t) {
    const In nxt = fetch(t - 1);
    const bool live = t < len;
    const bool ok = live && cval;
    const f32x4 r = sel4(ok, cur.r, Z4), u = sel4(ok, cur.u, Z4), c = sel4(ok, cur.c, Z4);
    const f32x4 hp = sel4(ok, cur.hp, Z4);
    f32x4 d = dh;
    if (has_ds) d += cur.ds;
    d = sel4(ok, d, Z4);
    const float keep = ATT ? 1.0f - cur.at : 1.0f;
    const f32x4 ue = u * keep;
    const f32x4 due = d * (hp - c);
    const f32x4 du = due * keep;
    const f32x4 dcp = d * (1.0f - ue) * (1.0f - c * c);
    f32x4 dhn = d * ue;
    if (ATT) {
      float sa = -(u.x * due.x + u.y * due.y + u.z * due.z + u.w * due.w);
      sa += __shfl_xor(sa, 16);
      sa += __shfl_xor(sa, 32);
      if (g == 0 && live && hvalid) atomicAdd(a.datt + h * (long)T + t, sa);
    }
    xb_publish<KC, PC>(bufA, col, j, colv, dcp);
    __syncthreads();
    f32x4 drh = Z4, drh1 = Z4;
    xmac2_lds<KC, PC>(drh, drh1, wc, bufA, lane, kcu);
    drh += drh1;
    const f32x4 drp = drh * hp * r * (1.0f - r);
    const f32x4 dup = du * u * (1.0f - u);
    dhn += drh * r;
    xb_publish<KC2, PC>(bufG, col, j, colv, drp);
    xb_publish<KC2, PC>(bufG, n + col, j, colv, dup);
    __syncthreads();
    f32x4 dh1 = Z4;
    xmac2_lds<KC2, PC>(dhn, dh1, wg, bufG, lane, kcu2);
    dhn += dh1;
    {
      const unsigned dp = (unsigned)((j * T + t) * a.lddp + col);
      st_dpin_b(rdp, ok, dp, drp, a.dpin_bf16); st_dpin_b(rdp, ok, dp + n, dup, a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 2 * n, dcp, a.dpin_bf16);
    }
    dh = sel4(live, dhn, dh);
    cur = nxt;
  }
  if (a.dh0 && cval) st4(a.dh0 + h * n + col, dh);
}

// FP: blocks i | j | f from x . kernel[0:D] + bias in the kernel; Pin then holds only o | tns | tls (3n wide: the product
// that also needs the time features, net.py "xw.t")
template <int RNT, bool FP, int PC>
__device__ __forceinline__ void t4lstm_fwd_x3(const T4Args& a, const int bx, bf16x8* xb) {
  constexpr int KC = (RNT * 16 + 31) / 32;
  constexpr int NP = FP ? 3 : 6;       // blocks read from Pin (PC: bf16 pieces per operand)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const int kcu = (n + 31) >> 5;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  constexpr bool SPL = FP && PC == 3;
  bf16x8* const spill = xb + 2 * PC * KC * 64 + (size_t)w * 7 * KC * 64 + lane;      // (SPL: per-lane third pieces, 7 blocks per wave)
  XA<KC, PC, SPL> wm[4];
#pragma unroll
  for (int gb = 0; gb < 4; ++gb) { wm[gb].sp = spill + gb * KC * 64; xa_load_fwd(wm[gb], a.Wm, a.ldm, gb * n, n, w, lane); }
  const int col = 16 * w + 4 * g;
  const bool cval = hvalid && col < n;
  f32x4 cs = Z4, mown = Z4;
  if (cval && a.st_in) { cs = ld4(a.st_in + h * 2 * n + col); mown = ld4(a.st_in + h * 2 * n + n + col); }
  const int len = hvalid ? min(a.seq_len[h * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  const int t0 = a.t0, tend = min(Tmax, a.t1);
  const float* pin = a.Pin + (hvalid ? h : 0) * (long)T * a.ldp + (cval ? col : 0);
  f32x4 pn[NP];
#pragma unroll
  for (int gb = 0; gb < NP; ++gb) pn[gb] = sel4(cval && t0 < len, ld4(pin + (long)t0 * a.ldp + gb * n), Z4);
  const int Dx = a.Dx, kcx = (Dx + 32) >> 5;     // (+ the bias row)
  XA<KC, PC, SPL> px[FP ? 3 : 1];
  f32x4 qn[3] = {Z4, Z4, Z4};
  f32x8 xr[KC];
  const float* xrow = FP ? a.X + (hvalid ? h : 0) * (long)T * a.ldx : nullptr;
  if (FP) {
#pragma unroll
    for (int gb = 0; gb < 3; ++gb) {
      px[gb].sp = spill + (4 + gb) * KC * 64;
      xa_load_fwd_k(px[gb], a.Wkx, a.ldm, gb * n, n, Dx, w, lane, a.bk + gb * n);
    }
    xload<KC>(xr, xrow + (long)t0 * a.ldx, Dx, g);
  }
  auto project = [&](int tnext) {
    XB<KC, PC> xv;
    xsplit<KC, PC>(xr, Dx, g, xv);
    const long tn = tnext < T ? tnext : T - 1;
    xload<KC>(xr, xrow + tn * a.ldx, Dx, g);
    if constexpr (FP) {
      f32x4 q[3] = {Z4, Z4, Z4};
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (c < kcx) xmn(q, px, xv, c);
      qn[0] = q[0]; qn[1] = q[1]; qn[2] = q[2];
    }
  };
  xb_zero(xb, 2 * PC * KC * 64);
  XB<KC, PC> ms;
  xb_publish<KC, PC>(xb + ((t0 + 1) & 1) * PC * KC * 64, col, j, true, mown);
  if (FP) project(t0 + 1);
  __syncthreads();
  xb_collect<KC, PC>(xb + ((t0 + 1) & 1) * PC * KC * 64, lane, ms);
  const bool tiled = a.act_tiled != 0;
  const rnn_rsrc_t ros = rnn_rsrc_blk(a.out_seq, bx, T, n), rac = rnn_rsrc_blk(a.act, bx, T, tiled ? T4_TILED_ROW(RNT) : 6 * n);
  const rnn_rsrc_t rcs = rnn_rsrc_blk(a.act && !tiled ? a.cst : nullptr, bx, T, n), rmp = rnn_rsrc_blk(a.act ? a.mprev : nullptr, bx, T, n);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 acc[4];
    if (FP) { acc[0] = qn[0]; acc[1] = qn[1]; acc[2] = qn[2]; acc[3] = pn[0]; }
    else {
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) acc[gb] = pn[gb];
    }
    const f32x4 tns = pn[NP - 2], tls = pn[NP - 1];
    {
      const bool nl = (t + 1) < len;
      const long tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int gb = 0; gb < NP; ++gb) pn[gb] = sel4(cval && nl, ld4(pin + tn * a.ldp + gb * n), Z4);
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c < kcu) xmn(acc, wm, ms, c);
    if (FP) project(t + 2);       // (behind the recurrent products, ahead of the gate arithmetic: see gru_fwd_x3)
    const f32x4 ig = sig4(acc[0]), jg = tanh4(acc[1]), fg = sig4(acc[2] + 1.0f);
    const f32x4 og = sig4(acc[3]), tn = sig4(tns), tlg = sig4(tls);
    const f32x4 cn = fg * tlg * cs + ig * tn * jg;
    const f32x4 mn = og * tanh4(cn);
    {
      const bool ok = live && cval;
      const unsigned pos = (unsigned)(j * T + t);
      st4_b(ros, ok, pos * n + col, mn);
      if (tiled) {   // private tile-major image: every store is one contiguous KB of the wave (see T4_TILED_ROW)
        const unsigned ap = (unsigned)(t * 7 * RNT + w) * 256u + lane * 4u, sb = RNT * 256u;
        st4_b(rac, true, ap, ig); st4_b(rac, true, ap + sb, jg); st4_b(rac, true, ap + 2 * sb, fg); st4_b(rac, true, ap + 3 * sb, og);
        st4_b(rac, true, ap + 4 * sb, tn); st4_b(rac, true, ap + 5 * sb, tlg); st4_b(rac, true, ap + 6 * sb, cn);
      } else {
        const unsigned ap = pos * 6 * n + col;
        st4_b(rac, ok, ap, ig); st4_b(rac, ok, ap + n, jg); st4_b(rac, ok, ap + 2 * n, fg); st4_b(rac, ok, ap + 3 * n, og);
        st4_b(rac, ok, ap + 4 * n, tn); st4_b(rac, ok, ap + 5 * n, tlg);
        st4_b(rcs, ok, pos * n + col, cn);
      }
      st4_b(rmp, ok, pos * n + col, mown);
    }
    cs = sel4(live, cn, cs);
    mown = sel4(live, mn, mown);
    bf16x8* buf = xb + (t & 1) * PC * KC * 64;     // double buffered: one barrier per step
    xb_publish<KC, PC>(buf, col, j, true, mown);
    __syncthreads();
    xb_collect<KC, PC>(buf, lane, ms);
  }
  if (cval) {
    for (int t = max(len, t0); t < a.t1; ++t) st4(a.out_seq + (h * T + t) * n + col, Z4);
    if (a.st_out) { st4(a.st_out + h * 2 * n + col, cs); st4(a.st_out + h * 2 * n + n + col, mown); }
  }
}

template <int RNT, int PC>
__device__ __forceinline__ void t4lstm_bwd_x3(const T4Args& a, const int bx, bf16x8* xb) {
  constexpr int KC4 = 2 * RNT;                 // k = gate * n + feature: 4n <= 64 RNT
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const int kcu = (4 * n + 31) >> 5;
  const long h = (long)bx * 16 + j;
  const bool hvalid = h < a.Hn;
  XA<KC4, PC> wm;
  xa_load_bwd(wm, a.Wm, a.ldm, 0, 4 * n, n, w, lane);
  const int col = 16 * w + 4 * g;
  const bool colv = col < n;
  const bool cval = hvalid && colv;
  f32x4 dc = Z4, dm = Z4;
  if (cval && a.dst_in) { dc = ld4(a.dst_in + h * 2 * n + col); dm = ld4(a.dst_in + h * 2 * n + n + col); }
  const int len = hvalid ? min(a.seq_len[h * a.len_stride], T) : 0;
  const int Tmax = wave_max_i(len);
  if (cval)
    for (int t = max(len, a.t0); t < a.t1; ++t) {
      const long dp = (h * T + t) * a.lddp + col;
#pragma unroll
      for (int gb = 0; gb < 6; ++gb) st_dpin(a.dPin, dp + gb * n, Z4, a.dpin_bf16);
    }
  xb_zero(xb, 2 * PC * KC4 * 64);
  const long hc = hvalid ? h : 0;
  const int colc = cval ? col : 0;
  const bool tiled = a.act_tiled != 0;
  const float* abase = tiled ? a.act + ((long)bx * T * 7 * RNT + w) * 256 + lane * 4 : a.act + hc * T * 6 * n + colc;
  const float* cbase = tiled ? abase + 6 * RNT * 256 : a.cst + hc * T * n + colc;
  const int sa = tiled ? 7 * RNT * 256 : 6 * n, sb = tiled ? RNT * 256 : n, sc = tiled ? 7 * RNT * 256 : n;
  const float* dsbase = a.dout_seq + hc * T * n + colc;
  const rnn_rsrc_t rdp = rnn_rsrc(a.dPin + (a.dpin_bf16 ? (long)bx * 16 * T * a.lddp / 2 : (long)bx * 16 * T * a.lddp));
  struct In { f32x4 ig, jg, fg, og, tn, tlg, cn, cp, ds; };
  auto fetch = [&](int t) {
    t = max(t, 0);
    In v;
    const float* ap = abase + (long)t * sa;
    v.ig = ld4(ap); v.jg = ld4(ap + sb); v.fg = ld4(ap + 2 * sb); v.og = ld4(ap + 3 * sb); v.tn = ld4(ap + 4 * sb);
    v.tlg = ld4(ap + 5 * sb);
    v.cn = ld4(cbase + (long)t * sc);
    v.cp = ld4(cbase + (long)max(t - 1, 0) * sc);
    v.ds = ld4(dsbase + (long)t * n);
    return v;
  };
#ifdef RNNX_NO_PF
  constexpr bool PF = false;
#else
  constexpr bool PF = RNT <= 4;
#endif
  const int tstart = min(Tmax, a.t1) - 1;
  In cur = fetch(tstart), nxt = cur;
  for (int t = tstart; t >= a.t0; --t) {
    if (PF) nxt = fetch(t - 1);
    else cur = fetch(t);
    const bool live = t < len;
    const bool ok = live && cval;
    const f32x4 ig = sel4(ok, cur.ig, Z4), jg = sel4(ok, cur.jg, Z4), fg = sel4(ok, cur.fg, Z4);
    const f32x4 og = sel4(ok, cur.og, Z4), tn = sel4(ok, cur.tn, Z4);
    const f32x4 tlg = sel4(ok, cur.tlg, Z4);
    const f32x4 cn = sel4(ok, cur.cn, Z4);
    const f32x4 cp = sel4(ok && t > 0, cur.cp, Z4);
    f32x4 d = dm + cur.ds;
    d = sel4(ok, d, Z4);
    const f32x4 tc = tanh4(cn);
    const f32x4 dcc = sel4(ok, dc + d * og * (1.0f - tc * tc), Z4);
    f32x4 dg[4];
    dg[3] = d * tc * og * (1.0f - og);
    dg[2] = dcc * tlg * cp * fg * (1.0f - fg);
    dg[0] = dcc * tn * jg * ig * (1.0f - ig);
    dg[1] = dcc * ig * tn * (1.0f - jg * jg);
    const f32x4 dtn = dcc * ig * jg * tn * (1.0f - tn);
    const f32x4 dtl = dcc * fg * cp * tlg * (1.0f - tlg);
    const f32x4 dcn = dcc * fg * tlg;
    {
      const unsigned dp = (unsigned)((j * T + t) * a.lddp + col);
      st_dpin_b(rdp, ok, dp, dg[0], a.dpin_bf16); st_dpin_b(rdp, ok, dp + n, dg[1], a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 2 * n, dg[2], a.dpin_bf16); st_dpin_b(rdp, ok, dp + 3 * n, dg[3], a.dpin_bf16);
      st_dpin_b(rdp, ok, dp + 4 * n, dtn, a.dpin_bf16); st_dpin_b(rdp, ok, dp + 5 * n, dtl, a.dpin_bf16);
    }
    bf16x8* buf = xb + (t & 1) * PC * KC4 * 64;
#pragma unroll
    for (int gb = 0; gb < 4; ++gb) xb_publish<KC4, PC>(buf, gb * n + col, j, colv, dg[gb]);
    __syncthreads();
    f32x4 dma = Z4, dmb = Z4;
    xmac2_lds<KC4, PC>(dma, dmb, wm, buf, lane, kcu);
    const f32x4 dmn = dma + dmb;
    dc = sel4(live, dcn, dc);
    dm = sel4(live, dmn, dm);
    if (PF) cur = nxt;
  }
  if (cval && a.dst_out) { st4(a.dst_out + h * 2 * n + col, dc); st4(a.dst_out + h * 2 * n + n + col, dm); }
}

// ------------------------------------------------------------------ fused multi-encoder launches
// The CLSR step runs up to three independent recurrences over the same histories (GRU
// short_term_intention, Time4LSTM / GRU short-term encoder, GRU causal2).  Each one only fills a
// quarter of the chip (Hn/16 single-wave workgroups), so they are dispatched as ONE grid:
// blockIdx.y selects the encoder, blockIdx.x the 16-history tile.
// Backward: the Time4LSTM chain is the longest of a launch (4 gate blocks + 2 time gates per step against the GRU's 3):
// its waves issue ahead of the GRU waves that share their SIMDs (s_setprio), the GRUs fill the gaps -- 300 -> 242 us for
// the three encoders alone (scripts/bench_rnn.py), ~10 us inside the step, where the saved activations come from HBM.
// The forward launch measured 3-8 % SLOWER with it and keeps equal priorities.
#ifdef RNN_NO_T4_PRIO
#define RNN_T4_PRIO()
#else
#define RNN_T4_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) rnn_multi_fwd_kernel(RnnMultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  CLSR_CHAIN_PRIO();
  // the Time4LSTM first (y = 0): workgroups are dispatched in y-major order, and the longest chain of the launch must not
  // be the one that waits for a free slot when the grid does not fit at once
  const int which = a.has_t4 ? (int)blockIdx.y - 1 : (int)blockIdx.y;
  if (X3) {
    if (which >= 0) gru_fwd_x3<RNT, false, false, XPC>(a.gru[which], blockIdx.x, XB8(xb));
    else t4lstm_fwd_x3<RNT, false, XPC>(a.t4, blockIdx.x, XB8(xb));
    return;
  }
  if (which >= 0) gru_fwd_body<RNT>(a.gru[which], blockIdx.x, xb);
  else t4lstm_fwd_body<RNT>(a.t4, blockIdx.x, xb);
}
// split-bf16 recurrences with the input projection fused (every encoder of the launch carries X)
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT, 2) rnn_multi_fwd_fp_kernel(RnnMultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  CLSR_CHAIN_PRIO();
  const int which = a.has_t4 ? (int)blockIdx.y - 1 : (int)blockIdx.y;      // (the Time4LSTM first: see rnn_multi_fwd_kernel)
  if (which >= 0) gru_fwd_x3<RNT, false, true, XPC>(a.gru[which], blockIdx.x, XB8(xb));
  else t4lstm_fwd_x3<RNT, true, XPC>(a.t4, blockIdx.x, XB8(xb));
}
#ifdef RNNX_WPE3
#define RNN_WPE __attribute__((amdgpu_waves_per_eu(RNT == 3 ? 3 : 2)))
#else
#define RNN_WPE
#endif
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) RNN_WPE rnn_multi_bwd_kernel(RnnMultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  CLSR_CHAIN_PRIO();
  const int which = a.has_t4 ? (int)blockIdx.y - 1 : (int)blockIdx.y;      // (the Time4LSTM first: see rnn_multi_fwd_kernel)
  if (X3) {
    if (which >= 0) { gru_bwd_x3<RNT, false, XPC>(a.gru[which], blockIdx.x, XB8(xb)); return; }
    RNN_T4_PRIO();
    t4lstm_bwd_x3<RNT, XPC>(a.t4, blockIdx.x, XB8(xb));
    return;
  }
  if (which >= 0) { gru_bwd_body<RNT>(a.gru[which], blockIdx.x, xb); return; }
  RNN_T4_PRIO();
  t4lstm_bwd_body<RNT>(a.t4, blockIdx.x, xb);
}


extern "C" int clsr_t4lstm_fwd(const float* Pin, int ldp, const float* Wm, int ldm, const int* seq_len,
                               int len_stride, int Hn, int T, int n, float* out_seq, float* act,
                               float* cst, float* mprev, void* stream) {
  CLSR_CHECK_ARG(Pin && Wm && seq_len && out_seq);
  CLSR_CHECK_ARG(!act || (cst && mprev));
  int rc = check_rnn_shape(Hn, T, n, ldp);
  if (rc) return rc;
  T4Args a = {};
  a.Pin = Pin; a.ldp = ldp; a.Wm = Wm; a.ldm = ldm; a.seq_len = seq_len; a.len_stride = len_stride;
  a.Hn = Hn; a.T = T; a.n = n; a.out_seq = out_seq; a.act = act; a.cst = cst; a.mprev = mprev; a.t0 = 0; a.t1 = T;
  RNN_LAUNCH(t4lstm_fwd_kernel, rnn_tiles(n), false /* the plain entry points: always the fp32 form */, dim3(clsr_cdiv(Hn, 16)), stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_t4lstm_bwd(const float* act, const float* cst, const float* Wm, int ldm,
                               const int* seq_len, int len_stride, int Hn, int T, int n,
                               const float* dout_seq, float* dPin, void* stream) {
  CLSR_CHECK_ARG(act && cst && Wm && seq_len && dout_seq && dPin);
  int rc = check_rnn_shape(Hn, T, n, ldm);
  if (rc) return rc;
  CLSR_CHECK_SUPPORTED(((uintptr_t)Wm % 16) == 0);
  T4Args a = {};
  a.act = const_cast<float*>(act); a.cst = const_cast<float*>(cst); a.Wm = Wm; a.ldm = ldm;
  a.seq_len = seq_len; a.len_stride = len_stride; a.Hn = Hn; a.T = T; a.n = n;
  a.dout_seq = dout_seq; a.dPin = dPin; a.lddp = 6 * n; a.t0 = 0; a.t1 = T;
  RNN_LAUNCH(t4lstm_bwd_kernel, rnn_tiles(n), false /* the plain entry points: always the fp32 form */, dim3(clsr_cdiv(Hn, 16)), stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ======================================================================= Time4LSTM time inputs
// TT[h,t,:] = [tanh(t_now*w1 + b1) (n) | tanh(t_first*w2 + b2) (n)]
// (rnn_cell_implement.py:200-205; inputs[:, -1] = time_to_now, inputs[:, -2] = time_from_first_action)
// thread -> (row slot, 16-byte column chunk): 32-bit index arithmetic, one division per row (the element-wise
// form of this kernel spent its time in 64-bit divisions: 131 us for 16 M tanh; this one is bound by its 65 MB store)
__global__ void __launch_bounds__(256) t4_time_inputs_fwd_kernel(
    const float* __restrict__ tnow, const float* __restrict__ tfirst, long row_stride, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, long Hn, int T, int n,
    float* __restrict__ TT, const float* __restrict__ hist, int D, float* __restrict__ XT, int ldxt, int col0) {
  const int QC = (2 * n) >> 2, rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  if (ty >= rpb) return;
  const int c = 4 * q;
  const bool first = c < n;
  const f32x4 w = ld4(first ? w1 + c : w2 + (c - n)), b = ld4(first ? b1 + c : b2 + (c - n));
  const float* tsrc = first ? tnow : tfirst;
  const int M = (int)(Hn * T);
  for (int row = blockIdx.x * rpb + ty; row < M; row += gridDim.x * rpb) {
    const int h = row / T, t = row - h * T;
    const float x = tsrc[(long)h * row_stride + t];
    f32x4 v;
    v.x = tanhf_(x * w.x + b.x); v.y = tanhf_(x * w.y + b.y); v.z = tanhf_(x * w.z + b.z); v.w = tanhf_(x * w.w + b.w);
    st4(TT + (long)row * (2 * n) + c, v);
    if (XT) {   // second image [hist | pad | TT] of the row: the input of the K-fused time-gate projection (net.py)
      st4(XT + (long)row * ldxt + col0 + c, v);
      if (c < D) st4(XT + (long)row * ldxt + c, ld4(hist + (long)row * D + c));
    }
  }
}

extern "C" int clsr_t4_time_inputs_fwd2(const float* tnow, const float* tfirst, long row_stride,
                                        const float* w1, const float* b1, const float* w2,
                                        const float* b2, long Hn, int T, int n, float* TT, const float* hist, int D,
                                        float* XT, int ldxt, int col0, void* stream) {
  CLSR_CHECK_ARG(tnow && tfirst && w1 && b1 && w2 && b2 && TT && Hn > 0 && T > 0 && n > 0);
  CLSR_CHECK_SUPPORTED(n % 4 == 0 && 2 * n <= 1024 && Hn * T < (1L << 31));
  CLSR_CHECK_ARG(!XT || (hist && D > 0 && D % 4 == 0 && D <= 2 * n && col0 >= D && col0 % 4 == 0 && ldxt >= col0 + 2 * n &&
                        ldxt % 4 == 0));
  int blocks = clsr_cdiv(Hn * T, (256 / ((2 * n) / 4)) * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(t4_time_inputs_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tnow,
                     tfirst, row_stride, w1, b1, w2, b2, Hn, T, n, TT, hist, D, XT, ldxt, col0);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_t4_time_inputs_fwd(const float* tnow, const float* tfirst, long row_stride,
                                       const float* w1, const float* b1, const float* w2,
                                       const float* b2, long Hn, int T, int n, float* TT, void* stream) {
  return clsr_t4_time_inputs_fwd2(tnow, tfirst, row_stride, w1, b1, w2, b2, Hn, T, n, TT, nullptr, 0, nullptr, 0, 0, stream);
}

// dpre = dTT * (1 - TT^2); partial[blk][0][c] = sum dpre * time ; partial[blk][1][c] = sum dpre
// (c < n: w1/b1 with time_to_now; c >= n: w2/b2 with time_from_first_action).  One column per
// thread: block = (256 / C2) rows x C2 columns.
__global__ void __launch_bounds__(256) t4_time_inputs_bwd_kernel(
    const float* __restrict__ dTT, const float* __restrict__ TT, const float* __restrict__ tnow,
    const float* __restrict__ tfirst, long row_stride, long Hn, int T, int n, float* __restrict__ partial, int t0, int tc) {
  __shared__ f32x4 red[2][256];
  const int C2 = 2 * n, QC = C2 >> 2;
  const int rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  f32x4 sw = {0.f, 0.f, 0.f, 0.f}, sb = sw;
  if (ty < rpb) {
    const int c = 4 * q;
    const float* tsrc = (c < n) ? tnow : tfirst;
    const int M = (int)(Hn * tc);     // the steps [t0, t0 + tc) of every history
    for (int vrow = blockIdx.x * rpb + ty; vrow < M; vrow += gridDim.x * rpb) {
      const int h = vrow / tc, t = t0 + vrow - h * tc;
      const long row = (long)h * T + t;
      const f32x4 y = ld4(TT + row * C2 + c);
      const f32x4 d = ld4(dTT + row * C2 + c) * ((f32x4){1.f, 1.f, 1.f, 1.f} - y * y);
      sw += d * tsrc[(long)h * row_stride + t];
      sb += d;
    }
  }
  red[0][threadIdx.x] = sw;
  red[1][threadIdx.x] = sb;
  __syncthreads();
  if (threadIdx.x < QC) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    for (int y = 0; y < rpb; ++y) { a += red[0][y * QC + threadIdx.x]; b += red[1][y * QC + threadIdx.x]; }
    st4(partial + ((long)blockIdx.x * 2 + 0) * C2 + 4 * threadIdx.x, a);
    st4(partial + ((long)blockIdx.x * 2 + 1) * C2 + 4 * threadIdx.x, b);
  }
}

static int t4_tbwd_blocks(long M, int n) {
  const int rpb = 256 / ((2 * n) / 4);
  long b = (M + rpb * 8 - 1) / (rpb * 8);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" int clsr_t4_time_inputs_bwd_parts(long Hn, int T, int n) { return t4_tbwd_blocks(Hn * T, n); }

// partial: [parts][2][2n] floats: row 0 = [d w1 | d w2], row 1 = [d b1 | d b2]
extern "C" int clsr_t4_time_inputs_bwd(const float* dTT, const float* TT, const float* tnow,
                                       const float* tfirst, long row_stride, long Hn, int T, int n,
                                       float* partial, void* stream) {
  CLSR_CHECK_ARG(dTT && TT && tnow && tfirst && partial && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(n > 0 && n % 4 == 0 && 2 * n <= 1024 && Hn * T < (1L << 31));
  hipLaunchKernelGGL(t4_time_inputs_bwd_kernel, dim3(t4_tbwd_blocks(Hn * T, n)), dim3(256), 0,
                     (hipStream_t)stream, dTT, TT, tnow, tfirst, row_stride, Hn, T, n, partial, 0, T);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// the same sums over the steps [t0, t1) only: clsr_t4_time_inputs_bwd_parts(Hn, t1 - t0, n) partial rows
extern "C" int clsr_t4_time_inputs_bwd_range(const float* dTT, const float* TT, const float* tnow,
                                             const float* tfirst, long row_stride, long Hn, int T, int t0, int t1, int n,
                                             float* partial, void* stream) {
  CLSR_CHECK_ARG(dTT && TT && tnow && tfirst && partial && Hn > 0 && T > 0 && 0 <= t0 && t0 < t1 && t1 <= T);
  CLSR_CHECK_SUPPORTED(n > 0 && n % 4 == 0 && 2 * n <= 1024 && Hn * T < (1L << 31));
  hipLaunchKernelGGL(t4_time_inputs_bwd_kernel, dim3(t4_tbwd_blocks(Hn * (t1 - t0), n)), dim3(256), 0,
                     (hipStream_t)stream, dTT, TT, tnow, tfirst, row_stride, Hn, T, n, partial, t0, t1 - t0);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- C-ABI descriptors: clsr_gru_desc / clsr_t4_desc from include/clsr_hip.h
// floats of the tile-major activation image of clsr_t4_desc.act_tiled (hidden size n decides the wave count of the launch)
extern "C" long clsr_t4_act_tiled_floats(long Hn, int T, int n) {
  return (long)clsr_cdiv(Hn, 16) * T * 7 * rnn_tiles(n) * 256;
}
extern "C" int clsr_sizeof_gru_desc(void) { return (int)sizeof(clsr_gru_desc); }
extern "C" int clsr_sizeof_t4_desc(void) { return (int)sizeof(clsr_t4_desc); }

static int fill_multi(RnnMultiArgs& m, const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4,
                      const int* seq_len, int len_stride, int Hn, int T, bool backward, int t0, int t1) {
  CLSR_CHECK_ARG(seq_len && Hn > 0 && T > 0 && ngru >= 0 && ngru <= RNN_MAX_GRU && (ngru > 0 || t4));
  CLSR_CHECK_ARG(0 <= t0 && t0 < t1 && t1 <= T);
  CLSR_CHECK_ARG(ngru == 0 || grus);
  m.ngru = ngru;
  m.has_t4 = t4 ? 1 : 0;
  m.products = t4 ? t4->products : 0;
  for (int i = 0; i < ngru; ++i) {
    const clsr_gru_desc& d = grus[i];
    int rc = check_rnn_shape(Hn, T, d.n, backward ? d.ldg : d.ldp);
    if (rc) return rc;
    CLSR_CHECK_ARG(d.Wgh && d.Wch && (backward ? (d.gates && d.hprev && d.dPin) : (d.Pin != nullptr || d.X != nullptr)));
    GruArgs& a = m.gru[i];
    a = GruArgs{};
    a.Pin = d.Pin; a.ldp = d.ldp; a.Wgh = d.Wgh; a.ldg = d.ldg; a.Wch = d.Wch; a.ldc = d.ldc;
    a.h0 = d.h0; a.h0_stride = d.h0_stride; a.seq_len = seq_len; a.len_stride = len_stride;
    a.Hn = Hn; a.T = T; a.n = d.n; a.hT = d.hT; a.out_seq = d.out_seq; a.hprev = d.hprev; a.gates = d.gates;
    a.dhT = d.dhT; a.dout_seq = d.dout_seq; a.dPin = d.dPin; a.dh0 = d.dh0;
    a.lddp = d.lddp > 0 ? d.lddp : 3 * d.n;
    a.dpin_bf16 = d.dpin_bf16;
    CLSR_CHECK_ARG(d.products >= 0 && d.products <= 3);
    if (d.products) m.products = d.products;
    CLSR_CHECK_SUPPORTED(a.lddp % 4 == 0);
    // the branch-free stores address a block's 16 histories with 32-bit BYTE offsets into one buffer resource
    CLSR_CHECK_SUPPORTED(16L * T * (a.lddp > 3 * d.n ? a.lddp : 3 * d.n) * 4 < 0x80000000L);
    a.att = d.att; a.datt = d.datt; a.in_div = d.in_div > 1 ? d.in_div : 1;
    a.X = d.X; a.ldx = d.ldx; a.Dx = d.Dx; a.Wgx = d.Wgx; a.Wcx = d.Wcx; a.bg = d.bg; a.bc = d.bc;
    if (!backward && d.X) {
      CLSR_CHECK_ARG(d.Wgx && d.Wcx && d.bg && d.bc && !d.att);
      CLSR_CHECK_SUPPORTED(d.Dx > 0 && d.Dx % 8 == 0 && d.Dx < 64 && d.ldx % 4 == 0 && d.n <= 48 && ((uintptr_t)d.X % 16) == 0);
    }
    a.t0 = t0; a.t1 = t1;
    CLSR_CHECK_ARG(!(backward && d.att && !d.datt));
  }
  if (t4) {
    int rc = check_rnn_shape(Hn, T, t4->n, backward ? t4->ldm : t4->ldp);
    if (rc) return rc;
    CLSR_CHECK_ARG(t4->Wm && (backward ? (t4->act && (t4->cst || t4->act_tiled) && t4->dout_seq && t4->dPin)
                                       : (t4->Pin && t4->out_seq && (!t4->act || ((t4->cst || t4->act_tiled) && t4->mprev)))));
    T4Args& a = m.t4;
    a = T4Args{};
    a.Pin = t4->Pin; a.ldp = t4->ldp; a.Wm = t4->Wm; a.ldm = t4->ldm; a.seq_len = seq_len;
    a.len_stride = len_stride; a.Hn = Hn; a.T = T; a.n = t4->n; a.out_seq = t4->out_seq; a.act = t4->act;
    a.cst = t4->cst; a.mprev = t4->mprev; a.act_tiled = t4->act_tiled; a.dout_seq = t4->dout_seq; a.dPin = t4->dPin;
    a.lddp = t4->lddp > 0 ? t4->lddp : 6 * t4->n;
    a.dpin_bf16 = t4->dpin_bf16;
    CLSR_CHECK_SUPPORTED(a.lddp % 4 == 0);
    CLSR_CHECK_SUPPORTED(16L * T * (a.lddp > 6 * t4->n ? a.lddp : 6 * t4->n) * 4 < 0x80000000L);
    a.t0 = t0; a.t1 = t1;
    a.st_in = t4->st_in; a.st_out = t4->st_out; a.dst_in = t4->dst_in; a.dst_out = t4->dst_out;
    a.X = t4->X; a.ldx = t4->ldx; a.Dx = t4->Dx; a.Wkx = t4->Wkx; a.bk = t4->bk;
    if (!backward && t4->X) {
      CLSR_CHECK_ARG(t4->Wkx && t4->bk);
      CLSR_CHECK_SUPPORTED(t4->Dx > 0 && t4->Dx % 8 == 0 && t4->Dx < 64 && t4->ldx % 4 == 0 && t4->n <= 48 && ((uintptr_t)t4->X % 16) == 0);
    }
  }
  return CLSR_OK;
}

// the attentional GRU runs alone (its inputs depend on everything the other encoders produce): own kernels
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) augru_fwd_kernel(GruArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) gru_fwd_x3<RNT, true, false, XPC>(a, blockIdx.x, XB8(xb));
  else gru_fwd_body<RNT, true>(a, blockIdx.x, xb);
}
template <int RNT, int X3>
__global__ void __launch_bounds__(64 * RNT) augru_bwd_kernel(GruArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 xb[];
  if (X3) gru_bwd_x3<RNT, true, XPC>(a, blockIdx.x, XB8(xb));
  else gru_bwd_body<RNT, true>(a, blockIdx.x, xb);
}

static int multi_tiles(const RnnMultiArgs& m) {
  int n = m.has_t4 ? m.t4.n : 0;
  for (int i = 0; i < m.ngru; ++i) n = m.gru[i].n > n ? m.gru[i].n : n;
  return rnn_tiles(n);
}

// the steps [t0, t1) of every recurrence of the launch (state carried through the descriptors: GRU h0 / hT, Time4LSTM
// st_in / st_out)
extern "C" int clsr_rnn_fwd_multi_range(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4,
                                        const int* seq_len, int len_stride, int Hn, int T, int t0, int t1, void* stream) {
  RnnMultiArgs m;
  int rc = fill_multi(m, grus, ngru, t4, seq_len, len_stride, Hn, T, false, t0, t1);
  if (rc) return rc;
  if (ngru > 0 && grus[0].att) {
    CLSR_CHECK_SUPPORTED(ngru == 1 && !t4);
    RNN_LAUNCH(augru_fwd_kernel, rnn_tiles(m.gru[0].n), rnn_x3(m.products), dim3(clsr_cdiv(Hn, 16)), stream, m.gru[0]);
    CLSR_CHECK_LAUNCH();
    return CLSR_OK;
  }
  {   // fused input projection: all encoders or none
    int nfp = (t4 && t4->X) ? 1 : 0;
    for (int i = 0; i < ngru; ++i) nfp += grus[i].X ? 1 : 0;
    if (nfp) {
      CLSR_CHECK_ARG(nfp == ngru + (t4 ? 1 : 0));
      CLSR_CHECK_SUPPORTED(rnn_x3(m.products) && multi_tiles(m) == 3);
      if (rnn_x3(m.products) == 3) {      // (+ the per-lane third pieces of the seven weight blocks of each of the three waves)
        const size_t lds_ = (size_t)2 * 3 * 2 * 64 * 16 + (size_t)3 * 7 * 2 * 64 * 16;
        CLSR_HIP(hipFuncSetAttribute((const void*)rnn_multi_fwd_fp_kernel<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));
        hipLaunchKernelGGL((rnn_multi_fwd_fp_kernel<3, 3>), dim3(clsr_cdiv(Hn, 16), ngru + (t4 ? 1 : 0)), dim3(192), lds_, (hipStream_t)stream, m);
      }
      else RNN_LAUNCH1(rnn_multi_fwd_fp_kernel, 3, 2, dim3(clsr_cdiv(Hn, 16), ngru + (t4 ? 1 : 0)), stream, m);
      CLSR_CHECK_LAUNCH();
      return CLSR_OK;
    }
  }
  // (the tile-major activation image is written by the split-product bodies only)
  CLSR_CHECK_SUPPORTED(!(t4 && t4->act_tiled) || (rnn_x3(m.products) != 0 && !(rnn_x3(m.products) == 3 && multi_tiles(m) != 3)));
  RNN_LAUNCH(rnn_multi_fwd_kernel, multi_tiles(m), rnn_x3(m.products), dim3(clsr_cdiv(Hn, 16), ngru + (t4 ? 1 : 0)), stream, m);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_rnn_fwd_multi(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4,
                                  const int* seq_len, int len_stride, int Hn, int T, void* stream) {
  return clsr_rnn_fwd_multi_range(grus, ngru, t4, seq_len, len_stride, Hn, T, 0, T, stream);
}

// backward through the steps [t0, t1), descending (gradients carried: GRU dhT / dh0, Time4LSTM dst_in / dst_out)
extern "C" int clsr_rnn_bwd_multi_range(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4,
                                        const int* seq_len, int len_stride, int Hn, int T, int t0, int t1, void* stream) {
  RnnMultiArgs m;
  int rc = fill_multi(m, grus, ngru, t4, seq_len, len_stride, Hn, T, true, t0, t1);
  if (rc) return rc;
  if (ngru > 0 && grus[0].att) {
    CLSR_CHECK_SUPPORTED(ngru == 1 && !t4);
    RNN_LAUNCH(augru_bwd_kernel, rnn_tiles(m.gru[0].n), rnn_x3(m.products), dim3(clsr_cdiv(Hn, 16)), stream, m.gru[0]);
    CLSR_CHECK_LAUNCH();
    return CLSR_OK;
  }
  CLSR_CHECK_SUPPORTED(!(t4 && t4->act_tiled) || (rnn_x3(m.products) != 0 && !(rnn_x3(m.products) == 3 && multi_tiles(m) != 3)));
  RNN_LAUNCH(rnn_multi_bwd_kernel, multi_tiles(m), rnn_x3(m.products), dim3(clsr_cdiv(Hn, 16), ngru + (t4 ? 1 : 0)), stream, m);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_rnn_bwd_multi(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4,
                                  const int* seq_len, int len_stride, int Hn, int T, void* stream) {
  return clsr_rnn_bwd_multi_range(grus, ngru, t4, seq_len, len_stride, Hn, T, 0, T, stream);
}
