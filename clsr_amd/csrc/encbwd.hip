// Fused tail of the backward pass through the sequence encoders of the default CLSR graph (gfx950, fp32 MFMA):
// ONE pass over dPin [Hn*T, 480] -- the gradients of the fused input projection that the backward-through-time launch
// leaves behind -- produces
//   * the seven weight gradients that contract it over the positions (reference: tf.gradients through GRUCell /
//     Time4LSTMCell.call, rnn_cell_implement.py:129-298, clsr.py:160-237):
//       hist^T . dPin                     [40, 480]  every input-side kernel + the bias sums of all encoders
//       hprev1^T . dPin[:, 0:80]          [40, 80]   short_term_intention gates/kernel, hidden rows
//       (hprev1 * r1)^T . dPin[:, 80:120] [40, 40]   short_term_intention candidate/kernel, hidden rows
//       hprev2^T . dPin[:, 120:200], (hprev2 * r2)^T . dPin[:, 200:240]          causal2
//       mprev^T . dPin[:, 240:400]        [40, 160]  time4lstm kernel, hidden rows
//       TT^T . dPin[:, 360:480]           [80, 120]  the four time kernels (block matrix o | tns | tls)
//     (column layout of net.py's fused input projection: GRU blocks first -- short_term_intention, causal2 -- then
//      the Time4LSTM block  kernel | time_kernel_w1 | time_kernel_w2)
//   * d(hist) += dPin . W_x^T             [Hn*T, 40]  (the product that back-propagates into the history rows).
// Round 2 ran these as a multi-job weight-gradient launch (7 168 blocks; every job stages its own slices of dPin: the
// 393 MB are read ~2.1 times, the 33 MB of hist six times) beside a ring-buffered GEMM that reads dPin a third time:
// 618 us + 279 us in the step, the last thing the update phase waits for.  Here a workgroup stages 16 positions of
// dPin and of the seven left operands in LDS once; its four waves own DISJOINT sets of 16-column tiles of dPin (no cross-wave
// reduction of the weight gradients), balanced by MFMA count (51 / 54 / 55 / 51 accumulator tiles); the d(hist) product
// reads the same tile (K split over the waves, one LDS exchange per stage).  Specialised to the default widths
// (D = Du = H = 40, three encoders); anything else keeps the generic kernels (clsr_enc_bwd_fused_supported).
#include "common.h"
#include "clsr_hip.h"
#include <utility>

// diagnosis builds (scripts/build_variant.sh): -DEB_ABL_NOMFMA (no matrix instructions), -DEB_ABL_NODH (no d(hist)),
// -DEB_ABL_NOFETCH (operands of the first stage only)
#ifdef EB_ABL_NOMFMA
#define EB_MFMA(acc, a, b) (acc)[0] += (a) * (b)
#else
#define EB_MFMA(acc, a, b) MFMA4(acc, a, b)
#endif
#define EB_NX 480                 // columns of dPin
#define EB_PS 484                 // LDS row stride of the dPin tile (484 % 64 = 36: rows land on different banks)
#define EB_XW 368                 // left-operand tile: hist@0 | hp1@48 | hp1*r1@96 | mprev@144 | TT@192 (80) | hp2@272 | hp2*r2@320
#define EB_XS 400                // (400 % 64 = 16: the four position rows of a b32 read land on different banks)
#define EB_WS 484                 // LDS row stride of W_x^T (like EB_PS: 16 rows x float4 without bank conflicts)
#define EB_DS 52                  // row stride of the d(hist) exchange area
#define EB_NPT 30                 // 16-column tiles of dPin
#define EB_NXT 23                 // 16-column tiles of the left-operand tile
#define EB_CHUNK (5 * 5 * 256 + 5 * 16)   // partial layout of clsr_dw_reduce_batch (csrc/linear.hip: DW_CHUNK)

struct EncBwdArgs {
  const float* dPin;              // [M, 480]
  const float* hist;              // [M, 40]
  const float* hp1; const float* g1;    // [M, 40], [M, 120] (r | u | c)
  const float* mprev;             // [M, 40]
  const float* TT;                // [M, 80]
  const float* hp2; const float* g2;
  const float* Wt; int Kp;        // packed W_x^T (clsr_pack_batch): 40 out rows, K = 480
  float* dhist;                   // [M, 40], accumulated into
  float* ws[7];                   // partial workspaces of the seven products, gridDim.x partial slots per chunk
  long M;
  // optional (clsr_enc_bwd_fused_fold): what the segmented sums of the history lookups would otherwise add per sorted entry --
  // the long-term branch's d(hist) and the mean / recent-k shares of the history prologue -- is added to d(hist) HERE, where
  // its rows are read and written anyway; the sums then run in their lean form (csrc/segsum.hip)
  const float* dhist2; const float* dmean; const float* drecent; const int* seq_len; int len_stride, T, recent_k;
};

// product p: rows = left-operand tile columns [x0, x0 + K), columns = dPin columns [c0, c0 + N)
__device__ __host__ constexpr int eb_x0(int p) { return p == 0 ? 0 : p == 1 ? 48 : p == 2 ? 96 : p == 3 ? 144 : p == 4 ? 192 : p == 5 ? 272 : 320; }
__device__ __host__ constexpr int eb_K(int p) { return p == 4 ? 80 : 40; }
__device__ __host__ constexpr int eb_c0(int p) { return p == 0 ? 0 : p == 1 ? 0 : p == 2 ? 80 : p == 3 ? 240 : p == 4 ? 360 : p == 5 ? 120 : 200; }
__device__ __host__ constexpr int eb_N(int p) { return p == 0 ? 480 : p == 1 ? 80 : p == 2 ? 40 : p == 3 ? 160 : p == 4 ? 120 : p == 5 ? 80 : 40; }
// product that left-operand tile xt belongs to
__device__ __host__ constexpr int eb_prod(int xt) { return xt < 3 ? 0 : xt < 6 ? 1 : xt < 9 ? 2 : xt < 12 ? 3 : xt < 17 ? 4 : xt < 20 ? 5 : 6; }
// does (left tile xt, dPin tile pt) hold a wanted block?  (a tile that straddles a product's column range counts)
__device__ __host__ constexpr bool eb_need(int xt, int pt) {
  const int p = eb_prod(xt);
  return 16 * pt + 16 > eb_c0(p) && 16 * pt < eb_c0(p) + eb_N(p);
}
// dPin tiles of wave w (weights: 6 accumulator tiles for most, 9 where two GRU products meet, 11 / 8 under the time
// kernels): 51 / 54 / 55 / 51 accumulator tiles
__device__ __host__ constexpr int eb_nt(int w) { return w == 0 ? 6 : 8; }
__device__ __host__ constexpr int eb_tile(int w, int pi) {
  constexpr int t[4][8] = {{22, 23, 24, 0, 1, 2, 0, 0}, {25, 26, 27, 3, 4, 5, 6, 8}, {28, 29, 7, 9, 10, 11, 13, 14},
                           {12, 15, 16, 17, 18, 19, 20, 21}};
  return t[w][pi];
}
// K split of the d(hist) product (12 MFMAs per tile): 8 / 7 / 7 / 8 tiles, so that every wave issues 300-304 MFMAs a stage
__device__ __host__ constexpr int eb_dnt(int w) { return (w == 0 || w == 3) ? 8 : 7; }
__device__ __host__ constexpr int eb_dtile(int w, int k) { return (w == 0 ? 0 : w == 1 ? 8 : w == 2 ? 15 : 22) + k; }
__device__ __host__ constexpr int eb_count(int w) {
  int n = 0;
  for (int pi = 0; pi < eb_nt(w); ++pi)
    for (int xt = 0; xt < EB_NXT; ++xt) n += eb_need(xt, eb_tile(w, pi)) ? 1 : 0;
  return n;
}
__device__ __host__ constexpr int eb_acc(int w, int pi, int xt) {
  int n = 0;
  for (int p = 0; p < eb_nt(w); ++p)
    for (int x = 0; x < EB_NXT; ++x) {
      if (p == pi && x == xt) return n;
      n += eb_need(x, eb_tile(w, p)) ? 1 : 0;
    }
  return n;
}
__device__ __host__ constexpr bool eb_xused(int w, int xt) {
  for (int pi = 0; pi < eb_nt(w); ++pi)
    if (eb_need(xt, eb_tile(w, pi))) return true;
  return false;
}

template <int W>
__device__ __forceinline__ void eb_wave(const EncBwdArgs& a, float* Ps, float* Xs, const float* Wl, float* Dx,
                                        f32x4 (&acc)[eb_count(W)], float (&bsum)[8], const int lane) {
  constexpr int NP = eb_nt(W);
  const int i = lane & 15, g = lane >> 4;
  // ---- weight gradients: contraction over the 16 positions of the stage, 4 per MFMA.  Software pipeline by hand: the
  //      operands of step s + 1 are read from LDS BEFORE the MFMAs of step s issue (one wave per SIMD: nobody else hides
  //      the LDS latency; fully unrolled, the compiler reads all four steps up front and spills)
  float avA[EB_NXT], bvA[NP], avB[EB_NXT], bvB[NP];
  auto load = [&](float (&av)[EB_NXT], float (&bv)[NP], const int s) {
    const float* xs = Xs + (4 * s + g) * EB_XS + i;
    const float* ps = Ps + (4 * s + g) * EB_PS + i;
#pragma unroll
    for (int xt = 0; xt < EB_NXT; ++xt)
      if (eb_xused(W, xt)) av[xt] = xs[16 * xt];
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) bv[pi] = ps[16 * eb_tile(W, pi)];
  };
  auto mfmas = [&](const float (&av)[EB_NXT], const float (&bv)[NP]) {
    int n = 0;
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
#pragma unroll
      for (int xt = 0; xt < EB_NXT; ++xt)
        if (eb_need(xt, eb_tile(W, pi))) { EB_MFMA(acc[n], av[xt], bv[pi]); ++n; }
      bsum[pi] += bv[pi];
    }
  };
  // d(hist) operands of dPin tile kk: lane (j = i, g) holds position j, columns 16 kk + 4 g .. + 3
  // (an offset the compiler cannot see through, fresh every stage: the weight reads are loop invariant and would
  //  otherwise be hoisted out of the stage loop into ~96 VGPRs -- an opaque POINTER would lose its LDS address space)
  int woff = 0;
  asm volatile("" : "+v"(woff));
  const float* Wv = Wl + woff + i * EB_WS + 4 * g;
  const float* Pv = Ps + i * EB_PS + 4 * g;
  struct DhOp { f32x4 b, w[3]; };
  auto dload = [&](DhOp& o, const int k) {
    const int kk = eb_dtile(W, k);
    o.b = ld4(Pv + 16 * kk);
#pragma unroll
    for (int t = 0; t < 3; ++t) o.w[t] = ld4(Wv + 16 * t * EB_WS + 16 * kk);
  };
  load(avA, bvA, 0);
#pragma unroll 1
  for (int s = 0; s < 4; s += 2) {
    load(avB, bvB, s + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(avA, bvA);
    __builtin_amdgcn_sched_barrier(0);
    load(avA, bvA, s == 0 ? 2 : 3);          // (second trip: a harmless re-read, keeps the loop body branch free)
    __builtin_amdgcn_sched_barrier(0);
    mfmas(avB, bvB);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef EB_ABL_NODH
  return;
#endif
  // ---- d(hist)[16 pos, 40] partial over this wave's K range of dPin columns: D[16 features][16 positions]; the same
  //      hand pipeline, and the three accumulators take turns (a dependent MFMA would wait 8 cycles for its input)
  f32x4 dh[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  DhOp oA, oB;
  auto dmfma = [&](const DhOp& o) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t) EB_MFMA(dh[t], o.w[t][r], o.b[r]);
  };
  dload(oA, 0);
#pragma unroll
  for (int k = 0; k < eb_dnt(W); k += 2) {
    if (k + 1 < eb_dnt(W)) dload(oB, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    dmfma(oA);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 2 < eb_dnt(W)) dload(oA, k + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 1 < eb_dnt(W)) dmfma(oB);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) st4(Dx + ((W * 16 + i) * EB_DS) + 16 * t + 4 * g, dh[t]);   // [wave][position][EB_DS]
}

// scatter one wave's accumulator tiles into the partial workspaces (layout of dw_body / clsr_dw_reduce_batch)
// ACCB (speed-mode kernel): the bias sums are row 40 of the hist product -- a row of ones in the left operand tile
template <int W, bool ACCB = false>
__device__ __forceinline__ void eb_store(float* const (&ws)[7], const f32x4 (&acc)[eb_count(W)], const float (&bsum)[8],
                                         const int lane, const int part, const int nparts) {
  const int i = lane & 15, g = lane >> 4;
  int n = 0;
#pragma unroll
  for (int pi = 0; pi < eb_nt(W); ++pi) {
    const int pt = eb_tile(W, pi);
#pragma unroll
    for (int xt = 0; xt < EB_NXT; ++xt) {
      if (!eb_need(xt, pt)) continue;
      const int p = eb_prod(xt);
      const int nl = 16 * pt + i - eb_c0(p);                       // column inside the product
      const int kb = 16 * xt - eb_x0(p) + 4 * g;                   // first of the lane's four rows
      if (nl >= 0 && nl < eb_N(p)) {
        const int nc = nl / 80, nn = nl - nc * 80;
        float* dst = ws[p] + ((long)nc * nparts + part) * EB_CHUNK + (nn >> 4) * 256 + (nn & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kb + r;                                    // (one K chunk: K <= 80)
          if (k < eb_K(p)) dst[((k >> 4) * 5) * 256 + (k & 15) * 16] = acc[n][r];
        }
      }
      if (ACCB && xt == 2 && g == 2) {       // row 40 = 16 * 2 + 4 * 2 + 0 of the hist tiles: sum over the positions
        const int nl0 = 16 * pt + i, nc = nl0 / 80, nn = nl0 - nc * 80;
        ws[0][((long)nc * nparts + part) * EB_CHUNK + 5 * 5 * 256 + nn] = acc[n][0];
      }
      ++n;
    }
    if (!ACCB) {
      // bias sums ride in the partial of the hist product: db[column] = sum over the positions
      const float b = col4_sum(bsum[pi]);
      if (g == 0) {
        const int nl = 16 * pt + i, nc = nl / 80, nn = nl - nc * 80;
        ws[0][((long)nc * nparts + part) * EB_CHUNK + 5 * 5 * 256 + nn] = b;
      }
    }
  }
}

// Buffer resource over rows [row0, row0 + rows) of a row-major fp32 tensor: lanes whose byte offset falls outside read
// zeros / have their store dropped, so the ragged last stage and the idle lanes of a staging pattern need no branches,
// and the per-stage address arithmetic is scalar (the per-lane offsets are loop invariant).
typedef __amdgpu_buffer_rsrc_t eb_rsrc_t;
#define EB_OOB 0x80000000u
__device__ __forceinline__ eb_rsrc_t eb_rsrc(const float* p, long row0, int rows, int stride) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + row0 * stride), 0, (unsigned)(rows * stride * 4), 0x00020000);
}
__device__ __forceinline__ f32x4 eb_ld(eb_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

template <int W>
__device__ __forceinline__ void eb_body(const EncBwdArgs& a, float* lds) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  float* Wl = lds;                                  // [48][EB_WS] packed W_x^T (rows >= 40: zero)
  float* Ps = Wl + 48 * EB_WS;                       // [16][EB_PS]
  float* Xs = Ps + 16 * EB_PS;                      // [16][EB_XS]
  float* Dx = Xs + 16 * EB_XS;                      // [4 waves][16 positions][EB_DS]
  const int lane = threadIdx.x & 63, tid = W * 64 + lane;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  {
    for (int e = tid; e < 48 * (EB_WS / 4); e += 256) {
      const int row = e / (EB_WS / 4), c = e - row * (EB_WS / 4);
      st4(Wl + row * EB_WS + 4 * c, (row < 40 && c < EB_NX / 4) ? ld4(a.Wt + (long)row * a.Kp + 4 * c) : z4);
    }
    for (int e = tid; e < 16 * EB_XS; e += 256) Xs[e] = 0.f;      // (padding columns of the left tile stay zero)
  }
  // Staging pattern of a thread: position row = tid / 16 of the stage, piece c = tid % 16 of that row --
  //   dPin:  float4 c + 15 j (j = 0..7) of the row's 120, lanes c < 15;   every 40-wide operand: float4 c, lanes c < 10
  // (byte offsets computed once; the stage's first row goes into the scalar buffer base)
  const int row = tid >> 4, c = tid & 15;
  const unsigned vP = c < 15 ? (unsigned)(row * EB_NX * 4 + c * 16) : EB_OOB;
  const unsigned v40 = c < 10 ? (unsigned)(row * 160 + c * 16) : EB_OOB;
  const unsigned v80 = c < 10 ? (unsigned)(row * 320 + c * 16) : EB_OOB;
  const unsigned v120 = c < 10 ? (unsigned)(row * 480 + c * 16) : EB_OOB;
  float* const pP = Ps + row * EB_PS + 4 * c;
  float* const pX = Xs + row * EB_XS + 4 * c;
  f32x4 pr[8], xh, x1, xm, xt0, xt1, x2, xr1, xr2, dr = z4;     // dr: the thread's float4 of the d(hist) rows of the stage
  f32x4 d2 = z4, smv = z4, src_ = z4;                           // (fold form: second d(hist), mean / recent rows of the history)
  int slen = 0, stt = 0;
  const bool fold = a.dhist2 != nullptr;
  const long nst = (a.M + 15) >> 4;
  auto fetch = [&](long st) {
    const long m0 = st * 16;
    const int rows = (int)min(16L, a.M - m0);
    const eb_rsrc_t rP = eb_rsrc(a.dPin, m0, rows, EB_NX);
#pragma unroll
    for (int j = 0; j < 8; ++j) pr[j] = eb_ld(rP, vP + 240u * j);
    xh = eb_ld(eb_rsrc(a.hist, m0, rows, 40), v40);
    x1 = eb_ld(eb_rsrc(a.hp1, m0, rows, 40), v40);
    xr1 = eb_ld(eb_rsrc(a.g1, m0, rows, 120), v120);
    xm = eb_ld(eb_rsrc(a.mprev, m0, rows, 40), v40);
    const eb_rsrc_t rT = eb_rsrc(a.TT, m0, rows, 80);
    xt0 = eb_ld(rT, v80);
    xt1 = eb_ld(rT, v80 + 160u);
    x2 = eb_ld(eb_rsrc(a.hp2, m0, rows, 40), v40);
    xr2 = eb_ld(eb_rsrc(a.g2, m0, rows, 120), v120);
    dr = eb_ld(eb_rsrc(a.dhist, m0, rows, 40), v40);
    if (fold) {
      d2 = eb_ld(eb_rsrc(a.dhist2, m0, rows, 40), v40);
      const long m = m0 + row < a.M ? m0 + row : a.M - 1;
      const int h = (int)(m / a.T);
      stt = (int)(m - (long)h * a.T);
      slen = a.seq_len[(long)h * a.len_stride];
      if (c < 10) {
        if (a.dmean) smv = ld4(a.dmean + (long)h * 40 + 4 * c);
        if (a.drecent) src_ = ld4(a.drecent + (long)h * 40 + 4 * c);
      }
    }
  };
  // d(hist) of the thread's piece as the segmented sums would have formed it (reference: clsr.py:145-150,157,173-177: the
  // masked mean / recent-k mean of the history embeddings back-propagate 1 / len resp. 1 / min(len, k) of their gradient)
  auto folded = [&]() {
    f32x4 v = dr;
    if (fold) {
      v += d2;
      if (stt < slen) {
        if (a.dmean) v += smv * (1.0f / (float)slen);
        if (a.drecent && stt >= slen - a.recent_k) v += src_ * (1.0f / (float)(slen < a.recent_k ? slen : a.recent_k));
      }
    }
    return v;
  };
  auto stage = [&]() {     // (rows past the end of the batch were read as zeros)
    if (c < 15) {
#pragma unroll
      for (int j = 0; j < 8; ++j) st4(pP + 60 * j, pr[j]);
    }
    if (c < 10) {
      st4(pX, xh); st4(pX + 48, x1); st4(pX + 96, x1 * xr1); st4(pX + 144, xm);
      st4(pX + 192, xt0); st4(pX + 232, xt1); st4(pX + 272, x2); st4(pX + 320, x2 * xr2);
    }
  };

  constexpr int NA = eb_count(W);
  f32x4 acc[NA];
#pragma unroll
  for (int n = 0; n < NA; ++n) acc[n] = z4;
  float bsum[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bsum[q] = 0.f;

  long st = blockIdx.x;
  if (st < nst) fetch(st);
  f32x4 dprev = z4;                  // d(hist) rows of the PREVIOUS stage, stored one stage late (see below)
  long st_prev = -1;
  auto store_prev = [&]() {
    if (st_prev >= 0)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dprev),
                                             eb_rsrc(a.dhist, st_prev * 16, (int)min(16L, a.M - st_prev * 16), 40), v40, 0, 0);
  };
  for (; st < nst; st += gridDim.x) {
    __syncthreads();                 // the previous stage's tiles / exchange area are free
    stage();
    const f32x4 dcur = folded();     // (fetched one stage ahead, like the operands)
    __syncthreads();
#ifdef EB_ABL_NOFETCH
    if (st == blockIdx.x)
#endif
    if (st + gridDim.x < nst) fetch(st + gridDim.x);      // next stage's operands fly behind the MFMAs below
    // the previous stage's d(hist) rows leave HERE, behind the loads: a store issued at the end of a stage would be
    // the youngest memory operation when the next stage waits for its operands (one counter for loads and stores)
    store_prev();
    eb_wave<W>(a, Ps, Xs, Wl, Dx, acc, bsum, lane);
#ifndef EB_ABL_NODH
    __syncthreads();
    // d(hist)[m0 + row, :] += sum of the four waves' partials: 16 rows x 10 float4
    if (c < 10) {
      const float* dx = Dx + row * EB_DS + 4 * c;
      dprev = dcur + (ld4(dx) + ld4(dx + 16 * EB_DS)) + (ld4(dx + 2 * 16 * EB_DS) + ld4(dx + 3 * 16 * EB_DS));
    }
    st_prev = st;
#endif
  }
  store_prev();
  const int part = blockIdx.x, nparts = gridDim.x;
  eb_store<W>(a.ws, acc, bsum, lane, part, nparts);
}


// every wave runs its own instance of the body: same staging code, its own column range and accumulators
__global__ void __launch_bounds__(256) enc_bwd_fused_kernel(EncBwdArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) eb_body<0>(a, lds);
  else if (wave == 1) eb_body<1>(a, lds);
  else if (wave == 2) eb_body<2>(a, lds);
  else eb_body<3>(a, lds);
}

// ------------------------------------------------------------------------------------ speed mode (bf16 matrix pipe)
// The seven weight gradients of the same tail from the bf16 dPin that the speed mode's backward-through-time launch
// writes, on v_mfma_f32_16x16x32_bf16 (contraction over 32 positions per instruction: A = left-operand tile
// [16 features][32 positions], B = dPin tile [32 positions][16 columns]).  Same tile ownership, accumulators and partial
// layout as the fp32 kernel; the operands are staged TRANSPOSED ([feature][position] bf16 -- the position is the fast
// index across the lanes of a wave, so the 2-byte LDS writes are conflict free and a lane's 8 consecutive positions are
// one ds_read_b128; csrc/hdw.hip) in stages of 64 positions; the left operands are fp32 in memory and rounded to bf16
// on the way (hp * r is formed in fp32 first), products are exact, accumulation fp32.  The bias sums are a row of ones in
// the hist tile.  d(hist) stays with clsr_hgemm_hf32 (its A operand is the ROW-major dPin tile: a second LDS image).
// Replaces clsr_hdw_partial_multi (seven jobs, 5 376 blocks, each staging its own slices of dPin: 480 us in the step).
typedef __bf16 ebh_bf16x8 __attribute__((ext_vector_type(8)));
#define EBH_LD 72                 // bf16 per LDS row: 64 positions + 8 (rows 144 B apart: 16 rows x 16 B without conflicts)
#define EBH_ST 64                 // positions per stage
struct EncBwdHArgs {
  const __bf16* dPin;             // [M, 480] bf16
  const float* hist; const float* hp1; const float* g1; const float* mprev; const float* TT; const float* hp2;
  const float* g2;
  float* ws[7];
  long M;
  const __bf16* Wh; int Kph;      // bf16 image of the packed W_x^T (clsr_pack_batch_bf16 row order) or NULL: no d(hist)
  float* dhist;                   // [M, 40], accumulated into
};

// MFMAs of one 32-position step, unrolled at COMPILE time (accumulator indices must be constants: a loop that the
// optimiser does not fold sends the 220 accumulators to scratch memory)
template <int W, int X, int PI>
__device__ __forceinline__ void ebh_one(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8& av, const ebh_bf16x8 (&bv)[eb_nt(W)]) {
  if constexpr (eb_need(X, eb_tile(W, PI))) {
    constexpr int n = eb_acc(W, PI, X);
    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[PI], acc[n], 0, 0, 0);
  }
}
template <int W, int X, int... PI>
__device__ __forceinline__ void ebh_row(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8& av, const ebh_bf16x8 (&bv)[eb_nt(W)],
                                        std::integer_sequence<int, PI...>) {
  (ebh_one<W, X, PI>(acc, av, bv), ...);
}
template <int W, int X>
__device__ __forceinline__ void ebh_x(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8 (&bv)[eb_nt(W)], const __bf16* xs) {
  if constexpr (eb_xused(W, X)) {
    const ebh_bf16x8 av = *reinterpret_cast<const ebh_bf16x8*>(xs + 16 * X * EBH_LD);
    ebh_row<W, X>(acc, av, bv, std::make_integer_sequence<int, eb_nt(W)>{});
  }
}
template <int W, int... X>
__device__ __forceinline__ void ebh_all(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8 (&bv)[eb_nt(W)], const __bf16* xs,
                                        std::integer_sequence<int, X...>) {
  (ebh_x<W, X>(acc, bv, xs), ...);
}

template <int W>
__device__ __forceinline__ void ebh_body(const EncBwdHArgs& a, __bf16* lds) {
  __bf16* Ps = lds;                                  // [480][EBH_LD]
  __bf16* Xs = Ps + EB_NX * EBH_LD;                  // [EB_XW][EBH_LD]: hist@0 (ones@40) | hp1@48 | hp1*r1@96 | mprev@144 | TT@192 | hp2@272 | hp2*r2@320
  const int lane = threadIdx.x & 63, tid = W * 64 + lane;
  const int i = lane & 15, g = lane >> 4;
  for (int e = tid; e < EB_XW * EBH_LD; e += 256) Xs[e] = (__bf16)0.f;      // (padding rows stay zero)
  // staging plan: lane = position of the stage; wave W takes the 16-byte pieces c = W + 4 q of every row
  ebh_bf16x8 pr[15];
  f32x4 xh[3], x1[3], xr1[3], xm[3], xt[5], x2[3], xr2[3];
  bool valid = false;
  const long nst = (a.M + EBH_ST - 1) / EBH_ST;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](long st) {
    long m = st * EBH_ST + lane;
    valid = m < a.M;
    m = valid ? m : a.M - 1;
    const __bf16* dp = a.dPin + m * EB_NX;
#pragma unroll
    for (int q = 0; q < 15; ++q) pr[q] = *reinterpret_cast<const ebh_bf16x8*>(dp + 8 * (W + 4 * q));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = W + 4 * q < 10 ? W + 4 * q : 0;      // (pieces past the 40 columns: loaded, not written)
      xh[q] = ld4(a.hist + m * 40 + 4 * c);
      x1[q] = ld4(a.hp1 + m * 40 + 4 * c);
      xr1[q] = ld4(a.g1 + m * 120 + 4 * c);
      xm[q] = ld4(a.mprev + m * 40 + 4 * c);
      x2[q] = ld4(a.hp2 + m * 40 + 4 * c);
      xr2[q] = ld4(a.g2 + m * 120 + 4 * c);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) xt[q] = ld4(a.TT + m * 80 + 4 * (W + 4 * q));
  };
  auto put4 = [&](int row0, f32x4 v) {      // four feature rows of this lane's position
    typedef __bf16 h4 __attribute__((ext_vector_type(4)));
    const h4 h = __builtin_convertvector(valid ? v : z4, h4);
#pragma unroll
    for (int e = 0; e < 4; ++e) Xs[(row0 + e) * EBH_LD + lane] = h[e];
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 15; ++q) {
      const int c0 = 8 * (W + 4 * q);
#pragma unroll
      for (int e = 0; e < 8; ++e) Ps[(c0 + e) * EBH_LD + lane] = valid ? pr[q][e] : (__bf16)0.f;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = W + 4 * q;
      if (c < 10) {
        put4(4 * c, xh[q]); put4(48 + 4 * c, x1[q]); put4(96 + 4 * c, x1[q] * xr1[q]); put4(144 + 4 * c, xm[q]);
        put4(272 + 4 * c, x2[q]); put4(320 + 4 * c, x2[q] * xr2[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) put4(192 + 4 * (W + 4 * q), xt[q]);
    if (W == 0) Xs[40 * EBH_LD + lane] = (__bf16)(valid ? 1.f : 0.f);     // the row of ones: bias sums
  };

  constexpr int NA = eb_count(W), NP = eb_nt(W);
  f32x4 acc[NA];
#pragma unroll
  for (int n = 0; n < NA; ++n) acc[n] = z4;

  long st = blockIdx.x;
  if (st < nst) fetch(st);
  for (; st < nst; st += gridDim.x) {
    __syncthreads();                 // the previous stage's tiles are free (first trip: the zero fill is complete)
    stage();
    __syncthreads();
    if (st + gridDim.x < nst) fetch(st + gridDim.x);      // next stage's operands fly behind the MFMAs below
#pragma unroll 1
    for (int s = 0; s < EBH_ST / 32; ++s) {
      const int mo = 32 * s + 8 * g;
      // the wave's dPin tiles stay in registers for the step; the left-operand tiles pass through one at a time (all of
      // them at once: 17 x 4 more registers on top of 220 accumulators and the 152 of the next stage's operands -- spills)
      ebh_bf16x8 bv[NP];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) bv[pi] = *reinterpret_cast<const ebh_bf16x8*>(Ps + (16 * eb_tile(W, pi) + i) * EBH_LD + mo);
      ebh_all<W>(acc, bv, Xs + i * EBH_LD + mo, std::make_integer_sequence<int, EB_NXT>{});
    }
    if (a.Wh) {
      // d(hist)[pos, :] += dPin[pos, :] . W_x^T for the 16 positions 16 W .. 16 W + 15 of the stage: the contraction runs
      // over the 480 COLUMNS here, so the A operand is the row-major dPin (lane (i, g): position i, columns 32 ks + 8 g ..:
      // one 16-byte read of lines the staging loads have just brought in) and B the bf16 W_x^T image (38 KB: cache
      // resident); no LDS.  Replaces the 230 us clsr_hgemm_hf32 launch on the compute stream.
      const long m = st * EBH_ST + 16 * W + i;
      const bool mv = m < a.M;
      const __bf16* ap = a.dPin + (mv ? m : a.M - 1) * EB_NX + 8 * g;
      const __bf16* bp[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int o = 16 * t + i;                                   // out feature (rows >= 40 of the image are zero)
        const int rho = 32 * (o / 32) + 16 * ((o % 8) / 4) + 4 * ((o % 32) / 8) + (o % 4);
        bp[t] = a.Wh + (long)rho * a.Kph + 8 * g;
      }
      f32x4 dh[3] = {z4, z4, z4};
      const ebh_bf16x8 zh = {};
      // (all 15 dPin pieces of the lane in flight at once, the weight pieces stream behind them: one wave per SIMD has
      //  nobody else to hide a load latency per k-step)
      ebh_bf16x8 av[EB_NX / 32];
#pragma unroll
      for (int ks = 0; ks < EB_NX / 32; ++ks) av[ks] = *reinterpret_cast<const ebh_bf16x8*>(ap + 32 * ks);
#pragma unroll
      for (int ks = 0; ks < EB_NX / 32; ++ks) {
        const ebh_bf16x8 x = mv ? av[ks] : zh;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dh[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, *reinterpret_cast<const ebh_bf16x8*>(bp[t] + 32 * ks), dh[t], 0, 0, 0);
      }
      // lane (j = i, g) holds features 16 t + i of the positions 4 g .. 4 g + 3 of the tile
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long mm = st * EBH_ST + 16 * W + 4 * g + r;
          if (mm < a.M && 16 * t + i < 40) a.dhist[mm * 40 + 16 * t + i] += dh[t][r];
        }
    }
  }
  const float nob[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  eb_store<W, true>(a.ws, acc, nob, lane, blockIdx.x, gridDim.x);
}

__global__ void __launch_bounds__(256) enc_bwd_fused_h_kernel(EncBwdHArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsh[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) ebh_body<0>(a, ldsh);
  else if (wave == 1) ebh_body<1>(a, ldsh);
  else if (wave == 2) ebh_body<2>(a, ldsh);
  else ebh_body<3>(a, ldsh);
}

static int ebh_grid(long M) {
  long st = (M + EBH_ST - 1) / EBH_ST;
  return (int)(st < 256 ? st : 256);
}
extern "C" int clsr_enc_bwd_fused_h_parts(long M) { return ebh_grid(M); }
extern "C" long clsr_enc_bwd_fused_h_workspace_floats(long M, int p) {
  if (p < 0 || p > 6) return 0;
  return (long)clsr_cdiv(eb_N(p), 80) * ebh_grid(M) * EB_CHUNK;
}
extern "C" int clsr_enc_bwd_fused_h(const void* dPin_bf16, const float* hist, const float* hprev1, const float* gates1,
                                    const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                    float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                                    float* ws_hp2, float* ws_hp2r, const void* Wt_bf16, int Kph, float* dhist, long M,
                                    void* stream) {
  CLSR_CHECK_ARG(dPin_bf16 && hist && hprev1 && gates1 && mprev && TT && hprev2 && gates2 && M > 0);
  CLSR_CHECK_ARG(!Wt_bf16 || (dhist && Kph >= EB_NX && Kph % 8 == 0));
  CLSR_CHECK_ARG(ws_hist && ws_hp1 && ws_hp1r && ws_mprev && ws_tt && ws_hp2 && ws_hp2r);
  CLSR_CHECK_SUPPORTED(((uintptr_t)dPin_bf16 % 16) == 0 && ((uintptr_t)hist % 16) == 0);
  EncBwdHArgs a = {};
  a.dPin = (const __bf16*)dPin_bf16; a.hist = hist; a.hp1 = hprev1; a.g1 = gates1; a.mprev = mprev; a.TT = TT;
  a.hp2 = hprev2; a.g2 = gates2; a.M = M;
  a.Wh = (const __bf16*)Wt_bf16; a.Kph = Kph; a.dhist = dhist;
  a.ws[0] = ws_hist; a.ws[1] = ws_hp1; a.ws[2] = ws_hp1r; a.ws[3] = ws_mprev; a.ws[4] = ws_tt; a.ws[5] = ws_hp2; a.ws[6] = ws_hp2r;
  const size_t shmem = (size_t)(EB_NX + EB_XW) * EBH_LD * sizeof(__bf16);
  CLSR_HIP(hipFuncSetAttribute((const void*)enc_bwd_fused_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(enc_bwd_fused_h_kernel, dim3(ebh_grid(M)), dim3(256), shmem, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------------------------ "fp32x3" mode (split-bf16 products)
// The seven weight gradients of the same tail from the fp32 dPin, as split-bf16 products on v_mfma_f32_16x16x32_bf16
// (every value = bf16 hi + bf16 lo, product = lo*hi + hi*lo + hi*hi, fp32 accumulation; see csrc/dw3.hip): the fp32-MFMA
// kernel above is bound by the matrix pipe (26 GFLOP at 0.46 of the fp32 peak = 320-350 us in the step, 6x its operand
// traffic at the HBM rate); three bf16 MFMAs do the work of eight fp32 ones at 16x the rate.  Same tile ownership,
// accumulators and partial layout as the other two kernels.  What differs is the LDS image: POSITION-major --
// [32 positions][368 left-operand features | 480 dPin columns] bf16, one image for the hi parts and one for the lo
// parts (2 x 54 KB) -- written with 8-byte stores by threads that walk the rows of the tensors linearly (every global
// load instruction covers whole contiguous row segments; the position-fast mapping of the bf16 kernel above asks for 16
// bytes out of 64 different rows per instruction), and read through ds_read_b64_tr_b16, the LDS transpose read of
// gfx950: the 16 lanes of a group pass the addresses of a [4 positions][16 features] block and every lane receives
// one feature's four positions -- two of them are an MFMA operand (k-slots = positions 4g..4g+3 and 16+4g..16+4g+3 of
// the stage; A and B operands use the same map).  Row stride 1 696 B = 424 dwords = 40 mod 64: the eight rows a
// half-wave reads land on disjoint bank groups.  The bias sums are a row of ones in the hist tile; d(hist) = dPin . W_x^T
// stays a launch of its own (clsr_pgemm3) beside this one.
#define EBX_ST 32
#define EBX_ROW (EB_XW + EB_NX)            // bf16 per position row
#define EBX_RB (EBX_ROW * 2)               // bytes per position row
#define EBX_IMG (EBX_ST * EBX_RB)          // bytes per image
struct EncBwdXArgs {
  const float* dPin;              // [M, 480] fp32
  const float* hist; const float* hp1; const float* g1; const float* mprev; const float* TT; const float* hp2;
  const float* g2;
  float* ws[7];
  long M;
};
typedef short ebx_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char ebx_lds_t;
// MFMA operand: features 16-wide block at byte offset ``off`` of the image rows, this lane's two transpose reads
__device__ __forceinline__ ebh_bf16x8 ebx_tr2(ebx_lds_t* p) {
  const ebx_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) ebx_s16x4*>(p));
  const ebx_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) ebx_s16x4*>(p + 16 * EBX_RB));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(ebh_bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}
// 4 fp32 -> 4 bf16 hi, 4 bf16 lo (two 8-byte words)
__device__ __forceinline__ void ebx_split4(const f32x4& v, unsigned long long& hi, unsigned long long& lo) {
  typedef __bf16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  unsigned h[2], l[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, h2));
    const float ah = __builtin_bit_cast(float, hp << 16), bh = __builtin_bit_cast(float, hp & 0xffff0000u);
    h[p] = hp;
    l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a - ah, b - bh}, h2));
  }
  hi = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
  lo = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
}

// MFMAs of one stage, unrolled at compile time; dPin tiles OUTER (their operands: 8 registers at a time -- all of a wave's
// tiles at once would be 64 on top of 220 accumulators and 100 registers of prefetched operands), the left-operand tiles
// are re-read from LDS for every dPin tile that needs them (4 transpose reads per 3 MFMAs: far below the LDS rate)
// NPC = bf16 pieces per operand: 2 (hi.lo + lo.hi + hi.hi: "fp32x3") or 3 (every piece pair whose indices sum to <= 2: fp32
// accuracy, precision="fp32"; the third image makes 3 x 54 272 = 162 816 bytes of LDS: the CU's 160 KB less one KB)
template <int W, int PI, int X, int NPC>
__device__ __forceinline__ void ebx_one(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8 (&bv)[NPC], ebx_lds_t* xh) {
  if constexpr (eb_need(X, eb_tile(W, PI))) {
    constexpr int n = eb_acc(W, PI, X);
    ebh_bf16x8 av[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) av[i] = ebx_tr2(xh + i * EBX_IMG + 32 * X);
#pragma unroll
    for (int sidx = NPC - 1; sidx >= 0; --sidx)
#pragma unroll
      for (int i = sidx; i >= 0; --i) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[sidx - i], acc[n], 0, 0, 0);
  }
}
template <int W, int PI, int NPC, int... X>
__device__ __forceinline__ void ebx_col(f32x4 (&acc)[eb_count(W)], const ebh_bf16x8 (&bv)[NPC], ebx_lds_t* xh,
                                        std::integer_sequence<int, X...>) {
  (ebx_one<W, PI, X, NPC>(acc, bv, xh), ...);
}
template <int W, int PI, int NPC>
__device__ __forceinline__ void ebx_p(f32x4 (&acc)[eb_count(W)], ebx_lds_t* th) {
  // (an offset the compiler cannot see through: the left-operand reads of different dPin tiles must not be merged into
  //  one set of 23 x 8 live registers)
  int o = 0;
  asm volatile("" : "+v"(o));
  ebh_bf16x8 bv[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) bv[i] = ebx_tr2(th + o + i * EBX_IMG + (EB_XW + 16 * eb_tile(W, PI)) * 2);
  ebx_col<W, PI, NPC>(acc, bv, th + o, std::make_integer_sequence<int, EB_NXT>{});
}
template <int W, int NPC, int... PI>
__device__ __forceinline__ void ebx_all(f32x4 (&acc)[eb_count(W)], ebx_lds_t* th, std::integer_sequence<int, PI...>) {
  (ebx_p<W, PI, NPC>(acc, th), ...);
}

template <int W, int NPC>
__device__ __forceinline__ void ebx_body(const EncBwdXArgs& a, unsigned char* lds) {
  ebx_lds_t* const Lh = (ebx_lds_t*)lds;        // hi image, then the lo image (C-style cast: generic -> LDS address space)
  const int lane = threadIdx.x & 63, tid = W * 64 + lane;
  const int c16 = lane & 15, g = lane >> 4;
  for (int e = tid; e < NPC * EBX_IMG / 16; e += 256) reinterpret_cast<f32x4*>(lds)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // ---- staging plan (stage invariant)
  //  dPin: thread = (position tid >> 3, pieces (tid & 7) + 8 q, q < 15): 8 lanes read 128 contiguous bytes of a row
  //  left operands (40- / 80-wide rows without gaps: a stage's 32 rows are ONE contiguous block, read linearly, piece
  //  lane + 64 i):  wave 0  hp1, r1 -> hp1, hp1 * r1 | wave 1  hp2, r2 | wave 2  hist, TT[0:320] | wave 3  mprev, TT[320:640]
  constexpr int NPP = 15, NXP = 5;
  const int posP = tid >> 3, subP = tid & 7;
  int posX[NXP], colX[NXP], posT[NXP], colT[NXP];
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int p = lane + 64 * i;
    posX[i] = p / 10;
    colX[i] = 4 * (p - posX[i] * 10);
    const int pt = p + (W == 3 ? 320 : 0);
    posT[i] = pt / 20;
    colT[i] = 4 * (pt - posT[i] * 20);
  }
  f32x4 pr[NPP], xa[NXP], xb[NXP];
  const long nst = (a.M + EBX_ST - 1) / EBX_ST;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](long st) {
    const long m0 = st * EBX_ST;
    const float* dp = a.dPin + min(m0 + posP, a.M - 1) * EB_NX + 4 * subP;
#pragma unroll
    for (int q = 0; q < NPP; ++q) pr[q] = ld4(dp + 32 * q);
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const long m = min(m0 + posX[i], a.M - 1), mt = min(m0 + posT[i], a.M - 1);
      if (W == 0) { xa[i] = ld4(a.hp1 + m * 40 + colX[i]); xb[i] = ld4(a.g1 + m * 120 + colX[i]); }
      if (W == 1) { xa[i] = ld4(a.hp2 + m * 40 + colX[i]); xb[i] = ld4(a.g2 + m * 120 + colX[i]); }
      if (W == 2) { xa[i] = ld4(a.hist + m * 40 + colX[i]); xb[i] = ld4(a.TT + mt * 80 + colT[i]); }
      if (W == 3) { xa[i] = ld4(a.mprev + m * 40 + colX[i]); xb[i] = ld4(a.TT + mt * 80 + colT[i]); }
    }
  };
  auto put = [&](int pos, int feat, const f32x4& v, bool ok) {
    unsigned long long hi, lo;
    const f32x4 vv = ok ? v : z4;
    ebx_split4(vv, hi, lo);
    ebx_lds_t* d = Lh + pos * EBX_RB + feat * 2;
    *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d) = hi;
    *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d + EBX_IMG) = lo;
    if constexpr (NPC > 2) {      // third piece: what the first two leave
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned hb = (unsigned)((hi >> (16 * e)) & 0xffffull) << 16, lb = (unsigned)((lo >> (16 * e)) & 0xffffull) << 16;
        r[e] = (vv[e] - __builtin_bit_cast(float, hb)) - __builtin_bit_cast(float, lb);
      }
      unsigned long long r3, dummy;
      ebx_split4(r, r3, dummy);
      *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d + 2 * EBX_IMG) = r3;
    }
  };
  auto stage = [&](long mbase) {
    const bool okP = mbase + posP < a.M;
#pragma unroll
    for (int q = 0; q < NPP; ++q) put(posP, EB_XW + 4 * subP + 32 * q, pr[q], okP);
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const bool ok = mbase + posX[i] < a.M, okt = mbase + posT[i] < a.M;
      if (W == 0) { put(posX[i], 48 + colX[i], xa[i], ok); put(posX[i], 96 + colX[i], xa[i] * xb[i], ok); }
      if (W == 1) { put(posX[i], 272 + colX[i], xa[i], ok); put(posX[i], 320 + colX[i], xa[i] * xb[i], ok); }
      if (W == 2) { put(posX[i], colX[i], xa[i], ok); put(posT[i], 192 + colT[i], xb[i], okt); }
      if (W == 3) { put(posX[i], 144 + colX[i], xa[i], ok); put(posT[i], 192 + colT[i], xb[i], okt); }
    }
    if (W == 1 && lane < EBX_ST)      // the row of ones (feature 40 of the hist tile): bias sums; hi = 1, lo = 0
      *reinterpret_cast<__attribute__((address_space(3))) unsigned short*>(Lh + lane * EBX_RB + 40 * 2) =
          (mbase + lane < a.M) ? (unsigned short)0x3f80 : (unsigned short)0;
  };

  constexpr int NA = eb_count(W), NP = eb_nt(W);
  f32x4 acc[NA];
#pragma unroll
  for (int n = 0; n < NA; ++n) acc[n] = z4;
  // this lane's transpose-read base: row 4 g + (c16 >> 2), 8-byte column piece c16 & 3
  ebx_lds_t* const th = Lh + (4 * g + (c16 >> 2)) * EBX_RB + (c16 & 3) * 8;

  long st = blockIdx.x;
  if (st < nst) fetch(st);
  for (; st < nst; st += gridDim.x) {
    __syncthreads();                 // the previous stage's image is free (first trip: the zero fill is complete)
    stage(st * EBX_ST);
    __syncthreads();
    if (st + gridDim.x < nst) fetch(st + gridDim.x);      // next stage's operands fly behind the MFMAs below
    ebx_all<W, NPC>(acc, th, std::make_integer_sequence<int, NP>{});
  }
  const float nob[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  eb_store<W, true>(a.ws, acc, nob, lane, blockIdx.x, gridDim.x);
}

template <int NPC>
__global__ void __launch_bounds__(256) enc_bwd_fused_x3_kernel(EncBwdXArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsx[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) ebx_body<0, NPC>(a, ldsx);
  else if (wave == 1) ebx_body<1, NPC>(a, ldsx);
  else if (wave == 2) ebx_body<2, NPC>(a, ldsx);
  else ebx_body<3, NPC>(a, ldsx);
}

static int ebx_grid(long M) {
  long st = (M + EBX_ST - 1) / EBX_ST;
  return (int)(st < 256 ? st : 256);
}
extern "C" int clsr_enc_bwd_fused_x3_parts(long M) { return ebx_grid(M); }
extern "C" long clsr_enc_bwd_fused_x3_workspace_floats(long M, int p) {
  if (p < 0 || p > 6) return 0;
  return (long)clsr_cdiv(eb_N(p), 80) * ebx_grid(M) * EB_CHUNK;
}
// the seven weight gradients (+ bias sums) of clsr_enc_bwd_fused as split-bf16 products; d(hist) is NOT part of this launch
static int enc_bwd_fused_xn(int pieces, const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                            const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                            float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                            float* ws_hp2, float* ws_hp2r, long M, void* stream) {
  CLSR_CHECK_ARG(dPin && hist && hprev1 && gates1 && mprev && TT && hprev2 && gates2 && M > 0);
  CLSR_CHECK_ARG(ws_hist && ws_hp1 && ws_hp1r && ws_mprev && ws_tt && ws_hp2 && ws_hp2r);
  CLSR_CHECK_SUPPORTED(((uintptr_t)dPin % 16) == 0 && ((uintptr_t)hist % 16) == 0 && ((uintptr_t)hprev1 % 16) == 0 &&
                       ((uintptr_t)gates1 % 16) == 0 && ((uintptr_t)mprev % 16) == 0 && ((uintptr_t)TT % 16) == 0 &&
                       ((uintptr_t)hprev2 % 16) == 0 && ((uintptr_t)gates2 % 16) == 0);
  EncBwdXArgs a = {};
  a.dPin = dPin; a.hist = hist; a.hp1 = hprev1; a.g1 = gates1; a.mprev = mprev; a.TT = TT; a.hp2 = hprev2; a.g2 = gates2;
  a.M = M;
  a.ws[0] = ws_hist; a.ws[1] = ws_hp1; a.ws[2] = ws_hp1r; a.ws[3] = ws_mprev; a.ws[4] = ws_tt; a.ws[5] = ws_hp2; a.ws[6] = ws_hp2r;
  const size_t shmem = (size_t)pieces * EBX_IMG;
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  if (pieces == 3) {
    CLSR_HIP(hipFuncSetAttribute((const void*)enc_bwd_fused_x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(enc_bwd_fused_x3_kernel<3>, dim3(ebx_grid(M)), dim3(256), shmem, (hipStream_t)stream, a);
  } else {
    CLSR_HIP(hipFuncSetAttribute((const void*)enc_bwd_fused_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(enc_bwd_fused_x3_kernel<2>, dim3(ebx_grid(M)), dim3(256), shmem, (hipStream_t)stream, a);
  }
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_enc_bwd_fused_x3(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                                     const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                     float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                                     float* ws_hp2, float* ws_hp2r, long M, void* stream) {
  return enc_bwd_fused_xn(2, dPin, hist, hprev1, gates1, mprev, TT, hprev2, gates2, ws_hist, ws_hp1, ws_hp1r, ws_mprev, ws_tt,
                          ws_hp2, ws_hp2r, M, stream);
}
// ... with THREE bf16 pieces per operand (fp32 accuracy: precision="fp32"); parts / workspaces as clsr_enc_bwd_fused_x3
extern "C" int clsr_enc_bwd_fused_x6(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                                     const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                     float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                                     float* ws_hp2, float* ws_hp2r, long M, void* stream) {
  return enc_bwd_fused_xn(3, dPin, hist, hprev1, gates1, mprev, TT, hprev2, gates2, ws_hist, ws_hp1, ws_hp1r, ws_mprev, ws_tt,
                          ws_hp2, ws_hp2r, M, stream);
}

static_assert(eb_count(0) + eb_count(1) + eb_count(2) + eb_count(3) == 211, "every wanted block has an owner");
static_assert(eb_count(0) <= 56 && eb_count(1) <= 56 && eb_count(2) <= 56 && eb_count(3) <= 56, "accumulator budget");

static int eb_grid(long M) {
  long st = (M + 15) / 16;
  return (int)(st < 256 ? st : 256);
}

// partial slots per chunk that the launch fills in every workspace (== its grid size)
extern "C" int clsr_enc_bwd_fused_parts(long M) { return eb_grid(M); }
// floats of workspace of product p (0 hist | 1 hp1 | 2 hp1*r1 | 3 mprev | 4 TT | 5 hp2 | 6 hp2*r2)
extern "C" long clsr_enc_bwd_fused_workspace_floats(long M, int p) {
  if (p < 0 || p > 6) return 0;
  return (long)clsr_cdiv(eb_N(p), 80) * eb_grid(M) * EB_CHUNK;
}
// 1 when the default-graph widths apply: D = n = 40, NX = 480 with the column layout of the header comment
extern "C" int clsr_enc_bwd_fused_supported(int D, int n, int NX) { return D == 40 && n == 40 && NX == EB_NX; }

static int enc_bwd_fused_any(const float* dhist2, const float* dmean, const float* drecent, const int* seq_len, int len_stride, int T,
                             int recent_k, const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                                  const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                  const float* Wt, int Kp, float* dhist, float* ws_hist, float* ws_hp1, float* ws_hp1r,
                                  float* ws_mprev, float* ws_tt, float* ws_hp2, float* ws_hp2r, long M, void* stream) {
  CLSR_CHECK_ARG(dPin && hist && hprev1 && gates1 && mprev && TT && hprev2 && gates2 && Wt && dhist && M > 0);
  CLSR_CHECK_ARG(ws_hist && ws_hp1 && ws_hp1r && ws_mprev && ws_tt && ws_hp2 && ws_hp2r);
  CLSR_CHECK_ARG(Kp >= EB_NX && Kp % 4 == 0);
  CLSR_CHECK_SUPPORTED(((uintptr_t)dPin % 16) == 0 && ((uintptr_t)hist % 16) == 0 && ((uintptr_t)dhist % 16) == 0 &&
                       ((uintptr_t)Wt % 16) == 0);
  EncBwdArgs a = {};
  a.dPin = dPin; a.hist = hist; a.hp1 = hprev1; a.g1 = gates1; a.mprev = mprev; a.TT = TT; a.hp2 = hprev2; a.g2 = gates2;
  a.Wt = Wt; a.Kp = Kp; a.dhist = dhist; a.M = M;
  a.dhist2 = dhist2; a.dmean = dmean; a.drecent = drecent; a.seq_len = seq_len; a.len_stride = len_stride; a.T = T; a.recent_k = recent_k;
  a.ws[0] = ws_hist; a.ws[1] = ws_hp1; a.ws[2] = ws_hp1r; a.ws[3] = ws_mprev; a.ws[4] = ws_tt; a.ws[5] = ws_hp2; a.ws[6] = ws_hp2r;
  const size_t shmem = ((size_t)48 * EB_WS + 16 * EB_PS + 16 * EB_XS + 4 * 16 * EB_DS) * sizeof(float);
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  CLSR_HIP(hipFuncSetAttribute((const void*)enc_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(enc_bwd_fused_kernel, dim3(eb_grid(M)), dim3(256), shmem, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_enc_bwd_fused(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                                  const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                  const float* Wt, int Kp, float* dhist, float* ws_hist, float* ws_hp1, float* ws_hp1r,
                                  float* ws_mprev, float* ws_tt, float* ws_hp2, float* ws_hp2r, long M, void* stream) {
  return enc_bwd_fused_any(nullptr, nullptr, nullptr, nullptr, 0, 1, 1, dPin, hist, hprev1, gates1, mprev, TT, hprev2, gates2, Wt, Kp, dhist,
                           ws_hist, ws_hp1, ws_hp1r, ws_mprev, ws_tt, ws_hp2, ws_hp2r, M, stream);
}
// ... with d(hist)[m, :] += dhist2[m, :] + the mean / recent-k shares of row m = (history h, step t) of [Hn, T] positions:
// dmean[h, :] / len_h when t < len_h, drecent[h, :] / min(len_h, recent_k) when len_h - recent_k <= t < len_h (dmean / drecent
// may be NULL; len_h = seq_len[h * len_stride]) -- the terms the segmented sums of the history lookups otherwise add per entry
extern "C" int clsr_enc_bwd_fused_fold(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                                       const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                                       const float* Wt, int Kp, float* dhist, const float* dhist2, const float* dmean,
                                       const float* drecent, const int* seq_len, int len_stride, int T, int recent_k,
                                       float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt, float* ws_hp2,
                                       float* ws_hp2r, long M, void* stream) {
  CLSR_CHECK_ARG(dhist2 && seq_len && T > 0 && recent_k > 0 && M % T == 0);
  CLSR_CHECK_SUPPORTED(((uintptr_t)dhist2 % 16) == 0 && ((uintptr_t)dmean % 16) == 0 && ((uintptr_t)drecent % 16) == 0);
  return enc_bwd_fused_any(dhist2, dmean, drecent, seq_len, len_stride, T, recent_k, dPin, hist, hprev1, gates1, mprev, TT, hprev2, gates2,
                           Wt, Kp, dhist, ws_hist, ws_hp1, ws_hp1r, ws_mprev, ws_tt, ws_hp2, ws_hp2r, M, stream);
}
