// Position-tiled fp32 MFMA linear layers (gfx950): forward / dX ("pgemm") and dW ("pgemm_dw").
//
// Replaces the reference's tf.tensordot / MatMul + BiasAdd sites on the CLSR path:
//   models/base_model.py:664,704 (_fcn_net layers), models/sequential/clsr.py:363 (attention_mat),
//   the input-side projections of GRUCell / Time4LSTMCell (rnn_cell_implement.py:207-231),
// and their gradients.  Exact fp32: v_mfma_f32_16x16x4_f32 is bit-for-bit an fmaf chain.
//
// Orientation ("features x positions"): one MFMA tile is D[16 out-features][16 positions].
//   A operand  = W^T tile from LDS   : lane (i = l&15, g = l>>4) holds Wt[o0 + i][kslot]
//   B operand  = activations         : lane (j = l&15, g)        holds X[pos j][kslot]
//   D          : lane (j, g) holds out features o0 + 4g + {0..3} of position j
// The k-slot -> input-feature map is free as long as A and B agree, so MFMA #r of k-tile kt
// uses feature 16*kt + 4g + r: every lane loads ONE float4 of its position's row per k-tile
// and stores ONE float4 of outputs per out-tile -- the output layout of a layer is exactly
// the input layout of the next one (16-byte vector global accesses, no LDS transposes).
#include <stdlib.h>
#include "common.h"
#include "clsr_hip.h"
#include <cstdlib>

struct PGemmArgs {
  const float* X; int ldx;
  int T; int G;                   // T>0: r = m / T, t = m % T; G>0: xrow = (r / G) * T + t else xrow = m
  const float* Xmul; int ldmul;   // optional multiplier row r (needs T>0 or row-level positions)
  const float* in_scale; const float* in_shift; int in_relu;  // optional per-input-feature affine (+relu)
  const float* Wt; int Kp;        // packed transposed weights [16*ceil(N/16)][ldw], zero padded; Kp = LDS row width
  int ldw;                        // global row stride of Wt (== Kp unless the K range is processed in chunks)
  const float* bias;
  const float* addU; int ldu;     // optional += addU[xrow][n]
  const float* addV; int ldv;     // optional += addV[r][n]
  float* Y; int ldy; int accumulate;
  double* stats;                  // optional per-block partial column sums [gridDim.x][2][N]
  int M, K, N;
  // optional BN+ReLU backward epilogue: v = (ez*e_scale + e_shift > 0) ? v : 0 and the column sums become
  // (sum v, sum v * xhat) with xhat = (ez - e_mean) * e_invstd  (ez = pre-BN activation at [m, n])
  const float* ez; int ldez; const float* e_scale; const float* e_shift; const float* e_mean; const float* e_invstd;
  int no_ring;                    // A/B switch (CLSR_PGEMM_NO_RING): wide-K plain products load one k-tile ahead only
  int no_compact;                 // A/B switch (CLSR_PGEMM_NO_COMPACT): four MFMAs for a last k-tile of 8 features
  // time-range form (clsr_pgemm_range): M = Hn * rm_tc virtual rows, virtual row v is the physical row
  // (v / rm_tc) * rm_T + rm_t0 + v % rm_tc of X and of Y -- the steps [t0, t0 + tc) of every history of [Hn, T, .] tensors
  int rm_tc, rm_T, rm_t0;
};

// virtual -> physical row of the time-range form (identity when rm_tc == 0)
__device__ __forceinline__ int range_row(int v, int rm_tc, int rm_T, int rm_t0) {
  if (rm_tc == 0) return v;
  const int q = (int)((unsigned)v / (unsigned)rm_tc);
  return q * rm_T + rm_t0 + (v - q * rm_tc);
}

template <int OT, bool STATS>
__global__ void __launch_bounds__(256) pgemm_generic_kernel(PGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int ntile_out = (a.N + 15) >> 4;
  const int ot0 = blockIdx.y * OT;
  const int otc = min(OT, ntile_out - ot0);
  const int n0 = ot0 * 16;
  const int KT = (a.K + 15) >> 4;
  const int Kp = a.Kp;
  {  // stage this block's W^T chunk: rows n0 .. n0 + 16*otc
    const float* src = a.Wt + (long)n0 * a.ldw;
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    const int Kq = Kp >> 2, cnt = 16 * otc * Kq;
    for (int e = tid; e < cnt; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      dst[e] = ld4(src + (long)row * a.ldw + 4 * c);
    }
  }
  __syncthreads();

  double dsum[STATS ? OT : 1][4], dsq[STATS ? OT : 1][4];
  if (STATS) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) { dsum[ot][r] = 0.0; dsq[ot][r] = 0.0; }
  }

  const int ntiles = (a.M + 15) >> 4;
  const float* ldsA = lds + (long)j * Kp + 4 * g;  // + ot*16*Kp + kt*16
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int mv = tile * 16 + j;
    const bool valid = mv < a.M;
    const int m = range_row(valid ? mv : 0, a.rm_tc, a.rm_T, a.rm_t0);
    long xrow = m, r = m;
    if (a.T > 0) {
      r = m / a.T;
      if (a.G > 0) xrow = (r / a.G) * a.T + (m - r * a.T);
    }
    const float* xp = a.X + xrow * a.ldx + 4 * g;
    const float* mp = a.Xmul ? a.Xmul + r * a.ldmul + 4 * g : nullptr;

    auto loadB = [&](int kt) -> f32x4 {
      const int kcol = kt * 16 + 4 * g;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid && kcol < a.K) {
        v = ld4(xp + kt * 16);
        if (mp) v *= ld4(mp + kt * 16);
        if (a.in_scale) {
          v = v * ld4(a.in_scale + kcol) + ld4(a.in_shift + kcol);
          if (a.in_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
        }
      }
      return v;
    };

    f32x4 acc[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 bnext = loadB(0);
    for (int kt = 0; kt < KT; ++kt) {
      const f32x4 b = bnext;
      if (kt + 1 < KT) bnext = loadB(kt + 1);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        if (ot < otc) {
          const f32x4 w = ld4(ldsA + (long)ot * 16 * Kp + kt * 16);
          MFMA4(acc[ot], w.x, b.x);
          MFMA4(acc[ot], w.y, b.y);
          MFMA4(acc[ot], w.z, b.z);
          MFMA4(acc[ot], w.w, b.w);
        }
      }
    }

    if (valid) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int n = n0 + ot * 16 + 4 * g;
        if (ot < otc && n < a.N) {
          f32x4 v = acc[ot];
          if (a.bias) v += ld4(a.bias + n);
          if (a.addU) v += ld4(a.addU + xrow * a.ldu + n);
          if (a.addV) v += ld4(a.addV + r * a.ldv + n);
          float* yp = a.Y + (long)m * a.ldy + n;
          if (a.accumulate) v += ld4(yp);
          f32x4 w2 = v;  // second factor of the squared-sum accumulator
          if (a.ez) {
            const f32x4 zz = ld4(a.ez + (long)m * a.ldez + n);
            const f32x4 y = zz * ld4(a.e_scale + n) + ld4(a.e_shift + n);
            v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f;
            v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
            w2 = (zz - ld4(a.e_mean + n)) * ld4(a.e_invstd + n);
          }
          st4(yp, v);
          if (STATS) {
            dsum[ot][0] += v.x; dsq[ot][0] += (double)v.x * w2.x;
            dsum[ot][1] += v.y; dsq[ot][1] += (double)v.y * w2.y;
            dsum[ot][2] += v.z; dsq[ot][2] += (double)v.z * w2.z;
            dsum[ot][3] += v.w; dsq[ot][3] += (double)v.w * w2.w;
          }
        }
      }
    }
  }

  if (STATS) {
    __syncthreads();  // all waves are done with the weight tile; reuse LDS for the block reduction
    double* red = reinterpret_cast<double*>(lds);  // [4 waves][2][OT*16]
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double s = row16_sum_d(dsum[ot][r]);
        const double q = row16_sum_d(dsq[ot][r]);
        if (j == 0) {
          red[(wave * 2 + 0) * (OT * 16) + ot * 16 + 4 * g + r] = s;
          red[(wave * 2 + 1) * (OT * 16) + ot * 16 + 4 * g + r] = q;
        }
      }
    __syncthreads();
    for (int e = tid; e < 2 * 16 * otc; e += 256) {
      const int which = e / (16 * otc), c = e - which * 16 * otc;
      const int n = n0 + c;
      if (n < a.N) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * (OT * 16) + c];
        a.stats[((long)blockIdx.x * 2 + which) * a.N + n] = s;
      }
    }
  }
}

// ---- specialised fast path ------------------------------------------------------------------------
// Same math and the same lane layout as pgemm_generic_kernel, restructured so that the compiler can keep
// loads in flight behind the MFMAs (the generic kernel's optional-feature branches force an s_waitcnt
// after every load):
//   * prologue / epilogue features are template parameters -- no branches inside the tile loop;
//   * out-of-range rows / k-columns are handled with clamped addresses + selects, never with branches;
//   * a wave owns 32 positions (two B operands): every LDS weight read feeds 8 MFMAs;
//   * the activations of k-tile kt+1 are loaded while k-tile kt is multiplied;
//   * XCD-aware tile order: workgroup b runs on XCD b % 8, and every XCD walks ONE contiguous range of
//     position tiles, so the (1 + ngs) rows of a group that re-read the same history rows (xrow) hit in
//     that XCD's L2.
#define PRO_PLAIN 0
#define PRO_MUL 1
#define PRO_AFF 2
#define EPI_NONE 0
#define EPI_UV 1
#define EPI_ACC 2
#define EPI_EZ 4

//   * KTT > 0 (K = 16*KTT exactly covered): ALL k-tiles of a wave-tile are loaded before its first MFMA -- the MFMA
//     chain of a tile then runs back to back instead of waiting for one load latency per k-tile.
template <int OT, int PRO, int EPI, bool STATS, int KTT = 0, bool RING = false>
__global__ void __launch_bounds__(256) pgemm_fast_kernel(PGemmArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int ntile_out = (a.N + 15) >> 4;
  const int ot0 = blockIdx.y * OT;
  const int otc = min(OT, ntile_out - ot0);
  const int n0 = ot0 * 16;
  const int KT = (a.K + 15) >> 4;
  const int Kp = a.Kp;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // COMPACT LAST K-TILE (K % 16 == 8: the 40-wide inputs of this model).  The last k-tile holds 8 features in the lane
  // groups g = 0, 1 and each of its four MFMAs is half empty.  Instead: lane groups 2, 3 take the features 2, 3 / 6, 7 of
  // the tile (the float4 of groups 0, 1 again, components z, w) and the weight tile is staged with those columns moved to k' = 8, 9 / 12, 13
  // -- then the components x, y of every lane cover all eight features and TWO MFMAs do the work of four: 10 instead
  // of 12 MFMAs per out-tile and row at K = 40 (the z, w components meet zero weights and are not issued).
  const bool cmp_last = (a.K & 15) == 8 && !a.no_compact;
  const int kt_last = KT - 1;
  {  // stage this block's W^T chunk (rows n0 .. n0 + 16*otc); missing out-tiles are zero-filled
    const float* src = a.Wt + (long)n0 * a.ldw;
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    const int Kq = Kp >> 2, cnt = 16 * otc * Kq, tot = 16 * OT * Kq;
    for (int e = tid; e < tot; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      f32x4 v = zero4;
      if (e < cnt) {
        const int cl = c - 4 * kt_last;            // chunk inside the last k-tile (0..3) or outside it
        if (cmp_last && cl >= 0 && cl < 4) {
          const f32x4 s4 = ld4(src + (long)row * a.ldw + 4 * (4 * kt_last + (cl & 1)));
          v = cl < 2 ? (f32x4){s4.x, s4.y, 0.f, 0.f} : (f32x4){s4.z, s4.w, 0.f, 0.f};
        } else {
          v = ld4(src + (long)row * a.ldw + 4 * c);
        }
      }
      dst[e] = v;
    }
  }
  __syncthreads();

  f32x4 biasr[OT];
  bool nok[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    const int n = n0 + ot * 16 + 4 * g;
    nok[ot] = n < a.N;
    biasr[ot] = (a.bias && nok[ot]) ? ld4(a.bias + n) : zero4;
  }
  const float relu_lo = a.in_relu ? 0.f : -3.0e38f;

  // column statistics: fp32 per-lane partials over at most STAT_FLUSH wave-tiles (32 values), then a
  // 16-lane reduction added into per-wave DOUBLE accumulators in LDS (behind the weight chunk)
#ifndef CLSR_STAT_FLUSH_N
#define CLSR_STAT_FLUSH_N 8      // (1: every partial goes straight into the double accumulators -- diagnosis builds)
#endif
  constexpr int STAT_FLUSH = CLSR_STAT_FLUSH_N;
  float fsum[STATS ? OT : 1][4], fsq[STATS ? OT : 1][4];
  double* red = reinterpret_cast<double*>(lds + (long)16 * OT * Kp);  // [4 waves][2][OT*16]
  int pending = 0;
  if (STATS) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) { fsum[ot][r] = 0.f; fsq[ot][r] = 0.f; }
    for (int e = tid; e < 4 * 2 * OT * 16; e += 256) red[e] = 0.0;
    __syncthreads();
  }
  auto flush = [&]() {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s_ = row16_sum(fsum[ot][r]);
        const float q_ = row16_sum(fsq[ot][r]);
        if (j == 0) {
          red[(wave * 2 + 0) * (OT * 16) + ot * 16 + 4 * g + r] += (double)s_;
          red[(wave * 2 + 1) * (OT * 16) + ot * 16 + 4 * g + r] += (double)q_;
        }
        fsum[ot][r] = 0.f;
        fsq[ot][r] = 0.f;
      }
  };

  const int ntiles = (a.M + 31) >> 5;
  const int nb = gridDim.x;
  const int nx = nb >= 8 ? 8 : 1;
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int nslots = (nb - xcd + nx - 1) / nx;
  const int chunk = (ntiles + nx - 1) / nx;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const float* ldsA = lds + (long)j * Kp + 4 * g;  // + ot*16*Kp + kt*16

  for (int tile = xcd * chunk + slot * 4 + wave; tile < t_end; tile += nslots * 4) {
    int mrow[2];
    bool valid[2];
    const float* xp[2];
    const float* mp[2];
    long xrow[2], rr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int m = tile * 32 + s * 16 + j;
      valid[s] = m < a.M;
      const int mc = range_row(valid[s] ? m : a.M - 1, a.rm_tc, a.rm_T, a.rm_t0);
      mrow[s] = mc;
      int r = mc, xr = mc;
      if (a.T > 0) {
        r = mc / a.T;
        if (a.G > 0) xr = (r / a.G) * a.T + (mc - r * a.T);
      }
      xrow[s] = xr;
      rr[s] = r;
      xp[s] = a.X + (long)xr * a.ldx;
      mp[s] = PRO == PRO_MUL ? a.Xmul + (long)r * a.ldmul : nullptr;
    }

    // raw loads of one k-tile (nothing consumes them here: the prologue arithmetic is applied one
    // iteration later, so the loads stay in flight behind the MFMAs of the current k-tile)
    struct Raw { f32x4 x0, x1, p0, p1; };
    auto issue = [&](int kt) -> Raw {
      // (compact tile: lane groups 2, 3 re-read the float4 of groups 0, 1 and use its z, w components -- aligned, in range)
      const int kcol = (cmp_last && kt == kt_last && g >= 2) ? kt * 16 + 4 * (g - 2) : kt * 16 + 4 * g;
      const int kc = kcol < a.K ? kcol : 0;
      Raw q;
      q.x0 = ld4(xp[0] + kc);
      q.x1 = ld4(xp[1] + kc);
      if (PRO == PRO_MUL) { q.p0 = ld4(mp[0] + kc); q.p1 = ld4(mp[1] + kc); }
      if (PRO == PRO_AFF) { q.p0 = ld4(a.in_scale + kc); q.p1 = ld4(a.in_shift + kc); }
      return q;
    };
    auto finish = [&](const Raw& q, int kt, f32x4& b0, f32x4& b1) {
      const bool ink = (cmp_last && kt == kt_last) || kt * 16 + 4 * g < a.K;
      f32x4 v0 = q.x0, v1 = q.x1;
      if (PRO == PRO_MUL) { v0 *= q.p0; v1 *= q.p1; }
      if (PRO == PRO_AFF) {
        v0 = v0 * q.p0 + q.p1;
        v1 = v1 * q.p0 + q.p1;
        v0.x = fmaxf(v0.x, relu_lo); v0.y = fmaxf(v0.y, relu_lo); v0.z = fmaxf(v0.z, relu_lo); v0.w = fmaxf(v0.w, relu_lo);
        v1.x = fmaxf(v1.x, relu_lo); v1.y = fmaxf(v1.y, relu_lo); v1.z = fmaxf(v1.z, relu_lo); v1.w = fmaxf(v1.w, relu_lo);
      }
      b0 = ink ? v0 : zero4;
      b1 = ink ? v1 : zero4;
      if (cmp_last && kt == kt_last && g >= 2) { b0.x = b0.z; b0.y = b0.w; b1.x = b1.z; b1.y = b1.w; }
    };

    Raw rawk[KTT > 0 ? KTT : 1];
#pragma unroll
    for (int kt = 0; kt < (KTT > 0 ? KTT : 1); ++kt) rawk[kt] = issue(kt);
    Raw raw = rawk[0];
    // accumulators start from everything that is added to the product (bias, addU + addV, previous Y):
    // these loads are in flight together with the first activation loads and need no extra registers
    f32x4 acc[2][OT];
    f32x4 ezr[(EPI & EPI_EZ) ? 2 : 1][(EPI & EPI_EZ) ? OT : 1];
    {
      constexpr bool UV = (EPI & EPI_UV) != 0, ACC = (EPI & EPI_ACC) != 0;
      f32x4 tu[UV ? 2 : 1][UV ? OT : 1], tv[(UV || ACC) ? 2 : 1][(UV || ACC) ? OT : 1];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int nc = nok[ot] ? n0 + ot * 16 + 4 * g : 0;
          if (UV) { tu[s][ot] = ld4(a.addU + xrow[s] * a.ldu + nc); tv[s][ot] = ld4(a.addV + rr[s] * a.ldv + nc); }
          if (ACC) tv[s][ot] = ld4(a.Y + (long)mrow[s] * a.ldy + nc);
          if (EPI & EPI_EZ) ezr[s][ot] = ld4(a.ez + (long)mrow[s] * a.ldez + nc);
        }
      __builtin_amdgcn_sched_barrier(0);  // all of the loads above are issued before the first one is consumed
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          f32x4 v = biasr[ot];
          if (UV) v += tu[s][ot] + tv[s][ot];
          if (ACC) v += tv[s][ot];
          acc[s][ot] = v;
        }
    }

    auto mfma_block = [&](int kt, const f32x4& b0, const f32x4& b1) {
      // all weight tiles of this k-tile first, then 2*OT independent accumulators per k-slot
      f32x4 wt[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) wt[ot] = ld4(ldsA + (long)ot * 16 * Kp + kt * 16);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].x, b0.x); MFMA4(acc[1][ot], wt[ot].x, b1.x); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].y, b0.y); MFMA4(acc[1][ot], wt[ot].y, b1.y); }
      if (cmp_last && kt == kt_last) return;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].z, b0.z); MFMA4(acc[1][ot], wt[ot].z, b1.z); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].w, b0.w); MFMA4(acc[1][ot], wt[ot].w, b1.w); }
    };
    if (KTT > 0) {
#pragma unroll
      for (int kt = 0; kt < (KTT > 0 ? KTT : 1); ++kt) {
        f32x4 b0, b1;
        finish(rawk[kt], kt, b0, b1);
        mfma_block(kt, b0, b1);
      }
    } else if (RING) {
      // any K, plain operand, FEW positions (a variant of its own: the ring costs 50-80 VGPRs, which the launches
      // with many wave-tiles per SIMD would pay in occupancy -- the 100M-catalogue step got 8 % slower with it
      // everywhere): a ring of four k-tiles in flight (slot = kt % 4, refilled right after it is consumed).
      // With one tile ahead a wide-K product is a chain of K/16 exposed load latencies: the [20480, 200] input of the
      // alpha-gate MLP (640 wave-tiles on 1024 SIMDs: nothing else to switch to) took 80 us for 0.65 GFLOP
      Raw ring[4];
      ring[0] = raw;
#pragma unroll
      for (int d = 1; d < 4; ++d) ring[d] = issue(min(d, KT - 1));
      for (int kt0 = 0; kt0 < KT; kt0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int kt = kt0 + d;
          if (kt < KT) {
            f32x4 b0, b1;
            finish(ring[d], kt, b0, b1);
            ring[d] = issue(min(kt + 4, KT - 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(kt, b0, b1);
          }
        }
      }
    } else {
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 b0, b1;
        finish(raw, kt, b0, b1);
        raw = issue(min(kt + 1, KT - 1));
        __builtin_amdgcn_sched_barrier(0);  // keep the loads above ahead of the MFMA block
        mfma_block(kt, b0, b1);
      }
    }

#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int n = n0 + ot * 16 + 4 * g;
        const int nc = nok[ot] ? n : 0;
        const bool ok = valid[s] && nok[ot];
        f32x4 v = acc[s][ot];
        float* yp = a.Y + (long)mrow[s] * a.ldy + nc;
        f32x4 w2 = v;  // second factor of the squared-sum accumulator
        if (EPI & EPI_EZ) {
          const f32x4 zz = ezr[s][ot];
          const f32x4 y = zz * ld4(a.e_scale + nc) + ld4(a.e_shift + nc);
          v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f;
          v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
          w2 = (zz - ld4(a.e_mean + nc)) * ld4(a.e_invstd + nc);
        }
        // the first attention layer's output (U + V + product epilogue: 65-330 MB, next read a kernel later) is
        // stored non-temporally: streamed through the write-back L2 / Infinity Cache it costs this GEMM 15 %
        // (213 -> 181 us alone) and evicts the history-level operands it re-reads.  Decided per variant at COMPILE
        // time: a run-time flag around the two store flavours gave the gain back, and the other large outputs
        // (dy0, daq, z1: read back by the very next kernel) lose with the same treatment -- all measured A/B
        if (EPI & EPI_UV) { if (ok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yp)); }
        else if (ok) st4(yp, v);
        if (STATS) {
          const f32x4 vm = ok ? v : zero4;
          fsum[ot][0] += vm.x; fsq[ot][0] = fmaf(vm.x, w2.x, fsq[ot][0]);
          fsum[ot][1] += vm.y; fsq[ot][1] = fmaf(vm.y, w2.y, fsq[ot][1]);
          fsum[ot][2] += vm.z; fsq[ot][2] = fmaf(vm.z, w2.z, fsq[ot][2]);
          fsum[ot][3] += vm.w; fsq[ot][3] = fmaf(vm.w, w2.w, fsq[ot][3]);
        }
      }
    }
    if (STATS && ++pending == STAT_FLUSH) {
      flush();
      pending = 0;
    }
  }

  if (STATS) {
    if (pending) flush();
    __syncthreads();
    for (int e = tid; e < 2 * 16 * otc; e += 256) {
      const int which = e / (16 * otc), c = e - which * 16 * otc;
      const int n = n0 + c;
      if (n < a.N) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * (OT * 16) + c];
        a.stats[((long)blockIdx.x * 2 + which) * a.N + n] = s;
      }
    }
  }
}


static int pgemm_grid_x(int M) {
  int ntiles = clsr_cdiv(M, 32);
  int gx = clsr_cdiv(ntiles, 4);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  return gx;
}

extern "C" int clsr_pgemm_stats_parts(int M) { return pgemm_grid_x(M); }

template <typename KernelT>
static int launch_kernel(KernelT kernel, const PGemmArgs& a, dim3 grid, size_t shmem, hipStream_t stream) {
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

template <int OT>
static int launch_pgemm(const PGemmArgs& a, hipStream_t stream) {
  const int ntile_out = clsr_cdiv(a.N, 16);
  dim3 grid(pgemm_grid_x(a.M), clsr_cdiv(ntile_out, OT));
  size_t wbytes = (size_t)16 * OT * a.Kp * sizeof(float);
  size_t sbytes = a.stats ? (size_t)4 * 2 * OT * 16 * sizeof(double) : 0;
  size_t shmem = wbytes + sbytes;
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  const int pro = a.Xmul ? PRO_MUL : (a.in_scale ? PRO_AFF : PRO_PLAIN);
  const int epi = ((a.addU && a.addV) ? EPI_UV : 0) | (a.accumulate ? EPI_ACC : 0) | (a.ez ? EPI_EZ : 0);
  // (the 8-out-tile BN-backward variant would spill registers: it stays on the generic kernel)
  const bool uv_ok = (a.addU != nullptr) == (a.addV != nullptr) && !(OT == 8 && epi == EPI_EZ);
  const bool st = a.stats != nullptr;
  const int KTn = clsr_cdiv(a.K, 16);
  const bool upfront = true;   // (measured against loads one k-tile ahead: all up front wins)
  // K = 128 over many positions (the K-fused time-gate projection [hist | TT] of the Time4LSTM) with all eight k-tiles up
  // front: measured 153 us against 123 us for the one-tile-ahead loop -- opt-in
  const bool k8 = false;
  // few positions, wide K: latency bound (see the kernel); also the skinny d(hist) = dPin . W^T product (K = 480 -> 40
  // columns: 30 k-tiles per wave-tile, 100 -> 132 VGPRs with the ring)
  const bool ring = !a.no_ring && KTn > 5 && (a.M <= 65536 || (OT == 3 && KTn >= 16));
#define CLSR_FAST(P, E, S)                                                                          \
  if (uv_ok && pro == (P) && epi == (E) && st == (S)) {                                             \
    /* (the X * Xmul variant holds twice the operands: with everything in flight it drops to one wave per SIMD and loses) */ \
    if (upfront && (P) != PRO_MUL && KTn == 5) return launch_kernel(pgemm_fast_kernel<OT, P, E, S, 5>, a, grid, shmem, stream); \
    if (upfront && (P) != PRO_MUL && KTn == 3) return launch_kernel(pgemm_fast_kernel<OT, P, E, S, 3>, a, grid, shmem, stream); \
    if (upfront && (P) == PRO_PLAIN && (E) == EPI_NONE && !(S) && KTn == 8 && k8) return launch_kernel(pgemm_fast_kernel<OT, PRO_PLAIN, EPI_NONE, false, 8>, a, grid, shmem, stream); \
    if ((P) == PRO_PLAIN && ring) return launch_kernel(pgemm_fast_kernel<OT, PRO_PLAIN, E, S, 0, true>, a, grid, shmem, stream); \
    return launch_kernel(pgemm_fast_kernel<OT, P, E, S>, a, grid, shmem, stream);                   \
  }
  CLSR_FAST(PRO_PLAIN, EPI_NONE, false)
  CLSR_FAST(PRO_PLAIN, EPI_NONE, true)
  CLSR_FAST(PRO_MUL, EPI_UV, false)
  CLSR_FAST(PRO_MUL, EPI_UV, true)
  CLSR_FAST(PRO_AFF, EPI_NONE, false)
  CLSR_FAST(PRO_AFF, EPI_NONE, true)
  CLSR_FAST(PRO_PLAIN, EPI_ACC, false)
  if (OT < 8) { CLSR_FAST(PRO_PLAIN, EPI_EZ, true) }
#undef CLSR_FAST
  // any other combination of the optional features: generic kernel (same results, branchy)
  if (st) return launch_kernel(pgemm_generic_kernel<OT, true>, a, grid, shmem, stream);
  return launch_kernel(pgemm_generic_kernel<OT, false>, a, grid, shmem, stream);
}

static int pgemm_dispatch(const PGemmArgs& a, hipStream_t s);  // splits the K range when W^T does not fit in LDS

// dy[m, :N] = mask(dY_next[m, :K] . W^T) with mask = (z*scale + shift > 0)  -- the product that
// back-propagates through a linear layer fused with the ReLU + batch-norm backward reduction of the
// layer below: stats receives per-block partial (sum dy, sum dy*xhat) for clsr_bn_bwd_coef.
extern "C" int clsr_pgemm_bnbwd(const float* X, int ldx, const float* Wt, int Kp, float* Y, int ldy,
                                const float* z, int ldz, const float* scale, const float* shift,
                                const float* mean, const float* invstd, double* stats, int M, int K, int N,
                                void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && z && scale && shift && mean && invstd && stats && M > 0 && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && Kp % 4 == 0);
  CLSR_CHECK_ARG(Kp >= 16 * clsr_cdiv(K, 16));
  PGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.Wt = Wt; a.Kp = Kp; a.ldw = Kp; a.Y = Y; a.ldy = ldy; a.stats = stats; a.M = M; a.K = K; a.N = N;
  a.in_relu = 1;
  a.ez = z; a.ldez = ldz; a.e_scale = scale; a.e_shift = shift; a.e_mean = mean; a.e_invstd = invstd;
  return pgemm_dispatch(a, (hipStream_t)stream);
}

// Y[m, :N] (=|+=) f(X)[xrow(m), :K] . W + bias + addU[xrow(m)] + addV[r(m)]
//   f = optional (* Xmul[r(m)]) then optional (x*in_scale + in_shift, relu)
// Wt is the packed transposed weight from clsr_pack_weight (row stride Kp).
// stats (optional): per-block partial column sums / sums of squares of the stored values,
//   [clsr_pgemm_stats_parts(M)][2][N] doubles, summed by clsr_bn_finalize.
extern "C" int clsr_pgemm(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                          const float* in_scale, const float* in_shift, int in_relu, const float* Wt,
                          int Kp, const float* bias, const float* addU, int ldu, const float* addV,
                          int ldv, float* Y, int ldy, int accumulate, double* stats, int M, int K,
                          int N, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && M >= 0 && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && Kp % 4 == 0);
  CLSR_CHECK_ARG(Kp >= 16 * clsr_cdiv(K, 16));
  CLSR_CHECK_ARG(!(Xmul && ldmul % 4) && !(addU && ldu % 4) && !(addV && ldv % 4));
  CLSR_CHECK_ARG(!(in_scale && !in_shift));
  CLSR_CHECK_SUPPORTED(!(Xmul && in_scale));
  CLSR_CHECK_SUPPORTED(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ldx >= K && (!Xmul || ldmul >= K));
  if (M == 0) return CLSR_OK;
  PGemmArgs a;
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.Wt = Wt; a.Kp = Kp; a.ldw = Kp; a.bias = bias; a.addU = addU; a.ldu = ldu; a.addV = addV; a.ldv = ldv;
  a.Y = Y; a.ldy = ldy; a.accumulate = accumulate; a.stats = stats; a.M = M; a.K = K; a.N = N;
  a.ez = nullptr; a.ldez = 0; a.e_scale = a.e_shift = a.e_mean = a.e_invstd = nullptr;
  a.no_ring = 0;
  a.no_compact = 0;
  a.rm_tc = a.rm_T = a.rm_t0 = 0;
  return pgemm_dispatch(a, (hipStream_t)stream);
}

// Y[p, :N] (=|+=) X[p, :K] . W + bias  for the rows p = h * T + t, h < Hn, t0 <= t < t1 of [Hn, T, .] tensors: one
// time range of a batched input projection / of the products that back-propagate dPin (see clsr_rnn_fwd_multi_range)
extern "C" int clsr_pgemm_range(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy,
                                int accumulate, int Hn, int T, int t0, int t1, int K, int N, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && Hn > 0 && T > 0 && 0 <= t0 && t0 < t1 && t1 <= T && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && Kp % 4 == 0);
  CLSR_CHECK_ARG(Kp >= 16 * clsr_cdiv(K, 16));
  CLSR_CHECK_SUPPORTED(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ldx >= K && (long)Hn * T < (1L << 31));
  PGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.Wt = Wt; a.Kp = Kp; a.ldw = Kp; a.bias = bias; a.Y = Y; a.ldy = ldy; a.accumulate = accumulate;
  a.M = Hn * (t1 - t0); a.K = K; a.N = N;
  a.no_ring = 0;
  if (t0 != 0 || t1 != T) { a.rm_tc = t1 - t0; a.rm_T = T; a.rm_t0 = t0; }
  return pgemm_dispatch(a, (hipStream_t)stream);
}

static int pgemm_out_tiles(int N) {
  const int nt = clsr_cdiv(N, 16);
  if (nt <= 3) return 3;
  if (nt <= 5) return 5;
  if (nt <= 8) return 8;
  const int waste5 = clsr_cdiv(nt, 5) * 5 - nt, waste8 = clsr_cdiv(nt, 8) * 8 - nt;
  return waste8 <= waste5 ? 8 : 5;
}
// narrow inputs (K <= 48: three k-tiles, all loaded up front): the five-tile variant keeps more waves in flight and wins
// unless it wastes two or more out-tiles over the eight-tile one (360 columns: 166 us with eight tiles per workgroup)
static int pgemm_out_tiles_k(int N, int K) {
  const int nt = clsr_cdiv(N, 16);
  if (nt > 8 && K <= 48) {
    const int waste5 = clsr_cdiv(nt, 5) * 5 - nt, waste8 = clsr_cdiv(nt, 8) * 8 - nt;
    if (waste5 - waste8 <= 1) return 5;
  }
  return pgemm_out_tiles(N);
}

static int pgemm_dispatch_one(const PGemmArgs& a, hipStream_t s) {
  int ot = pgemm_out_tiles_k(a.N, a.K);
  // the BN-backward epilogue has no 8-out-tile fast variant (register spills): two column chunks of the
  // 5-tile one (blockIdx.y) beat the generic kernel (31 -> ~17 us for the 20 480 x 64 -> 100 logit layer)
  if (ot == 8 && a.ez) ot = 5;
  // few positions (the row-level heads: 20 480 rows = 640 wave-tiles on 1 024 SIMDs): every wave owns ONE tile and its
  // MFMA chain is the kernel's critical path -- three out-tiles per wave and more column chunks (blockIdx.y) instead of
  // five or eight: 3.69 -> 3.65 ms per step (A/B switch CLSR_PGEMM_SMALLM_OT: 0 = off)
  constexpr int small_ot = 3;
  if (small_ot && a.M <= 32768 && ot > small_ot) ot = small_ot;
  if (ot == 3) return launch_pgemm<3>(a, s);
  if (ot == 5) return launch_pgemm<5>(a, s);
  return launch_pgemm<8>(a, s);
}

// ------------------------------------------------------------------------------------ wide-K products
// Y[m, :N] (=|+=) X[m, :K] . W (+ bias) for WIDE inputs (K >= 256, N <= 128: d(hist) = dPin . W_x^T at the 128-wide layer
// sizes of BASELINE configs[4], K = 1 536).  pgemm_fast keeps the whole W^T chunk of a workgroup in LDS and ran such a
// product as a chain of nine launches over K ranges, each re-reading and re-writing the output (4.7 ms of the 13.5 ms
// catalogue step, profiles/r04_catalogue_step_timeline.txt).  Here the K loop is INSIDE the kernel: a workgroup owns 128
// positions x all out-features (X is read once, Y written once), its four waves 64 positions x 64 features each
// (16 accumulator tiles), and 32-wide K stages of W^T and X rows are double-buffered through LDS -- the loads of stage
// s + 1 are in flight while stage s is multiplied, one barrier per stage.  Same orientation and k-slot map as above.
#define PKL_ST 36                 // LDS row stride of a 32-wide stage (floats): 9 x 16 B, conflict-free operand reads
struct PKLArgs { const float* X; int ldx; const float* Wt; int Kp; const float* bias; float* Y; int ldy; int accumulate; int M, K, N, opad; };

__global__ void __launch_bounds__(256, 2) pgemm_kloop_kernel(PKLArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pkl_lds[];
  float (*Ws)[128 * PKL_ST] = reinterpret_cast<float (*)[128 * PKL_ST]>(pkl_lds);
  float (*Xs)[128 * PKL_ST] = reinterpret_cast<float (*)[128 * PKL_ST]>(pkl_lds + 2 * 128 * PKL_ST);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const long p0 = (long)blockIdx.x * 128;
  const int wp = (wave & 1) * 64, wo = (wave >> 1) * 64;      // this wave's positions / out-features inside the tile
  // staging: thread -> (row = tid / 8 + 32 * q, 16-byte chunk tid % 8) of both operands
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  f32x4 wv[4], xv[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = sr + 32 * q;
      const bool kin = k0 + sc < a.K;            // (K % 4 == 0: a chunk is inside or outside)
      wv[q] = (kin && r < a.opad) ? ld4(a.Wt + (long)r * a.Kp + k0 + sc) : f32x4{0.f, 0.f, 0.f, 0.f};
      xv[q] = (kin && p0 + r < a.M) ? ld4(a.X + (p0 + r) * a.ldx + k0 + sc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = sr + 32 * q;
      st4(&Ws[b][r * PKL_ST + sc], wv[q]);
      st4(&Xs[b][r * PKL_ST + sc], xv[q]);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[o][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = (a.K + 31) / 32;
  const int not_ = min(4, (a.opad - wo + 15) / 16);      // out-tiles of this wave that exist (wave-uniform)
  fetch(0);
  stash(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int b = s & 1;
    if (s + 1 < nst) fetch(32 * (s + 1));
    if (not_ > 0) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4 wa[4], xb[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) wa[o] = ld4(&Ws[b][(wo + 16 * o + j) * PKL_ST + 16 * kt + 4 * g]);
#pragma unroll
        for (int p = 0; p < 4; ++p) xb[p] = ld4(&Xs[b][(wp + 16 * p + j) * PKL_ST + 16 * kt + 4 * g]);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (o < not_) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              MFMA4(acc[o][p], wa[o].x, xb[p].x);
              MFMA4(acc[o][p], wa[o].y, xb[p].y);
              MFMA4(acc[o][p], wa[o].z, xb[p].z);
              MFMA4(acc[o][p], wa[o].w, xb[p].w);
            }
          }
        }
      }
    }
    if (s + 1 < nst) stash(b ^ 1);       // (buffer b ^ 1 was last read in stage s - 1: every wave is past that barrier)
    __syncthreads();
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int f0 = wo + 16 * o + 4 * g;
    if (o < not_ && f0 < a.N) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = f32x4{a.bias[f0], a.bias[f0 + 1], a.bias[f0 + 2], a.bias[f0 + 3]};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const long m = p0 + wp + 16 * p + j;
        if (m < a.M) {
          float* y = a.Y + m * a.ldy + f0;
          f32x4 v = acc[o][p] + bv;
          if (a.accumulate) v += ld4(y);
          st4(y, v);
        }
      }
    }
  }
}

static bool pgemm_kloop_ok(const PGemmArgs& a) {
  return a.K >= 256 && a.N <= 128 && a.T == 0 && a.G == 0 && !a.Xmul && !a.in_scale && !a.addU && !a.addV && !a.stats &&
         !a.ez && a.rm_tc == 0 && a.M >= 128 * 256;
}
static int launch_pgemm_kloop(const PGemmArgs& a, hipStream_t s) {
  PKLArgs k;
  k.X = a.X; k.ldx = a.ldx; k.Wt = a.Wt; k.Kp = a.ldw ? a.ldw : a.Kp; k.bias = a.bias; k.Y = a.Y; k.ldy = a.ldy;
  k.accumulate = a.accumulate; k.M = a.M; k.K = a.K; k.N = a.N; k.opad = 16 * clsr_cdiv(a.N, 16);
  const size_t shmem = (size_t)4 * 128 * PKL_ST * sizeof(float);
  CLSR_HIP(hipFuncSetAttribute((const void*)pgemm_kloop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(pgemm_kloop_kernel, dim3(clsr_cdiv(a.M, 128)), dim3(256), shmem, s, k);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// W^T chunk of one workgroup = 16 * OT rows x Kp floats in LDS.  Wide inputs (K > ~290 for 8 out-tiles, the
// 128-wide layer sizes of BASELINE configs[4]) are processed as a chain of launches over K ranges that
// accumulate into Y: the first launch carries the bias / addU / addV terms, the last one the statistics and the
// BN-backward epilogue (they need the final values).
static int pgemm_dispatch(const PGemmArgs& a0, hipStream_t s) {
  PGemmArgs a = a0;
  if (a.ldw == 0) a.ldw = a.Kp;
  int ot = pgemm_out_tiles_k(a.N, a.K);
  if (ot == 8 && a.ez) ot = 5;
  // bytes of LDS for the weight chunk of one launch (A/B switch: CLSR_PGEMM_LDS_KB)
  constexpr size_t budget = (size_t)96 * 1024;
  const int kmax = (int)(budget / ((size_t)16 * ot * sizeof(float))) / 16 * 16;
  if (a.Kp <= kmax) return pgemm_dispatch_one(a, s);
  if (pgemm_kloop_ok(a)) return launch_pgemm_kloop(a, s);
  for (int k0 = 0; k0 < a.K; k0 += kmax) {
    PGemmArgs c = a;
    const bool first = k0 == 0, last = k0 + kmax >= a.K;
    c.K = last ? a.K - k0 : kmax;
    c.Kp = 16 * clsr_cdiv(c.K, 16);
    c.X = a.X + k0;
    c.Wt = a.Wt + k0;
    if (a.Xmul) c.Xmul = a.Xmul + k0;
    if (a.in_scale) { c.in_scale = a.in_scale + k0; c.in_shift = a.in_shift + k0; }
    if (!first) { c.accumulate = 1; c.bias = nullptr; c.addU = nullptr; c.addV = nullptr; }
    if (!last) { c.stats = nullptr; c.ez = nullptr; }
    int rc = pgemm_dispatch_one(c, s);
    if (rc) return rc;
  }
  return CLSR_OK;
}

// ------------------------------------------------------------------------------------ packing
// Wt[o][i] (o < 16*ceil(O/16), i < Ip) = s1 * A[o, i] + s2 * B[o, i], zero outside [O, I], where
// A[o, i] = transposed ? src1[o*ld1 + i] : src1[i*ld1 + o]  (same for B).
__global__ void pack_weight_kernel(const float* __restrict__ src1, int ld1, float s1,
                                   const float* __restrict__ src2, int ld2, float s2, int transposed,
                                   int O, int I, int Ip, int Opad, float* __restrict__ Wt) {
  const int total = Opad * Ip;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int o = e / Ip, i = e - o * Ip;
    float v = 0.f;
    if (o < O && i < I) {
      v = s1 * (transposed ? src1[(long)o * ld1 + i] : src1[(long)i * ld1 + o]);
      if (src2) v += s2 * (transposed ? src2[(long)o * ld2 + i] : src2[(long)i * ld2 + o]);
    }
    Wt[e] = v;
  }
}

extern "C" int clsr_pack_weight(const float* src1, int ld1, float s1, const float* src2, int ld2,
                                float s2, int transposed, int O, int I, int Ip, float* Wt,
                                void* stream) {
  CLSR_CHECK_ARG(src1 && Wt && O > 0 && I > 0 && Ip >= I && Ip % 4 == 0);
  const int Opad = 16 * clsr_cdiv(O, 16);
  int blocks = clsr_cdiv((long)Opad * Ip, 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src1, ld1,
                     s1, src2, ld2, s2, transposed, O, I, Ip, Opad, Wt);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

__global__ void __launch_bounds__(256) pack_batch_kernel(const clsr_pack_desc* __restrict__ descs) {
  const clsr_pack_desc d = descs[blockIdx.y];
  const int total = d.O * d.I;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int o = e / d.I, i = e - o * d.I;
    float v = d.s1 * (d.transposed ? d.src1[(long)o * d.ld1 + i] : d.src1[(long)i * d.ld1 + o]);
    if (d.src2) v += d.s2 * (d.transposed ? d.src2[(long)o * d.ld2 + i] : d.src2[(long)i * d.ld2 + o]);
    d.dst[(long)(d.o0 + o) * d.Kp + d.i0 + i] = v;
  }
}

extern "C" int clsr_sizeof_pack_desc(void) { return (int)sizeof(clsr_pack_desc); }

extern "C" int clsr_pack_batch(const clsr_pack_desc* descs_device, int n, int max_elems, void* stream) {
  CLSR_CHECK_ARG(descs_device && n > 0 && max_elems > 0);
  int bx = clsr_cdiv(max_elems, 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(pack_batch_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, descs_device);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------------------------ dW
// dW[k, n] = sum_m f(X)[xrow(m), k] * dY[m, n];  db[n] = sum_m dY[m, n]
// MFMA tile D[16 k][16 n] += A[16 k][4 m] . B[4 m][16 n]: the reduction (MFMA K) runs over
// positions.  Every wave owns a full 80x80 (5x5 tile) accumulator chunk and a strided subset of
// the positions; per-wave partial chunks go to `partial` and are summed deterministically by
// clsr_dw_reduce (no float atomics on the weights).
#define DW_T CLSR_DW_T
#define DW_CHUNK CLSR_DW_CHUNK

struct DwArgs {
  const void* X; int ldx; int T; int G; const float* Xmul; int ldmul;   // X / dY: fp32, or bf16 (XH / YH variants)
  const float* in_scale; const float* in_shift; int in_relu;
  const void* dY; int ldy;
  float* partial;
  int M, K, N;
  // time-range form: M = Hn * rm_tc virtual rows -> physical rows of X and dY (see PGemmArgs); the launch writes the
  // partials [poff, poff + gx) of the pstride partial slots that each (K, N) chunk of the workspace holds, so that
  // the launches over the ranges of one product fill ONE workspace for ONE reduction
  int rm_tc, rm_T, rm_t0, pstride, poff;
};

// Block = 4 waves, 64 positions per iteration.  The X tile (prologue applied) and the dY tile are
// staged through LDS with coalesced 16-byte global loads (register prefetch of the next tile runs
// behind the MFMAs of the current one); each wave then reduces its own 16 positions:
//   A operand (lane (i,g)) = Xs[16w + 4s + g][16kt + i],  B operand = Ys[16w + 4s + g][16nt + i].
// Row stride 80 floats == 16 (mod 32) banks => the two row groups of a half-wave hit disjoint banks.
#define DW_W (16 * DW_T)
// (bx, gx) = this block's index / the number of blocks along the positions, (by, bz, gz) = its K / N chunk: blockIdx and
// gridDim for a launch of its own, a job's share of the grid in a multi-job launch (dw_multi_kernel)
#define DW_LDS_FLOATS (64 * 2 * DW_W > 2 * DW_CHUNK ? 64 * 2 * DW_W : 2 * DW_CHUNK)
// KTC / NTC > 0: the chunk's tile counts are compile-time constants (single-chunk products of the common shapes): the
// MFMA block then has no scalar compare + branch per tile and schedules as one straight run
template <int MODE, bool XH = false, bool YH = false, int KTC = 0, int NTC = 0>  // 0 plain, 1 X*Xmul[r], 2 relu(X*in_scale + in_shift)
__device__ __forceinline__ void dw_body(const DwArgs& a, float* lds, const int bx, const int gx, const int by,
                                        const int bz, const int gz) {
  float* Xs = lds;
  float* Ys = lds + 64 * DW_W;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, g = lane >> 4;
  const int k0 = by * DW_W, n0 = bz * DW_W;
  const int ktc = KTC ? KTC : min(DW_T, ((a.K + 15) >> 4) - by * DW_T);
  const int ntc = NTC ? NTC : min(DW_T, ((a.N + 15) >> 4) - bz * DW_T);
  const int kmax4 = ((a.K + 3) & ~3) - 4, nmax4 = a.N - 4;

  f32x4 acc[DW_T][DW_T];
#pragma unroll
  for (int kt = 0; kt < DW_T; ++kt)
#pragma unroll
    for (int nt = 0; nt < DW_T; ++nt) acc[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[DW_T];
#pragma unroll
  for (int nt = 0; nt < DW_T; ++nt) bsum[nt] = 0.f;

  // staging assignment: 5 float4 of the X tile and 5 of the dY tile per thread (64 rows x 20 float4 each).
  // Loads are unconditional (clamped addresses) and issued back to back; masks / prologue are applied
  // when the registers are written to LDS, after the MFMAs of the previous tile.
  constexpr int Q = DW_W / 4;       // float4 per tile row
  constexpr int NLD = 64 * Q / 256; // loads per thread per tile
  f32x4 xr[NLD], yr[NLD], mr[MODE == 1 ? NLD : 1];
  unsigned vmask = 0;               // bit j: X piece valid, bit 8+j: dY piece valid
  f32x4 scv[MODE == 2 ? NLD : 1], shv[MODE == 2 ? NLD : 1];
  if (MODE == 2) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + 256 * j;
      const int c4 = idx - (idx / Q) * Q;
      int kcol = k0 + 4 * c4;
      kcol = kcol <= kmax4 ? kcol : kmax4;
      // K % 4 != 0 never happens together with an affine prologue (BN widths are multiples of 4)
      scv[j] = ld4(a.in_scale + kcol);
      shv[j] = ld4(a.in_shift + kcol);
    }
  }
  const int ntiles = (a.M + 63) >> 6;
  auto fetch = [&](int tile) {
    vmask = 0;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx / Q, c4 = idx - row * Q;
      int m = tile * 64 + row;
      const bool mv = (tile < ntiles) && (m < a.M);
      m = range_row(m < a.M ? m : a.M - 1, a.rm_tc, a.rm_T, a.rm_t0);
      int kcol = k0 + 4 * c4, ncol = n0 + 4 * c4;
      if (mv && kcol <= kmax4) vmask |= 1u << j;
      if (mv && ncol <= nmax4) vmask |= 1u << (8 + j);
      kcol = kcol <= kmax4 ? kcol : kmax4;
      ncol = ncol <= nmax4 ? ncol : nmax4;
      long xrow = m, r = m;
      if (a.T > 0) {
        const unsigned rr = (unsigned)m / (unsigned)a.T;
        r = rr;
        if (a.G > 0) xrow = (long)(rr / (unsigned)a.G) * a.T + (m - (int)rr * a.T);
      }
      xr[j] = load4e<XH>(a.X, xrow * a.ldx + kcol);
      if (MODE == 1) mr[j] = ld4(a.Xmul + r * a.ldmul + kcol);
      yr[j] = load4e<YH>(a.dY, (long)m * a.ldy + ncol);
    }
  };
  auto stage = [&]() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + 256 * j;
      f32x4 xv = xr[j];
      if (MODE == 1) xv *= mr[j];
      if (MODE == 2) {
        xv = xv * scv[j] + shv[j];
        if (a.in_relu) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
      }
      st4(Xs + 4 * idx, (vmask >> j) & 1u ? xv : z);
      st4(Ys + 4 * idx, (vmask >> (8 + j)) & 1u ? yr[j] : z);
    }
  };

  fetch(bx);
  for (int tile = bx; tile < ntiles; tile += gx) {
    __syncthreads();  // previous tile fully consumed
    stage();
    __syncthreads();
    fetch(tile + gx);  // global loads of the next tile fly behind the MFMAs below
    const float* xw = Xs + (16 * wave + g) * DW_W + i;
    const float* yw = Ys + (16 * wave + g) * DW_W + i;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float av[DW_T], bv[DW_T];
#pragma unroll
      for (int t = 0; t < DW_T; ++t) {
        av[t] = xw[(4 * s) * DW_W + 16 * t];
        bv[t] = yw[(4 * s) * DW_W + 16 * t];
      }
#pragma unroll
      for (int kt = 0; kt < DW_T; ++kt)
        if (kt < ktc) {
#pragma unroll
          for (int nt = 0; nt < DW_T; ++nt)
            if (nt < ntc) MFMA4(acc[kt][nt], av[kt], bv[nt]);
        }
#pragma unroll
      for (int nt = 0; nt < DW_T; ++nt) bsum[nt] += bv[nt];
    }
  }
#pragma unroll
  for (int nt = 0; nt < DW_T; ++nt) bsum[nt] = col4_sum(bsum[nt]);
  __syncthreads();  // staging buffers are dead from here on; reuse the LDS for the block reduction
  float* red = lds;

  // block-level reduction through LDS: waves 2,3 -> 0,1 ; wave 1 -> 0 ; wave 0 writes the partial
  auto store_to = [&](float* dst) {
#pragma unroll
    for (int kt = 0; kt < DW_T; ++kt)
#pragma unroll
      for (int nt = 0; nt < DW_T; ++nt) {
        float* t = dst + (kt * DW_T + nt) * 256 + (4 * g) * 16 + i;
        const f32x4 v = acc[kt][nt];
        t[0] = v.x; t[16] = v.y; t[32] = v.z; t[48] = v.w;
      }
    if (g == 0) {
#pragma unroll
      for (int nt = 0; nt < DW_T; ++nt) dst[DW_T * DW_T * 256 + nt * 16 + i] = bsum[nt];
    }
  };
  auto add_from = [&](const float* src) {
#pragma unroll
    for (int kt = 0; kt < DW_T; ++kt)
#pragma unroll
      for (int nt = 0; nt < DW_T; ++nt) {
        const float* t = src + (kt * DW_T + nt) * 256 + (4 * g) * 16 + i;
        acc[kt][nt] += (f32x4){t[0], t[16], t[32], t[48]};
      }
#pragma unroll
    for (int nt = 0; nt < DW_T; ++nt) bsum[nt] += src[DW_T * DW_T * 256 + nt * 16 + i];
  };
  if (wave >= 2) store_to(red + (wave - 2) * DW_CHUNK);
  __syncthreads();
  if (wave < 2) add_from(red + wave * DW_CHUNK);
  __syncthreads();
  if (wave == 1) store_to(red);
  __syncthreads();
  if (wave == 0) {
    add_from(red);
    const long chunk = (long)by * gz + bz;
    store_to(a.partial + (chunk * a.pstride + a.poff + bx) * DW_CHUNK);
  }
}

template <int MODE, bool XH = false, bool YH = false, int KTC = 0, int NTC = 0>
__global__ void __launch_bounds__(256) pgemm_dw_kernel(DwArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[DW_LDS_FLOATS];
  dw_body<MODE, XH, YH, KTC, NTC>(a, lds, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several weight gradients in ONE launch: job j owns the blocks [first[j], first[j+1]) of the 1-D grid.  The hidden-to-
// hidden / time-feature gradients of the sequence encoders are ~6 small products over the same Hn*T positions (30-100 us
// each, latency bound at 512 blocks): one after the other they were 350 us at the end of the backward pass, side by side
// they share the machine (and the slices of dPin they all read).
#define DWM_MAX 12
struct DwMultiArgs {
  DwArgs d[DWM_MAX];
  int first[DWM_MAX + 1];
  int gx[DWM_MAX];
  short nch[DWM_MAX];
  short mode[DWM_MAX];
  int n;
  int spec;    // 0: generic body only (CLSR_DW_GENERIC)
};
__global__ void __launch_bounds__(256) dw_multi_kernel(DwMultiArgs m) {
  __shared__ __attribute__((aligned(16))) float lds[DW_LDS_FLOATS];
  int j = 0;
  while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
  const int local = blockIdx.x - m.first[j];
  const int gx = m.gx[j], nch = m.nch[j];
  const int bx = local % gx, c = local / gx;
  const int by = c / nch, bz = c - by * nch;
  // this block's tile counts (uniform): the plain products of the common widths (K = 40 -> 3 tiles, full 80-wide N
  // chunks -> 5: the input-side / hidden-to-hidden gradients of the encoders) run the compile-time-count body
  const int ktc = min(DW_T, ((m.d[j].K + 15) >> 4) - by * DW_T), ntc = min(DW_T, ((m.d[j].N + 15) >> 4) - bz * DW_T);
  if (m.mode[j] == 0) {
    if (m.spec && ktc == 3 && ntc == 5) dw_body<0, false, false, 3, 5>(m.d[j], lds, bx, gx, by, bz, nch);
    else if (m.spec && ktc == 5 && ntc == 5) dw_body<0, false, false, 5, 5>(m.d[j], lds, bx, gx, by, bz, nch);
    else if (m.spec > 1 && ktc == 5 && ntc == 3) dw_body<0, false, false, 5, 3>(m.d[j], lds, bx, gx, by, bz, nch);
    else dw_body<0>(m.d[j], lds, bx, gx, by, bz, nch);
  } else if (m.mode[j] == 1) {
    if (m.spec > 1 && ktc == 3 && ntc == 3) dw_body<1, false, false, 3, 3>(m.d[j], lds, bx, gx, by, bz, nch);
    else dw_body<1>(m.d[j], lds, bx, gx, by, bz, nch);
  } else dw_body<2>(m.d[j], lds, bx, gx, by, bz, nch);
}

// dW[k*ldw + n] (=|+=) scale * sum_p partial[...]; db[n] likewise (from K-chunk 0).
// block = 64 output elements x 4 partial sub-ranges, combined through LDS.
__global__ void __launch_bounds__(1024) dw_reduce_kernel(const float* __restrict__ partial, int nparts, int K,
                                                        int N, int kchunks, int nchunks, float scale,
                                                        float* __restrict__ dW, int ldw,
                                                        float* __restrict__ db, int accumulate) {
  __shared__ float red[16][64];
  const int total = K * N + (db ? N : 0);
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  float s = 0.f;
  float* o = nullptr;
  if (e < total) {
    const float* p;
    if (e < K * N) {
      const int k = e / N, n = e - k * N;
      const int kc = k / (16 * DW_T), nc = n / (16 * DW_T);
      const int kk = k - kc * 16 * DW_T, nn = n - nc * 16 * DW_T;
      const long off = ((kk >> 4) * DW_T + (nn >> 4)) * 256 + (kk & 15) * 16 + (nn & 15);
      p = partial + ((long)(kc * nchunks + nc) * nparts) * DW_CHUNK + off;
      o = dW + (long)k * ldw + n;
    } else {
      const int n = e - K * N;
      const int nc = n / (16 * DW_T), nn = n - nc * 16 * DW_T;
      p = partial + ((long)nc * nparts) * DW_CHUNK + DW_T * DW_T * 256 + nn;
      o = db + n;
    }
    for (int w = sub; w < nparts; w += 16) s += p[(long)w * DW_CHUNK];
  }
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && e < total) {
    s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += red[u][threadIdx.x];
    s *= scale;
    *o = accumulate ? *o + s : s;
  }
}

static int dw_grid_x(int M) {
  int tiles = clsr_cdiv(M, 64);
  int gx = clsr_cdiv(tiles, 4);  // >= 4 position tiles per block before adding blocks
  constexpr int cap = 512;   // blocks per chunk of the fp32-MFMA kernels (384: +40 us per exact-mode step; the bf16 kernels use 384, csrc/hdw.hip)
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return gx;
}

// floats of workspace clsr_pgemm_dw needs
extern "C" long clsr_pgemm_dw_workspace_floats(int M, int K, int N) {
  const long chunks = (long)clsr_cdiv(K, 16 * DW_T) * clsr_cdiv(N, 16 * DW_T);
  return chunks * dw_grid_x(M) * DW_CHUNK;
}

static int dw_launch_partial(const void* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                             const float* in_scale, const float* in_shift, int in_relu, const void* dY, int ldy,
                             int M, int K, int N, float* workspace, hipStream_t s, int* gx_out, int x_bf16 = 0,
                             int dy_bf16 = 0) {
  CLSR_CHECK_ARG(X && dY && workspace && M >= 0 && K > 0 && N > 0);
  CLSR_CHECK_ARG(!(in_scale && !in_shift));
  // 16-byte staging loads: rows must be float4 addressable up to roundup(K,4) / N
  CLSR_CHECK_SUPPORTED(N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= ((K + 3) & ~3) &&
                       (!Xmul || (ldmul % 4 == 0 && ldmul >= ((K + 3) & ~3))) &&
                       ((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0);
  DwArgs a;
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.dY = dY; a.ldy = ldy; a.partial = workspace; a.M = M; a.K = K; a.N = N;
  const int kch = clsr_cdiv(K, 16 * DW_T), nch = clsr_cdiv(N, 16 * DW_T);
  const int gx = dw_grid_x(M);
  a.rm_tc = a.rm_T = a.rm_t0 = 0; a.pstride = gx; a.poff = 0;
  CLSR_CHECK_SUPPORTED(!(Xmul && in_scale) && !(in_scale && K % 4));
  if (x_bf16 || dy_bf16) {
    // speed mode (csrc/hgemm.hip): bf16 activations / gradients, fp32 MFMA accumulation as in the fp32 mode.
    // Built combinations: relu(bn(X bf16)) x dY bf16 (second attention layer), (X fp32 * Xmul) x dY bf16 (first layer)
    CLSR_CHECK_SUPPORTED(dy_bf16 && ((x_bf16 && in_scale) || (!x_bf16 && Xmul)));
    if (x_bf16) hipLaunchKernelGGL((pgemm_dw_kernel<2, true, true>), dim3(gx, kch, nch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pgemm_dw_kernel<1, false, true>), dim3(gx, kch, nch), dim3(256), 0, s, a);
    CLSR_CHECK_LAUNCH();
    *gx_out = gx;
    return CLSR_OK;
  }
  const int ktn = clsr_cdiv(K, 16), ntn = clsr_cdiv(N, 16);
  const bool one = kch == 1 && nch == 1;
#define DW_SPEC(MD, KT_, NT_)                                                                                          \
  if (one && ktn == KT_ && ntn == NT_) {                                                                               \
    hipLaunchKernelGGL((pgemm_dw_kernel<MD, false, false, KT_, NT_>), dim3(gx, 1, 1), dim3(256), 0, s, a);            \
    CLSR_CHECK_LAUNCH();                                                                                               \
    *gx_out = gx;                                                                                                      \
    return CLSR_OK;                                                                                                    \
  }
  if (Xmul) { DW_SPEC(1, 5, 5) DW_SPEC(1, 3, 3) }
  else if (in_scale) { DW_SPEC(2, 5, 3) DW_SPEC(2, 5, 5) }
  else { DW_SPEC(0, 5, 5) DW_SPEC(0, 3, 5) DW_SPEC(0, 3, 3) DW_SPEC(0, 5, 3) }
#undef DW_SPEC
  if (Xmul) hipLaunchKernelGGL(pgemm_dw_kernel<1>, dim3(gx, kch, nch), dim3(256), 0, s, a);
  else if (in_scale) hipLaunchKernelGGL(pgemm_dw_kernel<2>, dim3(gx, kch, nch), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(pgemm_dw_kernel<0>, dim3(gx, kch, nch), dim3(256), 0, s, a);
  CLSR_CHECK_LAUNCH();
  *gx_out = gx;
  return CLSR_OK;
}

// n weight-gradient partial products in one launch (same partial layout per job as clsr_pgemm_dw_partial; fp32 operands)
extern "C" int clsr_sizeof_dwjob(void) { return (int)sizeof(clsr_dwjob); }

extern "C" int clsr_pgemm_dw_partial_multi(const clsr_dwjob* jobs, int n, void* stream) {
  CLSR_CHECK_ARG(jobs && n > 0 && n <= DWM_MAX);
  DwMultiArgs m;
  m.n = n;
  int total = 0;
  for (int j = 0; j < n; ++j) {
    const clsr_dwjob& q = jobs[j];
    CLSR_CHECK_ARG(q.X && q.dY && q.workspace && q.M > 0 && q.K > 0 && q.N > 0 && !(q.in_scale && !q.in_shift));
    CLSR_CHECK_SUPPORTED(!q.x_bf16 && !q.dy_bf16);
    CLSR_CHECK_SUPPORTED(q.N % 4 == 0 && q.ldx % 4 == 0 && q.ldy % 4 == 0 && q.ldx >= ((q.K + 3) & ~3) &&
                         (!q.Xmul || (q.ldmul % 4 == 0 && q.ldmul >= ((q.K + 3) & ~3))) &&
                         ((uintptr_t)q.X % 16) == 0 && ((uintptr_t)q.dY % 16) == 0);
    CLSR_CHECK_SUPPORTED(!(q.Xmul && q.in_scale) && !(q.in_scale && q.K % 4));
    DwArgs& a = m.d[j];
    a.X = q.X; a.ldx = q.ldx; a.T = q.T; a.G = q.G; a.Xmul = q.Xmul; a.ldmul = q.ldmul;
    a.in_scale = q.in_scale; a.in_shift = q.in_shift; a.in_relu = q.in_relu;
    a.dY = q.dY; a.ldy = q.ldy; a.partial = q.workspace; a.M = q.M; a.K = q.K; a.N = q.N;
    const int kch = clsr_cdiv(q.K, 16 * DW_T), nch = clsr_cdiv(q.N, 16 * DW_T);
    m.first[j] = total;
    m.gx[j] = q.pgx > 0 ? q.pgx : dw_grid_x(q.M);
    CLSR_CHECK_ARG(q.rm_tc >= 0 && (q.rm_tc == 0 || (q.rm_T >= q.rm_t0 + q.rm_tc && q.rm_t0 >= 0 && q.M % q.rm_tc == 0 && !q.T)));
    CLSR_CHECK_ARG(q.pstride == 0 || (q.poff >= 0 && q.poff + m.gx[j] <= q.pstride));
    a.rm_tc = q.rm_tc; a.rm_T = q.rm_T; a.rm_t0 = q.rm_t0;
    a.pstride = q.pstride > 0 ? q.pstride : m.gx[j]; a.poff = q.pstride > 0 ? q.poff : 0;
    m.nch[j] = (short)nch;
    m.mode[j] = (short)(q.Xmul ? 1 : (q.in_scale ? 2 : 0));
    total += m.gx[j] * kch * nch;
  }
  m.first[n] = total;
  m.spec = 2;
  hipLaunchKernelGGL(dw_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, m);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_pgemm_dw(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                             const float* in_scale, const float* in_shift, int in_relu,
                             const float* dY, int ldy, int M, int K, int N, float scale, float* dW,
                             int ldw, float* db, int accumulate, float* workspace, void* stream) {
  CLSR_CHECK_ARG(dW && ldw >= N);
  hipStream_t s = (hipStream_t)stream;
  int gx = 0;
  int rc = dw_launch_partial(X, ldx, T, G, Xmul, ldmul, in_scale, in_shift, in_relu, dY, ldy, M, K, N, workspace, s,
                             &gx);
  if (rc) return rc;
  const int kch = clsr_cdiv(K, 16 * DW_T), nch = clsr_cdiv(N, 16 * DW_T);
  const int total = K * N + (db ? N : 0);
  hipLaunchKernelGGL(dw_reduce_kernel, dim3(clsr_cdiv(total, 64)), dim3(1024), 0, s, workspace, gx, K, N, kch,
                     nch, scale, dW, ldw, db, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Deferred form: only the per-block partial chunks are produced (own workspace per call); the weight gradients
// of a whole backward pass are then reduced by ONE clsr_dw_reduce_batch launch (blockIdx.y = descriptor) instead
// of ~20 small dependent launches.  Returns the number of partials per chunk (for the descriptor).
extern "C" int clsr_pgemm_dw_partial(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                                     const float* in_scale, const float* in_shift, int in_relu,
                                     const float* dY, int ldy, int M, int K, int N, float* workspace,
                                     void* stream) {
  int gx = 0;
  return dw_launch_partial(X, ldx, T, G, Xmul, ldmul, in_scale, in_shift, in_relu, dY, ldy, M, K, N, workspace,
                           (hipStream_t)stream, &gx);
}

// speed mode: X and / or dY stored as bf16 (see dw_launch_partial for the built combinations)
extern "C" int clsr_pgemm_dw_partial_h(const void* X, int x_bf16, int ldx, int T, int G, const float* Xmul, int ldmul,
                                       const float* in_scale, const float* in_shift, int in_relu,
                                       const void* dY, int dy_bf16, int ldy, int M, int K, int N, float* workspace,
                                       void* stream) {
  int gx = 0;
  return dw_launch_partial(X, ldx, T, G, Xmul, ldmul, in_scale, in_shift, in_relu, dY, ldy, M, K, N, workspace,
                           (hipStream_t)stream, &gx, x_bf16, dy_bf16);
}

extern "C" int clsr_pgemm_dw_parts(int M) { return dw_grid_x(M); }
extern "C" int clsr_sizeof_dw_desc(void) { return (int)sizeof(clsr_dw_desc); }

// Block = 64 consecutive floats of the partial layout (a quarter of one 16 x 16 tile: every wave load is one
// contiguous 256-byte segment per partial) x 16 waves that split the partials; the tail blocks of a descriptor
// sum its bias slots.  Work items beyond gridDim.x are picked up by a block-stride loop.
__global__ void __launch_bounds__(1024) dw_reduce_batch_kernel(const clsr_dw_desc* __restrict__ descs) {
  __shared__ float red[16][64];
  const clsr_dw_desc d = descs[blockIdx.y];
  const int K = d.K, N = d.N;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int ktiles = (K + 15) >> 4, ntiles = (N + 15) >> 4;
  const int nchunks = (N + 16 * DW_T - 1) / (16 * DW_T);
  const int wq = ktiles * ntiles * 4;
  const int items = wq + (d.db ? (N + 63) / 64 : 0);
  for (int it = blockIdx.x; it < items; it += gridDim.x) {   // block-uniform
    float s = 0.f;
    float* o = nullptr;
    const float* p = nullptr;
    const float* p2 = nullptr;
    if (it < wq) {
      const int tile = it >> 2, q = it & 3;
      const int ktg = tile / ntiles, ntg = tile - ktg * ntiles;
      const int kc = ktg / DW_T, kt = ktg - kc * DW_T, nc = ntg / DW_T, nt = ntg - nc * DW_T;
      const int w = q * 64 + lane;
      const int k = ktg * 16 + (w >> 4), n = ntg * 16 + (w & 15);
      if (k < K && n < N) {
        p = d.partial + ((long)(kc * nchunks + nc) * d.nparts) * DW_CHUNK + (kt * DW_T + nt) * 256 + w;
        if (d.partial2) p2 = d.partial2 + ((long)(kc * nchunks + nc) * d.nparts2) * DW_CHUNK + (kt * DW_T + nt) * 256 + w;
        o = d.dW + (long)k * d.ldw + n;
      }
    } else {
      const int n = (it - wq) * 64 + lane;
      if (n < N) {
        const int nc = n / (16 * DW_T), nn = n - nc * 16 * DW_T;
        p = d.partial + ((long)nc * d.nparts) * DW_CHUNK + DW_T * DW_T * 256 + nn;
        o = d.db + n;
      }
    }
    if (p) {
#pragma unroll 8
      for (int w = sub; w < d.nparts; w += 16) s += p[(long)w * DW_CHUNK];
      s *= d.scale;
    }
    if (p2) {
      float s2 = 0.f;
#pragma unroll 8
      for (int w = sub; w < d.nparts2; w += 16) s2 += p2[(long)w * DW_CHUNK];
      s += d.scale2 * s2;
    }
    red[sub][lane] = s;
    __syncthreads();
    if (sub == 0 && o) {
      s = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += red[u][lane];
      *o = d.accumulate ? *o + s : s;
    }
    __syncthreads();
  }
}

extern "C" int clsr_dw_reduce_batch(const clsr_dw_desc* descs_device, int n, int max_outputs, void* stream) {
  CLSR_CHECK_ARG(descs_device && n > 0 && max_outputs > 0);
  hipLaunchKernelGGL(dw_reduce_batch_kernel, dim3(clsr_cdiv(max_outputs, 64), n), dim3(1024), 0,
                     (hipStream_t)stream, descs_device);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
