// Position-tiled linear layers as SPLIT-bf16 (bf16 x 3) products on the bf16 matrix pipe (gfx950) -- the "fp32x3" twin of
// clsr_pgemm / clsr_pgemm_bnbwd (csrc/linear.hip), same arguments, same tensors (everything fp32 in HBM):
//   models/base_model.py:664,704 (_fcn_net layers), models/sequential/clsr.py:363 (attention_mat), the input-side
//   projections of GRUCell / Time4LSTMCell (rnn_cell_implement.py:207-231), and the products that back-propagate
//   through them.
// Every operand value v is split in registers into hi = bf16(v), lo = bf16(v - hi) and a product is taken as
// hi*hi + lo*hi + hi*lo: three v_mfma_f32_16x16x32_bf16 with fp32 accumulation (<= 2^-16 relative per product; see
// csrc/dw3.hip).  The fp32-input MFMA needs 256 cycles for the K = 32 that these three cover in 48, and blocks the
// issue port of its SIMD meanwhile: the exact kernels sat at 0.46-0.57 of the 157 TFLOP/s fp32 matrix peak; these are
// bound by their loads and stores.
//
// Orientation ("features x positions", as in linear.hip): one MFMA tile is D[16 out-features][16 positions].
//   A operand = W^T tile from LDS  : lane (i = l&15, g = l>>4) holds Wt[o0 + i][32 kt + 8 g + {0..7}]  (hi and lo images:
//                                    split ONCE per workgroup while the fp32 packed weights are staged)
//   B operand = activations        : lane (j = l&15, g) holds X[pos j][32 kt + 8 g + {0..7}]: two float4 loads, split
//                                    in registers (6 VALU per pair of values; every split operand feeds 3 * OT MFMAs)
//   D         : lane (j, g) holds out features o0 + 4 g + {0..3} of position j -> one float4 store per out tile; the
//               output layout of a layer is the input layout of the next one.
// A wave owns 32 positions (two B operands): every LDS weight read feeds 6 MFMAs.
#include <stdlib.h>
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct P3Args {
  const float* X; int ldx;
  int T; int G;                   // T>0: r = m / T, t = m % T; G>0: xrow = (r / G) * T + t else xrow = m
  const float* Xmul; int ldmul;   // optional multiplier row r
  const float* in_scale; const float* in_shift; int in_relu;  // optional per-input-feature affine (+relu)
  const float* Wt; int ldw;       // packed transposed fp32 weights [16*ceil(N/16)][ldw], zero padded (clsr_pack_batch)
  const float* bias;
  const float* addU; int ldu;     // optional += addU[xrow][n]
  const float* addV; int ldv;     // optional += addV[r][n]
  float* Y; int ldy; int accumulate;
  double* stats;                  // optional per-block partial column sums [gridDim.x][2][N]
  int M, K, N;
  // optional BN+ReLU backward epilogue (see PGemmArgs in linear.hip)
  const float* ez; int ldez; const float* e_scale; const float* e_shift; const float* e_mean; const float* e_invstd;
};

// 8 fp32 values -> bf16 hi / lo operand vectors
__device__ __forceinline__ void split8(const f32x8& v, bf16x8& hi, bf16x8& lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2v));
    const float ah = __builtin_bit_cast(float, hp << 16), bh = __builtin_bit_cast(float, hp & 0xffff0000u);
    h[p] = hp;
    l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a - ah, b - bh}, bf16x2v));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

#define P3_PLAIN 0
#define P3_MUL 1
#define P3_AFF 2
#define E3_NONE 0
#define E3_UV 1
#define E3_ACC 2
#define E3_EZ 4

__host__ __device__ constexpr int p3_kh(int KT) { return 32 * KT + 8; }   // bf16 per LDS weight row (conflict-free b128 reads)

// RD: k-tiles whose raw loads are in flight (ring); KTT > 0: K is exactly KTT k-tiles, all of them loaded up front
template <int OT, int PRO, int EPI, bool STATS, int KTT, int RD>
__global__ void __launch_bounds__(256) pgemm3_kernel(P3Args a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int ntile_out = (a.N + 15) >> 4;
  const int ot0 = blockIdx.y * OT;
  const int otc = min(OT, ntile_out - ot0);
  const int n0 = ot0 * 16;
  const int KT = KTT > 0 ? KTT : (a.K + 31) >> 5;
  const int Kh = p3_kh(KT), KTP = 32 * KT;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  __bf16* Wh = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Wl = Wh + 16 * OT * Kh;
  float* ptab = reinterpret_cast<float*>(Wl + 16 * OT * Kh);            // [2][KTP] scale | shift (P3_AFF)
  double* red = reinterpret_cast<double*>(ptab + (PRO == P3_AFF ? 2 * KTP : 0));   // [4 waves][2][OT*16]
  {  // stage + split this block's W^T chunk (rows n0 .. n0 + 16*otc; everything else zero)
    const int Kq = Kh >> 2;
    const int kw = (a.K + 3) & ~3;
    for (int e = tid; e < 16 * OT * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      f32x4 v = zero4;
      if (row < 16 * otc && 4 * c < kw) v = ld4(a.Wt + (long)(n0 + row) * a.ldw + 4 * c);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 h, l;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float x = v[2 * p], y = v[2 * p + 1];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x, y}, bf16x2v));
        const float xh = __builtin_bit_cast(float, hp << 16), yh = __builtin_bit_cast(float, hp & 0xffff0000u);
        h[p] = hp;
        l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x - xh, y - yh}, bf16x2v));
      }
      *reinterpret_cast<u32x2*>(Wh + row * Kh + 4 * c) = h;
      *reinterpret_cast<u32x2*>(Wl + row * Kh + 4 * c) = l;
    }
    if (PRO == P3_AFF) {
      for (int e = tid; e < 2 * KTP; e += 256) {
        const int which = e / KTP, k = e - which * KTP;
        ptab[e] = k < a.K ? (which ? a.in_shift[k] : a.in_scale[k]) : 0.f;
      }
    }
    if (STATS) for (int e = tid; e < 4 * 2 * OT * 16; e += 256) red[e] = 0.0;
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

  constexpr int STAT_FLUSH = 8;
  float fsum[STATS ? OT : 1][4], fsq[STATS ? OT : 1][4];
  int pending = 0;
  if (STATS) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) { fsum[ot][r] = 0.f; fsq[ot][r] = 0.f; }
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

  // XCD-aware tile order (see pgemm_fast_kernel)
  const int ntiles = (a.M + 31) >> 5;
  const int nb = gridDim.x;
  const int nx = nb >= 8 ? 8 : 1;
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int nslots = (nb - xcd + nx - 1) / nx;
  const int chunk = (ntiles + nx - 1) / nx;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const __bf16* ldsH = Wh + j * Kh + 8 * g;   // + ot*16*Kh + kt*32
  const __bf16* ldsL = Wl + j * Kh + 8 * g;

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
      const int mc = valid[s] ? m : a.M - 1;
      mrow[s] = mc;
      int r = mc, xr = mc;
      if (a.T > 0) {
        r = mc / a.T;
        if (a.G > 0) xr = (r / a.G) * a.T + (mc - r * a.T);
      }
      xrow[s] = xr;
      rr[s] = r;
      xp[s] = a.X + (long)xr * a.ldx;
      mp[s] = PRO == P3_MUL ? a.Xmul + (long)r * a.ldmul : nullptr;
    }

    // raw loads of one k-tile: two float4 pieces per position (clamped addresses: nothing sits under a branch)
    struct Raw { f32x4 x[2][2], p[PRO == P3_MUL ? 2 : 1][2]; };
    auto issue = [&](int kt) -> Raw {
      Raw q;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kcol = kt * 32 + 8 * g + 4 * h;
        const int kc = kcol < a.K ? kcol : 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          q.x[s][h] = ld4(xp[s] + kc);
          if (PRO == P3_MUL) q.p[s][h] = ld4(mp[s] + kc);
        }
      }
      return q;
    };
    auto finish = [&](const Raw& q, int kt, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kcol = kt * 32 + 8 * g + 4 * h;
          f32x4 x = q.x[s][h];
          if (PRO == P3_MUL) x *= q.p[s][h];
          if (PRO == P3_AFF) {
            x = x * ld4(ptab + kcol) + ld4(ptab + KTP + kcol);
            x.x = fmaxf(x.x, relu_lo); x.y = fmaxf(x.y, relu_lo); x.z = fmaxf(x.z, relu_lo); x.w = fmaxf(x.w, relu_lo);
          }
          v[h] = kcol < a.K ? x : zero4;
        }
        split8((f32x8){v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w}, bh[s], bl[s]);
      }
    };

    constexpr int NR = KTT > 0 ? KTT : RD;
    Raw ring[NR];
#pragma unroll
    for (int d = 0; d < NR; ++d) ring[d] = issue(KTT > 0 ? d : min(d, KT - 1));
    // accumulators start from everything that is added to the product (bias, addU + addV, previous Y)
    f32x4 acc[2][OT];
    f32x4 ezr[(EPI & E3_EZ) ? 2 : 1][(EPI & E3_EZ) ? OT : 1];
    {
      constexpr bool UV = (EPI & E3_UV) != 0, ACC = (EPI & E3_ACC) != 0;
      f32x4 tu[UV ? 2 : 1][UV ? OT : 1], tv[(UV || ACC) ? 2 : 1][(UV || ACC) ? OT : 1];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int nc = nok[ot] ? n0 + ot * 16 + 4 * g : 0;
          if (UV) { tu[s][ot] = ld4(a.addU + xrow[s] * a.ldu + nc); tv[s][ot] = ld4(a.addV + rr[s] * a.ldv + nc); }
          if (ACC) tv[s][ot] = ld4(a.Y + (long)mrow[s] * a.ldy + nc);
          if (EPI & E3_EZ) ezr[s][ot] = ld4(a.ez + (long)mrow[s] * a.ldez + nc);
        }
      __builtin_amdgcn_sched_barrier(0);
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

    auto mfma_block = [&](int kt, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
      // one image at a time (OT x 4 operand registers live, not 2 x): lo image first (small terms first), then the hi
      // image against both halves of the activations; 2 * OT independent accumulators between two MFMAs on the same one
      bf16x8 w[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) w[ot] = ld8h(ldsL + ot * 16 * Kh + kt * 32);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { HMFMA(acc[0][ot], w[ot], bh[0]); HMFMA(acc[1][ot], w[ot], bh[1]); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) w[ot] = ld8h(ldsH + ot * 16 * Kh + kt * 32);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { HMFMA(acc[0][ot], w[ot], bl[0]); HMFMA(acc[1][ot], w[ot], bl[1]); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { HMFMA(acc[0][ot], w[ot], bh[0]); HMFMA(acc[1][ot], w[ot], bh[1]); }
    };
    if (KTT > 0) {
#pragma unroll
      for (int kt = 0; kt < NR; ++kt) {
        bf16x8 bh[2], bl[2];
        finish(ring[kt], kt, bh, bl);
        mfma_block(kt, bh, bl);
      }
    } else {
      for (int kt0 = 0; kt0 < KT; kt0 += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
          const int kt = kt0 + d;
          if (kt < KT) {
            bf16x8 bh[2], bl[2];
            finish(ring[d], kt, bh, bl);
            ring[d] = issue(min(kt + RD, KT - 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(kt, bh, bl);
          }
        }
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
        f32x4 w2 = v;
        if (EPI & E3_EZ) {
          const f32x4 zz = ezr[s][ot];
          const f32x4 y = zz * ld4(a.e_scale + nc) + ld4(a.e_shift + nc);
          v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f;
          v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
          w2 = (zz - ld4(a.e_mean + nc)) * ld4(a.e_invstd + nc);
        }
        if (EPI & E3_UV) { if (ok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yp)); }
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

// same grid along the positions as clsr_pgemm: the statistics partials have clsr_pgemm_stats_parts(M) rows
static int p3_grid_x(int M) {
  int ntiles = clsr_cdiv(M, 32);
  int gx = clsr_cdiv(ntiles, 4);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  return gx;
}

template <typename KernelT>
static int p3_go(KernelT kernel, const P3Args& a, dim3 grid, size_t shmem, hipStream_t stream) {
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static size_t p3_shmem(int OT, int K, bool aff, bool stats) {
  const int KT = clsr_cdiv(K, 32);
  return (size_t)2 * 16 * OT * p3_kh(KT) * 2 + (aff ? (size_t)2 * 32 * KT * 4 : 0) + (stats ? (size_t)4 * 2 * OT * 16 * 8 : 0);
}

template <int OT>
static int p3_launch(const P3Args& a, hipStream_t stream) {
  const int ntile_out = clsr_cdiv(a.N, 16);
  dim3 grid(p3_grid_x(a.M), clsr_cdiv(ntile_out, OT));
  const int pro = a.Xmul ? P3_MUL : (a.in_scale ? P3_AFF : P3_PLAIN);
  const int epi = ((a.addU && a.addV) ? E3_UV : 0) | (a.accumulate ? E3_ACC : 0) | (a.ez ? E3_EZ : 0);
  const bool st = a.stats != nullptr;
  const size_t shmem = p3_shmem(OT, a.K, pro == P3_AFF, st);
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  const int KT = clsr_cdiv(a.K, 32);
#define P3_CASE(P, E, S)                                                                                        \
  if (pro == (P) && epi == (E) && st == (S)) {                                                                  \
    if ((P) != P3_MUL && KT == 2) return p3_go(pgemm3_kernel<OT, P, E, S, 2, 1>, a, grid, shmem, stream);        \
    if ((P) != P3_MUL && KT == 3) return p3_go(pgemm3_kernel<OT, P, E, S, 3, 1>, a, grid, shmem, stream);        \
    if (KT == 1) return p3_go(pgemm3_kernel<OT, P, E, S, 1, 1>, a, grid, shmem, stream);                         \
    return p3_go(pgemm3_kernel<OT, P, E, S, 0, ((P) == P3_MUL ? 2 : 3)>, a, grid, shmem, stream);                \
  }
  P3_CASE(P3_PLAIN, E3_NONE, false)
  P3_CASE(P3_PLAIN, E3_NONE, true)
  P3_CASE(P3_MUL, E3_UV, false)
  P3_CASE(P3_MUL, E3_UV, true)
  P3_CASE(P3_AFF, E3_NONE, false)
  P3_CASE(P3_AFF, E3_NONE, true)
  P3_CASE(P3_PLAIN, E3_ACC, false)
  P3_CASE(P3_PLAIN, E3_EZ, true)
#undef P3_CASE
  clsr_set_error("%s:%d: clsr_pgemm3 has no variant for prologue %d / epilogue %d / stats %d (use clsr_pgemm)", __FILE__,
                 __LINE__, pro, epi, (int)st);
  return CLSR_EUNSUPPORTED;
}

static int p3_out_tiles(int N, int M) {
  const int nt = clsr_cdiv(N, 16);
  int ot;
  if (nt <= 3) ot = 3;
  else if (nt <= 5) ot = 5;
  else {
    const int waste5 = clsr_cdiv(nt, 5) * 5 - nt, waste3 = clsr_cdiv(nt, 3) * 3 - nt;
    ot = waste3 < waste5 ? 3 : 5;
  }
  if (M <= 32768 && ot > 3) ot = 3;       // few positions: more column chunks instead (see pgemm_dispatch_one)
  return ot;
}

// wide inputs: a chain of launches over K ranges that accumulate into Y (see pgemm_dispatch in linear.hip); plain
// products only (the other variants have no accumulating form)
#define P3_LDS_BUDGET ((size_t)64 * 1024)      // LDS for the two weight images of one launch
static int p3_kmax(int ot) { return ((int)(P3_LDS_BUDGET / ((size_t)2 * 16 * ot * 2)) - 8) / 32 * 32; }
static int p3_dispatch(const P3Args& a, hipStream_t s) {
  const int ot = p3_out_tiles(a.N, a.M);
  const int kmax = p3_kmax(ot);
  if (a.K > kmax) CLSR_CHECK_SUPPORTED(!a.Xmul && !a.in_scale && !a.addU && !a.addV && !a.stats && !a.ez);
  for (int k0 = 0; k0 < a.K; k0 += kmax) {
    P3Args c = a;
    c.K = k0 + kmax >= a.K ? a.K - k0 : kmax;
    c.X = a.X + k0;
    c.Wt = a.Wt + k0;
    if (k0 > 0) { c.accumulate = 1; c.bias = nullptr; }
    const int rc = ot == 3 ? p3_launch<3>(c, s) : p3_launch<5>(c, s);
    if (rc) return rc;
  }
  return CLSR_OK;
}

// 1 when clsr_pgemm3 has a kernel for this combination of the optional features of clsr_pgemm and this shape
extern "C" int clsr_pgemm3_supported(int has_mul, int has_aff, int has_u, int has_v, int accumulate, int has_stats, int M,
                                     int K, int N) {
  if (has_mul && has_aff) return 0;
  if ((has_u != 0) != (has_v != 0)) return 0;
  if (K % 4 || N % 4 || K <= 0 || N <= 0) return 0;
  const int pro = has_mul ? P3_MUL : (has_aff ? P3_AFF : P3_PLAIN);
  const int epi = (has_u ? E3_UV : 0) | (accumulate ? E3_ACC : 0);
  const bool plain = pro == P3_PLAIN && epi == E3_NONE;
  if (K > p3_kmax(p3_out_tiles(N, M))) return (plain || (pro == P3_PLAIN && epi == E3_ACC)) && !has_stats;
  if (plain) return 1;
  if (pro == P3_MUL && epi == E3_UV) return 1;
  if (pro == P3_AFF && epi == E3_NONE) return 1;
  if (pro == P3_PLAIN && epi == E3_ACC && !has_stats) return 1;
  return 0;
}

// Same contract as clsr_pgemm (csrc/linear.hip): Y[m, :N] (=|+=) f(X)[xrow(m), :K] . W + bias + addU[xrow(m)] + addV[r(m)],
// Wt = the fp32 packed transposed weights of clsr_pack_batch (row stride Kp); split-bf16 products.
extern "C" int clsr_pgemm3(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                           const float* in_scale, const float* in_shift, int in_relu, const float* Wt,
                           int Kp, const float* bias, const float* addU, int ldu, const float* addV,
                           int ldv, float* Y, int ldy, int accumulate, double* stats, int M, int K,
                           int N, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && M >= 0 && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && Kp % 4 == 0);
  CLSR_CHECK_ARG(Kp >= ((K + 3) & ~3));
  CLSR_CHECK_ARG(!(Xmul && ldmul % 4) && !(addU && ldu % 4) && !(addV && ldv % 4));
  CLSR_CHECK_ARG(!(in_scale && !in_shift));
  CLSR_CHECK_SUPPORTED(!(Xmul && in_scale));
  CLSR_CHECK_SUPPORTED(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ldx >= K &&
                       (!Xmul || ldmul >= K));
  if (M == 0) return CLSR_OK;
  P3Args a = {};
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.Wt = Wt; a.ldw = Kp; a.bias = bias; a.addU = addU; a.ldu = ldu; a.addV = addV; a.ldv = ldv;
  a.Y = Y; a.ldy = ldy; a.accumulate = accumulate; a.stats = stats; a.M = M; a.K = K; a.N = N;
  return p3_dispatch(a, (hipStream_t)stream);
}

extern "C" int clsr_pgemm3_bnbwd_supported(int M, int K, int N) {
  return K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0 && K <= p3_kmax(p3_out_tiles(N, M));
}

// Same contract as clsr_pgemm_bnbwd: dy[m, :N] = mask(dY_next[m, :K] . W^T), mask = (z*scale + shift > 0), with the
// batch-norm backward partial sums (sum dy, sum dy * xhat) of the layer below.
extern "C" int clsr_pgemm3_bnbwd(const float* X, int ldx, const float* Wt, int Kp, float* Y, int ldy,
                                 const float* z, int ldz, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, double* stats, int M, int K, int N,
                                 void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && z && scale && shift && mean && invstd && stats && M > 0 && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && Kp % 4 == 0);
  CLSR_CHECK_ARG(Kp >= ((K + 3) & ~3));
  P3Args a = {};
  a.X = X; a.ldx = ldx; a.Wt = Wt; a.ldw = Kp; a.Y = Y; a.ldy = ldy; a.stats = stats; a.M = M; a.K = K; a.N = N;
  a.in_relu = 1;
  a.ez = z; a.ldez = ldz; a.e_scale = scale; a.e_shift = shift; a.e_mean = mean; a.e_invstd = invstd;
  return p3_dispatch(a, (hipStream_t)stream);
}
