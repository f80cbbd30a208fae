// Position-tiled bf16-MFMA linear layers of the attention block (gfx950) -- the SPEED mode of
// clsr_pgemm (csrc/linear.hip): the (row, step)-level activations of _attention_fcn
// (reference models/sequential/clsr.py:343-381, _fcn_net models/base_model.py:627-708) are stored as
// bf16 and multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Batch-norm statistics,
// biases, the U + V terms and every per-feature coefficient stay fp32.
//
// Same "features x positions" orientation as the fp32 kernels: one MFMA tile is
// D[16 out-features][16 positions]; lane (j = l & 15, g = l >> 4) owns position j.
//   B operand (activations): lane (j, g) holds k = 32*kt + 8g + {0..7} of position j   -> ONE 16-byte load
//   A operand (weights, LDS): lane (i, g) holds Wt[row i][32*kt + 8g + {0..7}]          -> ONE ds_read_b128
//   D: lane (j, g) holds rows 4g + {0..3} of the tile.
// Out tiles come in PAIRS whose packed rows are permuted (clsr_pack_batch_bf16) so that the 4 + 4 values a
// lane holds after the two MFMAs are the 8 CONSECUTIVE features 32p + 8g + {0..7} of its position: one
// 16-byte bf16 store per pair, and the stored row is exactly the next layer's B-operand layout.
#include "common.h"
#include "clsr_hip.h"

#include "hmma.h"

#define HP_BF 0    // X bf16, no prologue
#define HP_MUL 1   // X fp32 * Xmul[r] fp32
#define HP_AFF 2   // X bf16: relu(x * scale + shift)
#define HP_DY1 3   // X = z1 bf16: x = a1 * dy1 + a2 * z1 + a3 with dy1 = (z1*sc1 + sh1 > 0) ? w_out * ds[m] : 0
#define HP_F32 4   // X fp32, no prologue
#define HE_NONE 0  // Y = acc + bias
#define HE_UV 1    // Y = acc + addU[xrow] + addV[r]
#define HE_EZS 2   // dy = (ez*sc + sh > 0) ? acc : 0; column sums of dy, dy*xhat ONLY (nothing stored)
#define HE_EZA 3   // same mask, then Y = c1*dy + c2*ez + c3 (the complete BN backward)
#define HE_F32 4   // fp32 Y (=|+=) acc   (back-propagating products whose operands and results stay fp32 tensors)

struct HGemmArgs {
  const void* X; int ldx;
  int T, G;
  const float* Xmul; int ldmul;
  const float* pv[5];      // HP_AFF: scale, shift | HP_DY1: sc1, sh1, w_out, coef1 (a1|a2|a3, stride K), -
  int in_relu;
  const float* ds;
  void* dx_out; int lddx;  // HP_DY1: optional bf16 copy of the computed x (= dz1) [M, lddx]
  const __bf16* Wt; int Kp;
  const float* bias;
  const float* addU; int ldu; const float* addV; int ldv;
  const __bf16* ez; int ldez;
  const float* ev[5];      // HE_EZS: scale, shift, mean, invstd | HE_EZA: scale, shift, coef (c1|c2|c3, stride N)
  __bf16* Y; int ldy;
  float* Yf; int accumulate;   // HE_F32
  double* stats;
  int M, K, N;
};

// LDS: W [32*NP][Kp] bf16 | ptab [5][KTP] f32 (AFF / DY1) | etab [5][32*NP] f32 (EZ*) | red [4][2][32*NP] f64 (stats)
// KTT > 0: the number of 32-wide k-tiles is KTT and ALL of a tile's operand loads are issued before the first MFMA (a
// wave then has every byte of its tile in flight at once: these kernels are bound by memory latency x occupancy, not
// by the matrix pipe); KTT == 0: any K, loads one k-tile ahead
template <int NP, int PRO, int EPI, bool STATS, int S, int KTT>
__global__ void __launch_bounds__(256, (PRO == HP_MUL && STATS) ? 2 : 3) hgemm_kernel(HGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NT = 2 * NP, NR = 32 * NP;
  constexpr bool XF32 = PRO == HP_MUL || PRO == HP_F32;
  constexpr bool PTAB = PRO == HP_AFF || PRO == HP_DY1;
  constexpr bool EZ = EPI == HE_EZS || EPI == HE_EZA;
  constexpr bool ST = STATS || EPI == HE_EZS;   // the statistics pass of the BN backward always produces its sums
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int KT = (a.K + 31) >> 5, KTP = KT * 32;
  const int Kp = a.Kp;
  const int n0 = blockIdx.y * NR;
  __bf16* Wl = reinterpret_cast<__bf16*>(lds_raw);
  size_t off = ((size_t)NR * Kp * 2 + 15) & ~(size_t)15;
  float* ptab = reinterpret_cast<float*>(lds_raw + off);
  if (PTAB) off += (size_t)5 * KTP * 4;
  float* etab = reinterpret_cast<float*>(lds_raw + off);
  if (EZ) off += (size_t)5 * NR * 4;
  double* red = reinterpret_cast<double*>(lds_raw + off);
  const int nrows_w = min(NR, 32 * ((a.N + 31) >> 5) - n0);   // packed rows that exist for this column chunk
  {  // weights: rows n0 .. n0 + NR of the packed matrix (zero beyond the packed rows)
    const int Kq = Kp >> 3;   // 16-byte pieces per row
    const bf16x8 z8 = {};
    for (int e = tid; e < NR * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      reinterpret_cast<bf16x8*>(Wl)[e] = row < nrows_w ? ld8h(a.Wt + (long)(n0 + row) * Kp + 8 * c) : z8;
    }
    if (PTAB) {
      for (int e = tid; e < 5 * KTP; e += 256) {
        const int which = e / KTP, k = e - which * KTP;
        float v = 0.f;
        if (k < a.K) {
          if (PRO == HP_AFF) v = which < 2 ? a.pv[which][k] : 0.f;
          else {  // DY1 rows: sc1, sh1, p = a1 * w_out, a2, a3
            if (which < 2) v = a.pv[which][k];
            else if (which == 2) v = a.pv[3][k] * a.pv[2][k];
            else v = a.pv[3][(which - 2) * a.K + k];
          }
        }
        ptab[e] = v;
      }
    }
    if (EZ) {
      for (int e = tid; e < 5 * NR; e += 256) {
        const int which = e / NR, n = n0 + (e - which * NR);
        float v = 0.f;
        if (n < a.N) {
          if (which < 2) v = a.ev[which][n];
          else if (EPI == HE_EZS) v = which < 4 ? a.ev[which][n] : 0.f;
          else v = a.ev[2][(which - 2) * a.N + n];
        }
        etab[e] = v;
      }
    }
    if (ST) for (int e = tid; e < 4 * 2 * NR; e += 256) red[e] = 0.0;
  }
  __syncthreads();

  f32x8 biasr[NP];
  bool nok[NP];
  int ncl[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int n = n0 + 32 * p + 8 * g;
    nok[p] = n < a.N;
    ncl[p] = nok[p] ? n : 0;
    biasr[p] = (f32x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (EPI == HE_NONE && a.bias && nok[p]) biasr[p] = ld8f(a.bias + n);
  }
  const float relu_lo = a.in_relu ? 0.f : -3.0e38f;

  constexpr int STAT_FLUSH = 32 / S;   // fp32 per-lane partial sums of at most 32 values between flushes
  float fsum[ST ? NP : 1][8], fsq[ST ? NP : 1][8];
  int pending = 0;
  if (ST) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 8; ++r) { fsum[p][r] = 0.f; fsq[p][r] = 0.f; }
  }
  auto flush = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float s_ = row16_sum(fsum[p][r]);
        const float q_ = row16_sum(fsq[p][r]);
        if (j == 0) {
          red[(wave * 2 + 0) * NR + 32 * p + 8 * g + r] += (double)s_;
          red[(wave * 2 + 1) * NR + 32 * p + 8 * g + r] += (double)q_;
        }
        fsum[p][r] = 0.f;
        fsq[p][r] = 0.f;
      }
  };

  // XCD-aware tile order (see pgemm_fast_kernel): workgroup b runs on XCD b % 8; every XCD walks one contiguous
  // range of position tiles so the rows of a history group re-read their history-level operands from that XCD's L2
  const int ntiles = (a.M + 16 * S - 1) / (16 * S);
  const int nb = gridDim.x;
  const int nx = nb >= 8 ? 8 : 1;
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int nslots = (nb - xcd + nx - 1) / nx;
  const int chunk = (ntiles + nx - 1) / nx;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const __bf16* ldsA = Wl + (long)j * Kp + 8 * g;   // + (32p + 16half) * Kp + 32 kt

  for (int tile = xcd * chunk + slot * 4 + wave; tile < t_end; tile += nslots * 4) {
    int mrow[S];
    bool valid[S];
    long xrow[S], rr[S];
    float dsv[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int m = tile * (16 * S) + s * 16 + j;
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
      dsv[s] = PRO == HP_DY1 ? a.ds[mc] : 0.f;
    }

    struct Raw { f32x8 xf[S], mf[S]; bf16x8 xh[S]; };
    auto issue = [&](int kt) -> Raw {
      const int kcol = kt * 32 + 8 * g;
      const int kc = kcol < a.K ? kcol : 0;
      Raw q;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (XF32) q.xf[s] = ld8f(reinterpret_cast<const float*>(a.X) + xrow[s] * a.ldx + kc);
        else q.xh[s] = ld8h(reinterpret_cast<const __bf16*>(a.X) + xrow[s] * a.ldx + kc);
        if (PRO == HP_MUL) q.mf[s] = ld8f(a.Xmul + rr[s] * a.ldmul + kc);
      }
      return q;
    };
    auto finish = [&](const Raw& q, int kt, bf16x8* b) {
      const int kcol = kt * 32 + 8 * g;
      const bool ink = kcol < a.K;
      const bf16x8 z8 = {};
#pragma unroll
      for (int s = 0; s < S; ++s) {
        bf16x8 v;
        if (PRO == HP_BF) v = q.xh[s];
        if (PRO == HP_F32) v = to_h(q.xf[s]);
        if (PRO == HP_MUL) v = to_h(q.xf[s] * q.mf[s]);
        if (PRO == HP_AFF) {
          const f32x8 sc = ld8f(ptab + kcol), sh = ld8f(ptab + KTP + kcol);
          f32x8 x = to_f(q.xh[s]) * sc + sh;
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], relu_lo);
          v = to_h(x);
        }
        if (PRO == HP_DY1) {
          const f32x8 sc = ld8f(ptab + kcol), sh = ld8f(ptab + KTP + kcol), pp = ld8f(ptab + 2 * KTP + kcol);
          const f32x8 a2 = ld8f(ptab + 3 * KTP + kcol), a3 = ld8f(ptab + 4 * KTP + kcol);
          const f32x8 zz = to_f(q.xh[s]);
          const f32x8 y = zz * sc + sh;
          f32x8 x = a2 * zz + a3;
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += y[e] > 0.f ? pp[e] * dsv[s] : 0.f;
          v = to_h(x);
          if (a.dx_out && blockIdx.y == 0 && ink && valid[s])
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.dx_out) + (long)mrow[s] * a.lddx + kcol) = v;
        }
        b[s] = ink ? v : z8;
      }
    };

    Raw rawk[KTT > 0 ? KTT : 1];
#pragma unroll
    for (int kt = 0; kt < (KTT > 0 ? KTT : 1); ++kt) rawk[kt] = issue(kt);
    f32x4 acc[S][NT];
    bf16x8 ezr[EZ ? S : 1][EZ ? NP : 1];
    {
      constexpr bool INIT = EPI == HE_UV || EPI == HE_F32;
      f32x8 tu[INIT ? S : 1][INIT ? NP : 1], tv[EPI == HE_UV ? S : 1][EPI == HE_UV ? NP : 1];
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (EPI == HE_UV) {
            tu[s][p] = ld8f(a.addU + xrow[s] * a.ldu + ncl[p]);
            tv[s][p] = ld8f(a.addV + rr[s] * a.ldv + ncl[p]);
          }
          if (EPI == HE_F32) {
            tu[s][p] = (f32x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (a.accumulate) tu[s][p] = ld8f(a.Yf + (long)mrow[s] * a.ldy + ncl[p]);
          }
          if (EZ) ezr[s][p] = ld8h(a.ez + (long)mrow[s] * a.ldez + ncl[p]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          f32x8 v = biasr[p];
          if (EPI == HE_UV) v = tu[s][p] + tv[s][p];
          if (EPI == HE_F32) v = tu[s][p];
          acc[s][2 * p] = (f32x4){v[0], v[1], v[2], v[3]};
          acc[s][2 * p + 1] = (f32x4){v[4], v[5], v[6], v[7]};
        }
    }

    if (KTT > 0) {
#pragma unroll
      for (int kt = 0; kt < (KTT > 0 ? KTT : 1); ++kt) {
        bf16x8 b[S];
        finish(rawk[kt], kt, b);
        bf16x8 wt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wt[t] = ld8h(ldsA + (long)(16 * t) * Kp + kt * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int s = 0; s < S; ++s) HMFMA(acc[s][t], wt[t], b[s]);
      }
    } else {
      // any K: a ring of four k-tiles in flight (slot = kt % 4, refilled right after it is consumed)
      Raw ring[4];
      ring[0] = rawk[0];
#pragma unroll
      for (int d = 1; d < 4; ++d) ring[d] = issue(min(d, KT - 1));
      for (int kt0 = 0; kt0 < KT; kt0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int kt = kt0 + d;
          if (kt < KT) {
            bf16x8 b[S];
            finish(ring[d], kt, b);
            ring[d] = issue(min(kt + 4, KT - 1));
            bf16x8 wt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wt[t] = ld8h(ldsA + (long)(16 * t) * Kp + kt * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int s = 0; s < S; ++s) HMFMA(acc[s][t], wt[t], b[s]);
          }
        }
      }
    }

#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const bool ok = valid[s] && nok[p];
        const f32x4 lo = acc[s][2 * p], hi = acc[s][2 * p + 1];
        f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        f32x8 w2;
        bf16x8 out;
        if (EZ) {
          const int nl = 32 * p + 8 * g;
          const f32x8 zz = to_f(ezr[s][p]);
          const f32x8 y = zz * ld8f(etab + nl) + ld8f(etab + NR + nl);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (y[e] > 0.f && ok) ? v[e] : 0.f;
          if (EPI == HE_EZS) {
            w2 = (zz - ld8f(etab + 2 * NR + nl)) * ld8f(etab + 3 * NR + nl);
          } else {
            const f32x8 c1 = ld8f(etab + 2 * NR + nl), c2 = ld8f(etab + 3 * NR + nl), c3 = ld8f(etab + 4 * NR + nl);
            out = to_h(c1 * v + c2 * zz + c3);
            w2 = v;
          }
        } else {
          out = to_h(v);
          v = to_f(out);   // statistics of the STORED (rounded) values: the next layer normalises exactly those
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : 0.f;
          w2 = v;
        }
        if (EPI == HE_F32) {
          if (ok) {
            float* yp = a.Yf + (long)mrow[s] * a.ldy + ncl[p];
            st4(yp, lo);
            st4(yp + 4, hi);
          }
        } else if (EPI != HE_EZS && ok) {
          bf16x8* yp = reinterpret_cast<bf16x8*>(a.Y + (long)mrow[s] * a.ldy + ncl[p]);
          if (EPI == HE_UV) __builtin_nontemporal_store(out, yp);
          else *yp = out;
        }
        if (ST) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { fsum[p][e] += v[e]; fsq[p][e] = fmaf(v[e], w2[e], fsq[p][e]); }
        }
      }
    }
    if (ST && ++pending == STAT_FLUSH) {
      flush();
      pending = 0;
    }
  }

  if (ST) {
    if (pending) flush();
    __syncthreads();
    for (int e = tid; e < 2 * NR; e += 256) {
      const int which = e / NR, c = e - which * NR;
      const int n = n0 + c;
      if (n < a.N) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * NR + c];
        a.stats[((long)blockIdx.x * 2 + which) * a.N + n] = s;
      }
    }
  }
}

#define HG_S 1   // 16-position sets per wave iteration
static int hgemm_grid_x(int M) {
  int ntiles = clsr_cdiv(M, 16 * HG_S);
  int gx = clsr_cdiv(ntiles, 4);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  return gx;
}

extern "C" int clsr_hgemm_stats_parts(int M) { return hgemm_grid_x(M); }
extern "C" int clsr_hgemm_kp(int K) { return 32 * clsr_cdiv(K, 32) + 8; }

template <int NP, int PRO, int EPI, bool STATS>
static int hgemm_launch(const HGemmArgs& a, hipStream_t stream) {
  const int KT = clsr_cdiv(a.K, 32);
  size_t shmem = (((size_t)32 * NP * a.Kp * 2 + 15) & ~(size_t)15);
  if (PRO == HP_AFF || PRO == HP_DY1) shmem += (size_t)5 * KT * 32 * 4;
  if (EPI == HE_EZS || EPI == HE_EZA) shmem += (size_t)5 * 32 * NP * 4;
  if (STATS || EPI == HE_EZS) shmem += (size_t)4 * 2 * 32 * NP * 8;
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  dim3 grid(hgemm_grid_x(a.M), clsr_cdiv(a.N, 32 * NP));
#define HG_GO(KTTV)                                                                                                   \
  do {                                                                                                                \
    auto kernel = hgemm_kernel<NP, PRO, EPI, STATS, HG_S, KTTV>;                                                      \
    if (shmem > 64 * 1024)                                                                                            \
      CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));     \
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);                                                    \
  } while (0)
  if (KT == 2) HG_GO(2);
  else if (KT == 3) HG_GO(3);
  else HG_GO(0);
#undef HG_GO
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

template <int PRO, int EPI, bool STATS>
static int hgemm_np(const HGemmArgs& a, hipStream_t s) {
  if (a.N <= 64) return hgemm_launch<2, PRO, EPI, STATS>(a, s);
  if (a.N <= 96 || (a.N > 128 && a.N <= 192)) return hgemm_launch<3, PRO, EPI, STATS>(a, s);
  return hgemm_launch<4, PRO, EPI, STATS>(a, s);
}

static int hgemm_check(const void* X, const void* Wt, int Kp, int M, int K, int N, int ldx) {
  CLSR_CHECK_ARG(X && Wt && M >= 0 && K > 0 && N > 0);
  CLSR_CHECK_SUPPORTED(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && Kp % 8 == 0 && ((uintptr_t)X % 16) == 0);
  CLSR_CHECK_ARG(Kp >= 32 * clsr_cdiv(K, 32) && ldx >= K);
  return CLSR_OK;
}

// Y[m, :N] (bf16) = (a[xrow(m), :K] * q[r(m), :K]) . W + addU[xrow(m)] + addV[r(m)]      (fp32 a, q, U, V)
// the re-associated first attention layer (clsr.py:368-370 + base_model.py:664); stats = per-block partial column
// sums / sums of squares of the STORED values [clsr_hgemm_stats_parts(M)][2][N]
extern "C" int clsr_hgemm_mul_uv(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul, const void* Wt,
                                 int Kp, const float* addU, int ldu, const float* addV, int ldv, void* Y, int ldy,
                                 double* stats, int M, int K, int N, void* stream) {
  int rc = hgemm_check(X, Wt, Kp, M, K, N, ldx);
  if (rc) return rc;
  CLSR_CHECK_ARG(Xmul && addU && addV && Y && T > 0 && ldmul >= K && ldu >= N && ldv >= N && ldy >= N);
  CLSR_CHECK_SUPPORTED(ldmul % 4 == 0 && ldu % 4 == 0 && ldv % 4 == 0 && ldy % 8 == 0 && ldx % 4 == 0);
  if (M == 0) return CLSR_OK;
  HGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul; a.Wt = (const __bf16*)Wt; a.Kp = Kp;
  a.addU = addU; a.ldu = ldu; a.addV = addV; a.ldv = ldv; a.Y = (__bf16*)Y; a.ldy = ldy; a.stats = stats;
  a.M = M; a.K = K; a.N = N;
  if (stats) return hgemm_np<HP_MUL, HE_UV, true>(a, (hipStream_t)stream);
  return hgemm_np<HP_MUL, HE_UV, false>(a, (hipStream_t)stream);
}

// Y[m, :N] (bf16) = f(X[m, :K]) . W + bias,  X bf16,  f = relu(x * in_scale + in_shift) when in_scale is given
// (the layers behind a batch-norm: the normalised tensor is never materialised), identity otherwise
extern "C" int clsr_hgemm(const void* X, int ldx, const float* in_scale, const float* in_shift, int in_relu,
                          const void* Wt, int Kp, const float* bias, void* Y, int ldy, double* stats, int M, int K,
                          int N, void* stream) {
  int rc = hgemm_check(X, Wt, Kp, M, K, N, ldx);
  if (rc) return rc;
  CLSR_CHECK_ARG(Y && ldy >= N && !(in_scale && !in_shift));
  CLSR_CHECK_SUPPORTED(ldy % 8 == 0);
  if (M == 0) return CLSR_OK;
  HGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.Wt = (const __bf16*)Wt; a.Kp = Kp; a.bias = bias; a.Y = (__bf16*)Y; a.ldy = ldy;
  a.stats = stats; a.M = M; a.K = K; a.N = N; a.in_relu = in_relu; a.pv[0] = in_scale; a.pv[1] = in_shift;
  hipStream_t s = (hipStream_t)stream;
  if (in_scale) return stats ? hgemm_np<HP_AFF, HE_NONE, true>(a, s) : hgemm_np<HP_AFF, HE_NONE, false>(a, s);
  return stats ? hgemm_np<HP_BF, HE_NONE, true>(a, s) : hgemm_np<HP_BF, HE_NONE, false>(a, s);
}

// ------------------------------------------------------------------------------------ layer 0, one wave per history
// z0[r,t,:] (bf16) = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp with the G rows of a history group handled by ONE
// wave: it walks the history's 16-step tiles and, inside a tile, the rows of the group, so a[h,t,:] and U[h,t,:] are
// loaded once per tile (the position-tiled kernel above re-reads them for every row: 1.3 GB through L2 at 1M
// positions), q[r,:] and V[r,:] sit in LDS.  Same orientation, weights image and statistics as hgemm_kernel<MUL, UV>.
#define HL0_GMAX 8
struct HL0Args {
  const float* a; int lda;
  const float* q; int ldq;
  const __bf16* Wt; int Kp;
  const float* U; int ldu;
  const float* V; int ldv;
  __bf16* z0; int ldz;
  double* stats;
  long Hn;
  int G, T, Q, A0;
};

template <int NP, int KT, bool STATS>
__global__ void __launch_bounds__(256, 2) hgemm_l0g_kernel(HL0Args s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NT = 2 * NP, NR = 32 * NP, KTP = 32 * KT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int Kp = s.Kp;
  __bf16* Wl = reinterpret_cast<__bf16*>(lds_raw);
  size_t off = ((size_t)NR * Kp * 2 + 15) & ~(size_t)15;
  float* qs = reinterpret_cast<float*>(lds_raw + off) + (size_t)wave * HL0_GMAX * (KTP + NR);
  float* vs = qs + HL0_GMAX * KTP;
  off += (size_t)4 * HL0_GMAX * (KTP + NR) * 4;
  double* red = reinterpret_cast<double*>(lds_raw + off);   // [4 waves][2][NR]
  {
    const int Kq = Kp >> 3;
    const int nrows = 32 * ((s.A0 + 31) >> 5);
    const bf16x8 z8 = {};
    for (int e = tid; e < NR * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      reinterpret_cast<bf16x8*>(Wl)[e] = row < nrows ? ld8h(s.Wt + (long)row * Kp + 8 * c) : z8;
    }
    if (STATS) for (int e = tid; e < 4 * 2 * NR; e += 256) red[e] = 0.0;
  }
  __syncthreads();
  const __bf16* ldsA = Wl + (long)j * Kp + 8 * g;
  bool nok[NP];
  int ncl[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    nok[p] = 32 * p + 8 * g < s.A0;
    ncl[p] = nok[p] ? 32 * p + 8 * g : 0;
  }
  float fsum[STATS ? NP : 1][8], fsq[STATS ? NP : 1][8];
  int pending = 0;
  if (STATS) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 8; ++r) { fsum[p][r] = 0.f; fsq[p][r] = 0.f; }
  }
  auto flush = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float s_ = row16_sum(fsum[p][r]);
        const float q_ = row16_sum(fsq[p][r]);
        if (j == 0) {
          red[(wave * 2 + 0) * NR + 32 * p + 8 * g + r] += (double)s_;
          red[(wave * 2 + 1) * NR + 32 * p + 8 * g + r] += (double)q_;
        }
        fsum[p][r] = 0.f;
        fsq[p][r] = 0.f;
      }
  };
  // (the ragged last tile of a history is packed across the G rows when they fit into one tile: see att_l0_fwd_kernel)
  const int G = s.G, T = s.T, rem = T & 15;
  const bool packed = rem > 0 && G * rem <= 16;
  const int NTT = packed ? T >> 4 : (T + 15) >> 4;
  const bf16x8 z8 = {};

  for (long h = (long)blockIdx.x * 4 + wave; h < s.Hn; h += (long)gridDim.x * 4) {
    for (int e = lane; e < G * KTP; e += 64) {
      const int gg = e / KTP, n = e - gg * KTP;
      qs[e] = n < s.Q ? s.q[(h * G + gg) * s.ldq + n] : 0.f;
    }
    for (int e = lane; e < G * NR; e += 64) {
      const int gg = e / NR, n = e - gg * NR;
      vs[e] = n < s.A0 ? s.V[(h * G + gg) * s.ldv + n] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int tt = 0; tt < NTT; ++tt) {
      const int t = 16 * tt + j;
      const bool valid = t < T;
      const long xr = h * T + (valid ? t : T - 1);
      f32x8 at[KT], ut[NP];     // (a next-tile prefetch of these 48 registers was measured: 91 -> 97 us, dropped)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) at[kt] = ld8f(s.a + xr * s.lda + (32 * kt + 8 * g < s.Q ? 32 * kt + 8 * g : 0));
#pragma unroll
      for (int p = 0; p < NP; ++p) ut[p] = ld8f(s.U + xr * s.ldu + ncl[p]);
      // (arrive before the loop over the rows: see att_l0_fwd_kernel)
#if defined(__HIP_DEVICE_COMPILE__)   // ("v" is an x86 vector-register constraint in the host pass)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) asm volatile("" ::"v"(at[kt]));
#pragma unroll
      for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(ut[p]));
#endif
      for (int gg = 0; gg < G; ++gg) {
        f32x4 acc[NT];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f32x8 v = ut[p] + ld8f(vs + gg * NR + 32 * p + 8 * g);
          acc[2 * p] = (f32x4){v[0], v[1], v[2], v[3]};
          acc[2 * p + 1] = (f32x4){v[4], v[5], v[6], v[7]};
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const bf16x8 b = 32 * kt + 8 * g < s.Q ? to_h(at[kt] * ld8f(qs + gg * KTP + 32 * kt + 8 * g)) : z8;
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) HMFMA(acc[tl], ld8h(ldsA + (long)(16 * tl) * Kp + kt * 32), b);
        }
        const long m = (h * G + gg) * T + t;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const bool ok = valid && nok[p];
          const f32x4 lo = acc[2 * p], hi = acc[2 * p + 1];
          const f32x8 v0 = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          const bf16x8 out = to_h(v0);
          if (ok) __builtin_nontemporal_store(out, reinterpret_cast<bf16x8*>(s.z0 + m * s.ldz + ncl[p]));
          if (STATS) {
            const f32x8 v = to_f(out);   // statistics of the STORED (rounded) values
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = ok ? v[e] : 0.f;
              fsum[p][e] += x;
              fsq[p][e] = fmaf(x, x, fsq[p][e]);
            }
          }
        }
        if (STATS && ++pending == 32) {
          flush();
          pending = 0;
        }
      }
    }
    if (packed) {   // lane j = packed position j = (row j / rem, step T - rem + j % rem): every operand is per lane
      const int np = G * rem;
      const bool valid = j < np;
      const int gj = valid ? j / rem : 0, tj = T - rem + (valid ? j - gj * rem : 0);
      const long xr = h * T + tj;
      f32x4 acc[NT];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const f32x8 v = ld8f(s.U + xr * s.ldu + ncl[p]) + ld8f(vs + gj * NR + 32 * p + 8 * g);
        acc[2 * p] = (f32x4){v[0], v[1], v[2], v[3]};
        acc[2 * p + 1] = (f32x4){v[4], v[5], v[6], v[7]};
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int kc = 32 * kt + 8 * g;
        const bf16x8 b = (valid && kc < s.Q) ? to_h(ld8f(s.a + xr * s.lda + kc) * ld8f(qs + gj * KTP + kc)) : z8;
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) HMFMA(acc[tl], ld8h(ldsA + (long)(16 * tl) * Kp + kt * 32), b);
      }
      const long m = (h * G + gj) * T + tj;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const bool ok = valid && nok[p];
        const f32x4 lo = acc[2 * p], hi = acc[2 * p + 1];
        const f32x8 v0 = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const bf16x8 out = to_h(v0);
        if (ok) __builtin_nontemporal_store(out, reinterpret_cast<bf16x8*>(s.z0 + m * s.ldz + ncl[p]));
        if (STATS) {
          const f32x8 v = to_f(out);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = ok ? v[e] : 0.f;
            fsum[p][e] += x;
            fsq[p][e] = fmaf(x, x, fsq[p][e]);
          }
        }
      }
      if (STATS && ++pending == 32) {
        flush();
        pending = 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (STATS) {
    if (pending) flush();
    __syncthreads();
    for (int e = tid; e < 2 * NR; e += 256) {
      const int which = e / NR, c = e - which * NR;
      if (c < s.A0) {
        double t = 0.0;
        for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * NR + c];
        s.stats[((long)blockIdx.x * 2 + which) * s.A0 + c] = t;
      }
    }
  }
}

static int hl0_grid(long Hn) {
  long gx = (Hn + 3) / 4;
  return (int)(gx > 512 ? 512 : gx);
}
// 1 when clsr_hgemm_l0_group handles this shape (otherwise clsr_hgemm_mul_uv); only groups of several rows gain
extern "C" int clsr_hgemm_l0_group_supported(int G, int Q, int A0) {
  return G >= 2 && G <= HL0_GMAX && Q >= 8 && Q <= 96 && A0 >= 8 && A0 <= 96 && Q % 8 == 0 && A0 % 8 == 0;
}
extern "C" int clsr_hgemm_l0_group_stats_parts(long Hn) { return hl0_grid(Hn); }

template <int NP, int KT>
static int hl0_launch(const HL0Args& a, hipStream_t stream) {
  size_t shmem = (((size_t)32 * NP * a.Kp * 2 + 15) & ~(size_t)15) + (size_t)4 * HL0_GMAX * (32 * KT + 32 * NP) * 4 +
                 (size_t)4 * 2 * 32 * NP * 8;
  dim3 grid(hl0_grid(a.Hn));
  if (a.stats) {
    auto kernel = hgemm_l0g_kernel<NP, KT, true>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  } else {
    auto kernel = hgemm_l0g_kernel<NP, KT, false>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  }
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_hgemm_l0_group(const float* a, int lda, const float* q, int ldq, const void* Wt, int Kp,
                                   const float* U, int ldu, const float* V, int ldv, void* z0, int ldz, double* stats,
                                   long Hn, int G, int T, int Q, int A0, void* stream) {
  CLSR_CHECK_ARG(a && q && Wt && U && V && z0 && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_hgemm_l0_group_supported(G, Q, A0));
  CLSR_CHECK_SUPPORTED(lda % 4 == 0 && ldu % 4 == 0 && ldz % 8 == 0 && Kp % 8 == 0 && ((uintptr_t)a % 16) == 0 &&
                       ((uintptr_t)U % 16) == 0 && ((uintptr_t)z0 % 16) == 0);
  CLSR_CHECK_ARG(lda >= Q && ldq >= Q && Kp >= 32 * clsr_cdiv(Q, 32) && ldu >= A0 && ldv >= A0 && ldz >= A0);
  HL0Args s = {};
  s.a = a; s.lda = lda; s.q = q; s.ldq = ldq; s.Wt = (const __bf16*)Wt; s.Kp = Kp; s.U = U; s.ldu = ldu; s.V = V;
  s.ldv = ldv; s.z0 = (__bf16*)z0; s.ldz = ldz; s.stats = stats; s.Hn = Hn; s.G = G; s.T = T; s.Q = Q; s.A0 = A0;
  hipStream_t st = (hipStream_t)stream;
  const int np = clsr_cdiv(A0, 32), kt = clsr_cdiv(Q, 32);
#define HL0_GO(P, K) if (np == P && kt == K) return hl0_launch<P, K>(s, st)
  HL0_GO(1, 1); HL0_GO(1, 2); HL0_GO(1, 3); HL0_GO(2, 1); HL0_GO(2, 2); HL0_GO(2, 3); HL0_GO(3, 1); HL0_GO(3, 2); HL0_GO(3, 3);
#undef HL0_GO
  return CLSR_OK;
}

// Backward through the second attention layer and the batch-norm + ReLU below it, the [M, C1] gradient dz1 never
// round-tripping through memory:   x = dz1[m, :C1] = a1*dy1 + a2*z1 + a3  (BN-1 backward; dy1 = ds[m] * w_out where
// relu(bn1(z1)) > 0, recomputed from z1 and the score gradient ds), dh0 = x . W1^T, dy0 = dh0 where relu(bn0(z0)) > 0.
//   pass 1 (coef0 == NULL): per-block partial sums of dy0 and dy0 * xhat0 -> stats  (nothing else is written)
//   pass 2 (coef0 given):   dz0[m, :C0] (bf16) = c1*dy0 + c2*z0 + c3; dz1 is also stored (bf16) for the weight gradient
// Two passes over (z1, z0) instead of materialising dy0 and sweeping it again: 0.74 GB instead of 0.98 GB at 1M
// positions, and the MFMA work that is done twice is 1/16 of what it costs in fp32.
extern "C" int clsr_hgemm_att_l1_bwd(const void* z1, int ldz1, const float* ds, const float* scale1,
                                     const float* shift1, const float* w_out, const float* coef1, const void* Wt,
                                     int Kp, const void* z0, int ldz0, const float* scale0, const float* shift0,
                                     const float* mean0, const float* invstd0, const float* coef0, void* dz1,
                                     int lddz1, void* dz0, int lddz0, double* stats, int M, int C1, int C0,
                                     void* stream) {
  int rc = hgemm_check(z1, Wt, Kp, M, C1, C0, ldz1);
  if (rc) return rc;
  CLSR_CHECK_ARG(ds && scale1 && shift1 && w_out && coef1 && z0 && scale0 && shift0 && ldz0 >= C0 && M > 0);
  CLSR_CHECK_ARG(coef0 ? (dz0 && lddz0 >= C0) : (mean0 && invstd0 && stats));
  CLSR_CHECK_SUPPORTED(ldz0 % 8 == 0 && (!dz0 || lddz0 % 8 == 0) && (!dz1 || lddz1 % 8 == 0));
  HGemmArgs a = {};
  a.X = z1; a.ldx = ldz1; a.ds = ds; a.pv[0] = scale1; a.pv[1] = shift1; a.pv[2] = w_out; a.pv[3] = coef1;
  a.Wt = (const __bf16*)Wt; a.Kp = Kp; a.ez = (const __bf16*)z0; a.ldez = ldz0; a.ev[0] = scale0; a.ev[1] = shift0;
  a.M = M; a.K = C1; a.N = C0; a.stats = stats;
  hipStream_t s = (hipStream_t)stream;
  if (!coef0) {
    a.ev[2] = mean0; a.ev[3] = invstd0;
    return hgemm_np<HP_DY1, HE_EZS, true>(a, s);
  }
  a.ev[2] = coef0; a.Y = (__bf16*)dz0; a.ldy = lddz0; a.dx_out = dz1; a.lddx = lddz1;
  return hgemm_np<HP_DY1, HE_EZA, false>(a, s);
}

// Y[m, :N] (fp32, =|+=) X[m, :K] . W with fp32 X: the back-propagating products of the history-level / row-level layers
// in speed mode (operands rounded to bf16 in registers, fp32 accumulation and storage) -- gradients only, the forward
// values of those layers stay on the exact fp32 kernels
extern "C" int clsr_hgemm_f32(const float* X, int ldx, const void* Wt, int Kp, float* Y, int ldy, int accumulate,
                              int M, int K, int N, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && M >= 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
  CLSR_CHECK_SUPPORTED(K % 8 == 0 && N % 8 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && Kp % 8 == 0 &&
                       ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0);
  CLSR_CHECK_ARG(Kp >= 32 * clsr_cdiv(K, 32));
  if (M == 0) return CLSR_OK;
  HGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.Wt = (const __bf16*)Wt; a.Kp = Kp; a.Yf = Y; a.ldy = ldy; a.accumulate = accumulate;
  a.M = M; a.K = K; a.N = N;
  return hgemm_np<HP_F32, HE_F32, false>(a, (hipStream_t)stream);
}

// The same with a bf16 X (gradients that the producing kernel already stored as bf16: dPin of the sequence encoders)
extern "C" int clsr_hgemm_hf32(const void* X, int ldx, const void* Wt, int Kp, float* Y, int ldy, int accumulate,
                               int M, int K, int N, void* stream) {
  int rc = hgemm_check(X, Wt, Kp, M, K, N, ldx);
  if (rc) return rc;
  CLSR_CHECK_ARG(Y && ldy >= N);
  CLSR_CHECK_SUPPORTED(ldy % 4 == 0 && ((uintptr_t)Y % 16) == 0);
  if (M == 0) return CLSR_OK;
  HGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.Wt = (const __bf16*)Wt; a.Kp = Kp; a.Yf = Y; a.ldy = ldy; a.accumulate = accumulate;
  a.M = M; a.K = K; a.N = N;
  return hgemm_np<HP_BF, HE_F32, false>(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ packing
// bf16 image of the packed transposed weights for hgemm: row rho(o) of the image holds out-feature o with
// rho = 32*(o/32) + 16*((o%8)/4) + 4*((o%32)/8) + (o%4)   (pairs of MFMA tiles, see the file header), row stride Kp
// (bf16 elements), zero padding is never written (the buffer is zero-initialised by the caller).
__global__ void __launch_bounds__(256) pack_batch_bf16_kernel(const clsr_pack_desc* __restrict__ descs) {
  const clsr_pack_desc d = descs[blockIdx.y];
  const int total = d.O * d.I;
  __bf16* dst = reinterpret_cast<__bf16*>(d.dst);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int o = e / d.I, i = e - o * d.I;
    float v = d.s1 * (d.transposed ? d.src1[(long)o * d.ld1 + i] : d.src1[(long)i * d.ld1 + o]);
    if (d.src2) v += d.s2 * (d.transposed ? d.src2[(long)o * d.ld2 + i] : d.src2[(long)i * d.ld2 + o]);
    const int oo = d.o0 + o, w = oo & 31;
    const int rho = (oo & ~31) + 16 * ((w & 7) >> 2) + 4 * (w >> 3) + (w & 3);
    dst[(long)rho * d.Kp + d.i0 + i] = (__bf16)v;
  }
}

extern "C" int clsr_pack_batch_bf16(const clsr_pack_desc* descs_device, int n, int max_elems, void* stream) {
  CLSR_CHECK_ARG(descs_device && n > 0 && max_elems > 0);
  int bx = clsr_cdiv(max_elems, 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(pack_batch_bf16_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, descs_device);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// fp32 <-> bf16 row copies (tests, and tensors that cross between the two storage modes)
__global__ void cvt_f2h_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dst[e] = (__bf16)src[e];
}
__global__ void cvt_h2f_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dst[e] = (float)src[e];
}
extern "C" int clsr_cvt_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  CLSR_CHECK_ARG(src && dst && n >= 0);
  if (n == 0) return CLSR_OK;
  int blocks = clsr_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cvt_f2h_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_cvt_bf16_to_f32(const void* src, float* dst, long n, void* stream) {
  CLSR_CHECK_ARG(src && dst && n >= 0);
  if (n == 0) return CLSR_OK;
  int blocks = clsr_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cvt_h2f_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
