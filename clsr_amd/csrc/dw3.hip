// Weight gradients as SPLIT-bf16 (bf16 x 3) products on the bf16 matrix pipe (gfx950) -- the "fp32x3" twin of pgemm_dw
// (csrc/linear.hip):
//   dW[k, n] = sum_m f(X)[xrow(m), k] * dY[m, n],   db[n] = sum_m dY[m, n]
// for the layers of _fcn_net / _attention_fcn / the recurrent cells' input- and hidden-side kernels
// (reference models/base_model.py:627-708, models/sequential/clsr.py:343-381, rnn_cell_implement.py:207-231, through
// tf.gradients).  Operands stay fp32 in HBM.  When a 64-position stage is written to LDS every value v is split into
//   hi = bf16(v),  lo = bf16(v - hi)          (v = hi + lo up to 2^-17 |v|)
// and the product of a pair is taken as  hi*hi + lo*hi + hi*lo  -- three v_mfma_f32_16x16x32_bf16 with fp32
// accumulation; the dropped lo*lo term and the rounding of lo are <= 2^-16 relative per product, i.e. the same order
// as the fp32 rounding of the accumulation itself.  Three bf16 MFMAs cover K = 32 in 48 cycles; the fp32-input MFMA
// needs 8 x 32 = 256 cycles for the same K and blocks the issue port of its SIMD while it runs
// (profiles/r03_rnn_pmc.md): the exact kernels sat at 0.35-0.48 of the 157 TFLOP/s fp32 matrix peak with the pipe
// 63-70 % busy and slowed every kernel beside them (profiles/r03_contention.md).  This kernel is bound by its operand
// reads (2 x 4 bytes per position and feature: 655 MB per million positions at 80 x 80).
//
// Work split.  A workgroup (5 waves) walks 64-position stages of one 80 x 80 chunk (5 x 5 MFMA tiles).  Wave w owns
// ONE tile row (k-tile w, all n-tiles) -- or one tile column when the chunk has more n-tiles than k-tiles -- for ALL
// positions: 2 + 2 x 5 operand fetches feed 15 MFMAs, and no cross-wave reduction is needed at the end.  The LDS images
// and the transpose reads that feed the MFMAs are described at Dw3Lds below.  Partial chunks leave in pgemm_dw's
// layout and are summed by the same clsr_dw_reduce_batch launch (deterministic: fixed block -> tile assignment, fixed
// summation order of the bias row, no float atomics).
#include "common.h"
#include "clsr_hip.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define D3_T 5
#define D3_W (16 * D3_T)                         // 80 features per chunk
#define D3_CHUNK (D3_T * D3_T * 256 + D3_T * 16)
#define D3_RB 160                                // bytes per position row of an LDS image (80 bf16; 40 dwords = 8 x 5:
                                                 // the eight rows a half-wave's transpose read touches tile the 64 banks)
#define D3_IMG (64 * D3_RB)                      // bytes per image (64 positions)
#define D3_NT 320                                // threads per workgroup (5 waves)

struct Dw3Args {
  const float* X; int ldx; int T; int G; const float* Xmul; int ldmul;
  const float* in_scale; const float* in_shift; int in_relu;
  const float* dY; int ldy;
  float* partial;
  int M, K, N;
};

// LDS: four POSITION-major images [64 positions][80 features] bf16 -- X hi, X lo, dY hi, dY lo.  They are written with
// 8-byte stores by threads that walk the rows of the operands along the features (every global load instruction covers
// whole contiguous row segments; a position-fast mapping -- one row per lane -- asks for 16 bytes out of 64 different rows
// per instruction and the lines are fetched up to eight times through L2: the first version of this kernel ran at 3.3
// TB/s because of it) and read through ds_read_b64_tr_b16, the LDS transpose read of gfx950: the 16 lanes of a group
// pass the addresses of a [4 positions][16 features] block and each receives one feature's four positions; two such
// reads are one MFMA operand (k-slots = positions 4g..4g+3 and 16+4g..16+4g+3 of a 32-position half; A and B agree).
struct Dw3Lds {
  __attribute__((aligned(16))) unsigned char img[4 * D3_IMG];
  __attribute__((aligned(16))) float aff[2][D3_W];
  __attribute__((aligned(16))) float bred[D3_NT * 4];
};
typedef __attribute__((address_space(3))) unsigned char d3_lds_t;
typedef short d3_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 d3_tr2(d3_lds_t* p) {
  const d3_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) d3_s16x4*>(p));
  const d3_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) d3_s16x4*>(p + 16 * D3_RB));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}
// 4 fp32 -> 4 bf16 hi, 4 bf16 lo (two 8-byte words)
__device__ __forceinline__ void d3_split4(const f32x4& v, unsigned long long& hi, unsigned long long& lo) {
  unsigned h[2], l[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2v));
    const float ah = __builtin_bit_cast(float, hp << 16), bh = __builtin_bit_cast(float, hp & 0xffff0000u);
    h[p] = hp;
    l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a - ah, b - bh}, bf16x2v));
  }
  hi = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
  lo = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
}

#define MFMA_BF(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (acc), 0, 0, 0)

// threads per row for a chunk of `cpr` 16-byte pieces: the smallest divisor of 320 that is >= cpr (rows per pass = 320 / it)
__device__ __forceinline__ int d3_tpr(int cpr) {
  return cpr <= 1 ? 1 : cpr <= 2 ? 2 : cpr <= 4 ? 4 : cpr <= 5 ? 5 : cpr <= 8 ? 8 : cpr <= 10 ? 10 : cpr <= 16 ? 16 : 20;
}

// (bx, gx) = this block's index / the number of blocks along the positions, (by, bz, gz) = its K / N chunk
// DEPTH: stages whose raw loads are in flight (1 | 2)
template <int MODE, int DEPTH>   // MODE 0 plain, 1 X * Xmul[r], 2 relu?(X * in_scale + in_shift)
__device__ __forceinline__ void dw3_body(const Dw3Args& a, Dw3Lds& L, const int bx, const int gx, const int by,
                                         const int bz, const int gz) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const int k0 = by * D3_W, n0 = bz * D3_W;
  const int ktc = min(D3_T, ((a.K + 15) >> 4) - by * D3_T);
  const int ntc = min(D3_T, ((a.N + 15) >> 4) - bz * D3_T);
  // 16-byte pieces per row of this chunk of X / dY (K is padded to a multiple of 4 by its leading dimension)
  const int cprX = min(D3_W, ((a.K + 3) & ~3) - k0) >> 2, cprY = min(D3_W, a.N - n0) >> 2;
  // tile ownership: a row of the chunk (fixed k-tile) or a column (fixed n-tile), whichever dimension has more tiles
  const bool own_k = ktc >= ntc;
  const int oc = own_k ? ntc : ktc;                          // tiles this wave accumulates
  const bool active = wave < (own_k ? ktc : ntc);
  d3_lds_t* const img = (d3_lds_t*)L.img;
  // this lane's transpose-read base inside an image: row 4 g + (c16 >> 2), 8-byte column piece c16 & 3
  const int tro = (4 * g + (c16 >> 2)) * D3_RB + (c16 & 3) * 8;
  d3_lds_t* const fixh = img + (own_k ? 0 : 2 * D3_IMG) + tro + 32 * wave;
  d3_lds_t* const varh = img + (own_k ? 2 * D3_IMG : 0) + tro;

  f32x4 acc[D3_T];
#pragma unroll
  for (int u = 0; u < D3_T; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int e = tid; e < 4 * D3_IMG / 16; e += D3_NT) reinterpret_cast<f32x4*>(L.img)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (MODE == 2) {
    for (int e = tid; e < 2 * D3_W; e += D3_NT) {
      const int which = e / D3_W, c = e - which * D3_W;
      const int kc = min(k0 + c, a.K - 1);
      L.aff[which][c] = which ? a.in_shift[kc] : a.in_scale[kc];
    }
  }

  // staging plan: thread = (piece cX of the row, row group rgX); it handles the rows rgX + rpX * i of a stage, i < 4
  const int tprX = d3_tpr(cprX), tprY = d3_tpr(cprY);
  const int rpX = D3_NT / tprX, rpY = D3_NT / tprY;
  const int rgX = tid / tprX, cX = tid - rgX * tprX, rgY = tid / tprY, cY = tid - rgY * tprY;
  const bool useX = cX < cprX, useY = cY < cprY;
  const int colX = k0 + 4 * (useX ? cX : 0), colY = n0 + 4 * (useY ? cY : 0);
  struct Raw { f32x4 x[4], y[4], m[MODE == 1 ? 4 : 1]; };
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};        // exact fp32 column sums of dY (bias gradient): this thread's rows, piece cY
  const int ntiles = (a.M + 63) >> 6;
  const unsigned Tu = a.T > 0 ? (unsigned)a.T : 1u, Gu = a.G > 0 ? (unsigned)a.G : 1u;
  auto fetch = [&](Raw& R, int tile) {
    const int m0 = tile * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = rgX + rpX * i, py = rgY + rpY * i;
      if (px < 64) {
        int m = min(m0 + px, a.M - 1);
        m = m < 0 ? 0 : m;
        int xrow = m, r = m;
        if (a.T > 0) {
          const unsigned q_ = (unsigned)m / Tu;
          r = (int)q_;
          if (a.G > 0) xrow = (int)(q_ / Gu) * a.T + (m - (int)q_ * a.T);
        }
        R.x[i] = ld4(a.X + (long)xrow * a.ldx + colX);
        if (MODE == 1) R.m[i] = ld4(a.Xmul + (long)r * a.ldmul + colX);
      }
      if (py < 64) {
        int m = min(m0 + py, a.M - 1);
        m = m < 0 ? 0 : m;
        R.y[i] = ld4(a.dY + (long)m * a.ldy + colY);
      }
    }
  };
  auto stage = [&](const Raw& R, int tile) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int m0 = tile * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = rgX + rpX * i, py = rgY + rpY * i;
      if (px < 64 && useX) {
        f32x4 v = R.x[i];
        if (MODE == 1) v *= R.m[i];
        if (MODE == 2) {
          v = v * ld4(&L.aff[0][4 * cX]) + ld4(&L.aff[1][4 * cX]);
          if (a.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        unsigned long long hi, lo;
        d3_split4(m0 + px < a.M ? v : z, hi, lo);
        d3_lds_t* d = img + px * D3_RB + 8 * cX;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d) = hi;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d + D3_IMG) = lo;
      }
      if (py < 64 && useY) {
        const f32x4 v = m0 + py < a.M ? R.y[i] : z;
        bsum += v;
        unsigned long long hi, lo;
        d3_split4(v, hi, lo);
        d3_lds_t* d = img + 2 * D3_IMG + py * D3_RB + 8 * cY;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d) = hi;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(d + D3_IMG) = lo;
      }
    }
  };
  auto mfmas = [&]() {
    if (active) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int so = 32 * s * D3_RB;
        const bf16x8 fh = d3_tr2(fixh + so), fl = d3_tr2(fixh + D3_IMG + so);
#pragma unroll
        for (int u = 0; u < D3_T; ++u) {
          if (u < oc) {
            const bf16x8 vh = d3_tr2(varh + so + 32 * u), vl = d3_tr2(varh + D3_IMG + so + 32 * u);
            MFMA_BF(acc[u], fl, vh);     // small terms first
            MFMA_BF(acc[u], fh, vl);
            MFMA_BF(acc[u], fh, vh);
          }
        }
      }
    }
  };

  if (DEPTH == 2) {
    Raw r0, r1;
    fetch(r0, bx);
    fetch(r1, bx + gx);
    for (int tile = bx; tile < ntiles; tile += 2 * gx) {
      __syncthreads();   // previous stage fully consumed (and, first time, the zero fill / affine table written)
      stage(r0, tile);
      __syncthreads();
      fetch(r0, tile + 2 * gx);
      mfmas();
      if (tile + gx < ntiles) {      // (block-uniform)
        __syncthreads();
        stage(r1, tile + gx);
        __syncthreads();
        fetch(r1, tile + 3 * gx);
        mfmas();
      }
    }
  } else {
    Raw r0;
    fetch(r0, bx);
    for (int tile = bx; tile < ntiles; tile += gx) {
      __syncthreads();
      stage(r0, tile);
      __syncthreads();
      fetch(r0, tile + gx);   // the next stage's global loads fly behind the MFMAs below
      mfmas();
    }
  }

  // partial chunk in pgemm_dw's layout: tile (kt, nt) element (k, n) at (kt*5 + nt)*256 + k*16 + n
  const long chunk = (long)by * gz + bz;
  float* dst = a.partial + (chunk * gx + bx) * D3_CHUNK;
  if (active) {
#pragma unroll
    for (int u = 0; u < D3_T; ++u) {
      if (u < oc) {
        if (own_k) {          // rows of the MFMA result = k (4g + r), columns = n (c16)
          float* t = dst + (wave * D3_T + u) * 256 + (4 * g) * 16 + c16;
          t[0] = acc[u].x; t[16] = acc[u].y; t[32] = acc[u].z; t[48] = acc[u].w;
        } else {              // transposed product: rows = n (4g + r), columns = k (c16)
          st4(dst + (u * D3_T + wave) * 256 + c16 * 16 + 4 * g, acc[u]);
        }
      }
    }
  }
  // bias row: the threads cY, cY + tprY, ... hold the row groups of column piece cY -- summed in a fixed order
  __syncthreads();
  st4(&L.bred[4 * tid], bsum);
  __syncthreads();
  if (tid < D3_W) {
    const int c = tid >> 2, e = tid & 3;
    float t = 0.f;
    if (c < cprY)
      for (int k = 0; k < rpY; ++k) t += L.bred[4 * (c + tprY * k) + e];
    dst[D3_T * D3_T * 256 + tid] = t;
  }
}

#ifndef D3_DEPTH
#define D3_DEPTH 2
#endif
#define D3_DEPTH_OF(MODE) ((MODE) == 1 ? 1 : D3_DEPTH)   // (the X * Xmul form holds a third operand per stage: two stages spill)
template <int MODE>
__global__ void __launch_bounds__(D3_NT, 3) dw3_kernel(Dw3Args a) {
  __shared__ Dw3Lds L;
  dw3_body<MODE, D3_DEPTH_OF(MODE)>(a, L, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several weight gradients in ONE launch (see dw_multi_kernel in linear.hip): job j owns blocks [first[j], first[j+1]).
#define D3M_MAX 12
struct Dw3MultiArgs {
  Dw3Args d[D3M_MAX];
  int first[D3M_MAX + 1];
  int gx[D3M_MAX];
  short nch[D3M_MAX];
  short mode[D3M_MAX];
  int n;
};
__global__ void __launch_bounds__(D3_NT, 3) dw3_multi_kernel(Dw3MultiArgs m) {
  __shared__ Dw3Lds L;
  int jn = 0;
  while (jn + 1 < m.n && (int)blockIdx.x >= m.first[jn + 1]) ++jn;
  const int local = blockIdx.x - m.first[jn];
  const int gx = m.gx[jn], nch = m.nch[jn];
  const int bx = local % gx, c = local / gx;
  const int by = c / nch, bz = c - by * nch;
  switch (m.mode[jn]) {
    case 0: dw3_body<0, D3_DEPTH_OF(0)>(m.d[jn], L, bx, gx, by, bz, nch); break;
    case 1: dw3_body<1, D3_DEPTH_OF(1)>(m.d[jn], L, bx, gx, by, bz, nch); break;
    default: dw3_body<2, D3_DEPTH_OF(2)>(m.d[jn], L, bx, gx, by, bz, nch); break;
  }
}

static int dw3_grid_x(int M) {
  int tiles = clsr_cdiv(M, 64);
  int gx = clsr_cdiv(tiles, 4);
  static const int cap = getenv("CLSR_DW3_PARTS") ? atoi(getenv("CLSR_DW3_PARTS")) : 512;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return gx;
}
// number of partial chunks per 80 x 80 block that clsr_dw3_partial(_multi) writes (<= clsr_pgemm_dw_parts(M): the
// workspace of clsr_pgemm_dw_workspace_floats fits)
extern "C" int clsr_dw3_parts(int M) { return dw3_grid_x(M); }

static int dw3_check(const void* X, int ldx, const float* Xmul, int ldmul, const float* in_scale,
                     const float* in_shift, const void* dY, int ldy, int M, int K, int N, const void* workspace) {
  CLSR_CHECK_ARG(X && dY && workspace && M > 0 && K > 0 && N > 0);
  CLSR_CHECK_ARG(!(in_scale && !in_shift));
  CLSR_CHECK_SUPPORTED(N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= ((K + 3) & ~3) &&
                       (!Xmul || (ldmul % 4 == 0 && ldmul >= ((K + 3) & ~3))) &&
                       ((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0 && (!Xmul || ((uintptr_t)Xmul % 16) == 0));
  CLSR_CHECK_SUPPORTED(!(Xmul && in_scale) && !(in_scale && K % 4));
  return CLSR_OK;
}

// Same contract as clsr_pgemm_dw_partial (workspace floats = clsr_pgemm_dw_workspace_floats(M, K, N), partial count =
// clsr_dw3_parts(M), reduced by clsr_dw_reduce_batch); fp32 operands, split-bf16 products.
extern "C" int clsr_dw3_partial(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                                const float* in_scale, const float* in_shift, int in_relu, const float* dY, int ldy,
                                int M, int K, int N, float* workspace, void* stream) {
  const int rc = dw3_check(X, ldx, Xmul, ldmul, in_scale, in_shift, dY, ldy, M, K, N, workspace);
  if (rc != CLSR_OK) return rc;
  Dw3Args a;
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.dY = dY; a.ldy = ldy; a.partial = workspace; a.M = M; a.K = K; a.N = N;
  const dim3 grid(dw3_grid_x(M), clsr_cdiv(K, D3_W), clsr_cdiv(N, D3_W));
  hipStream_t s = (hipStream_t)stream;
  if (Xmul) hipLaunchKernelGGL((dw3_kernel<1>), grid, dim3(D3_NT), 0, s, a);
  else if (in_scale) hipLaunchKernelGGL((dw3_kernel<2>), grid, dim3(D3_NT), 0, s, a);
  else hipLaunchKernelGGL((dw3_kernel<0>), grid, dim3(D3_NT), 0, s, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// n weight-gradient partial products in one launch (fp32 operands; same partial layout as clsr_dw3_partial)
extern "C" int clsr_dw3_partial_multi(const clsr_dwjob* jobs, int n, void* stream) {
  CLSR_CHECK_ARG(jobs && n > 0 && n <= D3M_MAX);
  Dw3MultiArgs m;
  m.n = n;
  int total = 0;
  for (int jn = 0; jn < n; ++jn) {
    const clsr_dwjob& q = jobs[jn];
    CLSR_CHECK_SUPPORTED(!q.x_bf16 && !q.dy_bf16);
    CLSR_CHECK_SUPPORTED(q.rm_tc == 0 && q.pstride == 0 && q.pgx == 0);   // (time-range jobs: fp32-MFMA kernels only)
    const int rc = dw3_check(q.X, q.ldx, q.Xmul, q.ldmul, q.in_scale, q.in_shift, q.dY, q.ldy, q.M, q.K, q.N, q.workspace);
    if (rc != CLSR_OK) return rc;
    Dw3Args& a = m.d[jn];
    a.X = (const float*)q.X; a.ldx = q.ldx; a.T = q.T; a.G = q.G; a.Xmul = q.Xmul; a.ldmul = q.ldmul;
    a.in_scale = q.in_scale; a.in_shift = q.in_shift; a.in_relu = q.in_relu;
    a.dY = (const float*)q.dY; a.ldy = q.ldy; a.partial = q.workspace; a.M = q.M; a.K = q.K; a.N = q.N;
    const int kch = clsr_cdiv(q.K, D3_W), nch = clsr_cdiv(q.N, D3_W);
    m.first[jn] = total;
    m.gx[jn] = dw3_grid_x(q.M);
    m.nch[jn] = (short)nch;
    m.mode[jn] = (short)(q.Xmul ? 1 : (q.in_scale ? 2 : 0));
    total += m.gx[jn] * kch * nch;
  }
  m.first[n] = total;
  hipLaunchKernelGGL(dw3_multi_kernel, dim3(total), dim3(D3_NT), 0, (hipStream_t)stream, m);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
