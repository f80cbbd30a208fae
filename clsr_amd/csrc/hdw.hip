// Weight gradients on the bf16 matrix pipe (gfx950) -- the SPEED-mode counterpart of pgemm_dw (csrc/linear.hip):
//   dW[k, n] = sum_m f(X)[xrow(m), k] * dY[m, n],   db[n] = sum_m dY[m, n]
// for the layers of _fcn_net / _attention_fcn / the recurrent cells' input- and hidden-side kernels
// (reference models/base_model.py:627-708, models/sequential/clsr.py:343-381, rnn_cell_implement.py:207-231, through
// tf.gradients).  The contraction runs over POSITIONS: v_mfma_f32_16x16x32_bf16 with
//   A = X^T tile  [16 features k][32 positions],   B = dY tile [32 positions][16 features n],
// both read from LDS images that are stored TRANSPOSED ([feature][position], bf16), so that a lane's 8 consecutive
// positions are one ds_read_b128.  The fp32 kernel spends its time on the fp32 matrix pipe (13 GFLOP per million
// positions at 80 x 80: 123 us at 106 TFLOP/s); here the 16x faster pipe leaves the kernel bound by its operand reads.
// Operands are rounded to bf16 (8 mantissa bits) when staged, products are exact, accumulation is fp32 -- the rounding
// errors of ~1e6 independent products average out in the sum; fp32 mode keeps the exact kernel.
//
// Work split: a workgroup (4 waves) walks 64-position stages; every wave owns up to 7 of the 25 (k-tile, n-tile)
// accumulators of an 80 x 80 chunk for ALL positions -- no cross-wave reduction at the end; partial chunks are
// written in pgemm_dw's layout and summed by the same clsr_dw_reduce_batch launch.
#include "common.h"
#include "clsr_hip.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

#define HDW_T 5
#define HDW_W (16 * HDW_T)                       // 80 features per chunk
#define HDW_CHUNK (HDW_T * HDW_T * 256 + HDW_T * 16)
#define HDW_LD 72                                // bf16 per LDS row (64 positions + 8 pad; 16-byte aligned rows)
#define HDW_TPW 7                                // accumulator tiles per wave (4 x 7 >= 25)

struct HdwArgs {
  const void* X; int ldx; int T; int G; const float* Xmul; int ldmul;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* dY; int ldy;
  float* partial;
  int M, K, N;
};

template <bool H> struct RawPiece { typedef f32x4 type; };
template <> struct RawPiece<true> { typedef bf16x4v type; };
template <bool H>
__device__ __forceinline__ typename RawPiece<H>::type load_raw(const void* base, int off) {
  if constexpr (H) return *reinterpret_cast<const bf16x4v*>(reinterpret_cast<const __bf16*>(base) + off);
  else return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ f32x4 to_f32(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 to_f32(bf16x4v v) { return __builtin_convertvector(v, f32x4); }

// (bx, gx) = this block's index / the number of blocks along the positions, (by, bz, gz) = its K / N chunk: blockIdx and
// gridDim for a launch of its own, a job's share of the grid in a multi-job launch (hdw_multi_kernel)
struct HdwLds {
  __attribute__((aligned(16))) __bf16 Xs[HDW_W * HDW_LD];
  __attribute__((aligned(16))) __bf16 Ys[HDW_W * HDW_LD];
  __attribute__((aligned(16))) float aff[2][HDW_W];
};
template <int MODE, bool XH, bool YH>   // MODE 0 plain, 1 X * Xmul[r], 2 relu?(X * in_scale + in_shift)
__device__ __forceinline__ void hdw_body(const HdwArgs& a, HdwLds& L, const int bx, const int gx, const int by,
                                         const int bz, const int gz) {
  __bf16* Xs = L.Xs;
  __bf16* Ys = L.Ys;
  float (*aff)[HDW_W] = L.aff;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile ids / column chunks live in SGPRs
  const int j = lane & 15, g = lane >> 4;
  const int k0 = by * HDW_W, n0 = bz * HDW_W;
  const int ktc = min(HDW_T, ((a.K + 15) >> 4) - by * HDW_T);
  const int ntc = min(HDW_T, ((a.N + 15) >> 4) - bz * HDW_T);
  const int ntile = ktc * ntc;
  const int kmax4 = ((a.K + 3) & ~3) - 4, nmax4 = a.N - 4;

  // this wave's accumulator tiles: ids wave, wave + 4, ... (kt = id / ntc, nt = id % ntc)
  f32x4 acc[HDW_TPW];
  int tk[HDW_TPW], tn[HDW_TPW];
#pragma unroll
  for (int u = 0; u < HDW_TPW; ++u) {
    acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int id = wave + 4 * u;
    tk[u] = id < ntile ? id / ntc : -1;
    tn[u] = id < ntile ? id - (id / ntc) * ntc : 0;
  }
  if (MODE == 2) {
    for (int e = tid; e < 2 * HDW_W; e += 256) {
      const int which = e / HDW_W, c = e - which * HDW_W;
      const int kc = min(k0 + c, a.K - 1);
      aff[which][c] = which ? a.in_shift[kc] : a.in_scale[kc];
    }
  }

  // staging: the position is the fast index across lanes (conflict-free transposed LDS writes): thread -> row = tid & 63,
  // 16-byte column chunks c4 = (tid >> 6) + 4 * p  (p < 5: 20 chunks of 4 features).  Raw loads are kept as loaded
  // (bf16 pieces: 2 registers) and converted / masked when they are written to LDS, after the MFMAs of the stage before
  const int row = tid & 63;
  constexpr int NLD = 5;
  typename RawPiece<XH>::type xr[NLD];
  typename RawPiece<YH>::type yr[NLD];
  f32x4 mr[MODE == 1 ? NLD : 1];
  f32x4 bsum[NLD];     // exact fp32 column sums of dY (the bias gradient): this thread's row, its five column chunks
#pragma unroll
  for (int p = 0; p < NLD; ++p) bsum[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool mv = false;
  const int ntiles = (a.M + 63) >> 6;
  auto fetch = [&](int tile) {
    int m = tile * 64 + row;
    mv = (tile < ntiles) && (m < a.M);
    m = m < a.M ? m : a.M - 1;
    int xrow = m, r = m;
    if (a.T > 0) {
      const unsigned rr = (unsigned)m / (unsigned)a.T;
      r = (int)rr;
      if (a.G > 0) xrow = (int)(rr / (unsigned)a.G) * a.T + (m - (int)rr * a.T);
    }
    const int xo = xrow * a.ldx, yo = m * a.ldy, mo_ = MODE == 1 ? r * a.ldmul : 0;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int c4 = wave + 4 * p;
      const int kcol = min(k0 + 4 * c4, kmax4), ncol = min(n0 + 4 * c4, nmax4);
      xr[p] = load_raw<XH>(a.X, xo + kcol);
      if (MODE == 1) mr[p] = ld4(a.Xmul + mo_ + kcol);
      yr[p] = load_raw<YH>(a.dY, yo + ncol);
    }
  };
  auto stage = [&]() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int c4 = wave + 4 * p;
      f32x4 xv = to_f32(xr[p]);
      if (MODE == 1) xv *= mr[p];
      if (MODE == 2) {
        xv = xv * ld4(&aff[0][4 * c4]) + ld4(&aff[1][4 * c4]);
        if (a.in_relu) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
      }
      xv = (mv && k0 + 4 * c4 <= kmax4) ? xv : z;
      const f32x4 yv = (mv && n0 + 4 * c4 <= nmax4) ? to_f32(yr[p]) : z;
      bsum[p] += yv;
      const bf16x4v xh = __builtin_convertvector(xv, bf16x4v), yh = __builtin_convertvector(yv, bf16x4v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Xs[(4 * c4 + e) * HDW_LD + row] = xh[e];
        Ys[(4 * c4 + e) * HDW_LD + row] = yh[e];
      }
    }
  };

  fetch(bx);
  for (int tile = bx; tile < ntiles; tile += gx) {
    __syncthreads();   // previous stage fully consumed (and, first time, the affine table written)
    stage();
    __syncthreads();
    fetch(tile + gx);   // the next stage's global loads fly behind the MFMAs below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int mo = 32 * s + 8 * g;
#pragma unroll
      for (int u = 0; u < HDW_TPW; ++u) {
        if (tk[u] >= 0) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(Xs + (16 * tk[u] + j) * HDW_LD + mo);
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Ys + (16 * tn[u] + j) * HDW_LD + mo);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[u], 0, 0, 0);
        }
      }
    }
  }

  // partial chunk in pgemm_dw's layout: tile (kt, nt) element (k = 4g + r, n = j) at (kt*5 + nt)*256 + (4g + r)*16 + j
  const long chunk = (long)by * gz + bz;
  float* dst = a.partial + (chunk * gx + bx) * HDW_CHUNK;
#pragma unroll
  for (int u = 0; u < HDW_TPW; ++u) {
    if (tk[u] >= 0) {
      float* t = dst + (tk[u] * HDW_T + tn[u]) * 256 + (4 * g) * 16 + j;
      t[0] = acc[u].x; t[16] = acc[u].y; t[32] = acc[u].z; t[48] = acc[u].w;
    }
  }
  // bias row: the 64 lanes of a wave hold the 64 rows of the same column chunks -> wave sums
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    f32x4 v = bsum[p];
    v.x = wave_sum(v.x); v.y = wave_sum(v.y); v.z = wave_sum(v.z); v.w = wave_sum(v.w);
    if (lane == 0) st4(dst + HDW_T * HDW_T * 256 + 4 * (wave + 4 * p), v);
  }
}

template <int MODE, bool XH, bool YH>
__global__ void __launch_bounds__(256, 3) hdw_kernel(HdwArgs a) {
  __shared__ HdwLds L;
  hdw_body<MODE, XH, YH>(a, L, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several weight gradients in ONE launch (see dw_multi_kernel in linear.hip): job j owns blocks [first[j], first[j+1]).
#define HDWM_MAX 12
struct HdwMultiArgs {
  HdwArgs d[HDWM_MAX];
  int first[HDWM_MAX + 1];
  int gx[HDWM_MAX];
  short nch[HDWM_MAX];
  short key[HDWM_MAX];    // mode * 4 + 2 * x_bf16 + dy_bf16
  int n;
};
__global__ void __launch_bounds__(256, 3) hdw_multi_kernel(HdwMultiArgs m) {
  __shared__ HdwLds L;
  int j = 0;
  while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
  const int local = blockIdx.x - m.first[j];
  const int gx = m.gx[j], nch = m.nch[j];
  const int bx = local % gx, c = local / gx;
  const int by = c / nch, bz = c - by * nch;
  switch (m.key[j]) {   // (the operand combinations of the encoders' weight gradients: fp32 activations, fp32 or bf16 dY)
    case 0: hdw_body<0, false, false>(m.d[j], L, bx, gx, by, bz, nch); break;
    case 1: hdw_body<0, false, true>(m.d[j], L, bx, gx, by, bz, nch); break;
    case 4: hdw_body<1, false, false>(m.d[j], L, bx, gx, by, bz, nch); break;
    case 5: hdw_body<1, false, true>(m.d[j], L, bx, gx, by, bz, nch); break;
    default: hdw_body<2, false, false>(m.d[j], L, bx, gx, by, bz, nch); break;
  }
}

static int hdw_grid_x(int M);
// number of partial chunks per 80 x 80 block that clsr_hdw_partial(_multi) writes (the descriptor of the batched reduction)
extern "C" int clsr_hdw_parts(int M) { return hdw_grid_x(M); }
static int hdw_grid_x(int M) {
  int tiles = clsr_cdiv(M, 64);
  int gx = clsr_cdiv(tiles, 4);
  constexpr int cap = 384;   // blocks per chunk (512: 35 us more per speed-mode step in partial-sum traffic; 256: too few waves)
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return gx;
}

// Same contract as clsr_pgemm_dw_partial (workspace floats = clsr_pgemm_dw_workspace_floats(M, K, N), partial count =
// clsr_hdw_parts(M) <= clsr_pgemm_dw_parts(M), reduced by clsr_dw_reduce_batch); X / dY are fp32 or bf16 (x_bf16 / dy_bf16).
extern "C" int clsr_hdw_partial(const void* X, int x_bf16, int ldx, int T, int G, const float* Xmul, int ldmul,
                                const float* in_scale, const float* in_shift, int in_relu, const void* dY,
                                int dy_bf16, int ldy, int M, int K, int N, float* workspace, void* stream) {
  CLSR_CHECK_ARG(X && dY && workspace && M >= 0 && K > 0 && N > 0);
  CLSR_CHECK_ARG(!(in_scale && !in_shift));
  CLSR_CHECK_SUPPORTED(N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= ((K + 3) & ~3) &&
                       (!Xmul || (ldmul % 4 == 0 && ldmul >= ((K + 3) & ~3))) &&
                       ((uintptr_t)X % 8) == 0 && ((uintptr_t)dY % 8) == 0);
  CLSR_CHECK_SUPPORTED(!(Xmul && in_scale) && !(in_scale && K % 4));
  HdwArgs a;
  a.X = X; a.ldx = ldx; a.T = T; a.G = G; a.Xmul = Xmul; a.ldmul = ldmul;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.dY = dY; a.ldy = ldy; a.partial = workspace; a.M = M; a.K = K; a.N = N;
  const dim3 grid(hdw_grid_x(M), clsr_cdiv(K, HDW_W), clsr_cdiv(N, HDW_W));
  hipStream_t s = (hipStream_t)stream;
  const int mode = Xmul ? 1 : (in_scale ? 2 : 0);
  const int key = mode * 4 + (x_bf16 ? 2 : 0) + (dy_bf16 ? 1 : 0);
#define HDW_LAUNCH(MD, XB, YB) hipLaunchKernelGGL((hdw_kernel<MD, XB, YB>), grid, dim3(256), 0, s, a)
  switch (key) {
    case 0: HDW_LAUNCH(0, false, false); break;
    case 1: HDW_LAUNCH(0, false, true); break;
    case 2: HDW_LAUNCH(0, true, false); break;
    case 3: HDW_LAUNCH(0, true, true); break;
    case 4: HDW_LAUNCH(1, false, false); break;
    case 5: HDW_LAUNCH(1, false, true); break;
    case 8: HDW_LAUNCH(2, false, false); break;
    case 11: HDW_LAUNCH(2, true, true); break;
    default:
      clsr_set_error("%s:%d: unsupported operand combination for clsr_hdw_partial (mode %d, x_bf16 %d, dy_bf16 %d)",
                     __FILE__, __LINE__, mode, x_bf16, dy_bf16);
      return CLSR_EUNSUPPORTED;
  }
#undef HDW_LAUNCH
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// n weight-gradient partial products in one launch (fp32 X, fp32 or bf16 dY; same partial layout as clsr_hdw_partial)
extern "C" int clsr_hdw_partial_multi(const clsr_dwjob* jobs, int n, void* stream) {
  CLSR_CHECK_ARG(jobs && n > 0 && n <= HDWM_MAX);
  HdwMultiArgs m;
  m.n = n;
  int total = 0;
  for (int j = 0; j < n; ++j) {
    const clsr_dwjob& q = jobs[j];
    CLSR_CHECK_ARG(q.X && q.dY && q.workspace && q.M > 0 && q.K > 0 && q.N > 0 && !(q.in_scale && !q.in_shift));
    CLSR_CHECK_SUPPORTED(!q.x_bf16 && !(q.dy_bf16 && q.in_scale));
    CLSR_CHECK_SUPPORTED(q.rm_tc == 0 && q.pstride == 0 && q.pgx == 0);   // (time-range jobs: fp32 kernels only so far)
    CLSR_CHECK_SUPPORTED(q.N % 4 == 0 && q.ldx % 4 == 0 && q.ldy % 4 == 0 && q.ldx >= ((q.K + 3) & ~3) &&
                         (!q.Xmul || (q.ldmul % 4 == 0 && q.ldmul >= ((q.K + 3) & ~3))) &&
                         ((uintptr_t)q.X % 16) == 0 && ((uintptr_t)q.dY % 8) == 0);
    CLSR_CHECK_SUPPORTED(!(q.Xmul && q.in_scale) && !(q.in_scale && q.K % 4));
    HdwArgs& a = m.d[j];
    a.X = q.X; a.ldx = q.ldx; a.T = q.T; a.G = q.G; a.Xmul = q.Xmul; a.ldmul = q.ldmul;
    a.in_scale = q.in_scale; a.in_shift = q.in_shift; a.in_relu = q.in_relu;
    a.dY = q.dY; a.ldy = q.ldy; a.partial = q.workspace; a.M = q.M; a.K = q.K; a.N = q.N;
    const int kch = clsr_cdiv(q.K, HDW_W), nch = clsr_cdiv(q.N, HDW_W);
    m.first[j] = total;
    m.gx[j] = hdw_grid_x(q.M);
    m.nch[j] = (short)nch;
    m.key[j] = (short)((q.Xmul ? 1 : (q.in_scale ? 2 : 0)) * 4 + (q.dy_bf16 ? 1 : 0));
    total += m.gx[j] * kch * nch;
  }
  m.first[n] = total;
  hipLaunchKernelGGL(hdw_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, m);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
