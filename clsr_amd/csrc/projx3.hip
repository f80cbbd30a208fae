// Plain projection  Y[m, :N] = X[m, :K] . W + b  over M positions as split-bf16 products (gfx950): the K-fused time-gate
// projection of the Time4LSTM ([hist | TT] . [W_x ; W_t], K = 128 -> N = 120; reference rnn_cell_implement.py:214-236 --
// the input-side pre-activations of the o / T1 / T2 gates).  It sits on the step's dependent chain IN FRONT of the
// recurrences; the position-tiled fp32 kernel took 118 us for 6.3 GFLOP / 0.2 GB (256 fp32 MFMAs of 32 cycles per 16
// positions).  Here: A = the X tile (rows = positions, 8 consecutive features per lane and K = 32 chunk), B = bf16
// pieces of W from LDS, 96 bf16 MFMAs per 16 positions (two pieces per operand: the result feeds sigmoid gates, like the
// recurrences' own split products), a lane of the result holds four positions of one output feature and stores them as
// 4-byte pieces of 64-byte row segments (csrc/attl1fwd.hip).
#include <stdlib.h>
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

struct ProjArgs {
  const float* X; int ldx;
  const float* Wt; int Kp;        // packed W (clsr_pack_batch): row n = output feature (N rows), K inputs
  const float* bias;              // may be NULL
  float* Y; int ldy;
  int M, K, N;
  int acc;                        // != 0: Y += (a K slab of a wider product: K > 128 runs as slabs of 128 input features)
  // TT form (clsr_proj_x3_tt): X is the history embeddings [M, D]; the columns [col0, col0 + 2n) of the product's input are the
  // Time4LSTM's tanh time features, computed here from the two time scalars of the position (no [hist | TT] image in front
  // of the product: the launch leaves the dependent chain that starts the recurrences)
  const float* tnow; const float* tfirst; long row_stride; int T;
  const float* w1; const float* b1; const float* w2; const float* b2;
  int D, col0, n;
};

// NKC = 32-wide chunks of K, NT = 16-feature tiles of N, NP = bf16 pieces per operand
// Workgroups of 256 threads, or of 512 when the weight images leave room for only ONE workgroup per CU (three pieces of a
// 128 x 128 block: 104 KB): eight waves then share the image -- two per SIMD instead of one, which is what hides the latency
// of the X tile loads (one tile of prefetch) and of the stores.
template <int NKC, int NT, int NP, bool TTF = false>
__global__ void __launch_bounds__(512) proj_x3_kernel(ProjArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int WS = 32 * NKC + 8, NR = 16 * NT;
  const int nthr = blockDim.x, nwv = nthr >> 6;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  __bf16* Wi = reinterpret_cast<__bf16*>(lds_raw);          // [NP][NR][WS]
  float* ttab = reinterpret_cast<float*>(lds_raw + (size_t)NP * NR * WS * 2);      // TTF: [2][32 NKC] weight / bias of a time-feature column
  if (TTF) {
    for (int e = tid; e < 32 * NKC; e += nthr) {
      const int k = e - a.col0;
      float w = 0.f, b = 0.f;
      if (k >= 0 && k < a.n) { w = a.w1[k]; b = a.b1[k]; }
      else if (k >= a.n && k < 2 * a.n) { w = a.w2[k - a.n]; b = a.b2[k - a.n]; }
      ttab[e] = w; ttab[32 * NKC + e] = b;
    }
  }
  // wide outputs (the fused input projection of 128-wide encoders, N = 1 536): blockIdx.y owns a block of NR output
  // columns; X is re-read per block (L2 / Infinity Cache: 105 MB at configs[4]), every block writes its own columns
  const int n0 = blockIdx.y * NR;
  const int Nb = a.N - n0 < NR ? a.N - n0 : NR;
  {
    constexpr int C8 = WS / 8;
    for (int e = tid; e < NR * C8; e += nthr) {
      const int row = e / C8, k = 8 * (e - row * C8);
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < Nb && k < a.K) v = ld8f(a.Wt + (long)(n0 + row) * a.Kp + k);     // (K % 8 == 0)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const bf16x8 h = to_h(v);
        reinterpret_cast<bf16x8*>(Wi + (size_t)i * NR * WS)[e] = h;
        v -= to_f(h);
      }
    }
  }
  __syncthreads();
  float bias[NT];
  unsigned co[NT];
  constexpr unsigned SKIP = 0x40000000u;      // (out of range alone and in the sum of two: the resource spans exactly Y)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool ok = 16 * n + j < Nb;
    bias[n] = (ok && a.bias) ? a.bias[n0 + 16 * n + j] : 0.f;
    co[n] = ok ? (n0 + 16 * n + j) * 4u : SKIP;
  }
  const int wrow = j * WS + 8 * g;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ((unsigned)(a.M - 1) * (unsigned)a.ldy + (unsigned)a.N) * 4u, 0x00020000);
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int kofs[NKC];
  int kind[NKC];                // TTF: 0 = history columns (read), 1 = tanh(time_to_now ..), 2 = tanh(time_from_first ..), 3 = padding
#pragma unroll
  for (int c = 0; c < NKC; ++c) {
    const int k0 = 32 * c + 8 * g;
    kofs[c] = k0 < (TTF ? a.D : a.K) ? k0 : 0;
    kind[c] = !TTF ? 0 : k0 < a.D ? 0 : (k0 >= a.col0 && k0 < a.col0 + a.n) ? 1 : (k0 >= a.col0 + a.n && k0 < a.col0 + 2 * a.n) ? 2 : 3;
  }

  const int ntiles = (a.M + 15) >> 4;
  const int tstride = gridDim.x * nwv;
  int tile = blockIdx.x * nwv + wave;
  struct Raw { f32x8 x[NKC]; float tn, tf; };
  auto fetch = [&](int t) -> Raw {
    Raw r;
    const int m = t * 16 + j;
    const int mc = m < a.M ? m : a.M - 1;
    const float* p = a.X + (long)mc * a.ldx;
#pragma unroll
    for (int c = 0; c < NKC; ++c)
      if (!TTF || 32 * c < a.D) r.x[c] = ld8f(p + kofs[c]);        // (TTF: only the chunks that hold history columns; uniform)
    if (TTF) {
      const int h = mc / a.T, tt = mc - h * a.T;
      r.tn = a.tnow[(long)h * a.row_stride + tt];
      r.tf = a.tfirst[(long)h * a.row_stride + tt];
    }
    return r;
  };
  Raw cur = fetch(tile);
  for (; tile < ntiles; tile += tstride) {
    const Raw nxt = fetch(tile + tstride);
    const int m0 = tile * 16;
    const bool pv = m0 + j < a.M;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){bias[n], bias[n], bias[n], bias[n]};
    f32x4 old[NT];                // accumulate: the tile's present values, in flight underneath the products
    if (a.acc) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = m0 + 4 * g + e;
        const unsigned ro = p < a.M ? (unsigned)p * (unsigned)a.ldy * 4u : SKIP;
#pragma unroll
        for (int n = 0; n < NT; ++n)
          old[n][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, ro + co[n], 0, 0));
      }
    }
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      __builtin_amdgcn_sched_barrier(0);
      f32x8 y = (pv && 32 * c + 8 * g < a.K) ? cur.x[c] : z8;
      if (TTF) {
        // (same arithmetic as t4_time_inputs_fwd_kernel, csrc/rnn.hip, which still writes TT for the backward pass -- off the chain)
        if (32 * c >= a.D) y = z8;
        if (kind[c] != 0) {
          const float x = kind[c] == 1 ? cur.tn : cur.tf;
          const f32x8 w8 = ld8f(ttab + 32 * c + 8 * g), b8 = ld8f(ttab + 32 * NKC + 32 * c + 8 * g);
          f32x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tanhf_(x * w8[e] + b8[e]);
          y = (pv && kind[c] != 3) ? v : z8;
        } else if (!(pv && 32 * c + 8 * g < a.D)) {
          y = z8;
        }
      }
      bf16x8 xp[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        xp[i] = to_h(y);
        if (i + 1 < NP) y -= to_f(xp[i]);
      }
      // one weight piece in registers at a time (all NP pieces of the eight tiles were 96 VGPRs: with the accumulate form's
      // tile in flight the three-piece 128-column instance spilled); every piece product whose indices sum to <= NP - 1,
      // the smaller x pieces first
#pragma unroll
      for (int wp = NP - 1; wp >= 0; --wp) {
        bf16x8 w[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) w[n] = ld8h(Wi + (size_t)wp * NR * WS + wrow + 16 * n * WS + 32 * c);
#pragma unroll
        for (int i = NP - 1 - wp; i >= 0; --i)
#pragma unroll
          for (int n = 0; n < NT; ++n) HMFMA(acc[n], xp[i], w[n]);
      }
    }
    if (a.acc) {
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] += old[n];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = m0 + 4 * g + e;
      const unsigned ro = p < a.M ? (unsigned)p * (unsigned)a.ldy * 4u : SKIP;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float v = acc[n][e];      // (a copy: __builtin_bit_cast straight on the vector ELEMENT took element 0 for every e)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, ro + co[n], 0, 0);
      }
    }
    cur = nxt;
  }
}

extern "C" int clsr_proj_x3_supported(int M, int K, int N) {
  return M > 0 && K >= 8 && K <= 128 && K % 8 == 0 && N >= 4 && N <= 128 * 64 && N % 4 == 0;
}

template <int NKC, int NT, int NP, bool TTF = false>
static int proj_launch(const ProjArgs& a, hipStream_t stream) {
  constexpr int WS = 32 * NKC + 8, NR = 16 * NT;
  const size_t shmem = (size_t)NP * NR * WS * 2 + (TTF ? (size_t)2 * 32 * NKC * 4 : 0);
  const int threads = shmem > 80 * 1024 ? 512 : 256;         // (more than half of the CU's 160 KB: one workgroup per CU)
  int gx = clsr_cdiv(clsr_cdiv(a.M, 16), threads / 64);
  const int cap = threads == 512 ? 256 : 512;
  if (gx > cap) gx = cap;
  auto kernel = proj_x3_kernel<NKC, NT, NP, TTF>;
  if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(gx, clsr_cdiv(a.N, NR)), dim3(threads), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// pieces = 2: 2^-16 relative per product term; 3: 2^-23
static int proj_x3_any(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M,
                       int K, int N, int pieces, int accumulate, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && (pieces == 2 || pieces == 3));
  CLSR_CHECK_SUPPORTED(clsr_proj_x3_supported(M, K, N));
  CLSR_CHECK_ARG(ldx >= K && ldy >= N && Kp >= 16 * clsr_cdiv(K, 16));
  CLSR_CHECK_SUPPORTED(ldx % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Wt % 16) == 0 &&
                       ((uintptr_t)Y % 4) == 0 && (long)ldy * 4 * 16 < 0x40000000L);
  hipStream_t s = (hipStream_t)stream;
  const int nkc = clsr_cdiv(K, 32), nt = N <= 48 ? 3 : (N <= 80 ? 5 : 8);
  // the kernel addresses Y through a buffer resource with 32-bit byte offsets below 2^30: row ranges of at most that size
  // per launch (the history-replicated step projects 1M positions into a 480-column tensor: 1.97 GB)
  const long rows_max = ((0x40000000L - 1) / ((long)ldy * 4)) & ~15L;
  for (long m0 = 0; m0 < M; m0 += rows_max) {
    ProjArgs a = {};
    a.X = X + m0 * ldx; a.ldx = ldx; a.Wt = Wt; a.Kp = Kp; a.bias = bias; a.Y = Y + m0 * ldy; a.ldy = ldy;
    a.M = (int)(M - m0 < rows_max ? M - m0 : rows_max); a.K = K; a.N = N; a.acc = accumulate;
    int rc = CLSR_EUNSUPPORTED;
#define PJ_GO(C, T) \
    if (nkc == C && nt == T) rc = pieces == 2 ? proj_launch<C, T, 2>(a, s) : proj_launch<C, T, 3>(a, s)
    PJ_GO(1, 3); PJ_GO(2, 3); PJ_GO(3, 3); PJ_GO(4, 3);
    PJ_GO(1, 5); PJ_GO(2, 5); PJ_GO(3, 5); PJ_GO(4, 5);
    PJ_GO(1, 8); PJ_GO(2, 8); PJ_GO(3, 8); PJ_GO(4, 8);
#undef PJ_GO
    if (rc != CLSR_OK) return rc;
  }
  return CLSR_OK;
}

extern "C" int clsr_proj_x3(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M,
                            int K, int N, int pieces, void* stream) {
  return proj_x3_any(X, ldx, Wt, Kp, bias, Y, ldy, M, K, N, pieces, 0, stream);
}
// ---- K beyond 128 in ONE launch: the accumulators of a wave's position tiles stay in registers while the workgroup walks the
// K slabs of 128 input features -- per slab: the X pieces of the wave's tiles are requested, the workgroup re-stages the
// slab's weight block (bf16 pieces) in LDS between two barriers, then the products.  Against the chain of accumulating
// launches (clsr_proj_x3 per slab) the output is written once instead of read and written per slab, and d(hist) = dPin . W_x^T
// of the 128-wide encoders (K = 1 536: five K-range launches of clsr_pgemm3 before, each re-reading and re-writing the
// 105 MB output and 0.9 GB of dPin) reads dPin once.  TPW position tiles per wave: a slab's weight staging (64 KB from L2 +
// the splits) is shared by 8 x TPW tiles.
template <int NT, int NP, int TPW>
__global__ void __launch_bounds__(512) proj_x3_kloop_kernel(ProjArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NKC = 4, WS = 32 * NKC + 8, NR = 16 * NT, C8 = WS / 8;
  const int nthr = blockDim.x, nwv = nthr >> 6;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  __bf16* Wi = reinterpret_cast<__bf16*>(lds_raw);          // [NP][NR][WS]
  const int n0 = blockIdx.y * NR;
  const int Nb = a.N - n0 < NR ? a.N - n0 : NR;
  float bias[NT];
  unsigned co[NT];
  constexpr unsigned SKIP = 0x40000000u;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool ok = 16 * n + j < Nb;
    bias[n] = (ok && a.bias) ? a.bias[n0 + 16 * n + j] : 0.f;
    co[n] = ok ? (n0 + 16 * n + j) * 4u : SKIP;
  }
  const int wrow = j * WS + 8 * g;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ((unsigned)(a.M - 1) * (unsigned)a.ldy + (unsigned)a.N) * 4u, 0x00020000);
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ntiles = (a.M + 15) >> 4;
  const int ngroups = (ntiles + nwv * TPW - 1) / (nwv * TPW);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {      // (block-uniform: every wave meets every barrier)
    const int tile0 = (grp * nwv + wave) * TPW;
    f32x4 acc[TPW][NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){bias[n], bias[n], bias[n], bias[n]};
    for (int k0 = 0; k0 < a.K; k0 += 32 * NKC) {
      // the X pieces of this slab: in flight underneath the weight staging
      f32x8 xr[TPW][NKC];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int m = (tile0 + t) * 16 + j;
        const float* p = a.X + (long)(m < a.M ? m : a.M - 1) * a.ldx;
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
          const int k = k0 + 32 * c + 8 * g;
          xr[t][c] = ld8f(p + (k < a.K ? k : 0));
        }
      }
      __syncthreads();       // the previous slab's products have read their weights
      for (int e = tid; e < NR * C8; e += nthr) {
        const int row = e / C8, k = 8 * (e - row * C8);
        f32x8 v = z8;
        if (row < Nb && k < 32 * NKC && k0 + k < a.K) v = ld8f(a.Wt + (long)(n0 + row) * a.Kp + k0 + k);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const bf16x8 h = to_h(v);
          reinterpret_cast<bf16x8*>(Wi + (size_t)i * NR * WS)[e] = h;
          v -= to_f(h);
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 xp[TPW][NP];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const bool pv = (tile0 + t) * 16 + j < a.M;
          f32x8 y = (pv && k0 + 32 * c + 8 * g < a.K) ? xr[t][c] : z8;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            xp[t][i] = to_h(y);
            if (i + 1 < NP) y -= to_f(xp[t][i]);
          }
        }
#pragma unroll
        for (int wp = NP - 1; wp >= 0; --wp) {
          bf16x8 w[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) w[n] = ld8h(Wi + (size_t)wp * NR * WS + wrow + 16 * n * WS + 32 * c);
#pragma unroll
          for (int i = NP - 1 - wp; i >= 0; --i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
              for (int n = 0; n < NT; ++n) HMFMA(acc[t][n], xp[t][i], w[n]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int m0 = (tile0 + t) * 16;
      unsigned ro[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = m0 + 4 * g + e;
        ro[e] = p < a.M ? (unsigned)p * (unsigned)a.ldy * 4u : SKIP;
      }
      if (a.acc) {
        f32x4 old[NT];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            old[n][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, ro[e] + co[n], 0, 0));
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] += old[n];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float v = acc[t][n][e];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, ro[e] + co[n], 0, 0);
        }
    }
  }
}

template <int NT, int NP>
static int proj_kloop_launch(const ProjArgs& a, hipStream_t stream) {
  constexpr int TPW = 2, WS = 32 * 4 + 8, NR = 16 * NT;
  const size_t shmem = (size_t)NP * NR * WS * 2;
  const int threads = 512;
  int gx = clsr_cdiv(clsr_cdiv(a.M, 16), (threads / 64) * TPW);
  const int cap = shmem > 80 * 1024 ? 256 : 512;
  if (gx > cap) gx = cap;
  auto kernel = proj_x3_kloop_kernel<NT, NP, TPW>;
  if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(gx, clsr_cdiv(a.N, NR)), dim3(threads), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static int proj_x3_kloop(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M,
                         int K, int N, int pieces, int accumulate, void* stream) {
  CLSR_CHECK_ARG(X && Wt && Y && (pieces == 2 || pieces == 3));
  CLSR_CHECK_ARG(ldx >= K && ldy >= N && Kp >= 16 * clsr_cdiv(K, 16));
  CLSR_CHECK_SUPPORTED(ldx % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Wt % 16) == 0 &&
                       ((uintptr_t)Y % 4) == 0 && (long)ldy * 4 * 16 < 0x40000000L);
  hipStream_t s = (hipStream_t)stream;
  const int nt = N <= 48 ? 3 : (N <= 80 ? 5 : 8);
  const long rows_max = ((0x40000000L - 1) / ((long)ldy * 4)) & ~15L;
  for (long m0 = 0; m0 < M; m0 += rows_max) {
    ProjArgs a = {};
    a.X = X + m0 * ldx; a.ldx = ldx; a.Wt = Wt; a.Kp = Kp; a.bias = bias; a.Y = Y + m0 * ldy; a.ldy = ldy;
    a.M = (int)(M - m0 < rows_max ? M - m0 : rows_max); a.K = K; a.N = N; a.acc = accumulate;
    int rc = CLSR_EUNSUPPORTED;
#define PK_GO(T) if (nt == T) rc = pieces == 2 ? proj_kloop_launch<T, 2>(a, s) : proj_kloop_launch<T, 3>(a, s)
    PK_GO(3); PK_GO(5); PK_GO(8);
#undef PK_GO
    if (rc != CLSR_OK) return rc;
  }
  return CLSR_OK;
}

// K of any width (K % 8 == 0): slabs of 128 input features, the second and later ones accumulating into Y (launches on ONE
// stream: each reads what the one before wrote).  Kp = row stride of the packed weights (>= K rounded up to 16).
extern "C" int clsr_proj_x3_wide_supported(int M, int K, int N) {
  return M > 0 && K >= 8 && K % 8 == 0 && N >= 4 && N <= 128 * 64 && N % 4 == 0;
}
// accumulate != 0: Y += X . W + b (the first slab accumulates as well)
extern "C" int clsr_proj_x3_wide(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M,
                                 int K, int N, int pieces, int accumulate, void* stream) {
  CLSR_CHECK_SUPPORTED(clsr_proj_x3_wide_supported(M, K, N));
  if (K > 128) return proj_x3_kloop(X, ldx, Wt, Kp, bias, Y, ldy, M, K, N, pieces, accumulate, stream);
  for (int k0 = 0; k0 < K; k0 += 128) {
    const int kw = K - k0 < 128 ? K - k0 : 128;
    int rc = proj_x3_any(X + k0, ldx, Wt + k0, Kp, k0 ? nullptr : bias, Y, ldy, M, kw, N, pieces, (k0 || accumulate) ? 1 : 0, stream);
    if (rc != CLSR_OK) return rc;
  }
  return CLSR_OK;
}

// ------------------------------------------------------------------------------------------------ encoder tail
// Back-projections of the encoders' input-side gradients from ONE pass over dPin [M, NX] (gfx950, split-bf16 products):
//   dhist[m, :D]  += dPin[m, :NX] . W_x^T                              (d of the history embeddings: every encoder's share)
//   dTT[m, :2H]    = dPin[m, tcol0 : tcol0 + 3H] . W_t^T               (d of the Time4LSTM's tanh time features)
// (reference: tf.gradients through the input side of GRUCell / Time4LSTMCell, rnn_cell_implement.py:214-236.)  They were two
// position-tiled launches (clsr_pgemm3 134 us + clsr_pgemm 71 us alone) that each read the 393 MB of dPin beside the fused
// weight-gradient kernel, which reads it a third time: the tail of the step is bound by those reads.  A = the dPin tile
// (rows = positions; the K = 32 chunks of a row are loaded up front), B = bf16 pieces of W_x^T (all chunks) and of W_t^T
// (the chunks that overlap the time-gate columns) from LDS; result lanes = output features (csrc/attl1fwd.hip).
struct EncBackArgs {
  const float* dPin; int ldp;
  const float* WxT; int Kpx;      // packed W_x^T (clsr_pack_batch): row d = history feature (D rows), K = NX
  const float* WtT; int Kpt;      // packed W_t^T: row n = time feature (2H rows), K = 3H
  float* dhist; int ldh;
  float* dTT; int ldt;
  int M, NX, D, H2, H3, tcol0;
};

#define EB_TC 5      // K = 32 chunks the time-gate columns can overlap (3H <= 128 at any 8-aligned offset)

template <int NKC, int ND, int NT>      // chunks of NX, tiles of D, tiles of 2H
__global__ void __launch_bounds__(512, 1) enc_back_x3_kernel(EncBackArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int WSX = 32 * NKC + 8, WST = 32 * EB_TC + 8, DR = 16 * ND, TR = 16 * NT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  __bf16* Xi = reinterpret_cast<__bf16*>(lds_raw);        // [2][DR][WSX]
  __bf16* Ti = Xi + 2 * DR * WSX;                         // [2][TR][WST]: column k of the image = dPin column 32 c0 + k
  const int c0 = a.tcol0 >> 5;                            // first chunk that overlaps the time-gate columns
  {
    constexpr int C8 = WSX / 8;
    for (int e = tid; e < DR * C8; e += 512) {
      const int row = e / C8, k = 8 * (e - row * C8);
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < a.D && k < a.NX) v = ld8f(a.WxT + (long)row * a.Kpx + k);          // (NX % 8 == 0)
      const bf16x8 h = to_h(v);
      reinterpret_cast<bf16x8*>(Xi)[e] = h;
      reinterpret_cast<bf16x8*>(Xi + DR * WSX)[e] = to_h(v - to_f(h));
    }
    constexpr int T8 = WST / 8;
    for (int e = tid; e < TR * T8; e += 512) {
      const int row = e / T8, k = 8 * (e - row * T8);
      const int kl = 32 * c0 + k - a.tcol0;               // column of W_t^T (tcol0 % 8 == 0: whole groups of 8)
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < a.H2 && kl >= 0 && kl < a.H3) v = ld8f(a.WtT + (long)row * a.Kpt + kl);   // (3H % 8 == 0)
      const bf16x8 h = to_h(v);
      reinterpret_cast<bf16x8*>(Ti)[e] = h;
      reinterpret_cast<bf16x8*>(Ti + TR * WST)[e] = to_h(v - to_f(h));
    }
  }
  __syncthreads();
  constexpr unsigned SKIP = 0x40000000u;
  unsigned ho[ND], to_[NT];
#pragma unroll
  for (int n = 0; n < ND; ++n) ho[n] = 16 * n + j < a.D ? (16 * n + j) * 4u : SKIP;
#pragma unroll
  for (int n = 0; n < NT; ++n) to_[n] = 16 * n + j < a.H2 ? (16 * n + j) * 4u : SKIP;
  const int xrow = j * WSX + 8 * g, trow = j * WST + 8 * g;
  const __amdgpu_buffer_rsrc_t rh =
      __builtin_amdgcn_make_buffer_rsrc(a.dhist, 0, ((unsigned)(a.M - 1) * (unsigned)a.ldh + (unsigned)a.D) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rt =
      __builtin_amdgcn_make_buffer_rsrc(a.dTT, 0, ((unsigned)(a.M - 1) * (unsigned)a.ldt + (unsigned)a.H2) * 4u, 0x00020000);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int kofs[NKC];
#pragma unroll
  for (int c = 0; c < NKC; ++c) kofs[c] = 32 * c + 8 * g < a.NX ? 32 * c + 8 * g : 0;
  const int c1 = min(NKC, (a.tcol0 + a.H3 + 31) >> 5);     // one past the last chunk with time-gate columns

  const int ntiles = (a.M + 15) >> 4;
  for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
    const int m0 = tile * 16;
    const int m = m0 + j;
    const bool pv = m < a.M;
    const float* p = a.dPin + (long)(pv ? m : a.M - 1) * a.ldp;
    f32x8 x[NKC];
#pragma unroll
    for (int c = 0; c < NKC; ++c) x[c] = ld8f(p + kofs[c]);
    // d(hist) as it stands, in the result layout (4 positions of feature 16 n + j)
    unsigned roh[4], rot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = m0 + 4 * g + e;
      roh[e] = q < a.M ? (unsigned)q * (unsigned)a.ldh * 4u : SKIP;
      rot[e] = q < a.M ? (unsigned)q * (unsigned)a.ldt * 4u : SKIP;
    }
    f32x4 ah[ND], at[NT];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ah[n][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, roh[e] + ho[n], 0, 0));
#pragma unroll
    for (int n = 0; n < NT; ++n) at[n] = z4;
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      __builtin_amdgcn_sched_barrier(0);
      f32x8 y = (pv && 32 * c + 8 * g < a.NX) ? x[c] : z8;
      const bf16x8 xh = to_h(y);
      const bf16x8 xl = to_h(y - to_f(xh));
      bf16x8 wh[ND], wl[ND];
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        wh[n] = ld8h(Xi + xrow + 16 * n * WSX + 32 * c);
        wl[n] = ld8h(Xi + DR * WSX + xrow + 16 * n * WSX + 32 * c);
      }
#pragma unroll
      for (int n = 0; n < ND; ++n) HMFMA(ah[n], xh, wl[n]);
#pragma unroll
      for (int n = 0; n < ND; ++n) HMFMA(ah[n], xl, wh[n]);
#pragma unroll
      for (int n = 0; n < ND; ++n) HMFMA(ah[n], xh, wh[n]);
      if (c >= c0 && c < c1) {       // (uniform)
        const int ct = c - c0;
        bf16x8 th[NT], tl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          th[n] = ld8h(Ti + trow + 16 * n * WST + 32 * ct);
          tl[n] = ld8h(Ti + TR * WST + trow + 16 * n * WST + 32 * ct);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) HMFMA(at[n], xh, tl[n]);
#pragma unroll
        for (int n = 0; n < NT; ++n) HMFMA(at[n], xl, th[n]);
#pragma unroll
        for (int n = 0; n < NT; ++n) HMFMA(at[n], xh, th[n]);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        const float v = ah[n][e];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rh, roh[e] + ho[n], 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float v = at[n][e];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rt, rot[e] + to_[n], 0, 0);
      }
    }
  }
}

extern "C" int clsr_enc_back_x3_supported(int M, int NX, int D, int H2, int H3, int tcol0) {
  return M > 0 && NX >= 8 && NX <= 512 && NX % 8 == 0 && D >= 4 && D <= 48 && D % 4 == 0 && H2 >= 4 && H2 <= 80 &&
         H2 % 4 == 0 && H3 >= 8 && H3 <= 128 && H3 % 8 == 0 && tcol0 >= 0 && tcol0 % 8 == 0 && tcol0 + H3 <= NX &&
         ((tcol0 + H3 + 31) >> 5) - (tcol0 >> 5) <= EB_TC && (long)M * NX * 4 < 0x7fffffffL;
}

template <int NKC, int ND, int NT>
static int enc_back_launch(const EncBackArgs& a, hipStream_t stream) {
  constexpr int WSX = 32 * NKC + 8, WST = 32 * EB_TC + 8;
  const size_t shmem = (size_t)2 * 16 * ND * WSX * 2 + (size_t)2 * 16 * NT * WST * 2;
  int gx = clsr_cdiv(clsr_cdiv(a.M, 16), 8);
  if (gx > 256) gx = 256;
  auto kernel = enc_back_x3_kernel<NKC, ND, NT>;
  if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(gx), dim3(512), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_enc_back_x3(const float* dPin, int ldp, const float* WxT, int Kpx, const float* WtT, int Kpt, int tcol0,
                                float* dhist, int ldh, float* dTT, int ldt, int M, int NX, int D, int H2, int H3,
                                void* stream) {
  CLSR_CHECK_ARG(dPin && WxT && WtT && dhist && dTT);
  CLSR_CHECK_SUPPORTED(clsr_enc_back_x3_supported(M, NX, D, H2, H3, tcol0));
  CLSR_CHECK_ARG(ldp >= NX && ldh >= D && ldt >= H2 && Kpx >= 16 * clsr_cdiv(NX, 16) && Kpt >= 16 * clsr_cdiv(H3, 16));
  CLSR_CHECK_SUPPORTED(ldp % 4 == 0 && Kpx % 4 == 0 && Kpt % 4 == 0 && ((uintptr_t)dPin % 16) == 0 && ((uintptr_t)WxT % 16) == 0 &&
                       ((uintptr_t)WtT % 16) == 0 && (long)M * ldh * 4 < 0x40000000L && (long)M * ldt * 4 < 0x40000000L);
  EncBackArgs a = {};
  a.dPin = dPin; a.ldp = ldp; a.WxT = WxT; a.Kpx = Kpx; a.WtT = WtT; a.Kpt = Kpt; a.dhist = dhist; a.ldh = ldh; a.dTT = dTT;
  a.ldt = ldt; a.M = M; a.NX = NX; a.D = D; a.H2 = H2; a.H3 = H3; a.tcol0 = tcol0;
  hipStream_t s = (hipStream_t)stream;
  const int nkc = clsr_cdiv(NX, 32), nd = 3, nt = H2 <= 48 ? 3 : 5;
#define EB_GO(C) \
  if (nkc <= C) return nt == 3 ? enc_back_launch<C, 3, 3>(a, s) : enc_back_launch<C, 3, 5>(a, s)
  EB_GO(4); EB_GO(8); EB_GO(12); EB_GO(15); EB_GO(16);
#undef EB_GO
  (void)nd;
  clsr_set_error("%s:%d: no instance for NX = %d", __FILE__, __LINE__, NX);
  return CLSR_EUNSUPPORTED;
}

// The Time4LSTM's time-gate projection without its [hist | TT] input image:  Y = [hist | 0 | tanh(t_now w1 + b1) | tanh(t_first w2 + b2)] . W + b
// with the 2n time-feature columns computed in the prologue from the position's two time scalars (reference
// rnn_cell_implement.py:200-236).  K = col0 + 2n <= 128, D % 8 == 0, col0 % 8 == 0, n % 8 == 0.
extern "C" int clsr_proj_x3_tt_supported(int M, int D, int col0, int n, int N) {
  return M > 0 && D >= 8 && D % 8 == 0 && col0 >= D && col0 % 8 == 0 && n >= 8 && n % 8 == 0 && col0 + 2 * n <= 128 &&
         N >= 4 && N <= 128 && N % 4 == 0;
}
extern "C" int clsr_proj_x3_tt(const float* hist, int D, const float* tnow, const float* tfirst, long row_stride, int T,
                               const float* w1, const float* b1, const float* w2, const float* b2, int n, int col0,
                               const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M, int N, int pieces,
                               void* stream) {
  CLSR_CHECK_ARG(hist && tnow && tfirst && w1 && b1 && w2 && b2 && Wt && Y && (pieces == 2 || pieces == 3) && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_proj_x3_tt_supported(M, D, col0, n, N));
  const int K = col0 + 2 * n;
  CLSR_CHECK_ARG(ldy >= N && Kp >= 16 * clsr_cdiv(K, 16));
  CLSR_CHECK_SUPPORTED(Kp % 4 == 0 && ((uintptr_t)hist % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)Y % 4) == 0 &&
                       (long)ldy * 4 * 16 < 0x40000000L);
  hipStream_t s = (hipStream_t)stream;
  const int nkc = clsr_cdiv(K, 32), nt = N <= 48 ? 3 : (N <= 80 ? 5 : 8);
  const long rows_max = ((0x40000000L - 1) / ((long)ldy * 4)) / T * T;       // whole histories per launch (m -> (h, t) inside)
  CLSR_CHECK_SUPPORTED(rows_max >= T);
  for (long m0 = 0; m0 < M; m0 += rows_max) {
    ProjArgs a = {};
    a.X = hist + m0 * D; a.ldx = D; a.Wt = Wt; a.Kp = Kp; a.bias = bias; a.Y = Y + m0 * ldy; a.ldy = ldy;
    a.M = (int)(M - m0 < rows_max ? M - m0 : rows_max); a.K = K; a.N = N; a.acc = 0;
    a.tnow = tnow + (m0 / T) * row_stride; a.tfirst = tfirst + (m0 / T) * row_stride; a.row_stride = row_stride; a.T = T;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.D = D; a.col0 = col0; a.n = n;
    int rc = CLSR_EUNSUPPORTED;
#define PJT_GO(C, T_) \
    if (nkc == C && nt == T_) rc = pieces == 2 ? proj_launch<C, T_, 2, true>(a, s) : proj_launch<C, T_, 3, true>(a, s)
    PJT_GO(1, 3); PJT_GO(2, 3); PJT_GO(3, 3); PJT_GO(4, 3);
    PJT_GO(1, 5); PJT_GO(2, 5); PJT_GO(3, 5); PJT_GO(4, 5);
    PJT_GO(1, 8); PJT_GO(2, 8); PJT_GO(3, 8); PJT_GO(4, 8);
#undef PJT_GO
    if (rc != CLSR_OK) return rc;
  }
  return CLSR_OK;
}
