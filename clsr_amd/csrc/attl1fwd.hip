// Second attention layer, forward (gfx950):  z1[m, :] = relu(z0[m, :] * scale0 + shift0) . W1 + b1   + the per-feature sum /
// sum of squares of z1 for its batch-norm (reference _fcn_net, base_model.py:664-679; clsr.py:371-377).
// The position-tiled fp32 kernel it replaces (pgemm_fast_kernel with the BN-ReLU prologue) issues 60 fp32 MFMAs of 32 cycles
// per 16 positions and ran at 1.4 x its HBM floor.  Here the product is taken over THREE bf16 pieces per operand (x = p0 + p1 +
// p2, every piece product whose indices sum to <= 2: 2^-23 relative, the level of an fp32 fma chain -- this is a FORWARD
// product in front of a ReLU, see csrc/atthist.hip): 54 bf16 MFMAs of ~17 cycles per 16 positions.  Operands swapped like
// the backward kernels (csrc/attbwdx3.hip): A = the normalised z0 tile (rows = positions), B = W1 pieces from LDS, so a lane
// of the result holds four positions of ONE z1 feature: the bias is a per-lane register, the batch-norm sums are per-lane
// accumulators, z1 leaves as 4-byte stores of 64-byte row pieces.
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

struct L1FwdArgs {
  const void* z0; int ldz0;       // z0 / z1: float, or __bf16 (the kernel's ST: the speed mode's storage)
  const float* scale0; const float* shift0;
  const float* Wt; int Kp;        // packed W1 (clsr_pack_batch): row n = z1 feature (C1 rows), K = C0
  const float* bias;
  void* z1; int ldz1;
  double* stats;                  // [gridDim.x][2][C1] per-block partial sums (NULL: none)
  int M, C0, C1;
};

typedef __amdgpu_buffer_rsrc_t l1f_rsrc_t;

template <typename ST> struct L1fRaw;
template <> struct L1fRaw<float> { f32x8 v; };
template <> struct L1fRaw<__bf16> { bf16x8 v; };
__device__ __forceinline__ f32x8 l1f_f8(const L1fRaw<float>& r) { return r.v; }
__device__ __forceinline__ f32x8 l1f_f8(const L1fRaw<__bf16>& r) { return to_f(r.v); }
__device__ __forceinline__ L1fRaw<float> l1f_ld(const float* p) { L1fRaw<float> r; r.v = ld8f(p); return r; }
__device__ __forceinline__ L1fRaw<__bf16> l1f_ld(const __bf16* p) { L1fRaw<__bf16> r; r.v = ld8h(p); return r; }
__device__ __forceinline__ void l1f_st(float, l1f_rsrc_t rs, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
}
__device__ __forceinline__ void l1f_st(__bf16, l1f_rsrc_t rs, unsigned off, float v) {
  const __bf16 h = (__bf16)v;
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rs, off, 0, 0);
}

// NKC = 32-wide chunks of C0, NT = 16-feature tiles of C1, NP = bf16 pieces per operand (3: parity mode; 1: speed mode)
template <int NKC, int NT, int NP, typename ST>
__global__ void __launch_bounds__(256, 2) att_l1_fwd_kernel(L1FwdArgs a) {
  CLSR_CHAIN_PRIO();
  constexpr unsigned SB = sizeof(ST);
  const ST* z0p = reinterpret_cast<const ST*>(a.z0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int KCP = 32 * NKC, WS = KCP + 8, NR = 16 * NT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  __bf16* Wi = reinterpret_cast<__bf16*>(lds_raw);                                   // [NP][NR][WS]
  float* tab = reinterpret_cast<float*>(lds_raw + (size_t)NP * NR * WS * 2);         // [2][KCP]: scale0, shift0
  {
    constexpr int C8 = WS / 8;
    for (int e = tid; e < NR * C8; e += 256) {
      const int row = e / C8, k = 8 * (e - row * C8);
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < a.C1 && k < a.C0) v = ld8f(a.Wt + (long)row * a.Kp + k);            // (C0 % 8 == 0)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const bf16x8 h = to_h(v);
        reinterpret_cast<bf16x8*>(Wi + (size_t)i * NR * WS)[e] = h;
        v -= to_f(h);
      }
    }
    for (int e = tid; e < 2 * KCP; e += 256) {
      const int which = e / KCP, k = e - which * KCP;
      tab[e] = k < a.C0 ? (which ? a.shift0[k] : a.scale0[k]) : 0.f;
    }
  }
  __syncthreads();

  float bias[NT];
  unsigned co[NT];
  constexpr unsigned SKIP = 0x40000000u;      // (out of range alone and in the sum of two: the resource spans exactly z1)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool ok = 16 * n + j < a.C1;
    bias[n] = ok ? a.bias[16 * n + j] : 0.f;
    co[n] = ok ? (16 * n + j) * SB : SKIP;
  }
  const int wrow = j * WS + 8 * g;
  const l1f_rsrc_t rz1 = __builtin_amdgcn_make_buffer_rsrc(a.z1, 0, (unsigned)a.M * (unsigned)a.ldz1 * SB, 0x00020000);
  float s1[NT], s2[NT];
  double d1[NT], d2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { s1[n] = 0.f; s2[n] = 0.f; d1[n] = 0.0; d2[n] = 0.0; }
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int ntiles = (a.M + 31) >> 5;
  const int tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  struct Raw { L1fRaw<ST> x[2][NKC]; };
  auto fetch = [&](int t) -> Raw {
    Raw r;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int m = t * 32 + 16 * s + j;
      const long mr = m < a.M ? m : a.M - 1;
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        const int k0 = 32 * c + 8 * g;
        r.x[s][c] = l1f_ld(z0p + mr * a.ldz0 + (k0 < a.C0 ? k0 : 0));
      }
    }
    return r;
  };
  Raw cur = fetch(tile);
  int pending = 0;
  for (; tile < ntiles; tile += tstride) {
    const Raw nxt = fetch(tile + tstride);
    const int m0 = tile * 32;
    f32x4 acc[2][NT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[s][n] = (f32x4){bias[n], bias[n], bias[n], bias[n]};
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      __builtin_amdgcn_sched_barrier(0);
      const int k0 = 32 * c + 8 * g;
      const f32x8 sc = ld8f(tab + k0), sh = ld8f(tab + KCP + k0);      // (features beyond C0: 0, 0 -> relu(0) = 0)
      bf16x8 xp[2][NP];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x8 y = l1f_f8(cur.x[s][c]) * sc + sh;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
        if (m0 + 16 * s + j >= a.M) y = z8;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          xp[s][i] = to_h(y);
          if (i + 1 < NP) y -= to_f(xp[s][i]);
        }
      }
      bf16x8 w[NP][NT];
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) w[i][n] = ld8h(Wi + (size_t)i * NR * WS + wrow + 16 * n * WS + 32 * c);
#pragma unroll
      for (int sidx = NP - 1; sidx >= 0; --sidx)
#pragma unroll
        for (int i = 0; i <= sidx; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            HMFMA(acc[0][n], xp[0][i], w[sidx - i][n]);
            HMFMA(acc[1][n], xp[1][i], w[sidx - i][n]);
          }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = m0 + 16 * s + 4 * g + e;
        const bool pv = p < a.M;
        const unsigned ro = pv ? (unsigned)p * (unsigned)a.ldz1 * SB : SKIP;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = acc[s][n][e];
          if constexpr (sizeof(ST) == 2) v = (float)(__bf16)v;      // (bf16 storage: statistics of the STORED values)
          l1f_st(ST(), rz1, ro + co[n], v);
          const float vv = pv ? v : 0.f;
          s1[n] += vv;
          s2[n] = fmaf(vv, vv, s2[n]);
        }
      }
    if (++pending == 2) {       // fp32 per-lane partials of at most 16 values between flushes
#pragma unroll
      for (int n = 0; n < NT; ++n) { d1[n] += (double)s1[n]; d2[n] += (double)s2[n]; s1[n] = 0.f; s2[n] = 0.f; }
      pending = 0;
    }
    cur = nxt;
  }
  if (a.stats) {
#pragma unroll
    for (int n = 0; n < NT; ++n) { d1[n] += (double)s1[n]; d2[n] += (double)s2[n]; }
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds_raw);     // [16 = wave * 4 + g][2][NR]
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      red[((wave * 4 + g) * 2 + 0) * NR + 16 * n + j] = d1[n];
      red[((wave * 4 + g) * 2 + 1) * NR + 16 * n + j] = d2[n];
    }
    __syncthreads();
    for (int e = tid; e < 2 * NR; e += 256) {
      const int which = e / NR, c = e - which * NR;
      if (c < a.C1) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[(w * 2 + which) * NR + c];
        a.stats[((long)blockIdx.x * 2 + which) * a.C1 + c] = t;
      }
    }
  }
}

static int l1f_grid(int M) {
  int gx = clsr_cdiv(clsr_cdiv(M, 32), 4);
  if (gx > 512) gx = 512;
  return gx < 1 ? 1 : gx;
}

extern "C" int clsr_att_l1_fwd_supported(int C0, int C1) {
  return C0 >= 8 && C0 <= 96 && C0 % 8 == 0 && C1 >= 4 && C1 <= 48 && C1 % 4 == 0;
}
extern "C" int clsr_att_l1_fwd_stats_parts(int M) { return l1f_grid(M); }

template <int NKC, int NT, int NP, typename ST>
static int l1f_launch(const L1FwdArgs& a, hipStream_t stream) {
  constexpr int WS = 32 * NKC + 8, NR = 16 * NT;
  size_t shmem = (size_t)NP * NR * WS * 2 + (size_t)2 * 32 * NKC * 4;
  const size_t red = (size_t)16 * 2 * NR * 8;
  if (shmem < red) shmem = red;
  auto kernel = att_l1_fwd_kernel<NKC, NT, NP, ST>;
  if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(l1f_grid(a.M)), dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static int att_l1_fwd_any(const void* z0, bool half, int ldz0, const float* scale0, const float* shift0, const float* Wt, int Kp,
                          const float* bias, void* z1, int ldz1, double* stats, int M, int C0, int C1, void* stream) {
  CLSR_CHECK_ARG(z0 && scale0 && shift0 && Wt && bias && z1 && M > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l1_fwd_supported(C0, C1));
  CLSR_CHECK_ARG(ldz0 >= C0 && ldz1 >= C1 && Kp >= 16 * clsr_cdiv(C0, 16));
  CLSR_CHECK_SUPPORTED(ldz0 % (half ? 8 : 4) == 0 && Kp % 4 == 0 && ((uintptr_t)z0 % 16) == 0 && ((uintptr_t)Wt % 16) == 0 &&
                       ((uintptr_t)z1 % 4) == 0 && ((long)M * ldz1) * 4 < 0x40000000L);
  L1FwdArgs a = {};
  a.z0 = z0; a.ldz0 = ldz0; a.scale0 = scale0; a.shift0 = shift0; a.Wt = Wt; a.Kp = Kp; a.bias = bias; a.z1 = z1;
  a.ldz1 = ldz1; a.stats = stats; a.M = M; a.C0 = C0; a.C1 = C1;
  hipStream_t s = (hipStream_t)stream;
  const int nkc = clsr_cdiv(C0, 32), nt = C1 <= 32 ? 2 : 3;
#define L1F_GO(K, N) if (nkc == K && nt == N) return half ? l1f_launch<K, N, 1, __bf16>(a, s) : l1f_launch<K, N, 3, float>(a, s)
  L1F_GO(1, 2); L1F_GO(1, 3); L1F_GO(2, 2); L1F_GO(2, 3); L1F_GO(3, 2); L1F_GO(3, 3);
#undef L1F_GO
  return CLSR_OK;
}

extern "C" int clsr_att_l1_fwd(const float* z0, int ldz0, const float* scale0, const float* shift0, const float* Wt, int Kp,
                               const float* bias, float* z1, int ldz1, double* stats, int M, int C0, int C1, void* stream) {
  return att_l1_fwd_any(z0, false, ldz0, scale0, shift0, Wt, Kp, bias, z1, ldz1, stats, M, C0, C1, stream);
}
// speed mode: z0 / z1 stored as bf16 (uint16 bit patterns), ONE bf16 piece per operand; batch-norm sums of the stored (rounded) values
extern "C" int clsr_att_l1_fwd_x1_h(const void* z0, int ldz0, const float* scale0, const float* shift0, const float* Wt, int Kp,
                                    const float* bias, void* z1, int ldz1, double* stats, int M, int C0, int C1, void* stream) {
  return att_l1_fwd_any(z0, true, ldz0, scale0, shift0, Wt, Kp, bias, z1, ldz1, stats, M, C0, C1, stream);
}
