// Shared device/host helpers for the CLSR gfx950 kernels.
// Target: MI355X (gfx950, CDNA4) only -- 64-wide wavefronts, fp32 MFMA 16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CLSR_OK 0
#define CLSR_EINVAL (-1)
#define CLSR_ELAUNCH (-2)
#define CLSR_EUNSUPPORTED (-3)

void clsr_set_error(const char* fmt, ...);

#define CLSR_CHECK_ARG(cond)                                                              \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      clsr_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond);           \
      return CLSR_EINVAL;                                                                 \
    }                                                                                     \
  } while (0)

#define CLSR_CHECK_SUPPORTED(cond)                                                        \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      clsr_set_error("%s:%d: unsupported shape: %s", __FILE__, __LINE__, #cond);          \
      return CLSR_EUNSUPPORTED;                                                           \
    }                                                                                     \
  } while (0)

#define CLSR_CHECK_LAUNCH()                                                               \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) {                                                               \
      clsr_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__,               \
                     hipGetErrorString(e_));                                              \
      return CLSR_ELAUNCH;                                                                \
    }                                                                                     \
  } while (0)

#define CLSR_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      clsr_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,                   \
                     hipGetErrorString(e_));                                              \
      return CLSR_ELAUNCH;                                                                \
    }                                                                                     \
  } while (0)

long long clsr_p2p_timeout_ticks(void);   // csrc/p2p.hip: bounded waits of the peer-to-peer / grid-barrier kernels

static inline int clsr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Partial chunk of a weight gradient (csrc/linear.hip: pgemm_dw_kernel; summed by clsr_dw_reduce_batch): 5 x 5 tiles of
// 16 x 16 (tile (kt, nt) at (kt * 5 + nt) * 256, element (k & 15) * 16 + (n & 15)) followed by 5 x 16 bias sums
#define CLSR_DW_T 5
#define CLSR_DW_CHUNK (CLSR_DW_T * CLSR_DW_T * 256 + CLSR_DW_T * 16)

#ifdef __HIPCC__
// ---- wave64 reductions -------------------------------------------------------------
// float sums / maxima run on the DPP path: row_shr 1/2/4/8 = inclusive scan inside each row of 16 lanes, row_bcast 15 /
// 31 across the rows, total in lane 63, read back as a scalar (every lane gets it) -- ~12 VALU instructions instead of
// six DEPENDENT ds_bpermute round trips (__shfl_xor), which were the critical path of the per-row softmax kernels.
// -DCLSR_NO_DPP_REDUCE: the shuffle forms (A/B builds).
#define CLSR_DPP_F(x, old, ctrl, rm) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (x)), (ctrl), (rm), 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
#ifdef CLSR_NO_DPP_REDUCE
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  v += CLSR_DPP_F(v, 0.0f, 0x111, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x112, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x114, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x118, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x142, 0xa);
  v += CLSR_DPP_F(v, 0.0f, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef CLSR_NO_DPP_REDUCE
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#else
  // (a lane without a source keeps its own value: max(v, v))
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x111, 0xf));
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x112, 0xf));
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x114, 0xf));
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x118, 0xf));
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x142, 0xa));
  v = fmaxf(v, CLSR_DPP_F(v, v, 0x143, 0xc));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
#ifdef CLSR_NO_DPP_REDUCE
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  // butterfly inside a row of 16: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror -- every lane gets the total
  v += CLSR_DPP_F(v, 0.0f, 0xB1, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x4E, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x141, 0xf);
  v += CLSR_DPP_F(v, 0.0f, 0x140, 0xf);
  return v;
#endif
}
__device__ __forceinline__ double row16_sum_d(double v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 4 lanes {l, l^16, l^32, l^48} that share (lane & 15)
__device__ __forceinline__ float col4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// sum over a 256-thread block (result valid in thread 0); red = 4 doubles of LDS per value
__device__ __forceinline__ double block256_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// v_rcp_f32 (1 ulp) instead of an IEEE division: the gates of the T-serial recurrences spend most of their
// VALU time in these two functions
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh via exp; accurate to ~2e-7 relative on the range that matters, saturates cleanly
__device__ __forceinline__ float tanhf_(float x) {
  float ax = fabsf(x);
  float e = __expf(-2.0f * ax);
  float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// 4 consecutive elements of an fp32 or a bf16 (H) tensor as fp32 (element index idx: multiple of 4)
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <bool H>
__device__ __forceinline__ f32x4 load4e(const void* base, long idx) {
  if (H) return __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const __bf16*>(base) + idx), f32x4);
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
}

// One element / four elements of an embedding table stored as fp32 or as bf16 (H): read widened to fp32, written rounded
// to nearest-even (csrc/optim.hip, csrc/multi.hip: the bf16-table forms of the regulariser / Adam / row-gather kernels)
template <bool H> __device__ __forceinline__ float tbl_ld(const void* p, long e) {
  if (H) return (float)reinterpret_cast<const __bf16*>(p)[e];
  return reinterpret_cast<const float*>(p)[e];
}
template <bool H> __device__ __forceinline__ void tbl_st(void* p, long e, float v) {
  if (H) reinterpret_cast<__bf16*>(p)[e] = (__bf16)v;
  else reinterpret_cast<float*>(p)[e] = v;
}
template <bool H> __device__ __forceinline__ void tbl_st4(void* p, long e, f32x4 v) {
  if (H) *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p) + e) = __builtin_convertvector(v, bf16x4_t);
  else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + e) = v;
}

// Wave priority of the kernels on the step's dependency chain (everything except the weight-gradient partial-sum
// kernels, which run beside it on their own stream): their waves issue ahead of the MFMA-saturated weight-gradient waves
// that share their SIMDs.  In-step kernel times on the chain summed to 3.47 ms against 2.77 ms for the same kernels
// running alone (profiles/r03_contention.md) -- issue arbitration, not occupancy, is what the side kernels take away.
#ifdef CLSR_NO_CHAIN_PRIO
#define CLSR_CHAIN_PRIO()
#else
#define CLSR_CHAIN_PRIO() __builtin_amdgcn_s_setprio(2)
#endif
#define MFMA4(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)
#endif
