// bf16 matrix-pipe helpers shared by the speed-mode kernels (hgemm.hip, hattbwd.hip): v_mfma_f32_16x16x32_bf16 operand
// types, 16-byte loads of eight bf16 / two float4, conversions.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x8 ld8f(const float* p) {
  const f32x4 a = ld4(p), b = ld4(p + 4);
  return (f32x8){a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}
__device__ __forceinline__ bf16x8 ld8h(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32x8 to_f(bf16x8 v) { return __builtin_convertvector(v, f32x8); }
__device__ __forceinline__ bf16x8 to_h(f32x8 v) { return __builtin_convertvector(v, bf16x8); }
#define HMFMA(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (acc), 0, 0, 0)
