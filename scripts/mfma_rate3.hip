// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32, v_mfma_f32_16x16x16_bf16 and v_mfma_f32_16x16x32_bf16 on gfx950
// (10 independent accumulators, 1 and 2 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate3.hip -o build/mfma_rate3 && build/mfma_rate3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
  s16x4 a4 = {(short)a, 1, 2, 3}, b4 = {(short)b, 3, 2, 1};
  bf16x8 a8, b8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(a + e); b8[e] = (__bf16)(b - e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int KIND, int NACC>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s accumulators %2d  waves/SIMD %d : %8.3f ms  %6.2f ns per MFMA per SIMD\n", name, NACC, waves_per_simd, ms,
         ms * 1e6 / ((double)iters * 8 * NACC * waves_per_simd));
  hipFree(out);
}

int main() {
  run<0, 10>("v_mfma_f32_16x16x4_f32", 1); run<1, 10>("v_mfma_f32_16x16x16_bf16", 1); run<2, 10>("v_mfma_f32_16x16x32_bf16", 1);
  run<0, 10>("v_mfma_f32_16x16x4_f32", 2); run<1, 10>("v_mfma_f32_16x16x16_bf16", 2); run<2, 10>("v_mfma_f32_16x16x32_bf16", 2);
  run<1, 2>("v_mfma_f32_16x16x16_bf16", 1); run<2, 2>("v_mfma_f32_16x16x32_bf16", 1);
  return 0;
}
