// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 (independent and dependent chains).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o build/mfma_rate && build/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int NACC>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd waves on each of the 1024 SIMDs
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * 4 * iters * 8 * NACC;  // per wave
  const double flops = n_mfma * 2048.0;
  printf("accumulators %2d  waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s  %6.1f ns per MFMA per SIMD\n", NACC, waves_per_simd,
         ms, flops / ms / 1e9, ms * 1e6 / ((double)iters * 8 * NACC * waves_per_simd));
  hipFree(out);
}

int main() {
  run<1>(1); run<2>(1); run<4>(1); run<10>(1); run<10>(2); run<10>(4); run<1>(4);
  return 0;
}
