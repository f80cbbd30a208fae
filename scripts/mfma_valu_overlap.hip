// Micro-benchmark: does VALU work of the SAME wave hide under v_mfma_f32_16x16x4_f32 (fp32 inputs) on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o build/mfma_valu_overlap && build/mfma_valu_overlap
// One wave per SIMD.  Per loop iteration: M independent MFMAs (4 accumulators round robin) and V independent v_fma_f32
// (8 chains), interleaved V/M VALU per MFMA by construction (inline asm keeps the order).  If the two pipes overlap,
// time(M, V) ~ max(time(M, 0), time(0, V)); if the fp32 MFMA executes on the vector ALUs, it is the SUM.
// The same with v_mfma_f32_16x16x32_bf16 for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MF, int VA, bool BF>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x4 acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
  for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x + i;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (MF > 0 ? MF : 1); ++m) {
      if (MF > 0) {
        if (BF) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m & 3], 0, 0, 0);
        else acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < VA; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// two waves per SIMD (512-thread blocks, one per CU): waves 0-3 issue only MFMAs, waves 4-7 only v_fma
template <bool BF>
__global__ void __launch_bounds__(512) k2(float* out, int iters, float a, float b, int mode) {
  f32x4 acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
  for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x + i;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_m = w < 4 && (mode & 1), do_v = w >= 4 && (mode & 2);
  if (do_m)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (BF) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m & 3], 0, 0, 0);
        else acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
      }
    }
  if (do_v)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 48; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
    }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <bool BF>
void run2(const char* what, int mode) {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<BF>), dim3(256), dim3(512), 0, 0, out, 10, 1.0f, 0.5f, mode);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k2<BF>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.3f ms\n", what, ms);
  hipFree(out);
}

template <int MF, int VA, bool BF>
void run(const char* what) {
  const int blocks = 256;     // 4 waves per block, 256 CUs: one wave per SIMD
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, VA, BF>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MF, VA, BF>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int groups = MF > 0 ? MF : 1;
  printf("%-34s %7.3f ms = %6.1f ns per group of (%d MFMA + %d v_fma)\n", what, ms, ms * 1e6 / ((double)iters * groups),
         MF > 0 ? 1 : 0, VA);
  hipFree(out);
}

int main() {
  run<8, 0, false>("f32 MFMA only");
  run<0, 4, false>("4 v_fma only");
  run<8, 4, false>("f32 MFMA + 4 v_fma each");
  run<0, 6, false>("6 v_fma only");
  run<8, 6, false>("f32 MFMA + 6 v_fma each");
  run<8, 0, true>("bf16 16x16x32 MFMA only");
  run<8, 2, true>("bf16 MFMA + 2 v_fma each");
  run<0, 2, false>("2 v_fma only");
  printf("two waves per SIMD: one issues 8 MFMAs per iteration, the other 48 v_fma\n");
  run2<false>("f32 MFMA wave alone", 1);
  run2<false>("v_fma wave alone", 2);
  run2<false>("f32 MFMA wave + v_fma wave on the same SIMD", 3);
  run2<true>("bf16 MFMA wave alone", 1);
  run2<true>("bf16 MFMA wave + v_fma wave on the same SIMD", 3);
  return 0;
}
