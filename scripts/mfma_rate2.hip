// What slows the pgemm inner loop?  40 MFMAs per "k-tile" (5 out-tiles x 2 sub-tiles x 4 k-slots) with, optionally,
// A operands from LDS, B operands produced by VALU work on loaded registers, global loads / stores around them.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate2.hip -o build/mfma_rate2 && build/mfma_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

template <bool LDS_A, bool VALU_B, bool GLOAD, bool STORE, bool INIT>
__global__ void __launch_bounds__(256) loop(const float* __restrict__ X, float* __restrict__ out, int tiles, int KT,
                                            int K, float* __restrict__ Y, const float* __restrict__ U) {
  __shared__ __attribute__((aligned(16))) float lds[80 * 84];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  for (int e = tid; e < 80 * 84; e += 256) lds[e] = 0.001f * (e % 7);
  __syncthreads();
  const float* ldsA = lds + j * 84 + 4 * g;
  f32x4 acc[2][5];
  float total = 0.f;
  for (int tile = 0; tile < tiles; ++tile) {
    const long tix = ((long)(blockIdx.x * 4 + (tid >> 6)) * tiles + tile);
#pragma unroll
    for (int ot = 0; ot < 5; ++ot) { acc[0][ot] = (f32x4){0, 0, 0, 0}; acc[1][ot] = (f32x4){0, 0, 0, 1}; }
    if (INIT) {
      const float* up = U + (tix % 4096) * 32 * 80 + j * 80 + 4 * g;
      f32x4 t[2][5];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ot = 0; ot < 5; ++ot) t[s2][ot] = *(const f32x4*)(up + s2 * 16 * 80 + ot * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ot = 0; ot < 5; ++ot) acc[s2][ot] += t[s2][ot];
    }
    const float* xp0 = X + ((long)(blockIdx.x * 4 + (tid >> 6)) * tiles + tile) % 4096 * 32 * 80 + j * 80 + 4 * g;
    const float* xp1 = xp0 + 16 * 80;
    f32x4 r0 = GLOAD ? *(const f32x4*)xp0 : (f32x4){1, 2, 3, 4};
    f32x4 r1 = GLOAD ? *(const f32x4*)xp1 : (f32x4){4, 3, 2, 1};
    for (int kt = 0; kt < KT; ++kt) {
      f32x4 b0 = r0, b1 = r1;
      if (VALU_B) {
        const bool ink = kt * 16 + 4 * g < K;
        b0 = ink ? b0 * b1 : (f32x4){0, 0, 0, 0};
        b1 = ink ? b1 * 1.5f : (f32x4){0, 0, 0, 0};
      }
      if (GLOAD) {
        const int kn = kt + 1 < KT ? kt + 1 : kt;
        r0 = *(const f32x4*)(xp0 + kn * 16);
        r1 = *(const f32x4*)(xp1 + kn * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 wt[5];
#pragma unroll
      for (int ot = 0; ot < 5; ++ot)
        wt[ot] = LDS_A ? *(const f32x4*)(ldsA + ot * 16 * 84 + kt * 16) : (f32x4){1.f + ot, 2.f, 3.f, 4.f + kt};
#pragma unroll
      for (int ot = 0; ot < 5; ++ot) { MFMA4(acc[0][ot], wt[ot].x, b0.x); MFMA4(acc[1][ot], wt[ot].x, b1.x); }
#pragma unroll
      for (int ot = 0; ot < 5; ++ot) { MFMA4(acc[0][ot], wt[ot].y, b0.y); MFMA4(acc[1][ot], wt[ot].y, b1.y); }
#pragma unroll
      for (int ot = 0; ot < 5; ++ot) { MFMA4(acc[0][ot], wt[ot].z, b0.z); MFMA4(acc[1][ot], wt[ot].z, b1.z); }
#pragma unroll
      for (int ot = 0; ot < 5; ++ot) { MFMA4(acc[0][ot], wt[ot].w, b0.w); MFMA4(acc[1][ot], wt[ot].w, b1.w); }
    }
    if (STORE) {
      float* yp = Y + tix * 32 * 80 + j * 80 + 4 * g;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ot = 0; ot < 5; ++ot) *(f32x4*)(yp + s2 * 16 * 80 + ot * 16) = acc[s2][ot];
    } else {
#pragma unroll
      for (int ot = 0; ot < 5; ++ot) {
        const f32x4 s = acc[0][ot] + acc[1][ot];
        total += s.x + s.y + s.z + s.w;
      }
    }
  }
  out[blockIdx.x * 256 + tid] = total;
}

template <bool LDS_A, bool VALU_B, bool GLOAD, bool STORE, bool INIT>
void run(const char* name, const float* X, float* out, float* Y, const float* U) {
  const int blocks = 1024, tiles = 8, KT = 5;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<LDS_A, VALU_B, GLOAD, STORE, INIT>), dim3(blocks), dim3(256), 0, 0, X, out, 1, KT, 80, Y, U);
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop<LDS_A, VALU_B, GLOAD, STORE, INIT>), dim3(blocks), dim3(256), 0, 0, X, out, tiles, KT, 80, Y, U);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * tiles * KT * 40 * 2048.0;
  printf("%-46s %8.3f ms  %6.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
  float *X, *out;
  hipMalloc(&X, (size_t)4096 * 32 * 80 * 4 + 4096);
  hipMemset(X, 0, (size_t)4096 * 32 * 80 * 4 + 4096);
  hipMalloc(&out, 1024 * 256 * 4);
  float *Y, *U;
  hipMalloc(&Y, (size_t)1024 * 4 * 8 * 32 * 80 * 4);   // 335 MB like z0
  hipMalloc(&U, (size_t)4096 * 32 * 80 * 4);
  hipMemset(U, 0, (size_t)4096 * 32 * 80 * 4);
  run<false, false, false, false, false>("MFMA only (tile structure, acc re-init)", X, out, Y, U);
  run<true, true, false, false, false>("+ LDS A + VALU B", X, out, Y, U);
  run<true, true, true, false, false>("+ global loads of B (L2 resident)", X, out, Y, U);
  run<true, true, true, true, false>("+ stores of the 32x80 output tile (335 MB)", X, out, Y, U);
  run<true, true, true, false, true>("+ accumulator init loads (no stores)", X, out, Y, U);
  run<true, true, true, true, true>("+ stores + accumulator init loads", X, out, Y, U);
  return 0;
}
