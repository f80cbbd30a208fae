// Cost of a grid-wide barrier inside a persistent kernel on gfx950 (160 / 256 workgroups of 256 threads): sense-reversal
// barrier on two global words, agent-scope fences (cross-XCD L2 writeback + invalidate), and a data hand-over check.
//   hipcc --offload-arch=gfx950 -O2 scripts/grid_barrier_probe.hip -o build/abl/grid_barrier_probe && build/abl/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned* gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_thread_fence(__ATOMIC_RELEASE);          // this block's writes are visible before it arrives
    if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(2);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}
__global__ void __launch_bounds__(256) probe(unsigned* bar, float* data, int nphase, int* bad) {
  const int nb = gridDim.x, b = blockIdx.x;
  for (int p = 0; p < nphase; ++p) {
    // every block writes a 4 KB slice, then reads its neighbour's slice of the previous phase after the barrier
    for (int i = threadIdx.x; i < 1024; i += 256) data[(size_t)b * 1024 + i] = (float)(p * 1000 + b);
    grid_barrier(bar, bar + 32, nb);
    const int nbr = (b + nb / 2 + 1) % nb;
    float v = data[(size_t)nbr * 1024 + threadIdx.x];
    if (v != (float)(p * 1000 + nbr)) atomicAdd(bad, 1);
    grid_barrier(bar, bar + 32, nb);
  }
}
int main() {
  unsigned* bar; float* data; int* bad;
  hipMalloc(&bar, 256); hipMemset(bar, 0, 256);
  hipMalloc(&data, 256 * 1024 * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  for (int nb : {160, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int nphase = 200;
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, bar, data, nphase, bad);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("%d blocks: %.2f us per (write 4 KB, barrier, read, barrier) = %.2f us per barrier, mismatches %d\n", nb,
             ms * 1000 / nphase, ms * 1000 / nphase / 2, hb);
    }
  }
  return 0;
}
