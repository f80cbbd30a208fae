// Where do the waves of 3-wave workgroups land?  Records (XCC, SE, CU, SIMD) of every wave of a 256 x 3 grid of 192-thread
// blocks (the shape of the fused recurrence launch) and prints waves per SIMD index summed over the CUs.
//   hipcc --offload-arch=gfx950 -O2 scripts/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
__global__ void probe(unsigned* out, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;        // stay resident while the rest of the grid launches
  const int wave = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { out[2 * wave] = hw; out[2 * wave + 1] = xcc + (x == 123.f); }
}
int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 192;
  const int nw = 256 * 3 * (threads / 64);
  unsigned* d; hipMalloc(&d, nw * 8);
  hipLaunchKernelGGL(probe, dim3(256, 3), dim3(threads), 0, 0, d, 200000);
  hipDeviceSynchronize();
  unsigned* h = (unsigned*)malloc(nw * 8);
  hipMemcpy(h, d, nw * 8, hipMemcpyDeviceToHost);
  long simd[4] = {0, 0, 0, 0};
  std::map<unsigned, int> cu_waves;
  for (int i = 0; i < nw; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
    const unsigned simd_id = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    simd[simd_id]++;
    cu_waves[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  printf("threads/block %d: waves per SIMD index: %ld %ld %ld %ld   distinct CUs %zu\n", threads, simd[0], simd[1], simd[2], simd[3], cu_waves.size());
  int mn = 1 << 30, mx = 0;
  for (auto& kv : cu_waves) { mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
  printf("waves per CU: min %d max %d\n", mn, mx);
  return 0;
}
