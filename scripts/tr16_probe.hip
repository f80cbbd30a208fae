// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds L[e] = e as 16-bit values in a [64][80] row-major
// image; lane (4 j' + q) of 16-lane group g passes the address of row 8 g + j', columns 4 q .. 4 q + 3.  Prints which
// element every (lane, j) receives.   hipcc --offload-arch=gfx950 -O2 scripts/tr16_probe.hip -o build/abl/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short L[64 * 80];
  for (int e = threadIdx.x; e < 64 * 80; e += 64) L[e] = (short)e;
  __syncthreads();
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  const int jp = c >> 2, q = c & 3;
  auto p = reinterpret_cast<__attribute__((address_space(3))) s16x4*>((__attribute__((address_space(3))) short*)L + (8 * g + jp) * 80 + 4 * q);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (row %2d col %2d)", h[l * 4 + j] / 80, h[l * 4 + j] % 80);
    printf("\n");
  }
  return 0;
}
