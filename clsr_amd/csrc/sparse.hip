// Sorted segmented reduction of the history-gather gradient (gfx950).
//
// Reference semantics: the gradient of tf.nn.embedding_lookup(item_lookup, item_history)
// (sequential_base_model.py:392-404) is an IndexedSlices with one slice per (history, step); TF's
// sparse Adam first sums duplicate indices (unsorted_segment_sum).  Popular
// items (Zipf head) and the padding row 0 appear thousands of times per batch, so scattering
// every slice with atomics serialises on a few cache lines.  Here the (id, position) pairs are
// radix-sorted by id on the device (rocPRIM device primitive) and each thread group walks 64
// consecutive sorted entries, summing runs of equal ids in registers: one atomic per run per
// group instead of one per slice.  The sorted order is also the per-rank input of the
// segmented sparse row exchange for multi-GPU runs with catalogues too large for dense tables.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "common.h"

__global__ void prep_sort_kernel(const int* __restrict__ ids, long nrows, int ncols, long row_stride,
                                 int* __restrict__ keys, int* __restrict__ vals) {
  const long n = nrows * ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long r = e / ncols;
    keys[e] = ids[r * row_stride + (e - r * ncols)];
    vals[e] = (int)e;
  }
}

static size_t sort_temp_bytes(size_t n, int end_bit) {
  size_t bytes = 0;
  int* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, n, 0, end_bit, (hipStream_t)0);
  return (bytes + 255) & ~(size_t)255;
}

// bytes of workspace clsr_sort_ids needs for n = nrows*ncols pairs with ids < vocab
extern "C" long clsr_sort_ids_workspace_bytes(long n, long vocab) {
  if (n <= 0) return 0;
  (void)vocab;  // a counting sort was tried: its per-key cursor atomics serialise on Zipf-head ids
  return (long)(2 * ((size_t)n * sizeof(int) + 255) + sort_temp_bytes((size_t)n, 32) + 512);
}

// keys_out[p] = p-th smallest id of ids[r*row_stride + c] (r < nrows, c < ncols); perm_out[p] = r*ncols + c
extern "C" int clsr_sort_ids(const int* ids, long nrows, int ncols, long row_stride, long vocab,
                             int* keys_out, int* perm_out, void* workspace, long workspace_bytes,
                             void* stream) {
  CLSR_CHECK_ARG(ids && keys_out && perm_out && workspace && nrows > 0 && ncols > 0 && vocab > 0);
  const size_t n = (size_t)nrows * ncols;
  CLSR_CHECK_ARG(workspace_bytes >= clsr_sort_ids_workspace_bytes((long)n, vocab));
  char* ws = (char*)workspace;
  int end_bit = 1;
  while (end_bit < 32 && (1L << end_bit) < vocab) ++end_bit;
  const size_t seg = (n * sizeof(int) + 255) & ~(size_t)255;
  int* keys_in = (int*)ws;
  int* vals_in = (int*)(ws + seg);
  void* temp = ws + 2 * seg;
  size_t temp_bytes = (size_t)workspace_bytes - 2 * seg;
  hipStream_t s = (hipStream_t)stream;
  int blocks = clsr_cdiv((long)n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(prep_sort_kernel, dim3(blocks), dim3(256), 0, s, ids, nrows, ncols, row_stride, keys_in, vals_in);
  CLSR_CHECK_LAUNCH();
  CLSR_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, perm_out, n, 0, end_bit, s));
  return CLSR_OK;
}

// g[pos, :] = dhist[pos, col0:col0+C] + (t < len) dmean[h]/len + recent(t) drecent[h]/cnt   (pos = h*T + t)
// grad[key*ldg + gcol0 + c] += sum of g over the run of equal keys; sumsq += sum g^2 (IndexedSlices norm)
template <int CP>
__global__ void __launch_bounds__(256) gather_bwd_sorted_kernel(
    const float* __restrict__ dhist, const float* __restrict__ dmean, const float* __restrict__ drecent,
    const int* __restrict__ keys, const int* __restrict__ perm, const int* __restrict__ seq_len,
    int len_stride, long n, int T, int D, int col0, int C, int recent_k, float* __restrict__ grad, int ldg,
    int gcol0, double* __restrict__ sumsq) {
  __shared__ double red[4];
  constexpr int GPB = 256 / CP;   // thread groups per block
  constexpr int EPL = 64 / CP;    // chunk entries preloaded per lane
  const int c = threadIdx.x % CP;
  const long gid = (long)blockIdx.x * GPB + threadIdx.x / CP;
  const bool cok = c < C;
  const long p0 = gid * 64;
  // every lane preloads EPL entries of the chunk (coalesced), later broadcast inside the group
  int mk[EPL], mp[EPL], ml[EPL];
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
    const long p = p0 + u * CP + c;
    mk[u] = -1; mp[u] = 0; ml[u] = 1;
    if (p < n) {
      mk[u] = keys[p];
      mp[u] = perm[p];
      ml[u] = seq_len[(long)(mp[u] / T) * len_stride];
    }
  }
  int cur = -1;
  float acc = 0.f, local = 0.f;
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
#pragma unroll 8
    for (int q = 0; q < CP; ++q) {
      const int key = __shfl(mk[u], q, CP);
      const int pos = __shfl(mp[u], q, CP);
      const int len = __shfl(ml[u], q, CP);
      if (key < 0) continue;     // past the end of the array (uniform inside the group)
      const int h = pos / T, t = pos - h * T;
      float g = 0.f;
      if (cok) {
        g = dhist[(long)pos * D + col0 + c];
        if (t < len) {
          if (dmean) g += dmean[(long)h * D + col0 + c] / (float)len;
          if (drecent && t >= len - recent_k)
            g += drecent[(long)h * D + col0 + c] / (float)(len < recent_k ? len : recent_k);
        }
      }
      local += g * g;
      if (key != cur) {
        if (cur >= 0 && cok) atomicAdd(grad + (long)cur * ldg + gcol0 + c, acc);
        cur = key;
        acc = g;
      } else {
        acc += g;
      }
    }
  }
  if (cur >= 0 && cok) atomicAdd(grad + (long)cur * ldg + gcol0 + c, acc);
  if (sumsq) {
    const double tot = block256_sum_d((double)local, red);
    if (threadIdx.x == 0 && tot != 0.0) atomicAdd(sumsq, tot);
  }
}

extern "C" int clsr_gather_bwd_sorted(const float* dhist, const float* dmean, const float* drecent,
                                      const int* keys, const int* perm, const int* seq_len, int len_stride,
                                      long n, int T, int D, int col0, int C, int recent_k, float* grad,
                                      int ldg, int gcol0, double* sumsq, void* stream) {
  CLSR_CHECK_ARG(dhist && keys && perm && seq_len && grad && n > 0 && T > 0 && D > 0 && C > 0);
  CLSR_CHECK_SUPPORTED(C <= 64);
  const int CP = C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : 64));
  const long groups = (n + 63) / 64;
  const int blocks = clsr_cdiv(groups, 256 / CP);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_GBS(CPV)                                                                                     \
  hipLaunchKernelGGL(gather_bwd_sorted_kernel<CPV>, dim3(blocks), dim3(256), 0, s, dhist, dmean, drecent,   \
                     keys, perm, seq_len, len_stride, n, T, D, col0, C, recent_k, grad, ldg, gcol0, sumsq)
  if (CP == 8) LAUNCH_GBS(8);
  else if (CP == 16) LAUNCH_GBS(16);
  else if (CP == 32) LAUNCH_GBS(32);
  else LAUNCH_GBS(64);
#undef LAUNCH_GBS
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
