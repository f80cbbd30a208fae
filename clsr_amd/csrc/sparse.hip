// Sorted segmented reduction of the history-gather gradient (gfx950).
//
// Reference semantics: the gradient of tf.nn.embedding_lookup(item_lookup, item_history)
// (sequential_base_model.py:392-404) is an IndexedSlices with one slice per (history, step); TF's
// sparse Adam first sums duplicate indices (unsorted_segment_sum).  Popular
// items (Zipf head) and the padding row 0 appear thousands of times per batch, so scattering
// every slice with atomics serialises on a few cache lines.  Here the (id, position) pairs are
// grouped by id on the device (the counting sort below) and each thread group walks 64
// consecutive sorted entries, summing runs of equal ids in registers: one atomic per run per
// group instead of one per slice.  The sorted order is also the per-rank input of the
// segmented sparse row exchange for multi-GPU runs with catalogues too large for dense tables.
#include "common.h"
#include "clsr_hip.h"
#include <cstdlib>

// ---- grouping (id, position) pairs by id: hand-written counting sort, three launches for ALL tables of a step
// (blockIdx.y = table), no vendor library on the step:
//   1. group_hist     every workgroup aggregates its chunk of ids in an LDS hash table (popular ids -- the Zipf head,
//                     the padding row 0 -- meet in LDS, not in one L2 atomic), then adds one count per distinct id
//   2. group_scan     exclusive scan of the bucket counters (one workgroup per table; counters -> cursors)
//   3. group_scatter  the same LDS aggregation, one cursor claim per distinct id and workgroup, then every entry is
//                     written to  base + its rank inside the workgroup's share
// Bucket = id & (2^bits - 1) with bits <= GROUP_MAX_BITS: vocabularies up to 2^18 ids come out sorted by id;
// larger ones (100M-item catalogue) come out grouped by bucket -- different ids of a bucket may interleave, which
// the consumer (runs of EQUAL ids, one atomic per run) handles by construction.  The order inside a run of equal
// ids is not defined (claims race); neither was the order of the run-sum atomics before.
#define GROUP_MAX_BITS 18
#define GROUP_CHUNK 2048          // ids per workgroup pass
#define GROUP_HS 4096             // LDS hash slots (>= 2 x chunk: short probe sequences)
#define GROUP_EPT (GROUP_CHUNK / 256)

struct GroupArgs {
  clsr_sortids_desc d[CLSR_SORTIDS_MAX];
};

__device__ __forceinline__ int group_slot(int bucket) { return (int)(((unsigned)bucket * 2654435761u) >> 20) & (GROUP_HS - 1); }

// insert `bucket` into the LDS table; returns the slot, *rank = number of earlier arrivals with the same bucket
__device__ __forceinline__ int group_insert(int* hkey, int* hcnt, int bucket, int* rank) {
  int slot = group_slot(bucket);
  while (true) {
    const int prev = atomicCAS(&hkey[slot], -1, bucket);
    if (prev == -1 || prev == bucket) break;
    slot = (slot + 1) & (GROUP_HS - 1);
  }
  *rank = atomicAdd(&hcnt[slot], 1);
  return slot;
}

template <bool SCATTER>
__global__ void __launch_bounds__(256) group_pass_kernel(GroupArgs a) {
  __shared__ int hkey[GROUP_HS], hcnt[GROUP_HS];
  const clsr_sortids_desc d = a.d[blockIdx.y];
  const long n = d.nrows * d.ncols;
  const int mask = (1 << d.bits) - 1;
  for (long c0 = (long)blockIdx.x * GROUP_CHUNK; c0 < n; c0 += (long)gridDim.x * GROUP_CHUNK) {
    for (int e = threadIdx.x; e < GROUP_HS; e += 256) { hkey[e] = -1; hcnt[e] = 0; }
    __syncthreads();
    int id[GROUP_EPT], slot[GROUP_EPT], rank[GROUP_EPT];
#pragma unroll
    for (int u = 0; u < GROUP_EPT; ++u) {
      const long e = c0 + u * 256 + threadIdx.x;
      id[u] = -1;
      if (e < n) {
        const long r = e / d.ncols;
        id[u] = d.ids[r * d.row_stride + (e - r * d.ncols)];
        slot[u] = group_insert(hkey, hcnt, id[u] & mask, &rank[u]);
      }
    }
    __syncthreads();
    // one global atomic per distinct bucket of the chunk: the histogram count, or the claim of that many
    // consecutive output positions (the claimed base replaces the key's count in LDS)
    for (int e = threadIdx.x; e < GROUP_HS; e += 256) {
      if (hkey[e] >= 0) {
        const int base = atomicAdd(&d.counts[hkey[e]], hcnt[e]);
        if (SCATTER) hcnt[e] = base;
      }
    }
    if (SCATTER) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < GROUP_EPT; ++u) {
        if (id[u] >= 0) {
          const int p = hcnt[slot[u]] + rank[u];
          d.keys_out[p] = id[u];
          d.perm_out[p] = (int)(c0 + u * 256 + threadIdx.x);
        }
      }
    }
    __syncthreads();
  }
}

// counts[b] -> number of entries in buckets < b  (exclusive scan, in place; one 1024-thread workgroup per table walks
// the counters in coalesced tiles of 4096: 16-byte loads, wave scans by shuffles, one LDS exchange per tile)
__global__ void __launch_bounds__(1024) group_scan_kernel(GroupArgs a) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const clsr_sortids_desc d = a.d[blockIdx.x];
  const int nb = 1 << d.bits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 4096) {
    const int i = base + 4 * threadIdx.x;
    int v[4] = {0, 0, 0, 0};
    if (i + 3 < nb) {
      const int4 q = *reinterpret_cast<const int4*>(d.counts + i);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      for (int k = 0; k < 4; ++k) if (i + k < nb) v[k] = d.counts[i + k];
    }
    const int mine = v[0] + v[1] + v[2] + v[3];
    int inc = mine;                       // inclusive scan of the per-thread sums inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    int run = before + inc - mine;
    int o4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o4[k] = run; run += v[k]; }
    if (i + 3 < nb) *reinterpret_cast<int4*>(d.counts + i) = make_int4(o4[0], o4[1], o4[2], o4[3]);
    else for (int k = 0; k < 4; ++k) if (i + k < nb) d.counts[i + k] = o4[k];
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = run;
    __syncthreads();
  }
}

extern "C" int clsr_sizeof_sortids_desc(void) { return (int)sizeof(clsr_sortids_desc); }

// bucket bits used for a vocabulary (the caller provides (1 << bits) ZEROED int counters per table)
extern "C" int clsr_sort_ids_bits(long vocab) {
  int bits = 1;
  while (bits < GROUP_MAX_BITS && (1L << bits) < vocab) ++bits;
  return bits;
}

// keys_out / perm_out: the (id, position) pairs of ids[r*row_stride + c] (r < nrows, c < ncols; position = r*ncols + c)
// grouped by id (ascending ids when vocab <= 2^18).  counts: (1 << bits) int counters per table, ZERO on entry
// (they come back holding the end offset of every bucket).  Three launches whatever the number of tables.
extern "C" int clsr_sort_ids_multi(const clsr_sortids_desc* descs, int n, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_SORTIDS_MAX);
  GroupArgs a;
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    const clsr_sortids_desc& d = descs[i];
    CLSR_CHECK_ARG(d.ids && d.keys_out && d.perm_out && d.counts && d.nrows > 0 && d.ncols > 0);
    CLSR_CHECK_ARG(d.bits >= 1 && d.bits <= GROUP_MAX_BITS);
    CLSR_CHECK_SUPPORTED(d.nrows * d.ncols < (1L << 31));
    a.d[i] = d;
    const long e = d.nrows * d.ncols;
    mx = e > mx ? e : mx;
  }
  hipStream_t s = (hipStream_t)stream;
  int blocks = clsr_cdiv(mx, GROUP_CHUNK);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(group_pass_kernel<false>, dim3(blocks, n), dim3(256), 0, s, a);
  CLSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(group_scan_kernel, dim3(n), dim3(1024), 0, s, a);
  CLSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(group_pass_kernel<true>, dim3(blocks, n), dim3(256), 0, s, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// single-table form (workspace = the zeroed counters are NOT assumed: cleared here with one extra launch)
__global__ void zero_ints_kernel(int* __restrict__ p, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) p[e] = 0;
}

extern "C" long clsr_sort_ids_workspace_bytes(long n, long vocab) {
  if (n <= 0) return 0;
  return ((long)sizeof(int) << clsr_sort_ids_bits(vocab)) + 256;
}

extern "C" int clsr_sort_ids(const int* ids, long nrows, int ncols, long row_stride, long vocab,
                             int* keys_out, int* perm_out, void* workspace, long workspace_bytes,
                             void* stream) {
  CLSR_CHECK_ARG(ids && keys_out && perm_out && workspace && nrows > 0 && ncols > 0 && vocab > 0);
  CLSR_CHECK_ARG(workspace_bytes >= clsr_sort_ids_workspace_bytes(nrows * ncols, vocab));
  clsr_sortids_desc d;
  d.ids = ids; d.keys_out = keys_out; d.perm_out = perm_out; d.counts = (int*)workspace;
  d.nrows = nrows; d.row_stride = row_stride; d.ncols = ncols; d.bits = clsr_sort_ids_bits(vocab);
  const long nb = 1L << d.bits;
  hipLaunchKernelGGL(zero_ints_kernel, dim3(clsr_cdiv(nb, 256)), dim3(256), 0, (hipStream_t)stream, d.counts, nb);
  CLSR_CHECK_LAUNCH();
  return clsr_sort_ids_multi(&d, 1, stream);
}

// g[pos, :] = dhist[pos, col0:col0+C] + (t < len) dmean[h]/len + recent(t) drecent[h]/cnt   (pos = h*T + t)
// grad[key*ldg + gcol0 + c] += sum of g over the run of equal keys; sumsq += sum g^2 (IndexedSlices norm)
// VW = floats per lane (1 | 4).  VW = 4 (every offset a multiple of 4): a row slice is read as 16-byte pieces, CP lanes
// cover 4 * CP columns -- 4x fewer lanes per entry and 4x the bytes per load, i.e. 16x the bytes in flight per wave.
// The scalar form moved 211 MB in 308 us (0.09 of the HBM rate) over the three launches of the 100M-item catalogue's
// 384 B + 128 B rows: one float per lane, eight reads in flight.  HD: dhist / dhist2 are bf16 tensors (bf16
// activations -- n * D * 2 bytes of gradient read per row).
#define GBS_CHUNK(CP) ((CP) < 16 ? 16 : (CP))   // sorted entries per thread group
template <int VW, bool HD> struct GbsLoad;
template <> struct GbsLoad<1, false> {
  typedef float type;
  static __device__ __forceinline__ float ld(const void* p, long off) { return reinterpret_cast<const float*>(p)[off]; }
};
template <> struct GbsLoad<4, false> {
  typedef f32x4 type;
  static __device__ __forceinline__ f32x4 ld(const void* p, long off) { return ld4(reinterpret_cast<const float*>(p) + off); }
};
template <> struct GbsLoad<1, true> {
  typedef float type;
  static __device__ __forceinline__ float ld(const void* p, long off) { return (float)reinterpret_cast<const __bf16*>(p)[off]; }
};
template <> struct GbsLoad<4, true> {
  typedef f32x4 type;
  static __device__ __forceinline__ f32x4 ld(const void* p, long off) { return load4e<true>(p, off); }
};
__device__ __forceinline__ float gbs_sq(float v) { return v * v; }
__device__ __forceinline__ float gbs_sq(const f32x4& v) { return v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
__device__ __forceinline__ void gbs_atomic(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void gbs_atomic(float* p, const f32x4& v) {
  atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
template <int CP, int VW, bool HD>
__global__ void __launch_bounds__(256) gather_bwd_sorted_kernel(
    const void* __restrict__ dhist, const void* __restrict__ dhist2, const float* __restrict__ dmean,
    const float* __restrict__ drecent, const int* __restrict__ keys, const int* __restrict__ perm,
    const int* __restrict__ seq_len, int len_stride, long n, int T, int D, int col0, int C, int recent_k,
    float* __restrict__ grad, int ldg, int gcol0, double* __restrict__ sumsq) {
  typedef typename GbsLoad<VW, HD>::type vec_t;
  typedef GbsLoad<VW, false> LdF;
  typedef GbsLoad<VW, HD> LdH;
  __shared__ double red[4];
  constexpr int GPB = 256 / CP;              // thread groups per block
  constexpr int EPL = GBS_CHUNK(CP) / CP;    // chunk entries preloaded per lane
  const int c = (threadIdx.x % CP) * VW;     // first column of this lane
  const long gid = (long)blockIdx.x * GPB + threadIdx.x / CP;
  const bool cok = c < C;
  const long p0 = gid * GBS_CHUNK(CP);
  // every lane preloads EPL entries of the chunk (coalesced) and works out what only depends on the entry: the
  // history row and the divisors of the mean / recent terms (0 = term absent); broadcast inside the group later.
  // Short chunks (16 .. 64 sorted entries per thread group): the walk below is a serial chain per group, so the
  // launch gets its parallelism from the number of groups (12 800 at 204 800 positions), not from long chunks.
  int mk[EPL], mp[EPL], mh[EPL];
  float ml[EPL], mr[EPL];
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
    const long p = p0 + u * CP + threadIdx.x % CP;
    mk[u] = -1; mp[u] = 0; mh[u] = 0; ml[u] = 0.f; mr[u] = 0.f;
    if (p < n) {
      mk[u] = keys[p];
      const int pos = perm[p];
      const int h = pos / T, t = pos - h * T;
      const int len = seq_len[(long)h * len_stride];
      const bool in_len = t < len;
      mp[u] = pos; mh[u] = h;
      ml[u] = in_len ? (float)len : 0.f;
      mr[u] = (in_len && t >= len - recent_k) ? (float)(len < recent_k ? len : recent_k) : 0.f;
    }
  }
  int cur = -1;
  vec_t acc = vec_t(0.f);
  float local = 0.f;
  const long cc = col0 + (cok ? c : 0);      // this lane's column in the [., D] tensors
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
    // sub-chunks of 8 sorted entries: their (random-row) gradient reads are issued together, then the run
    // accumulation walks them in order -- one exposed memory latency per 8 entries instead of per entry
    constexpr int SUB = CP < 8 ? CP : 8;
#pragma unroll
    for (int q0 = 0; q0 < CP; q0 += SUB) {
      int key[SUB];
      vec_t g[SUB];
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        key[k] = __shfl(mk[u], q0 + k, CP);
        const int pos = __shfl(mp[u], q0 + k, CP);
        const int h = __shfl(mh[u], q0 + k, CP);
        const float flen = __shfl(ml[u], q0 + k, CP);
        const float frec = __shfl(mr[u], q0 + k, CP);
        vec_t v = LdH::ld(dhist, (long)pos * D + cc);
        if (dhist2) v += LdH::ld(dhist2, (long)pos * D + cc);
        if (dmean) v += flen > 0.f ? LdF::ld(dmean, (long)h * D + cc) * (1.0f / flen) : vec_t(0.f);
        if (drecent) v += frec > 0.f ? LdF::ld(drecent, (long)h * D + cc) * (1.0f / frec) : vec_t(0.f);
        g[k] = (cok && key[k] >= 0) ? v : vec_t(0.f);
      }
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        if (key[k] < 0) continue;     // past the end of the array (uniform inside the group)
        local += gbs_sq(g[k]);
        if (key[k] != cur) {
          if (cur >= 0 && cok) gbs_atomic(grad + (long)cur * ldg + gcol0 + c, acc);
          cur = key[k];
          acc = g[k];
        } else {
          acc += g[k];
        }
      }
    }
  }
  if (cur >= 0 && cok) gbs_atomic(grad + (long)cur * ldg + gcol0 + c, acc);
  if (sumsq) {
    const double tot = block256_sum_d((double)local, red);
    if (threadIdx.x == 0 && tot != 0.0) atomicAdd(sumsq, tot);
  }
}

static int gbs_launch(const void* dhist, const void* dhist2, int bf16, const float* dmean, const float* drecent,
                      const int* keys, const int* perm, const int* seq_len, int len_stride, long n, int T, int D,
                      int col0, int C, int recent_k, float* grad, int ldg, int gcol0, double* sumsq, void* stream) {
  CLSR_CHECK_ARG(dhist && keys && perm && seq_len && grad && n > 0 && T > 0 && D > 0 && C > 0);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && D % 4 == 0 && col0 % 4 == 0 && gcol0 % 4 == 0 && ldg % 4 == 0 &&
                   ((uintptr_t)dhist % 16) == 0 && (!dhist2 || ((uintptr_t)dhist2 % 16) == 0) &&
                   (!dmean || ((uintptr_t)dmean % 16) == 0) && (!drecent || ((uintptr_t)drecent % 16) == 0) &&
                   ((uintptr_t)grad % 16) == 0;
#define LAUNCH_GBS(CPV, VWV)                                                                                   \
  do {                                                                                                         \
    const long groups = (n + GBS_CHUNK(CPV) - 1) / GBS_CHUNK(CPV);                                             \
    const int blocks = clsr_cdiv(groups, 256 / (CPV));                                                         \
    if (bf16)                                                                                                  \
      hipLaunchKernelGGL((gather_bwd_sorted_kernel<CPV, VWV, true>), dim3(blocks), dim3(256), 0, s, dhist,     \
                         dhist2, dmean, drecent, keys, perm, seq_len, len_stride, n, T, D, col0, C, recent_k,  \
                         grad, ldg, gcol0, sumsq);                                                             \
    else                                                                                                       \
      hipLaunchKernelGGL((gather_bwd_sorted_kernel<CPV, VWV, false>), dim3(blocks), dim3(256), 0, s, dhist,    \
                         dhist2, dmean, drecent, keys, perm, seq_len, len_stride, n, T, D, col0, C, recent_k,  \
                         grad, ldg, gcol0, sumsq);                                                             \
  } while (0)
  if (vec) {
    CLSR_CHECK_SUPPORTED(C <= 256);
    const int q = C / 4;
    if (q <= 8) LAUNCH_GBS(8, 4);
    else if (q <= 16) LAUNCH_GBS(16, 4);
    else if (q <= 32) LAUNCH_GBS(32, 4);
    else LAUNCH_GBS(64, 4);
  } else {
    CLSR_CHECK_SUPPORTED(C <= 64 && !bf16);
    if (C <= 8) LAUNCH_GBS(8, 1);
    else if (C <= 16) LAUNCH_GBS(16, 1);
    else if (C <= 32) LAUNCH_GBS(32, 1);
    else LAUNCH_GBS(64, 1);
  }
#undef LAUNCH_GBS
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// widest column block one launch of clsr_gather_bwd_sorted2 takes for this layout (256 with 16-byte accesses, else 64)
extern "C" int clsr_gather_bwd_sorted_max_cols(int D, int col0, int C, int ldg, int gcol0) {
  return (C % 4 == 0 && D % 4 == 0 && col0 % 4 == 0 && gcol0 % 4 == 0 && ldg % 4 == 0) ? 256 : 64;
}

extern "C" int clsr_gather_bwd_sorted2(const float* dhist, const float* dhist2, const float* dmean,
                                       const float* drecent, const int* keys, const int* perm, const int* seq_len,
                                       int len_stride, long n, int T, int D, int col0, int C, int recent_k,
                                       float* grad, int ldg, int gcol0, double* sumsq, void* stream) {
  return gbs_launch(dhist, dhist2, 0, dmean, drecent, keys, perm, seq_len, len_stride, n, T, D, col0, C, recent_k, grad,
                    ldg, gcol0, sumsq, stream);
}
// the same with d(hist) (and its optional second addend) stored as bf16; dmean / drecent stay fp32
extern "C" int clsr_gather_bwd_sorted2_h(const void* dhist_bf16, const void* dhist2_bf16, const float* dmean,
                                         const float* drecent, const int* keys, const int* perm, const int* seq_len,
                                         int len_stride, long n, int T, int D, int col0, int C, int recent_k,
                                         float* grad, int ldg, int gcol0, double* sumsq, void* stream) {
  return gbs_launch(dhist_bf16, dhist2_bf16, 1, dmean, drecent, keys, perm, seq_len, len_stride, n, T, D, col0, C,
                    recent_k, grad, ldg, gcol0, sumsq, stream);
}

extern "C" int clsr_gather_bwd_sorted(const float* dhist, const float* dmean, const float* drecent,
                                      const int* keys, const int* perm, const int* seq_len, int len_stride,
                                      long n, int T, int D, int col0, int C, int recent_k, float* grad,
                                      int ldg, int gcol0, double* sumsq, void* stream) {
  return clsr_gather_bwd_sorted2(dhist, nullptr, dmean, drecent, keys, perm, seq_len, len_stride, n, T, D, col0, C,
                                 recent_k, grad, ldg, gcol0, sumsq, stream);
}

// =================================================================== touched-row exchange (multi-GPU)
// The gradient of an embedding table lives in a dense table plus a byte map of the touched ("involved")
// rows.  For the data-parallel exchange of a table whose touched rows are few compared with the vocabulary
// (user tables at Taobao scale, every table of a 100M-item catalogue) a rank ships only (ids, rows):
//   clsr_flags_compact : byte map -> ascending id list + count (bounded by cap)
//   clsr_rows_pack     : rows[i, :] = table[ids[i], :]
//   clsr_rows_unpack   : table[ids[i], :] (= 0 | += rows[i, :]) and flags[ids[i]] = 1
// All three read the count from device memory, so the host never synchronises; every rank applies the
// gathered lists in rank order, so the fp32 sums are bit-identical on all replicas.
#define CMP_SEG 1024  // flags per wave and trip: 64 lanes x 16 bytes
#define CMP_SPW 8     // segments per wave (loads in flight)
#define CMP_SPB (4 * CMP_SPW)   // segments per workgroup: 32 K flags

// sixteen flag bytes from v0 (a multiple of 16): one 16-byte load when the map is 16-byte aligned and whole, else bytes.
// Round 5 read the map one byte per lane (64 bytes per load instruction): 250 us per pass over the 100 MB map of a 100M-item
// catalogue, twice, with a single-workgroup scan of the 97 656 wave counts between them (267 us).
__device__ __forceinline__ uint4 cmp_ld16(const unsigned char* __restrict__ flags, long v0, long V, bool aligned) {
  if (aligned && v0 + 16 <= V) return *reinterpret_cast<const uint4*>(flags + v0);
  unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (v0 + i < V && flags[v0 + i] != 0) w[i >> 2] |= 1u << (8 * (i & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}
// bit 7 of every non-zero byte
__device__ __forceinline__ unsigned cmp_nz(unsigned x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
__device__ __forceinline__ int cmp_count(const uint4& u) {
  return __popc(cmp_nz(u.x)) + __popc(cmp_nz(u.y)) + __popc(cmp_nz(u.z)) + __popc(cmp_nz(u.w));
}
__device__ __forceinline__ int cmp_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// pass 1: workgroup b owns the CMP_SPB segments from b * CMP_SPB on; seg_count[segment], blk_count[b]
__global__ void __launch_bounds__(256) flags_count_kernel(const unsigned char* __restrict__ flags, long V, long nseg,
                                                          int* __restrict__ seg_count, int* __restrict__ blk_count, int aligned) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s0 = (long)blockIdx.x * CMP_SPB + wave * CMP_SPW;
  uint4 u[CMP_SPW];
#pragma unroll
  for (int i = 0; i < CMP_SPW; ++i) {
    const long v0 = (s0 + i) * CMP_SEG + 16 * lane;
    u[i] = (s0 + i < nseg && v0 < V) ? cmp_ld16(flags, v0, V, aligned != 0) : make_uint4(0u, 0u, 0u, 0u);
  }
  int tot = 0;
#pragma unroll
  for (int i = 0; i < CMP_SPW; ++i) {
    const int c = cmp_wave_sum(cmp_count(u[i]));
    if (lane == 0 && s0 + i < nseg) seg_count[s0 + i] = c;
    tot += c;
  }
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) blk_count[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// exclusive scan of the workgroup counts (one workgroup; a few thousand values: every thread owns a contiguous slice); total ->
// count_out[0] (clamped to cap), count_out[1] = 1 if the list had to be truncated (never happens when cap is a true bound)
__global__ void __launch_bounds__(1024) flags_scan_kernel(int* __restrict__ blk_count, int nblk, int cap, int* __restrict__ count_out) {
  __shared__ int part[1024];
  const int per = (nblk + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(nblk, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += blk_count[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the slice sums
    const int add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int b = b0; b < b1; ++b) {
    const int c = blk_count[b];
    blk_count[b] = run;
    run += c;
  }
  if (threadIdx.x == 1023) {
    const int total = part[1023];
    count_out[0] = total < cap ? total : cap;
    count_out[1] = total > cap ? 1 : 0;
  }
}

// pass 2: ids_out[offset of the workgroup + counts of its earlier segments + rank within the segment] = v
__global__ void __launch_bounds__(256) flags_write_kernel(const unsigned char* __restrict__ flags, long V, long nseg,
                                                          const int* __restrict__ seg_count, const int* __restrict__ blk_off,
                                                          int cap, int* __restrict__ ids_out, int id_offset, int aligned) {
  __shared__ int soff[CMP_SPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long sb = (long)blockIdx.x * CMP_SPB;
  if (threadIdx.x < 64) {      // exclusive scan of the workgroup's CMP_SPB (<= 64) segment counts
    const int c = (threadIdx.x < CMP_SPB && sb + threadIdx.x < nseg) ? seg_count[sb + threadIdx.x] : 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(inc, o, 64);
      if (lane >= o) inc += x;
    }
    if (threadIdx.x < CMP_SPB) soff[threadIdx.x] = blk_off[blockIdx.x] + inc - c;
  }
  __syncthreads();
  const long s0 = sb + wave * CMP_SPW;
  uint4 u[CMP_SPW];
#pragma unroll
  for (int i = 0; i < CMP_SPW; ++i) {
    const long v0 = (s0 + i) * CMP_SEG + 16 * lane;
    u[i] = (s0 + i < nseg && v0 < V) ? cmp_ld16(flags, v0, V, aligned != 0) : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int i = 0; i < CMP_SPW; ++i) {
    const int c = cmp_count(u[i]);
    if (__ballot(c > 0) == 0ull) continue;      // (a catalogue's map is almost empty: 0.2 % of the rows are touched)
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(inc, o, 64);
      if (lane >= o) inc += x;
    }
    int p = soff[wave * CMP_SPW + i] + inc - c;
    const long v0 = (s0 + i) * CMP_SEG + 16 * lane;
    const unsigned w[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) {
        if (p < cap) ids_out[p] = (int)(v0 + k) + id_offset;
        ++p;
      }
    }
  }
}

extern "C" long clsr_flags_compact_workspace_bytes(long V) {
  if (V <= 0) return 0;
  const long nseg = (V + CMP_SEG - 1) / CMP_SEG;
  return (long)((nseg + (nseg + CMP_SPB - 1) / CMP_SPB) * sizeof(int) + 256);
}

static int flags_compact_impl(const unsigned char* flags, long V, int* ids_out, int cap, int* count_out,
                              void* workspace, long workspace_bytes, int id_offset, void* stream) {
  CLSR_CHECK_ARG(flags && ids_out && count_out && workspace && V > 0 && cap > 0);
  CLSR_CHECK_SUPPORTED(V < (1L << 31));
  CLSR_CHECK_ARG(workspace_bytes >= clsr_flags_compact_workspace_bytes(V));
  const long nseg = (V + CMP_SEG - 1) / CMP_SEG;
  const int nblk = (int)((nseg + CMP_SPB - 1) / CMP_SPB);
  int* seg = (int*)workspace;
  int* blk = seg + nseg;
  hipStream_t s = (hipStream_t)stream;
  const int aligned = ((uintptr_t)flags % 16) == 0;
  hipLaunchKernelGGL(flags_count_kernel, dim3(nblk), dim3(256), 0, s, flags, V, nseg, seg, blk, aligned);
  CLSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(flags_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nblk, cap, count_out);
  CLSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(flags_write_kernel, dim3(nblk), dim3(256), 0, s, flags, V, nseg, seg, blk, cap, ids_out, id_offset, aligned);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_flags_compact(const unsigned char* flags, long V, int* ids_out, int cap, int* count_out,
                                  void* workspace, long workspace_bytes, void* stream) {
  return flags_compact_impl(flags, V, ids_out, cap, count_out, workspace, workspace_bytes, 0, stream);
}

// the same over the slice flags[0 .. V) of a larger byte map that starts at row id_offset: ids_out holds GLOBAL row ids
// (owner-routed exchange: every rank compacts the range of rows it owns)
extern "C" int clsr_flags_compact_off(const unsigned char* flags, long V, int id_offset, int* ids_out, int cap,
                                      int* count_out, void* workspace, long workspace_bytes, void* stream) {
  CLSR_CHECK_ARG(id_offset >= 0);
  return flags_compact_impl(flags, V, ids_out, cap, count_out, workspace, workspace_bytes, id_offset, stream);
}

// offsets[o] = number of ids (ascending list of count[0] entries) below bounds[o], o <= W: the slices of a sorted id list
// that fall into W contiguous ownership ranges
__global__ void range_offsets_kernel(const int* __restrict__ ids, const int* __restrict__ count,
                                     const int* __restrict__ bounds, int W, int* __restrict__ offsets) {
  const int o = threadIdx.x;
  if (o > W) return;
  const int n = count[0], b = bounds[o];
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ids[mid] < b) lo = mid + 1; else hi = mid;
  }
  offsets[o] = lo;
}

extern "C" int clsr_range_offsets(const int* ids, const int* count, const int* bounds, int W, int* offsets,
                                  void* stream) {
  CLSR_CHECK_ARG(ids && count && bounds && offsets && W > 0 && W < 1024);
  hipLaunchKernelGGL(range_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, count, bounds, W, offsets);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

__global__ void rows_pack_kernel(const float* __restrict__ table, const int* __restrict__ ids,
                                 const int* __restrict__ count, int C, float* __restrict__ rows) {
  const long total = (long)count[0] * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long i = e / C;
    const int c = (int)(e - i * C);
    rows[e] = table[(long)ids[i] * C + c];
  }
}

extern "C" int clsr_rows_pack(const float* table, const int* ids, const int* count, int cap, int C,
                              float* rows_out, void* stream) {
  CLSR_CHECK_ARG(table && ids && count && rows_out && cap > 0 && C > 0);
  int blocks = clsr_cdiv((long)cap * C, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rows_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, ids, count, C,
                     rows_out);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// mode 0: table rows = 0; mode 1: table rows += rows.  ids are unique within one list: no atomics.
__global__ void rows_unpack_kernel(const int* __restrict__ ids, const float* __restrict__ rows,
                                   const int* __restrict__ count, int C, int mode, float* __restrict__ table,
                                   unsigned char* __restrict__ flags) {
  const long total = (long)count[0] * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long i = e / C;
    const int c = (int)(e - i * C);
    const long v = ids[i];
    if (mode == 0) {
      table[v * C + c] = 0.f;
    } else {
      table[v * C + c] += rows[e];
      if (c == 0 && flags) flags[v] = 1;
    }
  }
}

extern "C" int clsr_rows_unpack(const int* ids, const float* rows, const int* count, int cap, int C, int mode,
                                float* table, unsigned char* flags, void* stream) {
  CLSR_CHECK_ARG(ids && count && table && cap > 0 && C > 0 && (mode == 0 || (mode == 1 && rows)));
  int blocks = clsr_cdiv((long)cap * C, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rows_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, rows, count, C,
                     mode, table, flags);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
