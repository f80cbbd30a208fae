// Deterministic gradients of the embedding lookups (gfx950): a STABLE radix sort of the (row id, slice) pairs of a
// lookup site and a segmented sum over the sorted list that adds every row's total to the gradient table exactly once,
// in a fixed order -- two runs of the same step give bit-identical gradient tables and clip norms.
//
// Reference semantics: the gradient of tf.nn.embedding_lookup (models/sequential/sequential_base_model.py:381-452,
// models/sequential/clsr.py:103-135) is an IndexedSlices with one slice per looked-up id; tf.train.AdamOptimizer sums the
// slices of equal ids (unsorted_segment_sum, models/base_model.py:263-276) and tf.clip_by_norm takes the norm of the
// un-summed slices (base_model.py:290-296).  The counting sort of csrc/sparse.hip groups equal ids but leaves the order
// inside a group to racing cursor claims, and its consumer combines partial sums with float atomics: the fp32 sums moved
// in the last bits from run to run.
//
// Sort: least-significant-digit radix sort, 8-bit digits, ceil(bits / 8) passes; a pass = histogram, scan and scatter
// launch for ALL tables of a step (blockIdx.y = table); a table's list may chain two id sources (history lookup + target rows).  (Counting the next pass's digits inside the scatter, with global
// integer atomics at the entries' destinations, saved a launch and cost 350 us: a popular id's entries all hit one counter.)  Stable: inside a workgroup's 2 048
// entries the rank of an entry among the entries with its digit follows the index order (per-wave match masks by ballots,
// per-segment counts scanned in segment order), workgroups are ordered by the scan.  Equal ids therefore stay in
// slice order, and the whole result is a function of the input alone.
//
// Segmented sum (clsr_segsum_multi): thread groups walk chunks of the sorted list and sum runs of equal ids in
// registers, in list order.  A run that lies inside one chunk is added to its row with a plain read-modify-write: no
// other group holds that id (the list is fully sorted), and the caller guarantees that no other launch writes the table
// meanwhile.  A run that crosses chunk borders leaves one partial per chunk in a workspace; a second launch adds the
// partials of such a run IN CHUNK ORDER and writes the row once.  The squared norms of the slices leave as one partial
// per workgroup and are added in workgroup order.  No float atomics anywhere.
#include "common.h"
#include "clsr_hip.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

// ===================================================================================================== stable radix sort
#define RS_EPB 2048                 // entries per workgroup
#define RS_EPT (RS_EPB / 256)       // entries per thread
#define RS_SEG (RS_EPT * 4)         // 64-entry segments per workgroup (round u, wave w) -> u * 4 + w

struct RsTable {
  const int* ids; long row_stride; int ncols;   // pass 0 source: ids[r * row_stride + c], position = r * ncols + c
  const int* ids2; long row_stride2; long n1;   // ... and, for positions >= n1, ids2[(position - n1) * row_stride2]
  int* k[2]; int* p[2];                         // ping-pong (key, position) buffers; the LAST pass writes k[1] / p[1] = outputs
  int* hist;                                    // [2][256 * nblocks] digit counters of the current / next pass
  long n; int nblocks; int passes; int first_dst;
};
struct RsArgs { RsTable t[CLSR_SORTIDS_MAX]; };

__device__ __forceinline__ int rs_key(const RsTable& t, int pass, long e, int src) {
  if (pass == 0) {
    if (e >= t.n1) return t.ids2[(e - t.n1) * t.row_stride2];
    const long r = e / t.ncols;
    return t.ids[r * t.row_stride + (e - r * t.ncols)];
  }
  return t.k[src][e];
}

// digit counts of a pass: hist[block * 256 + digit]
__global__ void __launch_bounds__(256) rs_hist_kernel(RsArgs a, int pass) {
  __shared__ int h[256];
  const RsTable& t = a.t[blockIdx.y];
  if (pass >= t.passes || (int)blockIdx.x >= t.nblocks) return;
  const int src = (t.first_dst + pass + 1) & 1;
  h[threadIdx.x] = 0;
  __syncthreads();
  const long e0 = (long)blockIdx.x * RS_EPB;
  const int shift = 8 * pass;
#pragma unroll
  for (int u = 0; u < RS_EPT; ++u) {
    const long e = e0 + u * 256 + threadIdx.x;
    if (e < t.n) atomicAdd(&h[(rs_key(t, pass, e, src) >> shift) & 255], 1);
  }
  __syncthreads();
  t.hist[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// exclusive scan of the 256 * nblocks counters of one table (block-major: hist[block * 256 + digit]) in digit-major
// order, in place: offset[b][d] = sum of the totals of the digits below d + the counts of digit d in the blocks before b.
// One workgroup of 1024 threads per table: thread (q, d) owns digit d of a quarter of the blocks; its counters are read
// eight at a time (independent loads in flight), the 4 x 256 partial sums and the 256 digit totals are scanned in LDS,
// and a second walk writes the running offsets.  (Tiled scans with barriers per tile, and per-digit wave scans, both
// took 80-140 us per pass for 28 000 counters: chains of dependent trips to memory.)
__global__ void __launch_bounds__(1024) rs_scan_kernel(RsArgs a, int pass) {
  __shared__ int psum[4][256], dbase[256];
  const RsTable& t = a.t[blockIdx.x];
  if (pass >= t.passes) return;
  int* c = t.hist;
  const int nbk = t.nblocks;
  const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int nbq = (nbk + 3) >> 2;
  const int b0 = q * nbq, b1 = min(nbk, b0 + nbq);
  int s = 0;
  for (int b = b0; b < b1; b += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = b + u < b1 ? c[(b + u) * 256 + d] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  psum[q][d] = s;
  __syncthreads();
  if (threadIdx.x < 64) {      // exclusive scan of the 256 digit totals: 4 per lane
    const int lane = threadIdx.x;
    int v[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int dd = 4 * lane + k;
      v[k] = psum[0][dd] + psum[1][dd] + psum[2][dd] + psum[3][dd];
      mine += v[k];
    }
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(inc, o, 64);
      if (lane >= o) inc += x;
    }
    int run = inc - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) { dbase[4 * lane + k] = run; run += v[k]; }
  }
  __syncthreads();
  int run = dbase[d];
  for (int qq = 0; qq < q; ++qq) run += psum[qq][d];
  for (int b = b0; b < b1; b += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = b + u < b1 ? c[(b + u) * 256 + d] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b + u < b1) c[(b + u) * 256 + d] = run;
      run += v[u];
    }
  }
}

// one pass: entries of workgroup b go to offset[digit][b] + (stable rank among the workgroup's entries with that digit)
__global__ void __launch_bounds__(256) rs_scatter_kernel(RsArgs a, int pass) {
  __shared__ int segcnt[RS_SEG][256];
  __shared__ int gbase[256];
  const RsTable& t = a.t[blockIdx.y];
  if (pass >= t.passes || (int)blockIdx.x >= t.nblocks) return;
  const int src = (t.first_dst + pass + 1) & 1, dst = (t.first_dst + pass) & 1;
  const int* hist = t.hist;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < RS_SEG * 256; e += 256) (&segcnt[0][0])[e] = 0;
  gbase[threadIdx.x] = hist[blockIdx.x * 256 + threadIdx.x];
  __syncthreads();
  const long e0 = (long)blockIdx.x * RS_EPB;
  const int shift = 8 * pass;
  int key[RS_EPT], pos[RS_EPT], rank[RS_EPT];
#pragma unroll
  for (int u = 0; u < RS_EPT; ++u) {
    const long e = e0 + u * 256 + threadIdx.x;
    const bool valid = e < t.n;
    key[u] = valid ? rs_key(t, pass, e, src) : 0;
    pos[u] = valid ? (pass == 0 ? (int)e : t.p[src][e]) : 0;
    const int d = (key[u] >> shift) & 255;
    // lanes of this wave with the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int r = __popcll(peers & ((1ull << lane) - 1ull));
    rank[u] = valid ? r : -1;
    if (valid && r == 0) segcnt[u * 4 + wave][d] = __popcll(peers);
  }
  __syncthreads();
  {  // per digit: exclusive scan over the segments, in segment (= index) order
    int run = 0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) {
      const int c = segcnt[s][threadIdx.x];
      segcnt[s][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < RS_EPT; ++u) {
    if (rank[u] < 0) continue;
    const int d = (key[u] >> shift) & 255;
    const int o = gbase[d] + segcnt[u * 4 + wave][d] + rank[u];
    t.k[dst][o] = key[u];
    t.p[dst][o] = pos[u];
  }
}

__global__ void rs_zero_kernel(int* __restrict__ p, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) p[e] = 0;
}

static long rs_table_ints(long n) {
  const long nb = (n + RS_EPB - 1) / RS_EPB;
  return 2 * n + 2 * 256 * nb + 64;        // tmp keys, tmp positions, two digit-counter arrays
}
// bytes of workspace for sorting n_tables tables of total_entries entries in all together
extern "C" long clsr_sort_ids_stable_workspace_bytes(long total_entries, int n_tables) {
  if (total_entries < 0 || n_tables <= 0) return 0;
  return (2 * total_entries + 2 * 256 * (total_entries / RS_EPB + n_tables) + 64L * n_tables) * (long)sizeof(int) + 256;
}

// keys_out / perm_out: the (id, position) pairs of ids[r*row_stride + c] (r < nrows, c < ncols; position = r*ncols + c)
// in ascending id order, equal ids in ascending position order.  desc.bits = significant bits of the ids
// (ceil(log2(vocab))), desc.counts is not used.  3 * ceil(max bits / 8) launches whatever the number of tables.
extern "C" int clsr_sort_ids_stable_multi(const clsr_sortids_desc* descs, int n, void* workspace, long workspace_bytes,
                                          void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_SORTIDS_MAX && workspace);
  RsArgs a;
  int* w = (int*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  long used = 0, mxb = 1, hist_ints = 0;
  int maxp = 0;
  for (int i = 0; i < n; ++i) {
    const clsr_sortids_desc& d = descs[i];
    CLSR_CHECK_ARG(d.ids && d.keys_out && d.perm_out && d.nrows > 0 && d.ncols > 0 && d.bits >= 1 && d.bits <= 31);
    CLSR_CHECK_ARG(d.nrows2 >= 0 && (d.nrows2 == 0 || d.ids2));
    const long e = d.nrows * d.ncols + d.nrows2;
    CLSR_CHECK_SUPPORTED(e < (1L << 31) - RS_EPB);
    RsTable& t = a.t[i];
    t.ids = d.ids; t.row_stride = d.row_stride; t.ncols = d.ncols; t.n = e;
    t.ids2 = d.ids2; t.row_stride2 = d.row_stride2; t.n1 = d.nrows * d.ncols;
    t.nblocks = (int)((e + RS_EPB - 1) / RS_EPB);
    t.passes = (d.bits + 7) / 8;
    t.k[1] = d.keys_out; t.p[1] = d.perm_out;
    t.k[0] = w + used; t.p[0] = w + used + e;
    t.hist = w + used + 2 * e;
    // pass p writes buffer (first_dst + p) & 1; the last pass must write buffer 1
    t.first_dst = (t.passes & 1) ? 1 : 0;
    used += rs_table_ints(e);
    hist_ints += 2L * 256 * t.nblocks;
    mxb = t.nblocks > mxb ? t.nblocks : mxb;
    maxp = t.passes > maxp ? t.passes : maxp;
  }
  CLSR_CHECK_ARG(workspace_bytes >= used * (long)sizeof(int) + 16);
  hipStream_t s = (hipStream_t)stream;
  // (the first pass's counters are written by rs_hist_kernel; every scan launch clears the other half before its scatter
  //  accumulates the next pass's counts into it)
  for (int p = 0; p < maxp; ++p) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3((int)mxb, n), dim3(256), 0, s, a, p);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(n), dim3(1024), 0, s, a, p);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3((int)mxb, n), dim3(256), 0, s, a, p);
    CLSR_CHECK_LAUNCH();
  }
  (void)hist_ints;
  return CLSR_OK;
}

// ===================================================================================================== segmented sums
// One lookup SITE of a launch: rows `key` of `grad` receive the sums of the slices  g[e, :] = src[pos(e), col0 : col0 + C]
// (+ src2, + the mean / recent-k terms of the history prologue when dmean / drecent are given: pos = h * T + t).
//
// ONE launch (round 6; rounds 4-5: a chunk walk + a border launch).  Thread groups walk chunks of SS_CHUNK sorted entries
// and sum runs of equal ids in registers, in list order; a run that lies inside one chunk is written to its row once.
// Runs that cross a chunk border:
//  * SHORT continuation -- the run ends within the first SS_E entries of the next chunk (decided from the sorted keys
//    alone, the same answer on both sides of the border): the LEFT chunk keeps walking those entries, the right chunk skips
//    them.  No partial, no communication; on lists whose ids repeat a few times (the benchmark's catalogue feed) this is every
//    border.
//  * LONG runs (popular ids of a Zipf list: the longest spans ~800 chunks): every chunk of the run leaves ONE partial in the
//    workspace with device-coherent stores and then raises its ready word; the chunk in which the run ENDS (the tail) owns the
//    row: its whole workgroup finds the run's extent from the key list, waits for the ready words of the EARLIER chunks
//    (decoupled look-back: workgroups are dispatched in index order and a chunk only ever waits for lower-numbered ones, so
//    the wait always ends), adds the partials in a fixed order and writes the row.  A ready word has exactly ONE reader -- the
//    tail of the run its chunk feeds -- which clears it again: the workspace needs no clearing between launches and a replayed
//    launch plan / hipGraph passes no counter.  (A launch-wide counter of finished workgroups -- the first version: epochs as
//    ready values, the last workgroup folds the squared norms -- cost 8 us: ~900 device-scope atomics on one word at the end
//    of a launch whose workgroups all finish together.)
// The squared norms of the slices leave as one partial per workgroup and are added in workgroup order by a one-workgroup
// launch behind the walk (only when a site asks for them).
// The workspace must be ZERO when it is first used and must not be shared by two launches in flight.  No float atomics;
// every addition has a fixed place in a fixed order: bit-identical runs.
#ifndef SS_BU
#define SS_BU 8          // partial rows of a lane slot in flight in the tail combine (power of two)
#endif
#ifndef SS_CHUNK
#define SS_CHUNK 32                 // sorted entries per thread group
#endif
#ifndef SS_LEAN_B
#define SS_LEAN_B 8                 // entries of a chunk in flight per thread group in the LEAN instantiation (8 or 16; SS_CHUNK % SS_LEAN_B == 0)
#endif
#define SS_E 8                      // a continuation of at most this many entries is walked by the chunk the run comes from
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct SsHdr { unsigned err; unsigned pad[3]; };
struct SsSite {
  const void* src; const void* src2; int src_bf16; const float* dmean; const float* drecent;
  const int* keys; const int* perm; const int* seq_len; int len_stride;
  long n; int T; int D; int col0; int C; int recent_k;
  float* grad; int ldg; int gcol0;
  double* sumsq;                  // += sum of the squared slice values (may be NULL)
  int assign;                     // row totals are stored, not added
  const float* src_b; double* sumsq_b; long n1; int ldb; int colb;   // second source (entries with perm >= n1), n1 = 0: none
  float* bnd; int* meta; double* ssp;   // workspace: chunk-border partials [nchunks][2][Cp], [nchunks][4] ints, [blocks][2] doubles
  int first_block; int nblocks; int cp; int vw;
};
struct SsArgs { SsSite s[CLSR_SEGSUM_MAX]; int n; SsHdr* hdr; };

template <int VW> struct SsVec { typedef float type; };
template <> struct SsVec<4> { typedef f32x4 type; };
template <int VW>
__device__ __forceinline__ typename SsVec<VW>::type ss_ld(const void* p, int bf16, long off) {
  if constexpr (VW == 4) {
    return bf16 ? load4e<true>(p, off) : load4e<false>(p, off);
  } else {
    return bf16 ? (float)reinterpret_cast<const __bf16*>(p)[off] : reinterpret_cast<const float*>(p)[off];
  }
}
__device__ __forceinline__ float ss_sq(float x) { return x * x; }
__device__ __forceinline__ float ss_sq(const f32x4& x) { return x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }

// device-coherent accesses (write-through, L2-bypassing -- the pattern of csrc/headsfused.hip): only the few partial rows of
// LONG runs and the per-workgroup squared norms travel this way; a release fence would write back an L2 full of gradient rows
#define SS_WAIT_MEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ void ss_cst(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ss_cst(float* p, const f32x4& v) { ss_cst(p, v.x); ss_cst(p + 1, v.y); ss_cst(p + 2, v.z); ss_cst(p + 3, v.w); }
__device__ __forceinline__ void ss_csti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ss_cldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int VW> __device__ __forceinline__ typename SsVec<VW>::type ss_cld(const float* p) {
  if constexpr (VW == 4) {
    f32x4 v;
    v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  } else {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// meta[chunk] (published by chunks whose LAST run goes on into the next chunk only): 0 first key, 1 last key, 2 flags (1: the
// first run comes from the previous chunk, 2: the last run goes on into the next chunk, 4: the whole chunk is one run), 3 ready
// (1 while the partial waits for its reader)
// eight consecutive ints a[q0 .. q0 + 7] (fill beyond ``pe``): two 16-byte loads when the batch is whole and aligned
__device__ __forceinline__ void ss_ld8(const int* __restrict__ a, const long q0, const long pe, const int fill, int out[8]) {
  if (q0 + 8 <= pe && ((reinterpret_cast<uintptr_t>(a + q0) & 15) == 0)) {
    const int4 u = *reinterpret_cast<const int4*>(a + q0), w = *reinterpret_cast<const int4*>(a + q0 + 4);
    out[0] = u.x; out[1] = u.y; out[2] = u.z; out[3] = u.w;
    out[4] = w.x; out[5] = w.y; out[6] = w.z; out[7] = w.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = q0 + k < pe ? a[q0 + k] : fill;
  }
}

// SB consecutive ints a[q0 .. q0 + SB - 1] (fill beyond ``pe``)
template <int SB>
__device__ __forceinline__ void ss_ldn(const int* __restrict__ a, const long q0, const long pe, const int fill, int (&out)[SB]) {
  if constexpr (SB % 8 == 0) {
#pragma unroll
    for (int b8 = 0; b8 < SB; b8 += 8) ss_ld8(a, q0 + b8, pe, fill, out + b8);
  } else {
#pragma unroll
    for (int k = 0; k < SB; ++k) out[k] = q0 + k < pe ? a[q0 + k] : fill;
  }
}

// LEAN: the site has no second gradient tensor and no mean / recent shares (src2, dmean, drecent all NULL) and its row totals
// are stored (assign): four registers per entry in flight instead of twenty -- the kernel is bound by the number of row
// reads it keeps in flight (profiles/r04_embed_kernel_trace.md: traffic = the algorithmic bytes at 3 TB/s), and with 240
// VGPRs only two waves per SIMD were resident.
// bf16 source rows whose slices start on 16-byte boundaries (launch-uniform): the PAIR instantiation of the lean walk
__device__ __forceinline__ bool ss_pair16(const SsSite& s) {
  return s.src_bf16 && (s.D & 7) == 0 && (s.col0 & 7) == 0 && (reinterpret_cast<uintptr_t>(s.src) & 15) == 0;
}
template <int VW, bool LEAN, int SB = 8, bool PAIR = false>
__device__ __forceinline__ void ss_chunks(const SsSite& s, const int local_block, double* red, int* sh, float* shf, SsHdr* hdr) {
  typedef typename SsVec<VW>::type vec_t;
  // SB = 8: four batches cover the chunk, a short continuation (<= 8 entries) is one more batch -- on ~7 % of the chunks of the
  // benchmark's catalogue feed; every wave is resident from the start, so the launch lasts as long as its LONGEST chain: five
  // batches instead of four (42 us against the 36 us of a run-free walk).  SB = 9 (instantiable: the four batches have 36
  // slots, the last four hold a continuation of <= 4 entries, EVERY chunk is four dependent batches) measured 52.7 us: 140
  // VGPRs (three waves per SIMD), or 128 with 15 spilled, and unaligned key loads.
  constexpr int NBATCH = SS_CHUNK / (SB >= 16 ? SB : 8);
  constexpr bool WIDE = SB * NBATCH > SS_CHUNK;
  constexpr int E = WIDE ? SB * NBATCH - SS_CHUNK : SS_E;
  static_assert(E <= SB && E <= SS_CHUNK, "a short continuation fits one batch");
  const int CP = s.cp;                         // lanes per thread group (power of two, 8..64)
  const int gpb = 256 / CP;
  const int lig = threadIdx.x & (CP - 1);      // lane in group
  const int gi = threadIdx.x / CP;             // group in workgroup
  const int c = lig * VW;
  const bool cok = c < s.C;
  const long chunk = (long)local_block * gpb + gi;
  const long nchunks = (s.n + SS_CHUNK - 1) / SS_CHUNK;
  const long p0 = chunk * SS_CHUNK;
  const int Cp = CP * VW;
  float local = 0.f, local_b = 0.f;
  int flags = 0, first_key = -1, cur = -1;
  vec_t pfirst = vec_t(0.f);                   // this chunk's share of a run that comes from the previous chunk
  if (chunk < nchunks) {
    const long cc = s.col0 + (cok ? c : 0);
    const long pe = p0 + SS_CHUNK < s.n ? p0 + SS_CHUNK : s.n;
    // the keys on both sides of the two borders: one round trip
    const int kprev = p0 > 0 ? s.keys[p0 - 1] : -1;
    const int kpe = p0 + E < s.n ? s.keys[p0 + E] : -2;          // (-2: no such entry)
    const int klast = s.keys[pe - 1];
    const int knx = pe < s.n ? s.keys[pe] : -1;
    const int kne = pe + E < s.n ? s.keys[pe + E] : -2;
    constexpr bool PIPED = LEAN && SB * (SS_CHUNK / (SB >= 16 ? SB : 8)) <= SS_CHUNK && VW == 4;    // (the pipelined walk below)
    int keyN[SB], posN[SB];
    if constexpr (PIPED) {
      keyN[0] = s.keys[p0];                      // (the walk takes its keys / slice numbers from LDS)
    } else {
      ss_ldn<SB>(s.keys, p0, pe, -1, keyN);
      ss_ldn<SB>(s.perm, p0, pe, 0, posN);
    }
    // a run over the border at p0 / pe that ends within SS_E entries behind it belongs to the chunk it comes from
    const bool short_prev = kprev >= 0 && keyN[0] == kprev && kpe != kprev;
    const bool short_next = knx >= 0 && klast == knx && kne != knx;
    const int prev_key = short_prev ? -2 : kprev;      // (-2: never the key of an entry)
    const int next_key = short_next ? -2 : knx;
    const long pe_walk = WIDE ? p0 + (long)SB * NBATCH : (short_next ? pe + SB : pe);     // (not WIDE: one more batch for the continuation)
    const long lim = short_next ? (pe + E < s.n ? pe + E : s.n) : pe;         // entries that may be read
    if constexpr (!PIPED) {
      if (short_prev) {
#pragma unroll
        for (int k = 0; k < E; ++k) if (keyN[k] == kprev) keyN[k] = -1;         // (a prefix: the list is sorted)
      }
    }
    int nruns = 0;
    vec_t acc = vec_t(0.f);
    float* bnd = s.bnd + chunk * 2 * Cp;
    constexpr bool pair16 = PAIR;       // (an instantiation of its own, chosen by ss_pair16: the fp32 walk keeps its code)
    auto flush = [&](bool last) {
      // the run `cur` ends here (last: at the end of the chunk)
      const bool from_prev = nruns == 0 && cur == prev_key;
      const bool to_next = last && cur == next_key;
      if (from_prev || to_next) {
        if (from_prev) pfirst = acc;
        if (cok) ss_cst(bnd + (from_prev ? 0 : Cp) + c, acc);
      } else if (cok) {
        vec_t* g = reinterpret_cast<vec_t*>(s.grad + (long)cur * s.ldg + s.gcol0 + c);
        *g = (LEAN || s.assign) ? acc : *g + acc;
      }
      ++nruns;
    };
    if constexpr (PIPED) {
      // ---- LEAN walk, software-pipelined: the row reads of batch b + 1 are in flight while batch b is summed (keys two batches
      //      ahead).  With the loads of a batch issued only after the previous batch had been consumed, a chunk was a chain of
      //      4-5 dependent row-read round trips: with every row read a cache hit and no row stores the launch still took 27 us
      //      of its 43 (scripts/bench_segsum.py, -DSS_ABL_NOLOAD / -DSS_ABL_NOSTORE builds).
      const int cs = cok ? c : 0;
      // keys / slice numbers of the walk ([p0, p0 + SS_CHUNK + E)) staged in LDS once, in the round trip of the border keys
      // above: the batches read them from there (no key registers held across the row reads: 162 registers -> three waves per
      // SIMD with them, and the launch no longer fitted the chip at once)
      __shared__ int kp_all[32 * (2 * (SS_CHUNK + SS_E))];
      int* lk = kp_all + gi * (2 * (SS_CHUNK + SS_E));
      int* lp = lk + SS_CHUNK + SS_E;
      // (every entry that exists: which of those behind the chunk may be READ -- ``lim`` -- hangs on the border keys above; with
      //  the bound applied when a batch is fetched these loads leave together with those, not one round trip behind them)
      for (int e = lig; e < SS_CHUNK + SS_E; e += CP) {
        const long q = p0 + e;
        lk[e] = q < s.n ? s.keys[q] : -1;
        lp[e] = q < s.n ? s.perm[q] : 0;
      }
      // (a thread group lies inside ONE wave: its LDS writes are visible to its own later reads once they have completed)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      auto fetch = [&](const long q0, int (&kk)[SB], int (&pp)[SB]) {
        if (q0 < pe_walk) {
          const int o = (int)(q0 - p0);
#pragma unroll
          for (int k = 0; k < SB; ++k) {
            const bool in = q0 + k < lim;
            kk[k] = in ? lk[o + k] : -1;
            pp[k] = in ? lp[o + k] : 0;
          }
          if (q0 == p0 && short_prev) {
#pragma unroll
            for (int k = 0; k < E; ++k) if (kk[k] == kprev) kk[k] = -1;       // (a prefix: the list is sorted)
          }
          if (q0 >= pe) {                    // the continuation batch: the entries of the run that crosses the border only
#pragma unroll
            for (int k = 0; k < SB; ++k) if (kk[k] != knx) kk[k] = -1;
          }
        } else {
#pragma unroll
          for (int k = 0; k < SB; ++k) { kk[k] = -1; pp[k] = 0; }
        }
      };
      auto issue = [&](const int (&pp)[SB], vec_t (&rv)[SB], unsigned& secm) {
        secm = 0u;
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int pos = pp[k];
          const bool second = s.n1 > 0 && pos >= s.n1;          // (uniform inside the thread group)
          secm |= second ? 1u << k : 0u;
          if constexpr (PAIR) {
            // bf16 source: ONE 16-byte load per entry whatever its source -- an fp32 slice of the second list, or the 16 bytes
            // that hold this lane's four bf16 values and its pair lane's (the half is picked when the batch is summed)
            const char* p = second ? reinterpret_cast<const char*>(s.src_b + (long)(pos - s.n1) * s.ldb + s.colb + cs)
                                   : reinterpret_cast<const char*>(s.src) + ((((long)pos * s.D + cc) * 2) & ~15L);
            rv[k] = *reinterpret_cast<const f32x4*>(p);
          } else {
            const float* p = second ? s.src_b + (long)(pos - s.n1) * s.ldb + s.colb + cs
                                    : reinterpret_cast<const float*>(s.src) + (long)pos * s.D + cc;
            rv[k] = *reinterpret_cast<const vec_t*>(p);
          }
        }
      };
      auto consume = [&](const long q0, const unsigned secm, const vec_t (&rv)[SB]) {
        int key[SB], pu[SB];
        fetch(q0, key, pu);
        const int knext = (q0 + SB < pe_walk && q0 + SB < lim) ? lk[(int)(q0 + SB - p0)] : -1;      // (the continuation batch starts with the run's key)
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          if (key[k] < 0) continue;
          vec_t v = rv[k];
          const bool second = (secm >> k) & 1u;
          if constexpr (PAIR) {
            const u32x4_t raw = __builtin_bit_cast(u32x4_t, v);
            const unsigned lo = (lig & 1) ? raw.z : raw.x, hi = (lig & 1) ? raw.w : raw.y;
            const f32x4 vb = {__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                              __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
            v = second ? v : vb;
          }
          if (!cok) v = vec_t(0.f);
          if (second) local_b += ss_sq(v);
          else local += ss_sq(v);
          if (key[k] != cur) {
            if (cur < 0) first_key = key[k];
            cur = key[k];
            acc = v;
          } else {
            acc += v;
          }
          int nk = key[k];
          if (k < SB - 1) { if (key[k + 1] >= 0) nk = key[k + 1]; }
          else if (q0 + SB < pe_walk && knext >= 0) nk = knext;
          if (nk != key[k]) {
            const bool from_prev = nruns == 0 && cur == prev_key;
            if (from_prev) pfirst = acc;
            if (cok) {
              if (from_prev) ss_cst(bnd + c, acc);
              else *reinterpret_cast<vec_t*>(s.grad + (long)cur * s.ldg + s.gcol0 + c) = acc;
            }
            ++nruns;
          }
        }
      };
      int ku[SB], pp[SB];
      vec_t rA[SB], rB[SB];
      unsigned sA = 0u, sB = 0u;
      fetch(p0, ku, pp);
      issue(pp, rA, sA);
      fetch(p0 + SB, ku, pp);
      issue(pp, rB, sB);
      for (long q0 = p0; q0 < pe_walk; q0 += 2 * SB) {
        consume(q0, sA, rA);
        if (q0 + 2 * SB < pe_walk) { fetch(q0 + 2 * SB, ku, pp); issue(pp, rA, sA); }
        if (q0 + SB < pe_walk) consume(q0 + SB, sB, rB);
        if (q0 + 3 * SB < pe_walk) { fetch(q0 + 3 * SB, ku, pp); issue(pp, rB, sB); }
      }
    } else
    // keys / slice numbers of a batch of eight are fetched ONE BATCH AHEAD (two 16-byte loads each): the row reads of a
    // batch depend on them, and with both fetched in the batch itself every batch paid two memory round trips in a row
    for (long q0 = p0; q0 < pe_walk; q0 += SB) {
      int key[SB], posk[SB];
      bool sec[SB];
      vec_t g[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k) { key[k] = keyN[k]; posk[k] = posN[k]; }
      // every load of the batch is UNCONDITIONAL, from a selected / clamped address, and the shares are combined afterwards:
      // with a branch per source (second list, mean / recent shares behind the sequence length) the loads of the eight
      // entries left one branch at a time -- in the step (mean and recent shares present) this launch took 84 us against
      // 28 us for the bare rows
      vec_t rv[SB], r2[SB], mv[SB], rc[SB];
      int ln[SB], tt[SB];
      const bool has_mr = !LEAN && (s.dmean || s.drecent);          // (launch-uniform, like src2 / src_bf16)
      const int cs = cok ? c : 0;
      if constexpr (pair16) {
        // LEAN, bf16 source: ONE 16-byte load per entry whatever its source -- an fp32 slice of the second list, or the 16
        // bytes that hold this lane's four bf16 values and its pair lane's (the same lines as 8-byte loads; the half is
        // picked in the combine loop below).  The branch sits OUTSIDE the loop over the entries: inside it (the form below)
        // the compiler converts each bf16 load where it is issued, behind an s_waitcnt vmcnt(0) -- the eight row reads of a
        // batch left one round trip at a time, 62 us for the item site of configs[4] where the fp32 form takes 43
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int pos = posk[k];
          const bool second = s.n1 > 0 && pos >= s.n1;
          sec[k] = second; tt[k] = 0; ln[k] = 0;
          const char* p = second ? reinterpret_cast<const char*>(s.src_b + (long)(pos - s.n1) * s.ldb + s.colb + cs)
                                 : reinterpret_cast<const char*>(s.src) + ((((long)pos * s.D + cc) * 2) & ~15L);
          if constexpr (VW == 4) rv[k] = *reinterpret_cast<const f32x4*>(p);
        }
      } else
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int pos = posk[k];
        const bool second = s.n1 > 0 && pos >= s.n1;          // (uniform inside the thread group)
        sec[k] = second;
        const int pp = second ? 0 : pos;
        const int h = pp / s.T;
        tt[k] = pp - h * s.T;
        ln[k] = has_mr ? s.seq_len[(long)h * s.len_stride] : 0;
        if (s.src_bf16) {
          // (both sources unconditionally, from clamped rows, selected afterwards: under `if (second)` -- uniform in a thread
          // group, not in a wave -- the eight loads of a batch left one branch at a time and the bf16 form of this launch,
          // which moves 0.83 x the bytes, took 63 us against 56 us for fp32)
          const vec_t vb = ss_ld<VW>(s.src, 1, (long)pp * s.D + cc);
          const vec_t vs = s.n1 > 0 ? *reinterpret_cast<const vec_t*>(s.src_b + (long)(second ? pos - s.n1 : 0) * s.ldb + s.colb + cs)
                                    : vec_t(0.f);
          rv[k] = second ? vs : vb;
          if (!LEAN && s.src2) r2[k] = second ? vec_t(0.f) : ss_ld<VW>(s.src2, 1, (long)pp * s.D + cc);
        } else {
          const float* p = second ? s.src_b + (long)(pos - s.n1) * s.ldb + s.colb + cs
                                  : reinterpret_cast<const float*>(s.src) + (long)pos * s.D + cc;
          rv[k] = *reinterpret_cast<const vec_t*>(p);
          if (!LEAN && s.src2) r2[k] = *reinterpret_cast<const vec_t*>(reinterpret_cast<const float*>(s.src2) + (long)pp * s.D + cc);
        }
        if (!LEAN && s.dmean) mv[k] = *reinterpret_cast<const vec_t*>(s.dmean + (long)h * s.D + cc);
        if (!LEAN && s.drecent) rc[k] = *reinterpret_cast<const vec_t*>(s.drecent + (long)h * s.D + cc);
      }
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        vec_t v = rv[k];
        if constexpr (VW == 4 && LEAN) {
          if constexpr (pair16) {        // (selects, no branch: the raw 16 bytes of a bf16 row -> this lane's four values)
            const u32x4_t raw = __builtin_bit_cast(u32x4_t, v);
            const unsigned lo = (lig & 1) ? raw.z : raw.x, hi = (lig & 1) ? raw.w : raw.y;
            const f32x4 vb = {__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                              __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
            v = sec[k] ? v : vb;
          }
        }
        if (!sec[k]) {
          if (!LEAN && s.src2) v += r2[k];
          const int len = ln[k];
          if (has_mr && tt[k] < len) {
            if (!LEAN && s.dmean) v += mv[k] * (1.0f / (float)len);
            if (!LEAN && s.drecent && tt[k] >= len - s.recent_k) v += rc[k] * (1.0f / (float)(len < s.recent_k ? len : s.recent_k));
          }
        }
        g[k] = (cok && key[k] >= 0) ? v : vec_t(0.f);
      }
      if (q0 + SB < pe_walk) {
        ss_ldn<SB>(s.keys, q0 + SB, lim, -1, keyN);
        ss_ldn<SB>(s.perm, q0 + SB, lim, 0, posN);
        if constexpr (WIDE) {
          if (q0 + 2 * SB > pe) {          // slots behind the chunk: the entries of the run that crosses the border only
#pragma unroll
            for (int k = 0; k < SB; ++k) if (q0 + SB + k >= pe && keyN[k] != knx) keyN[k] = -1;
          }
        } else if (q0 + SB >= pe) {        // the continuation batch: the entries of the run that crosses the border only
#pragma unroll
          for (int k = 0; k < SB; ++k) if (keyN[k] != knx) keyN[k] = -1;
        }
      } else {
#pragma unroll
        for (int k = 0; k < SB; ++k) { keyN[k] = -1; posN[k] = 0; }
      }
      const int knext = keyN[0];
      // the gradient rows of all eight entries' ids are requested NOW, whether or not a run ends there: a run that ends
      // inside the chunk is added to its row with a read-modify-write, and one dependent row read per run end, one after
      // the other along the walk, made this launch 110 us for 640 chunks
      // (assign: the rows are known to be zero and this call is their only writer -- no row read at all)
      vec_t gv[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k)
        gv[k] = (LEAN || s.assign) ? vec_t(0.f)
                         : *reinterpret_cast<const vec_t*>(s.grad + (long)(key[k] < 0 ? 0 : key[k]) * s.ldg + s.gcol0 + (cok ? c : 0));
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        if (key[k] < 0) continue;
        if (sec[k]) local_b += ss_sq(g[k]);
        else local += ss_sq(g[k]);
        if (key[k] != cur) {
          if (cur < 0) first_key = key[k];
          cur = key[k];
          acc = g[k];
        } else {
          acc += g[k];
        }
        // does the run end at this entry?  (the last entry of the walk leaves its run open: see below)
        int nk = key[k];
        if (k < SB - 1) { if (key[k + 1] >= 0) nk = key[k + 1]; }
        else if (q0 + SB < pe_walk && knext >= 0) nk = knext;
        if (nk != key[k]) {
          const bool from_prev = nruns == 0 && cur == prev_key;
          if (from_prev) pfirst = acc;
          if (cok) {
            if (from_prev) ss_cst(bnd + c, acc);
            else *reinterpret_cast<vec_t*>(s.grad + (long)cur * s.ldg + s.gcol0 + c) = gv[k] + acc;
          }
          ++nruns;
        }
      }
    }
    if (cur >= 0) {
      const bool single = nruns == 0;                 // the chunk is one run
      const bool to_next = cur == next_key;
      flush(true);
      if (first_key == prev_key && prev_key >= 0) flags |= 1;
      if (to_next) flags |= 2;
      if (single) flags |= 4;
      if (!(flags & 3)) flags = 0;                    // (no share of a long run: nothing to publish)
    }
    if (flags & 2) {                                  // a later chunk will read this chunk's partial
      SS_WAIT_MEM();                                  // the partial rows of this wave have left
      if (lig == 0) {
        int* m = s.meta + chunk * 4;
        ss_csti(m, first_key); ss_csti(m + 1, cur); ss_csti(m + 2, flags);
      }
      SS_WAIT_MEM();
      if (lig == 0) ss_csti(s.meta + chunk * 4 + 3, 1);
    }
  }
  // ---- the chunk in which a LONG run ends owns its row.  Extent of the run: the chunks before this one whose LAST entry
  //      carries the key (read from the key list -- no communication); a run that started within the CP chunks before this
  //      one is finished by the thread group itself (no barrier: the groups of a wave are in lockstep): its lanes wait for the
  //      ready words of those chunks, then the partials are added in chunk order.  Longer runs: the whole workgroup (below).
  const bool is_tail = chunk < nchunks && (flags & 1) && !((flags & 4) && (flags & 2));
  bool deferred = false;
  if (is_tail) {
    const long cb = chunk - 1 - lig;
    const bool member = cb >= 0 && s.keys[(cb + 1) * SS_CHUNK - 1] == first_key;
    const unsigned long long bm = __ballot(member) >> ((threadIdx.x & 63) & ~(CP - 1));
    const unsigned long long mine = CP == 64 ? bm : (bm & ((1ull << CP) - 1ull));
    const int L = mine == (CP == 64 ? ~0ull : ((1ull << CP) - 1ull)) ? CP : __builtin_ctzll(~mine);
    if (L >= CP) {
      deferred = true;
    } else {
      // lane i < L waits for chunk - 1 - i (a LOWER-numbered chunk: its workgroup was dispatched before this one and waits
      // for nothing this workgroup has to give; bounded all the same -- two seconds of the 100 MHz clock)
      const long long t0 = wall_clock64();
      if (lig < L) {
        while (ss_cldi(s.meta + (chunk - 1 - lig) * 4 + 3) != 1) {
          if (wall_clock64() - t0 > 200000000LL) { __hip_atomic_store(&hdr->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int cs = cok ? c : 0;
      // head (chunk - L): the partial of its LAST run (slot 1); chunks in between: their only run (slot 0); then this chunk's
      vec_t tot = ss_cld<VW>(s.bnd + (chunk - L) * 2 * Cp + Cp + cs);
      for (int i0 = 1; i0 < L; i0 += 8) {
        vec_t pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = ss_cld<VW>(s.bnd + (chunk - L + (i0 + u < L ? i0 + u : L - 1)) * 2 * Cp + cs);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u < L) tot += pv[u];
      }
      tot += pfirst;
      if (cok) {
        vec_t* g = reinterpret_cast<vec_t*>(s.grad + (long)first_key * s.ldg + s.gcol0 + c);
        *g = (LEAN || s.assign) ? tot : *g + tot;
      }
      if (lig < L) ss_csti(s.meta + (chunk - 1 - lig) * 4 + 3, 0);      // (read: the words are clear again for the next launch)
    }
  }
  if (lig == 0) { sh[2 * gi] = deferred ? flags : 0; sh[2 * gi + 1] = first_key; }
  // ---- squared norms: one partial per workgroup and source, added in workgroup order by ss_fold_kernel
  if (s.sumsq) {
    const double tot = block256_sum_d((double)local, red);
    if (threadIdx.x == 0) s.ssp[2 * local_block] = tot;
  }
  if (s.sumsq_b) {
    __syncthreads();
    const double tot = block256_sum_d((double)local_b, red);
    if (threadIdx.x == 0) s.ssp[2 * local_block + 1] = tot;
  }
  if (!__syncthreads_or(deferred ? 1 : 0)) return;
  // ---- tails of runs that span CP chunks or more (the popular ids of a Zipf list: up to ~800 chunks): the whole workgroup
  //      adds the partials (slot j of 256 / CP takes the chunks j, j + S, ... of the run, SS_BU rows in flight; the slots are
  //      added in order)
  const int S = gpb;
  for (int t = 0; t < gpb; ++t) {
    const int fl = sh[2 * t];
    if (!fl) continue;                                      // (uniform)
    const long ck = (long)local_block * gpb + t;
    const int key = sh[2 * t + 1];
    // extent: chunks ck - L .. ck - 1 hold entries of the run -- their LAST entries carry the key (256 borders per trip)
    long L = 0;
    for (bool open = true; open;) {
      const long cb = ck - 1 - L - threadIdx.x;             // this thread's chunk of the trip
      const bool member = cb >= 0 && s.keys[(cb + 1) * SS_CHUNK - 1] == key;
      // (first non-member of the trip, in thread order: ballots per wave, the waves in order through LDS)
      const unsigned long long bm = __ballot(member);
      const int nw = bm == ~0ull ? 64 : __builtin_ctzll(~bm);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sh[2 * gpb + (threadIdx.x >> 6)] = nw;
      __syncthreads();
      int tot = 0;
      bool all = true;
      for (int w = 0; w < 4 && all; ++w) { const int x = sh[2 * gpb + w]; tot += x; all = x == 64; }
      L += tot;
      open = all;
    }
    // members i = 0 .. L - 1: chunk ck - L + i; the head (i = 0) left the partial of its LAST run (slot 1), the others the
    // partial of their only run (slot 0); the tail's own share is still in the registers of its thread group
    vec_t aN[SS_BU];
#pragma unroll
    for (int u = 0; u < SS_BU; ++u) aN[u] = vec_t(0.f);
    const int cs = cok ? c : 0;
    const long long t0 = wall_clock64();
    for (long i0 = gi; i0 < L; i0 += (long)SS_BU * S) {
      vec_t pv[SS_BU];
#pragma unroll
      for (int u = 0; u < SS_BU; ++u) {
        const long i = i0 + (long)u * S;
        const long cm = ck - L + (i < L ? i : L - 1);
        while (ss_cldi(s.meta + cm * 4 + 3) != 1) {
          if (wall_clock64() - t0 > 200000000LL) { __hip_atomic_store(&hdr->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          __builtin_amdgcn_s_sleep(1);
        }
        pv[u] = ss_cld<VW>(s.bnd + cm * 2 * Cp + (i == 0 ? Cp : 0) + cs);
      }
#pragma unroll
      for (int u = 0; u < SS_BU; ++u)
        if (cok && i0 + (long)u * S < L) aN[u] += pv[u];
    }
#pragma unroll
    for (int w = 1; w < SS_BU; w *= 2)
#pragma unroll
      for (int u = 0; u + w < SS_BU; u += 2 * w) aN[u] += aN[u + w];
    __syncthreads();                       // (every wait above has ended: the ready words can be cleared)
    for (long i = threadIdx.x; i < L; i += 256) ss_csti(s.meta + (ck - L + i) * 4 + 3, 0);
    if constexpr (VW == 4) {
      shf[(gi * CP + lig) * 4 + 0] = aN[0].x; shf[(gi * CP + lig) * 4 + 1] = aN[0].y;
      shf[(gi * CP + lig) * 4 + 2] = aN[0].z; shf[(gi * CP + lig) * 4 + 3] = aN[0].w;
    } else {
      shf[gi * CP + lig] = aN[0];
    }
    __syncthreads();
    if (gi == t && cok) {                  // the tail's own thread group: the slots in order, then its own share
      vec_t tot = vec_t(0.f);
      for (int j = 0; j < S; ++j) {
        if constexpr (VW == 4) tot += (f32x4){shf[(j * CP + lig) * 4], shf[(j * CP + lig) * 4 + 1], shf[(j * CP + lig) * 4 + 2], shf[(j * CP + lig) * 4 + 3]};
        else tot += shf[j * CP + lig];
      }
      tot += pfirst;
      vec_t* g = reinterpret_cast<vec_t*>(s.grad + (long)key * s.ldg + s.gcol0 + c);
      *g = s.assign ? tot : *g + tot;
    }
  }
}

// squared norms of every site: the workgroup partials of the walk in workgroup order (lane-strided sums, then the lanes in
// order -- a fixed order).  Block (site, source); launched only when a site wants its norms.
__global__ void __launch_bounds__(64) ss_fold_kernel(SsArgs a) {
  const SsSite& s = a.s[blockIdx.x >> 1];
  const int which = blockIdx.x & 1, lane = threadIdx.x;
  double* dst = which == 0 ? s.sumsq : s.sumsq_b;
  if (!dst) return;
  double t = 0.0;
  // (eight partials of a lane in flight; the same order of additions as one after the other)
  for (int b0 = lane; b0 < s.nblocks; b0 += 64 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 64 * u;
      v[u] = b < s.nblocks ? s.ssp[2 * b + which] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  double tot = 0.0;
  for (int l = 0; l < 64; ++l) tot += __shfl(t, l, 64);
  if (lane == 0) *dst += tot;
}

__global__ void __launch_bounds__(256) ss_chunks_kernel(SsArgs a) {
  __shared__ double red[4];
  __shared__ int sh[2 * 32 + 4];
  __shared__ float shf[256 * 4];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.s[i + 1].first_block) ++i;
  const SsSite& s = a.s[i];
  if (s.vw == 4) ss_chunks<4, false>(s, blockIdx.x - s.first_block, red, sh, shf, a.hdr);
  else ss_chunks<1, false>(s, blockIdx.x - s.first_block, red, sh, shf, a.hdr);
}
// every site of the launch is LEAN (see ss_chunks): half the registers, twice the resident waves
__global__ void __launch_bounds__(256, 4) ss_chunks_lean_kernel(SsArgs a) {
  __shared__ double red[4];
  __shared__ int sh[2 * 32 + 4];
  __shared__ float shf[256 * 4];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.s[i + 1].first_block) ++i;
  const SsSite& s = a.s[i];
  if (s.vw == 4 && ss_pair16(s)) ss_chunks<4, true, SS_LEAN_B, true>(s, blockIdx.x - s.first_block, red, sh, shf, a.hdr);
  else if (s.vw == 4) ss_chunks<4, true, SS_LEAN_B>(s, blockIdx.x - s.first_block, red, sh, shf, a.hdr);
  else ss_chunks<1, true>(s, blockIdx.x - s.first_block, red, sh, shf, a.hdr);
}

static void ss_shape(const clsr_segsum_desc& d, int* cp, int* vw) {
  const bool vec = d.C % 4 == 0 && d.D % 4 == 0 && d.col0 % 4 == 0 && d.gcol0 % 4 == 0 && d.ldg % 4 == 0 &&
                   ((uintptr_t)d.src % 16) == 0 && (!d.src2 || ((uintptr_t)d.src2 % 16) == 0) &&
                   (!d.dmean || ((uintptr_t)d.dmean % 16) == 0) && (!d.drecent || ((uintptr_t)d.drecent % 16) == 0) &&
                   ((uintptr_t)d.grad % 16) == 0;
  *vw = vec ? 4 : 1;
  const int q = (d.C + *vw - 1) / *vw;
  *cp = q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
}
static long ss_site_bytes(long n, int cp, int vw, int* blocks) {
  const long nchunks = (n + SS_CHUNK - 1) / SS_CHUNK;
  const int nb = clsr_cdiv(nchunks, 256 / cp);
  if (blocks) *blocks = nb;
  long b = nchunks * 2 * cp * vw * (long)sizeof(float);
  b = (b + 15) & ~15L;
  b += nchunks * 4 * (long)sizeof(int);
  b = (b + 15) & ~15L;
  b += (long)nb * 2 * sizeof(double);
  return (b + 15) & ~15L;
}

extern "C" int clsr_sizeof_segsum_desc(void) { return (int)sizeof(clsr_segsum_desc); }
extern "C" long clsr_segsum_workspace_bytes(const clsr_segsum_desc* descs, int n) {
  long tot = 256;      // (header: error word; alignment slack)
  for (int i = 0; i < n; ++i) {
    int cp, vw;
    ss_shape(descs[i], &cp, &vw);
    tot += ss_site_bytes(descs[i].n, cp, vw, nullptr);
  }
  return tot;
}
// != 0: a tail of some launch on this workspace gave up waiting for an earlier chunk's partial (never expected: see above);
// synchronous
extern "C" int clsr_segsum_error(const void* workspace) {
  if (!workspace) return -1;
  const unsigned char* w = (const unsigned char*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  SsHdr h;
  if (hipMemcpy(&h, w, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)h.err;
}

// Sums of the slices of n lookup sites into their gradient tables (ONE launch).  The sites of one call must write
// different tables (or disjoint columns), and nothing else may write those tables while the call runs.  ``workspace``: zero
// when first used; not shared by launches in flight.
extern "C" int clsr_segsum_multi(const clsr_segsum_desc* descs, int n, void* workspace, long workspace_bytes, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_SEGSUM_MAX && workspace);
  SsArgs a;
  a.n = n;
  unsigned char* w = (unsigned char*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  a.hdr = (SsHdr*)w;
  long used = 64;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const clsr_segsum_desc& d = descs[i];
    CLSR_CHECK_ARG(d.src && d.keys && d.perm && d.grad && d.n > 0 && d.D > 0 && d.C > 0);
    CLSR_CHECK_ARG(!(d.dmean || d.drecent) || (d.seq_len && d.T > 0));
    SsSite& s = a.s[i];
    ss_shape(d, &s.cp, &s.vw);
    CLSR_CHECK_SUPPORTED(d.C <= 64 * s.vw);
    CLSR_CHECK_SUPPORTED(!(d.src_bf16 && s.vw == 4 && ((uintptr_t)d.src % 8)));
    s.src = d.src; s.src2 = d.src2; s.src_bf16 = d.src_bf16; s.dmean = d.dmean; s.drecent = d.drecent;
    s.keys = d.keys; s.perm = d.perm; s.seq_len = d.seq_len; s.len_stride = d.len_stride;
    s.n = d.n; s.T = d.T > 0 ? d.T : 1; s.D = d.D; s.col0 = d.col0; s.C = d.C; s.recent_k = d.recent_k;
    s.grad = d.grad; s.ldg = d.ldg; s.gcol0 = d.gcol0; s.sumsq = d.sumsq;
    s.assign = d.assign; s.src_b = d.src_b; s.sumsq_b = d.sumsq_b; s.n1 = d.src_b ? d.n1 : 0; s.ldb = d.ldb; s.colb = d.colb;
    CLSR_CHECK_ARG(!d.src_b || (d.n1 > 0 && d.ldb > 0));
    CLSR_CHECK_SUPPORTED(!d.src_b || s.vw == 1 || (d.ldb % 4 == 0 && d.colb % 4 == 0 && ((uintptr_t)d.src_b % 16) == 0));
    int nb;
    const long bytes = ss_site_bytes(d.n, s.cp, s.vw, &nb);
    const long nchunks = (d.n + SS_CHUNK - 1) / SS_CHUNK;
    s.bnd = (float*)(w + used);
    long o = (nchunks * 2 * s.cp * s.vw * (long)sizeof(float) + 15) & ~15L;
    s.meta = (int*)(w + used + o);
    o += (nchunks * 4 * (long)sizeof(int) + 15) & ~15L;
    s.ssp = (double*)(w + used + o);
    used += bytes;
    s.first_block = total;
    s.nblocks = nb;
    total += nb;
  }
  CLSR_CHECK_ARG(workspace_bytes >= used + 16);
  hipStream_t st = (hipStream_t)stream;
  bool lean = true;
  for (int i = 0; i < n; ++i) lean = lean && !descs[i].src2 && !descs[i].dmean && !descs[i].drecent && descs[i].assign;
  if (lean) hipLaunchKernelGGL(ss_chunks_lean_kernel, dim3(total), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ss_chunks_kernel, dim3(total), dim3(256), 0, st, a);
  CLSR_CHECK_LAUNCH();
  bool norms = false;
  for (int i = 0; i < n; ++i) norms = norms || descs[i].sumsq || descs[i].sumsq_b;
  if (norms) {
    hipLaunchKernelGGL(ss_fold_kernel, dim3(2 * n), dim3(64), 0, st, a);
    CLSR_CHECK_LAUNCH();
  }
  return CLSR_OK;
}
