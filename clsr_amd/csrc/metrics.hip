// Evaluation metrics on the device (gfx950): the scores of a whole validation / test file stay in HBM and only a
// handful of doubles come back.
//
// Reference: reco_utils/recommender/deeprec/deeprec_utils.py:554-821 (mrr_score, ndcg_score, hit_score, cal_metric,
// cal_weighted_metric) as called by SequentialBaseModel.run_eval / run_weighted_eval
// (models/sequential/sequential_base_model.py:204-292): auc + logloss over all lines, mean_mrr / ndcg@k / hit@k /
// group_auc over groups of 1 + num_ngs consecutive lines, wauc = per-user roc_auc weighted by the user's line count.
//
// No sort anywhere: ROC-AUC with averaged tie ranks equals (#{pos > neg} + 0.5 #{pos == neg}) / (n_pos n_neg) -- integer
// pair counts, exact -- and a positive's rank inside its group is a count as well.  Pair counting is P x N work for
// the global AUC (1e10 compare-and-count operations for a 1M-line test file: milliseconds on 256 CUs) and
// P_u x n_u inside a user / a group.  Rank ties (equal scores inside a group) break like a STABLE ascending sort read
// backwards: among equal scores the LATER line ranks first.  (numpy's default argsort, which the reference uses, is
// not stable above 16 elements, so the reference itself does not define that order.)
#include "common.h"
#include "clsr_hip.h"

typedef unsigned long long u64;

__device__ __forceinline__ double clip_d(double p, double lo, double hi) { return p < lo ? lo : (p > hi ? hi : p); }

// Order-independent accumulation of the per-block partial sums: double atomics add in whatever order the blocks
// finish, and a metric that sits exactly on a 4-decimal rounding boundary (0.58125: small test files produce such
// ratios) then rounds either way from run to run.  The partials are added as 64-bit FIXED-POINT integers instead
// (integer addition commutes exactly); a one-block kernel converts the slots back to doubles.  The slots must be
// zero on entry (the bit pattern of 0.0 is the integer 0) and hold the final sums on return.
__device__ __forceinline__ void fx_add(double* slot, double v, double scale) {
  atomicAdd(reinterpret_cast<u64*>(slot), (u64)__double2ll_rn(v * scale));
}
__global__ void fx_finalize_kernel(double* out, int n, double inv_scale) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (double)(long long)reinterpret_cast<const u64*>(out)[i] * inv_scale;
}
static void fx_finalize(double* out, int n, double scale, hipStream_t s) {
  hipLaunchKernelGGL(fx_finalize_kernel, dim3(1), dim3(64), 0, s, out, n, 1.0 / scale);
}
#define FX_LOGLOSS 67108864.0              // 2^26: the sum is at most 27 N (N < 2^31 lines)
#define FX_GROUP 1073741824.0              // 2^30: sums of per-group values in [0, 1]
#define FX_WAUC 1152921504606846976.0      // 2^60: the weighted sum is at most 1

// out[0] = sum over lines of -(y log p + (1 - y) log(1 - p)), p clipped to [1e-11, 1 - 1e-11] (cal_metric "logloss")
__global__ void __launch_bounds__(256) eval_logloss_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ labels, long N,
                                                           double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < N; e += (long)gridDim.x * 256) {
    const double p = clip_d((double)pred[e], 10e-12, 1.0 - 10e-12);
    const double y = (double)labels[e];
    s -= y * log(p) + (1.0 - y) * log(1.0 - p);
  }
  s = block256_sum_d(s, red);
  if (threadIdx.x == 0) fx_add(out, s, FX_LOGLOSS);
}

extern "C" int clsr_eval_logloss(const float* pred, const float* labels, long N, double* out, void* stream) {
  CLSR_CHECK_ARG(pred && labels && out && N > 0);
  int blocks = clsr_cdiv(N, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(eval_logloss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, labels, N, out);
  fx_finalize(out, 1, FX_LOGLOSS, (hipStream_t)stream);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// scores of the positive lines, compacted (any order); count[0] must be zero on entry
__global__ void __launch_bounds__(256) eval_compact_pos_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ labels, long N,
                                                               float* __restrict__ pos, int* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  for (long e0 = (long)blockIdx.x * 256; e0 < N; e0 += (long)gridDim.x * 256) {
    const long e = e0 + threadIdx.x;
    const bool on = e < N && labels[e] == 1.0f;
    const u64 m = __ballot(on);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (on) pos[base + __popcll(m & ((1ull << lane) - 1ull))] = pred[e];
  }
}

extern "C" int clsr_eval_compact_pos(const float* pred, const float* labels, long N, float* pos_out, int* count,
                                     void* stream) {
  CLSR_CHECK_ARG(pred && labels && pos_out && count && N > 0);
  CLSR_CHECK_SUPPORTED(N < (1L << 31));
  int blocks = clsr_cdiv(N, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(eval_compact_pos_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, labels, N, pos_out,
                     count);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[0] += #{(p, n): pos_p > neg_n}, out[1] += #{pos_p == neg_n}, out[2] += number of negative lines
#define AUC_NPT 4       // negatives per thread
#define AUC_PT 2048     // positives per LDS tile
__global__ void __launch_bounds__(256) eval_auc_pairs_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ labels, long N,
                                                             const float* __restrict__ pos,
                                                             const int* __restrict__ count, u64* __restrict__ out) {
  __shared__ float ps[AUC_PT];
  __shared__ u64 red[3][4];
  const int P = count[0];
  u64 gt = 0, eq = 0, nneg = 0;
  for (long e0 = (long)blockIdx.x * (256 * AUC_NPT); e0 < N; e0 += (long)gridDim.x * (256 * AUC_NPT)) {
    float s[AUC_NPT];
    bool neg[AUC_NPT];
#pragma unroll
    for (int u = 0; u < AUC_NPT; ++u) {
      const long e = e0 + u * 256 + threadIdx.x;
      neg[u] = e < N && labels[e] != 1.0f;
      s[u] = neg[u] ? pred[e] : INFINITY;     // +inf: no finite positive is greater or equal
      nneg += neg[u] ? 1 : 0;
    }
    for (int p0 = 0; p0 < P; p0 += AUC_PT) {
      const int pc = min(AUC_PT, P - p0);
      __syncthreads();
      for (int i = threadIdx.x; i < pc; i += 256) ps[i] = pos[p0 + i];
      __syncthreads();
      unsigned g[AUC_NPT] = {0, 0, 0, 0}, q[AUC_NPT] = {0, 0, 0, 0};
      for (int i = 0; i < pc; ++i) {
        const float v = ps[i];
#pragma unroll
        for (int u = 0; u < AUC_NPT; ++u) { g[u] += v > s[u] ? 1u : 0u; q[u] += v == s[u] ? 1u : 0u; }
      }
#pragma unroll
      for (int u = 0; u < AUC_NPT; ++u) { gt += g[u]; eq += q[u]; }
    }
  }
  // block reduction of three 64-bit counters
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 v[3] = {gt, eq, nneg};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) red[k][wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const u64 t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t) atomicAdd(out + threadIdx.x, t);
  }
}

extern "C" int clsr_eval_auc_pairs(const float* pred, const float* labels, long N, const float* pos,
                                   const int* count, void* out_u64x3, void* stream) {
  CLSR_CHECK_ARG(pred && labels && pos && count && out_u64x3 && N > 0);
  int blocks = clsr_cdiv(N, 256 * AUC_NPT);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eval_auc_pairs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, labels, N, pos,
                     count, (u64*)out_u64x3);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- group metrics: one wave per group of G consecutive lines
// out[0] += mrr, out[1] += group_auc, out[2 + i] += ndcg@ks[i], out[2 + nk + i] += hit@ks[i]; err[0] += groups without
// a positive or without a negative (the reference raises / divides by zero there)
#define GM_MAXK 8
struct GroupMetricArgs {
  const float* pred; const float* labels; long n_groups; int G; int nk; int ks[GM_MAXK]; int want_auc;
  double* out; int* err;
};

__global__ void __launch_bounds__(256) eval_group_metrics_kernel(GroupMetricArgs a) {
  extern __shared__ float gs[];   // [4 waves][G] scores, then [4][G] labels
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = a.G;
  float* sc = gs + (long)wave * G;
  float* lb = gs + (long)4 * G + (long)wave * G;
  double acc[2 + 2 * GM_MAXK];
#pragma unroll
  for (int i = 0; i < 2 + 2 * GM_MAXK; ++i) acc[i] = 0.0;
  int bad = 0;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < a.n_groups; grp += (long)gridDim.x * 4) {
    const float* p = a.pred + grp * G;
    const float* l = a.labels + grp * G;
    for (int i = lane; i < G; i += 64) { sc[i] = p[i]; lb[i] = l[i]; }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // every lane takes the lines i = lane, lane + 64, ...; positives compute their rank by a count over the group
    double mrr = 0.0, dcg[GM_MAXK], hit[GM_MAXK];
#pragma unroll
    for (int k = 0; k < GM_MAXK; ++k) { dcg[k] = 0.0; hit[k] = 0.0; }
    int npos = 0;
    u64 gt = 0, eq = 0;
    for (int i = lane; i < G; i += 64) {
      if (lb[i] != 1.0f) continue;
      ++npos;
      const float s = sc[i];
      int above = 0;
      for (int j = 0; j < G; ++j) {
        const float t = sc[j];
        above += (t > s || (t == s && j > i)) ? 1 : 0;
        if (a.want_auc && lb[j] != 1.0f) { gt += s > t ? 1 : 0; eq += s == t ? 1 : 0; }
      }
      const int rank = above + 1;
      mrr += 1.0 / (double)rank;
#pragma unroll
      for (int k = 0; k < GM_MAXK; ++k)
        if (k < a.nk && rank <= a.ks[k]) { dcg[k] += 1.0 / log2((double)rank + 1.0); hit[k] = 1.0; }
    }
    // wave reductions
    mrr = wave_sum_d(mrr);
    int np_all = npos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np_all += __shfl_xor(np_all, o, 64);
    double gtd = (double)gt, eqd = (double)eq;
    gtd = wave_sum_d(gtd);
    eqd = wave_sum_d(eqd);
    const int nneg = G - np_all;
    if (np_all == 0 || (a.want_auc && nneg == 0)) ++bad;
    if (np_all > 0) {
      acc[0] += mrr / (double)np_all;
      if (a.want_auc && nneg > 0) acc[1] += (gtd + 0.5 * eqd) / ((double)np_all * (double)nneg);
#pragma unroll
      for (int k = 0; k < GM_MAXK; ++k) {
        if (k < a.nk) {
          const double d = wave_sum_d(dcg[k]);
          double h = hit[k];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) h = fmax(h, __shfl_xor(h, o, 64));
          double ideal = 0.0;
          const int kk = min(min(a.ks[k], G), np_all);
          for (int r = 1; r <= kk; ++r) ideal += 1.0 / log2((double)r + 1.0);
          acc[2 + k] += d / ideal;
          acc[2 + a.nk + k] += h;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    for (int i = 0; i < 2 + 2 * a.nk; ++i)
      if (acc[i] != 0.0) fx_add(a.out + i, acc[i], FX_GROUP);
    if (bad) atomicAdd(a.err, bad);
  }
}

extern "C" int clsr_eval_group_metrics(const float* pred, const float* labels, long n_groups, int G, const int* ks_host,
                                       int nk, int want_auc, double* out, int* err, void* stream) {
  CLSR_CHECK_ARG(pred && labels && out && err && n_groups > 0 && G > 0 && nk >= 0 && (nk == 0 || ks_host));
  CLSR_CHECK_SUPPORTED(nk <= GM_MAXK && G <= 4096);
  GroupMetricArgs a;
  a.pred = pred; a.labels = labels; a.n_groups = n_groups; a.G = G; a.nk = nk; a.want_auc = want_auc;
  a.out = out; a.err = err;
  for (int i = 0; i < GM_MAXK; ++i) a.ks[i] = i < nk ? ks_host[i] : 0;
  int blocks = clsr_cdiv(n_groups, 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eval_group_metrics_kernel, dim3(blocks), dim3(256), (size_t)8 * G * sizeof(float),
                     (hipStream_t)stream, a);
  fx_finalize(out, 2 + 2 * nk, FX_GROUP, (hipStream_t)stream);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- wauc: lines grouped by user (clsr_sort_ids_multi on the user ids: perm + the end offset of every user's
// segment), one wave per user: out[0] += (n_u / N) * roc_auc(user u); err[0] += users whose lines are all one class
#define UA_PT 256
__global__ void __launch_bounds__(256) eval_user_auc_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ labels,
                                                            const int* __restrict__ perm, const int* __restrict__ ends,
                                                            int nb, long N, double* __restrict__ out,
                                                            int* __restrict__ err) {
  __shared__ float ps[4][UA_PT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  int bad = 0;
  for (long b = (long)blockIdx.x * 4 + wave; b < nb; b += (long)gridDim.x * 4) {
    const int lo = b ? ends[b - 1] : 0, hi = ends[b];
    const int n = hi - lo;
    if (n <= 0) continue;
    // positives of the segment, UA_PT at a time, against every negative of the segment
    u64 gt = 0, eq = 0;
    int npos = 0, done = 0;     // done: positives consumed so far (in segment order)
    while (true) {
      // gather the next UA_PT positives: a wave-wide ordered scan over the segment
      int got = 0, seen = 0;
      for (int i0 = 0; i0 < n && got < UA_PT; i0 += 64) {
        const int i = i0 + lane;
        const bool on = i < n && labels[perm[lo + i]] == 1.0f;
        const u64 m = __ballot(on);
        const int before = seen + __popcll(m & ((1ull << lane) - 1ull));   // index of this positive in the segment
        if (on && before >= done && before - done < UA_PT) ps[wave][before - done] = pred[perm[lo + i]];
        seen += __popcll(m);
        got = min(UA_PT, max(0, seen - done));
      }
      if (done == 0) {        // total number of positives: finish the scan once
        int tot = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
          const int i = i0 + lane;
          tot += __popcll(__ballot(i < n && labels[perm[lo + i]] == 1.0f));
        }
        npos = tot;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      const int pc = min(UA_PT, npos - done);
      if (pc <= 0) break;
      for (int i = lane; i < n; i += 64) {
        const int row = perm[lo + i];
        if (labels[row] == 1.0f) continue;
        const float s = pred[row];
        unsigned g = 0, q = 0;
        for (int k = 0; k < pc; ++k) { const float v = ps[wave][k]; g += v > s ? 1u : 0u; q += v == s ? 1u : 0u; }
        gt += g; eq += q;
      }
      done += pc;
      __builtin_amdgcn_wave_barrier();
      if (done >= npos) break;
    }
    double gtd = wave_sum_d((double)gt), eqd = wave_sum_d((double)eq);
    const int nneg = n - npos;
    if (npos == 0 || nneg == 0) { ++bad; continue; }
    acc += ((double)n / (double)N) * (gtd + 0.5 * eqd) / ((double)npos * (double)nneg);
  }
  if (lane == 0) {
    if (acc != 0.0) fx_add(out, acc, FX_WAUC);
    if (bad) atomicAdd(err, bad);
  }
}

extern "C" int clsr_eval_user_auc(const float* pred, const float* labels, const int* perm, const int* ends, int nb,
                                  long N, double* out, int* err, void* stream) {
  CLSR_CHECK_ARG(pred && labels && perm && ends && out && err && nb > 0 && N > 0);
  int blocks = clsr_cdiv(nb, 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eval_user_auc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, labels, perm, ends,
                     nb, N, out, err);
  fx_finalize(out, 1, FX_WAUC, (hipStream_t)stream);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
