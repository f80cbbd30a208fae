// The row-level heads of the CLSR training step as TWO persistent launches (gfx950).
//
// Reference: models/sequential/clsr.py:239-275 (concat_all -> fcn_alpha -> sigmoid -> user_embed fusion ->
// model_output), models/base_model.py:653-708 (_fcn_net: two dense layers, each followed by
// tf.layers.batch_normalization + relu, then w_nn_output / b_nn_output), base_model.py:215-235 (softmax data loss),
// and the gradients of all of it.
//
// Why: at the turn of the step (last forward kernel -> first backward kernel) the chip is idle behind a chain of 22
// dependent launches over B = 5 120 rows (alpha-gate input, 4 dense layers, 4 batch-norm finalisations, output layers,
// fusion, loss, and the same again backwards): ~245 us of the 3.5 ms step at ~10 us per link, each link a few
// microseconds of work (profiles/r04_fp32_timeline.txt).  The only reason they are separate launches is that
// batch normalisation needs the statistics of ALL rows between two layers.  Here a workgroup keeps its rows in LDS
// through every layer and the workgroups meet at a GRID BARRIER where a launch boundary used to be:
//   launch 1  alpha-gate input -> alpha MLP -> alpha, model_output -> logit MLP -> logits, softmax loss ->
//             logit-MLP backward -> d(model_output)                                              (6 barriers)
//   (the contrastive-loss kernel of the side stream joins here: it accumulates into the same dL / dS)
//   launch 2  fusion backward -> alpha-MLP backward -> d(alpha-gate input) scattered to dfs / dtarget / dL / dS
//                                                                                                 (2 barriers)
// <= 256 workgroups of rows_per_block = G * ceil(groups / 256) <= 80 rows (whole softmax groups; five 16-position MFMA
// tiles -- B = 20 480 rows of BASELINE configs[1] are exactly 256 x 80 -- the padding rows are zero), all co-resident
// (one per CU: launch_bounds(256, 1) + 144 / 121 KB of LDS), so a spin
// barrier on a device-scope counter cannot deadlock: every workgroup that is waited for is running or will be
// scheduled as soon as kernels of OTHER streams (which never wait for this one) drain.
//
// Products: v_mfma_f32_16x16x4_f32 in the "features x positions" orientation of csrc/linear.hip (A = packed W^T rows,
// B = the rows' activations from LDS, D = 4 consecutive output features of one position per lane).  A wave owns the
// out-feature tiles wave, wave + 4, ...; its slice of the NEXT layer's packed weights is fetched into registers BEFORE
// it waits at the barrier, so the weight fetch hides under the wait.
// Batch-norm sums: every workgroup writes its fp64 column sums, and after the barrier every workgroup folds all of them
// in the same fixed order (no atomics: identical statistics in every workgroup and in every run).  The weight gradients
// of the four dense layers stay in the step's batched dW launch (they read the activations / dz this kernel writes);
// the output-layer gradients leave as per-workgroup partial rows like clsr_mlp_out_bwd's.
//
// Widths are compile-time (the reference's clsr.yaml: D = 40, final-state width 40, att_fcn_layer_sizes [80, 40],
// layer_sizes [100, 64]); clsr_heads_fused_supported() says no for anything else and the host keeps the multi-launch
// path (also for synchronised batch-norm statistics across ranks, which need a collective between the layers).
#include <stdlib.h>
#include <stdio.h>
#include "common.h"
#include "clsr_hip.h"

#define HF_D 40
#define HF_NFS 40
#define HF_AIN 161
#define HF_LD 164
#define HF_A0 80
#define HF_A1 40
#define HF_L0 100
#define HF_L1 64
#define HF_MT 5                  // 16-position MFMA tiles per workgroup
#define HF_RP (16 * HF_MT)       // padded rows per workgroup
#define HF_MAXBLK 256
#define HF_MAXG 8
#define HF_NSTAGE 8
#define HF_GS 16                 // workgroups per group of the two-level sums
#define HF_NGRP (HF_MAXBLK / HF_GS)
#define HF_ROW 256               // doubles per partial row (2 x features <= 200)
// workspace: counters | per-workgroup partial rows [stage][workgroup][256] | per-group rows [stage][group][256] | debug
#define HF_CTR_BYTES 1024        // u32 gctr[8][16] at 0, u32 top[8] at 512 (cleared by the step); the STICKY error words
                                 // sit behind the partial sums, at the head of the debug area (hf_err_words)
#define HF_PART_BYTES ((long)HF_NSTAGE * HF_MAXBLK * HF_ROW * 8)
#define HF_GPART_BYTES ((long)HF_NSTAGE * HF_NGRP * HF_ROW * 8)
#define HF_DBG_BYTES (1024 + 65536)

// LDS row strides (floats): 16 * ceil(K / 16) + 4 like the packed weights (conflict-free 16-byte operand reads)
#define HS_A 180
#define HS_80 84
#define HS_40 52
#define HS_100 116
#define HS_64 68

// launch 1, 144 KB.  X is three things in turn: the alpha-gate input rows (first alpha layer only) -> the second alpha
// layer's z1 (until alpha is known) -> model_output rows + the first logit layer's z0; the dy / dz rows of the second logit
// layer take model_output's place once the first logit layer has consumed it.
struct HfLds {
  union {
    float A[HF_RP * HS_A];
    struct { float MO[HF_RP * HS_80]; float Z0L[HF_RP * HS_100]; } m;
    struct { float DY1[HF_RP * HS_64]; float pad_[HF_RP * (HS_80 - HS_64)]; float Z1A[HF_RP * HS_40]; } e;
  } X;
  float ACT[HF_RP * HS_100];      // relu(bn(z)) of the layer being consumed; later dy / dz of the first logit layer
  float Z0A[HF_RP * HS_80];       // first alpha layer's z0; the second logit layer's z1 (stride HS_64) once alpha is known
  float bn[4][4][112];            // layer x (scale, shift, mean, invstd)
  float cf[3][112];
  float wout[64];
  float alpha[HF_RP], logit[HF_RP], dl[HF_RP], lossp[HF_RP];
  double red[512];
  double tot[256];
  int flag[4];
};
// launch 2, 147 KB.  W holds the second alpha layer's z1 and dy / dz rows (and the row sums' scratch), then d(alpha-gate input)
struct HfLds2 {
  union {
    float DAIN[HF_RP * HS_A];
    struct { float Z1A[HF_RP * HS_40]; float DY1[HF_RP * HS_40]; float tmp[HF_RP * HF_D]; } e;
  } W;
  float Z0A[HF_RP * HS_80];
  float DY0[HF_RP * HS_80];
  float DMO[HF_RP * 2 * HF_D];
  float bn[2][4][112];
  float cf[3][112];
  float wout[64];
  float alpha[HF_RP], dal[HF_RP];
  double red[512];
  double tot[256];
  int flag[4];
};
static_assert(sizeof(HfLds) <= 160 * 1024 && sizeof(HfLds2) <= 160 * 1024, "LDS budget");

// ---------------------------------------------------------------------------------------------- grid-wide sums
// The batch-norm sums of a layer over ALL workgroups, and the grid barrier that comes with them, WITHOUT cache maintenance:
// agent-scope fences (buffer_wbl2 / buffer_inv: write back and invalidate the L2 of the XCD) cost 12-17 us per barrier here,
// with megabytes of activations of this and the neighbouring streams' kernels dirty in the L2s (profiles/r04_heads_fused.md).
// Only the few KB of partial sums cross workgroups inside a launch, so only THEY use device-coherent accesses (agent-scope
// atomic stores / loads: write-through, L2-bypassing), ordered by s_waitcnt + relaxed device-scope counters:
//   arrive  every workgroup stores its row of sums, waits for the stores (vmcnt(0)), then counts itself into its GROUP of 16;
//           the last arrival of a group adds the group's 16 rows in workgroup order, stores the group row, counts the group;
//   wait    until all groups are counted; every workgroup adds the <= 16 group rows in group order.
// Two levels: 16 + 16 serialised counter updates instead of 256 on one address, 32 row reads per workgroup instead of 256;
// the order of every addition is fixed by (workgroup index, group index): the same sums in every workgroup and every run.
// Data-parallel runs with synchronised batch-norm statistics: the group rows of ALL ranks are summed.  Every
// rank owns an exchange buffer (uncached device memory, mapped into its peers by hipIpc handles like csrc/p2p.hip's); the
// last arrival of a group PUSHES the group row into the slot (stage, own rank, group) of every rank's buffer and counts
// the group in every rank's counter of the stage; a workgroup waits until its own counter has reached
// the number of group rows pushed to it so far (the counters are never cleared: the running total -- groups x world per step,
// batches of different sizes have different group counts -- lives in the communicator)
// and adds world x groups rows in (rank, group) order -- the same sums, bit for bit, on every rank.
#define HF_MAXW 8
struct HfXchg {
  unsigned long long top[HF_NSTAGE];
  unsigned long long pad_[8];
  double gpart[HF_NSTAGE][HF_MAXW][HF_NGRP][HF_ROW];
};
struct HfComm { int rank, world; unsigned long long pushed; HfXchg* x[HF_MAXW]; };
struct HfPeers { HfXchg* x[HF_MAXW]; int rank, world; unsigned long long target; long long timeout_ticks; double* abort_flag; };

struct HfSync {
  unsigned* gctr; unsigned* top; unsigned* err; double* part; double* gpart; int nb; int* flag;
  const HfPeers* peers;
#ifdef HF_TIMING
  long long* dbg; int dbg0;
#endif
};
__device__ __forceinline__ HfSync hf_sync_init(void* workspace, int nb, int* flag, int dbg0, const HfPeers* peers) {
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  HfSync S;
  S.gctr = reinterpret_cast<unsigned*>(w); S.top = S.gctr + 128;
  S.err = reinterpret_cast<unsigned*>(w + HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES);   // not in the range the step clears
  S.part = reinterpret_cast<double*>(w + HF_CTR_BYTES);
  S.gpart = reinterpret_cast<double*>(w + HF_CTR_BYTES + HF_PART_BYTES);
  S.nb = nb; S.flag = flag; S.peers = peers;
#ifdef HF_TIMING
  S.dbg = reinterpret_cast<long long*>(w + HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES + 1024); S.dbg0 = dbg0;
#endif
  (void)dbg0;
  return S;
}
__device__ __forceinline__ void hf_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double hf_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define HF_WAIT_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// thread `has` contributes v0 to element i0 and v1 to element i1 of this workgroup's row of n2 sums
__device__ __forceinline__ void hf_arrive(const HfSync& S, int stage, int n2, bool has, int i0, double v0, int i1, double v1) {
  double* row = S.part + ((long)stage * HF_MAXBLK + blockIdx.x) * HF_ROW;
  if (has) { hf_st(row + i0, v0); hf_st(row + i1, v1); }
  HF_WAIT_STORES();
  __syncthreads();
  const int grp = blockIdx.x / HF_GS, ngrp = (S.nb + HF_GS - 1) / HF_GS;
  const int gsize = grp == ngrp - 1 ? S.nb - grp * HF_GS : HF_GS;
  if (threadIdx.x == 0) {
#ifdef HF_TIMING
    S.dbg[((S.dbg0 + stage) * 256 + blockIdx.x) * 2] = wall_clock64();
#endif
    const unsigned old = __hip_atomic_fetch_add(S.gctr + stage * HF_NGRP + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S.flag[0] = old == (unsigned)(gsize - 1);
  }
  __syncthreads();
  if (S.flag[0]) {          // (workgroup-uniform)
    if ((int)threadIdx.x < n2) {
      const double* p = S.part + ((long)stage * HF_MAXBLK + grp * HF_GS) * HF_ROW + threadIdx.x;
      double v[HF_GS];
#pragma unroll
      for (int k = 0; k < HF_GS; ++k) v[k] = k < gsize ? hf_ld(p + (long)k * HF_ROW) : 0.0;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < HF_GS; ++k) s += v[k];
      if (S.peers->world > 1) {
        for (int r = 0; r < S.peers->world; ++r)
          __hip_atomic_store(&S.peers->x[r]->gpart[stage][S.peers->rank][grp][threadIdx.x], s, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        hf_st(S.gpart + ((long)stage * HF_NGRP + grp) * HF_ROW + threadIdx.x, s);
      }
    }
    HF_WAIT_STORES();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (S.peers->world > 1) {
        for (int r = 0; r < S.peers->world; ++r)
          __hip_atomic_fetch_add(&S.peers->x[r]->top[stage], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        __hip_atomic_fetch_add(S.top + stage, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// tot[0:n2] = the sums over all workgroups (of all ranks)
__device__ __forceinline__ void hf_wait(const HfSync& S, int stage, int n2, double* tot) {
  const int ngrp = (S.nb + HF_GS - 1) / HF_GS;
  const int world = S.peers->world;
  if (threadIdx.x == 0) {
    // (bounded: a launch with more workgroups than the device can hold at once would otherwise hang it; two seconds of the
    // 100 MHz wall clock -- CLSR_P2P_TIMEOUT_S across ranks, which drift apart by seconds in their first steps -- then the
    // error word of the workspace is raised: clsr_heads_fused_error)
    const long long t0 = wall_clock64();
    if (world > 1) {
      const HfXchg* mine = S.peers->x[S.peers->rank];
      const unsigned long long target = S.peers->target;
      while (__hip_atomic_load(&mine->top[stage], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > S.peers->timeout_ticks) {      // (err: 1 + stage; err[1], err[2]: counter seen / wanted)
          __hip_atomic_store(S.err, 1u + (unsigned)stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          S.err[1] = (unsigned)__hip_atomic_load(&mine->top[stage], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          S.err[2] = (unsigned)target;
          if (S.peers->abort_flag) __hip_atomic_store(S.peers->abort_flag, 1.0 + stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    } else {
      while (__hip_atomic_load(S.top + stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ngrp) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000LL) {
          __hip_atomic_store(S.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (S.peers->abort_flag) __hip_atomic_store(S.peers->abort_flag, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
#ifdef HF_TIMING
    S.dbg[((S.dbg0 + stage) * 256 + blockIdx.x) * 2 + 1] = wall_clock64();
#endif
  }
  __syncthreads();
  if ((int)threadIdx.x < n2) {
    double s = 0.0;
    if (world > 1) {
      const HfXchg* mine = S.peers->x[S.peers->rank];
      for (int r = 0; r < world; ++r) {
        double v[HF_NGRP];
#pragma unroll
        for (int g = 0; g < HF_NGRP; ++g)
          v[g] = g < ngrp ? __hip_atomic_load(&mine->gpart[stage][r][g][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
#pragma unroll
        for (int g = 0; g < HF_NGRP; ++g) s += v[g];
      }
    } else {
      const double* p = S.gpart + (long)stage * HF_NGRP * HF_ROW + threadIdx.x;
      double v[HF_NGRP];
#pragma unroll
      for (int g = 0; g < HF_NGRP; ++g) v[g] = g < ngrp ? hf_ld(p + (long)g * HF_ROW) : 0.0;
#pragma unroll
      for (int g = 0; g < HF_NGRP; ++g) s += v[g];
    }
    tot[threadIdx.x] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- products
template <int KT, int NTW>
struct HfW { f32x4 w[NTW][KT]; };

// the wave's slice of a packed transposed weight (rows = out features, row stride Kp, zero padded)
template <int KT, int NTW>
__device__ __forceinline__ void hf_load_w(HfW<KT, NTW>& s, const float* __restrict__ Wt, int Kp, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tile = wave + 4 * t;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
      s.w[t][kt] = tile < ntiles ? ld4(Wt + (long)(16 * tile + i) * Kp + 16 * kt + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// epi(first feature of the lane's four, position, values)
template <int KT, int NTW, class Epi>
__device__ __forceinline__ void hf_gemm(const HfW<KT, NTW>& s, const float* X, int ldx, int ntiles, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  f32x4 acc[NTW][HF_MT];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int m = 0; m < HF_MT; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x4 x[HF_MT];
#pragma unroll
    for (int m = 0; m < HF_MT; ++m) x[m] = ld4(X + (16 * m + j) * ldx + 16 * kt + 4 * g);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (wave + 4 * t < ntiles) {
#pragma unroll
        for (int m = 0; m < HF_MT; ++m) MFMA4(acc[t][m], s.w[t][kt].x, x[m].x);
#pragma unroll
        for (int m = 0; m < HF_MT; ++m) MFMA4(acc[t][m], s.w[t][kt].y, x[m].y);
#pragma unroll
        for (int m = 0; m < HF_MT; ++m) MFMA4(acc[t][m], s.w[t][kt].z, x[m].z);
#pragma unroll
        for (int m = 0; m < HF_MT; ++m) MFMA4(acc[t][m], s.w[t][kt].w, x[m].w);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tile = wave + 4 * t;
    if (tile < ntiles) {
#pragma unroll
      for (int m = 0; m < HF_MT; ++m) epi(16 * tile + 4 * g, 16 * m + j, acc[t][m]);
    }
  }
}

// forward epilogue: z = acc + bias -> LDS rows (padding features stay zero) and the global activation tensor
struct HfEpiFwd {
  float* z; int ldz; const float* bias; int N; float* gout; long row0; int Rv;
  __device__ __forceinline__ void operator()(int f0, int pos, f32x4 v) const {
    if (f0 < N) v += f32x4{bias[f0], bias[f0 + 1], bias[f0 + 2], bias[f0 + 3]};   // (views of the flat parameter buffer: no 16-byte alignment)
    st4(z + pos * ldz + f0, v);
    if (f0 < N && pos < Rv) st4(gout + (row0 + pos) * N + f0, v);
  }
};
// backward epilogue: dy = (z * scale + shift > 0) ? acc : 0 into LDS
struct HfEpiMask {
  float* dy; int ld; const float* z; const float* sc; const float* sh;
  __device__ __forceinline__ void operator()(int f0, int pos, f32x4 v) const {
    const f32x4 y = ld4(z + pos * ld + f0) * ld4(sc + f0) + ld4(sh + f0);
    v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
    st4(dy + pos * ld + f0, v);
  }
};

// ---------------------------------------------------------------------------------------------- column sums
// thread (sub, c): the rows sub, sub + nsub, ... of column c; the sub-sums are added in sub order by thread c < N, which
// hands them to hf_arrive.  stats: (sum z, sum z^2); backward: (sum dy, sum dy * xhat)
__device__ __forceinline__ void hf_stats_arrive(const HfSync& S, int stage, const float* z, int ldz, int N, int Rv, double* red) {
  const int nsub = 256 / N, sub = threadIdx.x / N, c = threadIdx.x - sub * N;
  if (sub < nsub) {
    double s = 0.0, q = 0.0;
    for (int r = sub; r < Rv; r += nsub) { const double v = z[r * ldz + c]; s += v; q += v * v; }
    red[sub * 2 * N + c] = s; red[sub * 2 * N + N + c] = q;
  }
  __syncthreads();
  double s = 0.0, q = 0.0;
  const bool has = (int)threadIdx.x < N;
  if (has)
    for (int k = 0; k < nsub; ++k) { s += red[k * 2 * N + threadIdx.x]; q += red[k * 2 * N + N + threadIdx.x]; }
  hf_arrive(S, stage, 2 * N, has, threadIdx.x, s, N + threadIdx.x, q);
}
__device__ __forceinline__ void hf_bwd_sums_arrive(const HfSync& S, int stage, const float* dy, const float* z, int ld, int N,
                                                   int Rv, const float* mean, const float* invstd, double* red) {
  const int nsub = 256 / N, sub = threadIdx.x / N, c = threadIdx.x - sub * N;
  if (sub < nsub) {
    const float mu = mean[c], is = invstd[c];
    double s1 = 0.0, s2 = 0.0;
    for (int r = sub; r < Rv; r += nsub) {
      const float d = dy[r * ld + c];
      s1 += d; s2 += (double)d * ((z[r * ld + c] - mu) * is);
    }
    red[sub * 2 * N + c] = s1; red[sub * 2 * N + N + c] = s2;
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  const bool has = (int)threadIdx.x < N;
  if (has)
    for (int k = 0; k < nsub; ++k) { s1 += red[k * 2 * N + threadIdx.x]; s2 += red[k * 2 * N + N + threadIdx.x]; }
  hf_arrive(S, stage, 2 * N, has, threadIdx.x, s1, N + threadIdx.x, s2);
}

// batch statistics (tot: sums over all rows) -> (scale, shift, mean, invstd) in LDS; workgroup 0 also writes them out and
// moves the moving statistics
__device__ __forceinline__ void hf_bn_finalize(const double* tot, int N, double count, const clsr_bn_ptrs& bn, float momentum,
                                               float eps, float (*o)[112]) {
  const int c = threadIdx.x;
  if (c < 112) {
    float sc = 0.f, sh = 0.f, mean = 0.f, invstd = 0.f;
    if (c < N) {
      const double m = tot[c] / count;
      double v = tot[N + c] / count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      const float var = (float)v;
      invstd = 1.0f / sqrtf(var + eps);
      sc = bn.gamma[c] * invstd;
      sh = bn.beta[c] - mean * sc;
      if (blockIdx.x == 0) {
        bn.moving_mean[c] = bn.moving_mean[c] * momentum + mean * (1.0f - momentum);
        bn.moving_var[c] = bn.moving_var[c] * momentum + var * (1.0f - momentum);
        bn.scale[c] = sc; bn.shift[c] = sh; bn.mean[c] = mean; bn.invstd[c] = invstd;
      }
    }
    o[0][c] = sc; o[1][c] = sh; o[2][c] = mean; o[3][c] = invstd;      // (padding features: 0 -> relu mask false)
  }
  __syncthreads();
}
// backward sums -> dz = a1 * dy + a2 * z + a3 coefficients in LDS; workgroup 0 writes coef / dgamma / dbeta
__device__ __forceinline__ void hf_bn_coef(const double* tot, int N, double count, double gscale, const clsr_bn_ptrs& bn,
                                           const float (*o)[112], float (*cf)[112]) {
  const int c = threadIdx.x;
  if (c < 112) {
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < N) {
      const double s1 = tot[c], s2 = tot[N + c];
      const float g = bn.gamma[c], is = o[3][c], mu = o[2][c];
      const float c1 = (float)(s1 / count), c2 = (float)(s2 / count);
      a1 = g * is;
      a2 = -g * is * is * c2;
      a3 = -a1 * c1 - a2 * mu;
      if (blockIdx.x == 0) {
        bn.coef[c] = a1; bn.coef[N + c] = a2; bn.coef[2 * N + c] = a3;
        // (global sums under synchronised statistics: pre-divided by the world size, the gradient all-reduce restores them)
        bn.dgamma[c] = (float)(s2 * gscale);
        bn.dbeta[c] = (float)(s1 * gscale);
      }
    }
    cf[0][c] = a1; cf[1][c] = a2; cf[2][c] = a3;
  }
  __syncthreads();
}
// dy (LDS, in place) -> dz = a1 * dy + a2 * z + a3 for the valid rows and the N real features (zero elsewhere: the
// buffer is the next product's operand), and out to the global dz tensor
__device__ __forceinline__ void hf_bn_apply(float* dy, const float* z, int ld, int N, int Rv, const float (*cf)[112],
                                            float* gout, long row0) {
  const int QC = (ld - 4) >> 2;
  for (int e = threadIdx.x; e < HF_RP * QC; e += 256) {
    const int r = e / QC, c = 4 * (e - r * QC);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < Rv && c < N) {
      v = ld4(cf[0] + c) * ld4(dy + r * ld + c) + ld4(cf[1] + c) * ld4(z + r * ld + c) + ld4(cf[2] + c);
      st4(gout + (row0 + r) * N + c, v);
    }
    st4(dy + r * ld + c, v);
  }
}
// act = relu(z * scale + shift) for the N real features of all rows, zero padding features
__device__ __forceinline__ void hf_act(float* act, const float* z, int ld, int N, const float (*o)[112]) {
  const int QC = (ld - 4) >> 2;
  for (int e = threadIdx.x; e < HF_RP * QC; e += 256) {
    const int r = e / QC, c = 4 * (e - r * QC);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
      v = ld4(z + r * ld + c) * ld4(o[0] + c) + ld4(o[1] + c);
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    st4(act + r * ld + c, v);
  }
}


// -DHF_TIMING (diagnosis builds, scripts/heads_phases.py): workgroup 0 stamps the 100 MHz wall clock at the phase boundaries
#ifdef HF_TIMING
#define HF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) stamps[(i)] = wall_clock64(); } while (0)
#else
#define HF_STAMP(i) do { } while (0)
#endif
// global rows -> LDS rows with ALL loads of a thread in flight before the first store (a workgroup is one wave per SIMD:
// nothing else hides the latency of a dependent load -> store loop; the alpha-gate input took 43 us that way)
#define HF_NIT(n) (((n) + 255) / 256)

// ============================================================================================== launch 1
__global__ void __launch_bounds__(256, 1) heads_fused_k1(clsr_heads_desc a, int R, int nb, HfPeers peers) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hf_raw[];
  HfLds& L = *reinterpret_cast<HfLds*>(hf_raw);
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * R;
  const int G = a.G;
  const int Rv = (int)((a.B - row0) < R ? (a.B - row0) : R);
  const HfSync S = hf_sync_init(a.workspace, nb, L.flag, 0, &peers);
  const double count = (double)a.B * peers.world;      // (synchronised statistics: the rows of all ranks)
  const double gscale = 1.0 / peers.world;
  long long* stamps = reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(a.workspace) + HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES);
  (void)stamps;
  HF_STAMP(0);
  float* A = L.X.A;
  float* Z1A = L.X.e.Z1A;
  float* MO = L.X.m.MO;
  float* Z0L = L.X.m.Z0L;
  float* DY1 = L.X.e.DY1;                  // (stride HS_64)
  float* Z1L = L.Z0A;                      // (stride HS_64: the first alpha layer's z0 is dead once alpha is known)

  // ---- F0: alpha-gate input rows [final state | target | long-term | short-term | time to now], first alpha layer
  HfW<11, 2> w0;
  hf_load_w(w0, a.al_w0, a.kp_al_w0, HF_A0 / 16);
  {
    constexpr int QA = HS_A / 4, NIT = HF_NIT(HF_RP * QA);
    f32x4 v[NIT];
    float tn[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int e = tid + 256 * k, r = e / QA, c = 4 * (e - r * QA);
      // (branch-free: one unconditional 16-byte load per element from a selected source row -- with an if / else chain per
      // source the loads of the fifteen elements left one branch at a time, 17 us)
      const bool ok = r < Rv && c < HF_NFS + 3 * HF_D;
      const long b = ok ? row0 + r : row0, h = b / G;
      const int cc = ok ? c : 0;
      const float* src = cc < HF_NFS ? a.fs + h * HF_NFS + cc
                       : cc < HF_NFS + HF_D ? a.target + b * HF_D + (cc - HF_NFS)
                       : cc < HF_NFS + 2 * HF_D ? a.att_long + h * HF_D + (cc - HF_NFS - HF_D)
                       : a.att_short + b * HF_D + (cc - HF_NFS - 2 * HF_D);
      const f32x4 x = ld4(src);
      v[k] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
      tn[k] = a.tnow[((r < Rv ? row0 + r : row0) / a.tnow_group) * a.tnow_stride + a.tnow_col];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int e = tid + 256 * k, r = e / QA, c = 4 * (e - r * QA);
      if (r < HF_RP) {
        if (r < Rv && c == HF_NFS + 3 * HF_D) v[k].x = tn[k];
        st4(A + r * HS_A + c, v[k]);
        if (r < Rv && c < HF_LD) st4(a.ain + (row0 + r) * HF_LD + c, v[k]);
      }
    }
  }
  if (tid < 64) L.wout[tid] = tid < HF_A1 ? a.al_wout[tid] : 0.f;
  __syncthreads();
  HF_STAMP(1);
  hf_gemm(w0, A, HS_A, HF_A0 / 16, HfEpiFwd{L.Z0A, HS_80, a.al_b0, HF_A0, a.al_z0, row0, Rv});
  HF_STAMP(2);
  __syncthreads();
  hf_stats_arrive(S, 0, L.Z0A, HS_80, HF_A0, Rv, L.red);
  HfW<5, 1> w1;
  hf_load_w(w1, a.al_w1, a.kp_al_w1, 3);
  HF_STAMP(3);
  hf_wait(S, 0, 2 * HF_A0, L.tot);
  HF_STAMP(4);

  // ---- F1: second alpha layer
  hf_bn_finalize(L.tot, HF_A0, count, a.bn[0], a.momentum, a.eps, L.bn[0]);
  hf_act(L.ACT, L.Z0A, HS_80, HF_A0, L.bn[0]);
  __syncthreads();
  HF_STAMP(5);
  hf_gemm(w1, L.ACT, HS_80, 3, HfEpiFwd{Z1A, HS_40, a.al_b1, HF_A1, a.al_z1, row0, Rv});
  HF_STAMP(6);
  __syncthreads();
  hf_stats_arrive(S, 1, Z1A, HS_40, HF_A1, Rv, L.red);
  HfW<5, 2> w2;
  hf_load_w(w2, a.lg_w0, a.kp_lg_w0, 7);
  // the rows of model_output that do not depend on alpha are fetched under the wait as well
  constexpr int QM = HS_80 / 4, NITM = HF_NIT(HF_RP * QM);
  f32x4 mv[NITM], ms[NITM];
#pragma unroll
  for (int k = 0; k < NITM; ++k) {
    const int e = tid + 256 * k, r = e / QM, c = 4 * (e - r * QM);
    mv[k] = f32x4{0.f, 0.f, 0.f, 0.f}; ms[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < Rv && c < 2 * HF_D) {
      const long b = row0 + r;
      if (c < HF_D) { mv[k] = ld4(a.att_long + (b / G) * HF_D + c); ms[k] = ld4(a.att_short + b * HF_D + c); }
      else mv[k] = ld4(a.target + b * HF_D + (c - HF_D));
    }
  }
  HF_STAMP(7);
  hf_wait(S, 1, 2 * HF_A1, L.tot);
  HF_STAMP(8);

  // ---- F2: alpha, model_output = [alpha * long + (1 - alpha) * short | target], first logit layer
  hf_bn_finalize(L.tot, HF_A1, count, a.bn[1], a.momentum, a.eps, L.bn[1]);
  if (tid < Rv) {
    float s = a.al_bout[0];
    for (int c = 0; c < HF_A1; ++c) s += fmaxf(Z1A[tid * HS_40 + c] * L.bn[1][0][c] + L.bn[1][1][c], 0.f) * L.wout[c];
    const float al = sigmoidf_(s);
    L.alpha[tid] = al;
    a.alpha[row0 + tid] = al;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NITM; ++k) {
    const int e = tid + 256 * k, r = e / QM, c = 4 * (e - r * QM);
    if (r < HF_RP) {
      f32x4 v = mv[k];
      if (r < Rv && c < HF_D) { const float al = L.alpha[r]; v = mv[k] * al + ms[k] * (1.0f - al); }
      st4(MO + r * HS_80 + c, v);
      if (r < Rv && c < 2 * HF_D) st4(a.mo + (row0 + r) * 2 * HF_D + c, v);
    }
  }
  if (tid < 64) L.wout[tid] = a.lg_wout[tid];
  __syncthreads();
  HF_STAMP(9);
  hf_gemm(w2, MO, HS_80, 7, HfEpiFwd{Z0L, HS_100, a.lg_b0, HF_L0, a.lg_z0, row0, Rv});
  HF_STAMP(10);
  __syncthreads();
  hf_stats_arrive(S, 2, Z0L, HS_100, HF_L0, Rv, L.red);
  HfW<7, 1> w3;
  hf_load_w(w3, a.lg_w1, a.kp_lg_w1, 4);
  HF_STAMP(11);
  hf_wait(S, 2, 2 * HF_L0, L.tot);
  HF_STAMP(12);

  // ---- F3: second logit layer
  hf_bn_finalize(L.tot, HF_L0, count, a.bn[2], a.momentum, a.eps, L.bn[2]);
  hf_act(L.ACT, Z0L, HS_100, HF_L0, L.bn[2]);
  __syncthreads();
  HF_STAMP(13);
  hf_gemm(w3, L.ACT, HS_100, 4, HfEpiFwd{Z1L, HS_64, a.lg_b1, HF_L1, a.lg_z1, row0, Rv});
  HF_STAMP(14);
  __syncthreads();
  hf_stats_arrive(S, 3, Z1L, HS_64, HF_L1, Rv, L.red);
  HfW<4, 2> w4;
  hf_load_w(w4, a.lg_w1T, a.kp_lg_w1T, 7);
  float lab[HF_MAXG];                      // labels of this thread's softmax group
#pragma unroll
  for (int g = 0; g < HF_MAXG; ++g) lab[g] = (tid < HF_RP && g < G && tid * G + g < Rv) ? a.labels[row0 + (long)tid * G + g] : 0.f;
  HF_STAMP(15);
  hf_wait(S, 3, 2 * HF_L1, L.tot);
  HF_STAMP(16);

  // ---- F4 / B0: logits, softmax loss over the groups of G rows, d(output layer)
  hf_bn_finalize(L.tot, HF_L1, count, a.bn[3], a.momentum, a.eps, L.bn[3]);
  if (tid < Rv) {
    float s = a.lg_bout[0];
    for (int c = 0; c < HF_L1; ++c) s += fmaxf(Z1L[tid * HS_64 + c] * L.bn[3][0][c] + L.bn[3][1][c], 0.f) * L.wout[c];
    L.logit[tid] = s;
    a.logit[row0 + tid] = s;
  }
  __syncthreads();
  if (tid < HF_RP) {
    float local = 0.f;
    if (tid * G < Rv) {
      const float* lg = L.logit + tid * G;
      float mx = -INFINITY;
      for (int g = 0; g < G; ++g) mx = fmaxf(mx, lg[g]);
      float sum = 0.f;
      for (int g = 0; g < G; ++g) sum += __expf(lg[g] - mx);
      const float lse = mx + __logf(sum);
      int npos = 0;
#pragma unroll
      for (int g = 0; g < HF_MAXG; ++g)
        if (g < G && lab[g] == 1.0f) { ++npos; local -= (lg[g] - lse); }
#pragma unroll
      for (int g = 0; g < HF_MAXG; ++g)
        if (g < G) {
          const float dl = a.lscale * ((float)npos * __expf(lg[g] - lse) - (lab[g] == 1.0f ? 1.0f : 0.0f));
          L.dl[tid * G + g] = dl;
          if (a.dlogit) a.dlogit[row0 + tid * G + g] = dl;
        }
    }
    L.lossp[tid] = local;
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f;
    for (int p = 0; p < HF_RP; ++p) l += L.lossp[p];
    if (l != 0.f) atomicAdd(a.loss, (double)l * a.lscale);
  }
  for (int e = tid; e < HF_RP * (HS_64 / 4); e += 256) {
    const int r = e / (HS_64 / 4), c = 4 * (e - r * (HS_64 / 4));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < Rv && c < HF_L1) {
      const f32x4 y = ld4(Z1L + r * HS_64 + c) * ld4(L.bn[3][0] + c) + ld4(L.bn[3][1] + c);
      const f32x4 w = ld4(L.wout + c) * L.dl[r];
      v.x = y.x > 0.f ? w.x : 0.f; v.y = y.y > 0.f ? w.y : 0.f; v.z = y.z > 0.f ? w.z : 0.f; v.w = y.w > 0.f ? w.w : 0.f;
    }
    st4(DY1 + r * HS_64 + c, v);
  }
  __syncthreads();
  hf_bwd_sums_arrive(S, 4, DY1, Z1L, HS_64, HF_L1, Rv, L.bn[3][2], L.bn[3][3], L.red);
  if (tid < HF_L1) {           // d w_out partial of this workgroup: sum over its rows of relu(y) * dlogit
    float sw = 0.f;
    for (int r = 0; r < Rv; ++r)
      sw += fmaxf(Z1L[r * HS_64 + tid] * L.bn[3][0][tid] + L.bn[3][1][tid], 0.f) * L.dl[r];
    a.lg_wp[(long)blockIdx.x * (HF_L1 + 4) + tid] = sw;
  } else if (tid == HF_L1) {
    float sb = 0.f;
    for (int r = 0; r < Rv; ++r) sb += L.dl[r];
    a.lg_wp[(long)blockIdx.x * (HF_L1 + 4) + HF_L1] = sb;
  }
  HF_STAMP(17);
  hf_wait(S, 4, 2 * HF_L1, L.tot);
  HF_STAMP(18);

  // ---- B1: dz of the second logit layer, back through W1
  hf_bn_coef(L.tot, HF_L1, count, gscale, a.bn[3], L.bn[3], L.cf);
  hf_bn_apply(DY1, Z1L, HS_64, HF_L1, Rv, L.cf, a.lg_dz1, row0);
  __syncthreads();
  HF_STAMP(19);
  hf_gemm(w4, DY1, HS_64, 7, HfEpiMask{L.ACT, HS_100, Z0L, L.bn[2][0], L.bn[2][1]});
  HF_STAMP(20);
  __syncthreads();
  hf_bwd_sums_arrive(S, 5, L.ACT, Z0L, HS_100, HF_L0, Rv, L.bn[2][2], L.bn[2][3], L.red);
  HfW<7, 2> w5;
  hf_load_w(w5, a.lg_w0T, a.kp_lg_w0T, 5);
  HF_STAMP(21);
  hf_wait(S, 5, 2 * HF_L0, L.tot);
  HF_STAMP(22);

  // ---- B2: dz of the first logit layer, d(model_output)
  hf_bn_coef(L.tot, HF_L0, count, gscale, a.bn[2], L.bn[2], L.cf);
  hf_bn_apply(L.ACT, Z0L, HS_100, HF_L0, Rv, L.cf, a.lg_dz0, row0);
  __syncthreads();
  float* dmo = a.dmo;
  HF_STAMP(23);
  hf_gemm(w5, L.ACT, HS_100, 5, [=](int f0, int pos, f32x4 v) {
    if (pos < Rv) st4(dmo + (row0 + pos) * 2 * HF_D + f0, v);
  });
  HF_STAMP(40);
}

// ============================================================================================== launch 2
__global__ void __launch_bounds__(256, 1) heads_fused_k2(clsr_heads_desc a, int R, int nb, HfPeers peers) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hf_raw[];
  HfLds2& L = *reinterpret_cast<HfLds2*>(hf_raw);
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * R;
  const int G = a.G;
  const int Rv = (int)((a.B - row0) < R ? (a.B - row0) : R);
  const HfSync S = hf_sync_init(a.workspace, nb, L.flag, 0, &peers);
  const double count = (double)a.B * peers.world;      // (synchronised statistics: the rows of all ranks)
  const double gscale = 1.0 / peers.world;
  long long* stamps = reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(a.workspace) + HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES) + 64;
  (void)stamps;
  HF_STAMP(0);
  float* Z1A = L.W.e.Z1A;
  float* DY1 = L.W.e.DY1;
  float* tmp = L.W.e.tmp;
  float* DAIN = L.W.DAIN;

  HfW<3, 2> w0;
  hf_load_w(w0, a.al_w1T, a.kp_al_w1T, 5);
  // ---- the alpha MLP's activations and statistics and d(model_output) (written by launch 1): all loads, then all stores
  {
    constexpr int Q1 = HS_40 / 4, N1 = HF_NIT(HF_RP * Q1), Q0 = HS_80 / 4, N0 = HF_NIT(HF_RP * Q0), QD = 2 * HF_D / 4,
                  ND = HF_NIT(HF_RP * QD);
    f32x4 v1[N1], v0[N0], vd[ND];
#pragma unroll
    for (int k = 0; k < N1; ++k) {
      const int e = tid + 256 * k, r = e / Q1, c = 4 * (e - r * Q1);
      v1[k] = (r < Rv && c < HF_A1) ? ld4(a.al_z1 + (row0 + r) * HF_A1 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < N0; ++k) {
      const int e = tid + 256 * k, r = e / Q0, c = 4 * (e - r * Q0);
      v0[k] = (r < Rv && c < HF_A0) ? ld4(a.al_z0 + (row0 + r) * HF_A0 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int e = tid + 256 * k, r = e / QD, c = 4 * (e - r * QD);
      vd[k] = r < Rv ? ld4(a.dmo + (row0 + r) * 2 * HF_D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < N1; ++k) {
      const int e = tid + 256 * k, r = e / Q1, c = 4 * (e - r * Q1);
      if (r < HF_RP) st4(Z1A + r * HS_40 + c, v1[k]);
    }
#pragma unroll
    for (int k = 0; k < N0; ++k) {
      const int e = tid + 256 * k, r = e / Q0, c = 4 * (e - r * Q0);
      if (r < HF_RP) st4(L.Z0A + r * HS_80 + c, v0[k]);
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int e = tid + 256 * k, r = e / QD, c = 4 * (e - r * QD);
      if (r < HF_RP) st4(L.DMO + r * 2 * HF_D + c, vd[k]);
    }
  }
  if (tid < 112) {
    const bool v0 = tid < HF_A0, v1 = tid < HF_A1;
    L.bn[0][0][tid] = v0 ? a.bn[0].scale[tid] : 0.f; L.bn[0][1][tid] = v0 ? a.bn[0].shift[tid] : 0.f;
    L.bn[0][2][tid] = v0 ? a.bn[0].mean[tid] : 0.f; L.bn[0][3][tid] = v0 ? a.bn[0].invstd[tid] : 0.f;
    L.bn[1][0][tid] = v1 ? a.bn[1].scale[tid] : 0.f; L.bn[1][1][tid] = v1 ? a.bn[1].shift[tid] : 0.f;
    L.bn[1][2][tid] = v1 ? a.bn[1].mean[tid] : 0.f; L.bn[1][3][tid] = v1 ? a.bn[1].invstd[tid] : 0.f;
  }
  if (tid < 64) L.wout[tid] = tid < HF_A1 ? a.al_wout[tid] : 0.f;
  if (tid < HF_RP) L.alpha[tid] = tid < Rv ? a.alpha[row0 + tid] : 0.f;
  // fusion backward, alpha share: dalpha_logit[b] = alpha (1 - alpha) * sum_c due[b, c] * (long[h, c] - short[b, c])
  {
    constexpr int NT = HF_NIT(HF_RP * HF_D / 4);
    f32x4 lv[NT], sv[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int e = tid + 256 * k, r = e / (HF_D / 4), c = 4 * (e - r * (HF_D / 4));
      lv[k] = f32x4{0.f, 0.f, 0.f, 0.f}; sv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r < Rv) {
        const long b = row0 + r;
        lv[k] = ld4(a.att_long + (b / G) * HF_D + c); sv[k] = ld4(a.att_short + b * HF_D + c);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int e = tid + 256 * k, r = e / (HF_D / 4), c = 4 * (e - r * (HF_D / 4));
      if (r < HF_RP) st4(tmp + r * HF_D + c, ld4(L.DMO + r * 2 * HF_D + c) * (lv[k] - sv[k]));
    }
  }
  __syncthreads();
  if (tid < HF_RP) {
    float s = 0.f;
    if (tid < Rv) {
      for (int c = 0; c < HF_D; ++c) s += tmp[tid * HF_D + c];
      const float al = L.alpha[tid];
      s = s * al * (1.0f - al);
    }
    L.dal[tid] = s;
  }
  __syncthreads();
  for (int e = tid; e < HF_RP * (HS_40 / 4); e += 256) {
    const int r = e / (HS_40 / 4), c = 4 * (e - r * (HS_40 / 4));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < Rv && c < HF_A1) {
      const f32x4 y = ld4(Z1A + r * HS_40 + c) * ld4(L.bn[1][0] + c) + ld4(L.bn[1][1] + c);
      const f32x4 w = ld4(L.wout + c) * L.dal[r];
      v.x = y.x > 0.f ? w.x : 0.f; v.y = y.y > 0.f ? w.y : 0.f; v.z = y.z > 0.f ? w.z : 0.f; v.w = y.w > 0.f ? w.w : 0.f;
    }
    st4(DY1 + r * HS_40 + c, v);
  }
  __syncthreads();
  hf_bwd_sums_arrive(S, 6, DY1, Z1A, HS_40, HF_A1, Rv, L.bn[1][2], L.bn[1][3], L.red);
  if (tid < HF_A1) {
    float sw = 0.f;
    for (int r = 0; r < Rv; ++r)
      sw += fmaxf(Z1A[r * HS_40 + tid] * L.bn[1][0][tid] + L.bn[1][1][tid], 0.f) * L.dal[r];
    a.al_wp[(long)blockIdx.x * (HF_A1 + 4) + tid] = sw;
  } else if (tid == HF_A1) {
    float sb = 0.f;
    for (int r = 0; r < Rv; ++r) sb += L.dal[r];
    a.al_wp[(long)blockIdx.x * (HF_A1 + 4) + HF_A1] = sb;
  }
  HF_STAMP(1);
  hf_wait(S, 6, 2 * HF_A1, L.tot);
  HF_STAMP(2);

  // ---- dz of the second alpha layer, back through W1
  hf_bn_coef(L.tot, HF_A1, count, gscale, a.bn[1], L.bn[1], L.cf);
  hf_bn_apply(DY1, Z1A, HS_40, HF_A1, Rv, L.cf, a.al_dz1, row0);
  __syncthreads();
  HF_STAMP(3);
  hf_gemm(w0, DY1, HS_40, 5, HfEpiMask{L.DY0, HS_80, L.Z0A, L.bn[0][0], L.bn[0][1]});
  HF_STAMP(4);
  __syncthreads();
  hf_bwd_sums_arrive(S, 7, L.DY0, L.Z0A, HS_80, HF_A0, Rv, L.bn[0][2], L.bn[0][3], L.red);
  HfW<5, 3> w1;
  hf_load_w(w1, a.al_w0T, a.kp_al_w0T, 10);
  // the accumulators this launch adds to are fetched under the wait: row-level dtarget / dS, history-level dL / dfs
  constexpr int NR = HF_NIT(HF_RP * HF_D / 4), NH = HF_NIT(HF_RP * HF_D / 4);
  f32x4 gt[NR], gs[NR], gl[NH], gf[NH];
  const int ng = Rv / G;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int e = tid + 256 * k, r = e / (HF_D / 4), c = 4 * (e - r * (HF_D / 4));
    gt[k] = f32x4{0.f, 0.f, 0.f, 0.f}; gs[k] = gt[k]; gl[k] = gt[k]; gf[k] = gt[k];
    if (r < Rv) { gt[k] = ld4(a.dtarget + (row0 + r) * HF_D + c); gs[k] = ld4(a.dS + (row0 + r) * HF_D + c); }
    if (r < ng) { gl[k] = ld4(a.dL + (row0 / G + r) * HF_D + c); gf[k] = ld4(a.dfs + (row0 / G + r) * HF_NFS + c); }
  }
  HF_STAMP(5);
  hf_wait(S, 7, 2 * HF_A0, L.tot);
  HF_STAMP(6);

  // ---- dz of the first alpha layer, d(alpha-gate input), scattered to its sources together with the fusion's shares
  hf_bn_coef(L.tot, HF_A0, count, gscale, a.bn[0], L.bn[0], L.cf);
  hf_bn_apply(L.DY0, L.Z0A, HS_80, HF_A0, Rv, L.cf, a.al_dz0, row0);
  __syncthreads();
  float* dain = DAIN;
  HF_STAMP(7);
  hf_gemm(w1, L.DY0, HS_80, 10, [=](int f0, int pos, f32x4 v) { st4(dain + pos * HS_A + f0, v); });
  HF_STAMP(8);
  __syncthreads();
  // fusion backward (d user_embed = dmo[:, :D]: alpha * due to the long-term interest, (1 - alpha) * due to the short-term
  // one; dmo[:, D:] to the target) and d(alpha-gate input), added to their sources in one read-modify-write each
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int e = tid + 256 * k, r = e / (HF_D / 4), c = 4 * (e - r * (HF_D / 4));
    if (r < Rv) {             // row-level: dtarget, dS
      const float al = L.alpha[r];
      st4(a.dtarget + (row0 + r) * HF_D + c, gt[k] + (ld4(L.DMO + r * 2 * HF_D + HF_D + c) + ld4(DAIN + r * HS_A + HF_NFS + c)));
      st4(a.dS + (row0 + r) * HF_D + c,
          gs[k] + (ld4(L.DMO + r * 2 * HF_D + c) * (1.0f - al) + ld4(DAIN + r * HS_A + HF_NFS + 2 * HF_D + c)));
    }
    if (r < ng) {             // history-level: dL, dfs (sums over the group's rows, in row order)
      f32x4 sl = {0.f, 0.f, 0.f, 0.f}, sf = {0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < G; ++g) {
        const int rr = r * G + g;
        sl += ld4(L.DMO + rr * 2 * HF_D + c) * L.alpha[rr] + ld4(DAIN + rr * HS_A + HF_NFS + HF_D + c);
        sf += ld4(DAIN + rr * HS_A + c);
      }
      st4(a.dL + (row0 / G + r) * HF_D + c, gl[k] + sl);
      st4(a.dfs + (row0 / G + r) * HF_NFS + c, gf[k] + sf);
    }
  }
  HF_STAMP(40);
}

// ---------------------------------------------------------------------------------------------- host
static int hf_rows(long B, int G) {
  const long groups = B / G;
  const long gpb = (groups + HF_MAXBLK - 1) / HF_MAXBLK;
  return (int)(gpb * G);
}
extern "C" int clsr_sizeof_heads_desc(void) { return (int)sizeof(clsr_heads_desc); }
extern "C" int clsr_heads_fused_parts(long B, int G) {
  const int R = hf_rows(B, G);
  return (int)((B + R - 1) / R);
}
// compute units of the current device: every workgroup of a launch needs a CU of its own for the whole launch (one
// workgroup per CU by LDS footprint) -- on a partitioned device with fewer CUs the grid barriers would never complete
static int hf_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
extern "C" int clsr_heads_fused_supported(long B, int G, int D, int nfs, int a_in, int A0, int A1, int L0, int L1) {
  if (B <= 0 || G < 1 || G > HF_MAXG || B % G) return 0;
  if (D != HF_D || nfs != HF_NFS || a_in != HF_AIN || A0 != HF_A0 || A1 != HF_A1 || L0 != HF_L0 || L1 != HF_L1) return 0;
  if (hf_rows(B, G) > HF_RP) return 0;
  return clsr_heads_fused_parts(B, G) <= hf_cu_count();
}
extern "C" long clsr_heads_fused_workspace_bytes(void) {
  return (long)HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES + HF_DBG_BYTES;   // (debug: phase stamps / barrier times of -DHF_TIMING builds)
}
extern "C" long clsr_heads_fused_counter_bytes(void) { return HF_CTR_BYTES; }
// != 0: a workgroup gave up waiting at a grid barrier in some launch since the error words were last cleared
// (clsr_heads_fused_clear_error; the step's own zero fill does NOT reach them): the results of that step are invalid
static const unsigned* hf_err_words(const void* workspace) {
  return reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(workspace) + HF_CTR_BYTES + HF_PART_BYTES + HF_GPART_BYTES);
}
extern "C" int clsr_heads_fused_clear_error(void* workspace) {
  CLSR_CHECK_ARG(workspace);
  CLSR_HIP(hipMemset(const_cast<unsigned*>(hf_err_words(workspace)), 0, 3 * sizeof(unsigned)));
  return CLSR_OK;
}
extern "C" int clsr_heads_fused_error(const void* workspace) {
  if (!workspace) return -1;
  unsigned e[3] = {0, 0, 0};
  if (hipMemcpy(e, hf_err_words(workspace), sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (e[0] && getenv("CLSR_HEADS_DEBUG")) fprintf(stderr, "[clsr_heads_fused] barrier timeout: stage %u, counter %u of %u\n", e[0] - 1, e[1], e[2]);
  return (int)e[0];
}

static int hf_check(const clsr_heads_desc* d) {
  CLSR_CHECK_ARG(d && d->fs && d->target && d->att_long && d->att_short && d->tnow && d->labels && d->tnow_group > 0);
  CLSR_CHECK_SUPPORTED(clsr_heads_fused_supported(d->B, d->G, d->D, d->nfs, d->a_in, d->A0, d->A1, d->L0, d->L1));
  CLSR_CHECK_ARG(d->ld == HF_LD);
  CLSR_CHECK_ARG(d->al_w0 && d->al_w1 && d->lg_w0 && d->lg_w1 && d->al_w0T && d->al_w1T && d->lg_w0T && d->lg_w1T);
  CLSR_CHECK_ARG(d->kp_al_w0 == 16 * 11 + 4 && d->kp_al_w1 == HS_80 && d->kp_lg_w0 == HS_80 && d->kp_lg_w1 == HS_100 &&
                 d->kp_al_w0T == HS_80 && d->kp_al_w1T == HS_40 && d->kp_lg_w0T == HS_100 && d->kp_lg_w1T == HS_64);
  CLSR_CHECK_ARG(d->al_b0 && d->al_b1 && d->al_wout && d->al_bout && d->lg_b0 && d->lg_b1 && d->lg_wout && d->lg_bout);
  for (int i = 0; i < 4; ++i) {
    const clsr_bn_ptrs& b = d->bn[i];
    CLSR_CHECK_ARG(b.gamma && b.beta && b.moving_mean && b.moving_var && b.scale && b.shift && b.mean && b.invstd && b.coef &&
                   b.dgamma && b.dbeta);
  }
  CLSR_CHECK_ARG(d->ain && d->al_z0 && d->al_z1 && d->alpha && d->mo && d->lg_z0 && d->lg_z1 && d->logit && d->loss);
  CLSR_CHECK_ARG(d->lg_dz1 && d->lg_dz0 && d->dmo && d->al_dz1 && d->al_dz0 && d->lg_wp && d->al_wp && d->dL && d->dS &&
                 d->dtarget && d->dfs);
  CLSR_CHECK_ARG(d->workspace && d->workspace_bytes >= clsr_heads_fused_workspace_bytes());
  CLSR_CHECK_ARG(((uintptr_t)d->workspace & 15) == 0);
  return CLSR_OK;
}

// ---- communicator for synchronised statistics across ranks (host creates it once; see HfXchg above)
extern "C" long clsr_heads_comm_buffer_bytes(void) { return (long)sizeof(HfXchg); }
extern "C" int clsr_heads_comm_max_world(void) { return HF_MAXW; }
// uncached (fine-grained) device allocation of the exchange buffer, zeroed; freed by clsr_comm_free, shared by
// clsr_comm_ipc_handle / clsr_comm_ipc_open / clsr_comm_ipc_close (csrc/p2p.hip: they are size-agnostic)
extern "C" int clsr_heads_comm_alloc(void** buf_out) {
  CLSR_CHECK_ARG(buf_out);
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, sizeof(HfXchg), hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, sizeof(HfXchg), hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    clsr_set_error("%s:%d: exchange buffer allocation failed: %s", __FILE__, __LINE__, hipGetErrorString(e));
    return CLSR_ELAUNCH;
  }
  CLSR_HIP(hipMemset(p, 0, sizeof(HfXchg)));
  CLSR_HIP(hipDeviceSynchronize());
  *buf_out = p;
  return CLSR_OK;
}
// bufs[r] = the exchange buffer of rank r as THIS process addresses it (own allocation at [rank], opened handles elsewhere)
extern "C" int clsr_heads_comm_create(int rank, int world, void* const* bufs, void** comm_out) {
  CLSR_CHECK_ARG(bufs && comm_out && world >= 1 && world <= HF_MAXW && rank >= 0 && rank < world);
  HfComm* c = new HfComm();
  c->rank = rank; c->world = world; c->pushed = 0;
  for (int r = 0; r < HF_MAXW; ++r) c->x[r] = nullptr;
  for (int r = 0; r < world; ++r) {
    if (!bufs[r]) { delete c; clsr_set_error("%s:%d: exchange buffer of rank %d missing", __FILE__, __LINE__, r); return CLSR_EINVAL; }
    c->x[r] = (HfXchg*)bufs[r];
  }
  *comm_out = c;
  return CLSR_OK;
}
extern "C" int clsr_heads_comm_destroy(void* comm) {
  delete (HfComm*)comm;
  return CLSR_OK;
}
static HfPeers hf_peers(const clsr_heads_desc* d, bool new_step, int nb) {
  HfPeers p;
  for (int r = 0; r < HF_MAXW; ++r) p.x[r] = nullptr;
  p.rank = 0; p.world = 1; p.target = 0;
  p.timeout_ticks = clsr_p2p_timeout_ticks();
  p.abort_flag = d->abort_flag;
  if (d->comm) {
    HfComm* c = (HfComm*)d->comm;
    // (every rank issues the same sequence of step1 / step2 calls with the same B: the same totals everywhere)
    if (new_step) c->pushed += (unsigned long long)((nb + HF_GS - 1) / HF_GS) * c->world;
    for (int r = 0; r < c->world; ++r) p.x[r] = c->x[r];
    p.rank = c->rank; p.world = c->world; p.target = c->pushed;
  }
  return p;
}

// Start-up check of a communicator: a small launch that runs the SAME arrive / wait code as the heads launches through all
// eight stages across the ranks (nblocks workgroups per rank, known contributions) and verifies the sums on the device.
// ok_out (device int): 1 = every stage gave the expected sums on this rank, 0 = a sum was wrong or a wait timed out
// (timeout_s, short: a rank that cannot see its peers' pushes must not hold the job for the production timeout).
__global__ void __launch_bounds__(256) heads_comm_selftest_kernel(void* workspace, HfPeers peers, int nb, int* ok_out) {
  __shared__ double tot[256];
  __shared__ int flag[4];
  const HfSync S = hf_sync_init(workspace, nb, flag, 0, &peers);
  bool ok = true;
  double ranks = 0.0;
  for (int r = 0; r < peers.world; ++r) ranks += r + 1;
  const double blocks = 0.5 * nb * (nb + 1.0);
  for (int stage = 0; stage < HF_NSTAGE; ++stage) {
    const int t = threadIdx.x;
    // element i of a workgroup's row: (rank + 1) * (block + 1) * (i + 1 + stage)
    hf_arrive(S, stage, 8, t < 4, t, (peers.rank + 1.0) * (blockIdx.x + 1.0) * (t + 1.0 + stage), t + 4,
              (peers.rank + 1.0) * (blockIdx.x + 1.0) * (t + 5.0 + stage));
    hf_wait(S, stage, 8, tot);
    if (t < 8 && tot[t] != ranks * blocks * (t + 1.0 + stage)) ok = false;
    __syncthreads();
  }
  if (!ok || __hip_atomic_load(S.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicAnd(ok_out, 0);
}
extern "C" int clsr_heads_comm_self_test(void* comm, int nblocks, void* workspace, long workspace_bytes, int* ok_out,
                                         float timeout_s, void* stream) {
  CLSR_CHECK_ARG(comm && workspace && ok_out && nblocks >= 1 && nblocks <= HF_MAXBLK && timeout_s > 0.f);
  CLSR_CHECK_ARG(workspace_bytes >= clsr_heads_fused_workspace_bytes() && ((uintptr_t)workspace & 15) == 0);
  HfComm* c = (HfComm*)comm;
  CLSR_HIP(hipMemsetAsync(workspace, 0, HF_CTR_BYTES, (hipStream_t)stream));
  int one = 1;
  CLSR_HIP(hipMemcpyAsync(ok_out, &one, sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
  CLSR_HIP(hipStreamSynchronize((hipStream_t)stream));
  HfPeers p;
  for (int r = 0; r < HF_MAXW; ++r) p.x[r] = r < c->world ? c->x[r] : nullptr;
  p.rank = c->rank; p.world = c->world;
  c->pushed += (unsigned long long)((nblocks + HF_GS - 1) / HF_GS) * c->world;
  p.target = c->pushed;
  p.timeout_ticks = (long long)(timeout_s * 1e8f);
  p.abort_flag = nullptr;
  hipLaunchKernelGGL(heads_comm_selftest_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, workspace, p, nblocks, ok_out);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Launch 1.  The first clsr_heads_fused_counter_bytes() bytes of the workspace (the barrier counters) must be ZERO on
// entry: the caller clears them once per step, before this launch (the step's zero-fill launch does).
extern "C" int clsr_heads_fused_step1(const clsr_heads_desc* d, void* stream) {
  const int rc = hf_check(d);
  if (rc != CLSR_OK) return rc;
  const int R = hf_rows(d->B, d->G), nb = clsr_heads_fused_parts(d->B, d->G);
  const size_t shmem = sizeof(HfLds);
  CLSR_HIP(hipFuncSetAttribute((const void*)heads_fused_k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(heads_fused_k1, dim3(nb), dim3(256), shmem, (hipStream_t)stream, *d, R, nb, hf_peers(d, true, nb));
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
// Launch 2: after launch 1 on the same stream AND after every other writer of dL / dS / dtarget / dfs so far (the
// contrastive loss): it adds to them with plain read-modify-writes.
extern "C" int clsr_heads_fused_step2(const clsr_heads_desc* d, void* stream) {
  const int rc = hf_check(d);
  if (rc != CLSR_OK) return rc;
  const int R = hf_rows(d->B, d->G), nb = clsr_heads_fused_parts(d->B, d->G);
  const size_t shmem = sizeof(HfLds2);
  CLSR_HIP(hipFuncSetAttribute((const void*)heads_fused_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(heads_fused_k2, dim3(nb), dim3(256), shmem, (hipStream_t)stream, *d, R, nb, hf_peers(d, false, nb));
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
