// Embedding gathers / scatters of the CLSR step (gfx950).
//
// Reference call sites replaced: tf.nn.embedding_lookup at
//   models/sequential/sequential_base_model.py:384-437 (item / cate history + target rows),
//   models/sequential/clsr.py:108-116 (user long / short rows),
// the history prologue at models/sequential/clsr.py:145-150,157,173-177
//   (concat, hist_mean, hist_recent), and the IndexedSlices gradients of those lookups
//   (dense apply semantics, models/base_model.py:263-264).
//
// Layout: tables row-major fp32 [V, C]; history tensors [Hn, T, D] with D = Di + Dc,
// item part first.  Histories are "history-level" rows: in training every positive owns one
// history that the reference replicates (1 + train_num_ngs) times; the caller passes the
// row-level index arrays with a row stride (idx_row_stride = G*T) instead of copies.
#include "common.h"

// One wave per history.  lane -> (t-slot = lane / QD, q = lane % QD), QD = D/4 float4 columns.
// Each lane copies 16-byte pieces of embedding rows (coalesced 128 B / 32 B row reads, fully
// coalesced [T*D] output) and accumulates the masked mean / recent-k mean for its column chunk.
template <int U>
__global__ void __launch_bounds__(256) gather_hist_fwd_kernel(
    const float* __restrict__ item_tbl, const float* __restrict__ cate_tbl,
    const int* __restrict__ item_idx, const int* __restrict__ cate_idx, long idx_row_stride,
    const int* __restrict__ seq_len, int len_stride, int Hn, int T, int Di, int Dc, int recent_k,
    float* __restrict__ hist, float* __restrict__ hist_mean, float* __restrict__ hist_recent) {
  __shared__ f32x4 red[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + wave;
  const int D = Di + Dc, QD = D >> 2, QI = Di >> 2;
  const int tpar = 64 / QD;
  const int tslot = lane / QD, q = lane - tslot * QD;
  const bool active = (h < Hn) && (tslot < tpar);
  f32x4 msum = {0.f, 0.f, 0.f, 0.f}, rsum = {0.f, 0.f, 0.f, 0.f};
  int len = 0;
  if (active) {
    len = seq_len[(long)h * len_stride];
    const int* ii = item_idx + (long)h * idx_row_stride;
    const int* ci = cate_idx + (long)h * idx_row_stride;
    float* out = hist + (long)h * T * D;
    const int rlo = len - recent_k;
    // this lane copies 16-byte piece q of the rows of steps tslot, tslot + tpar, ...  Chunks of U steps: all U
    // indices first, then U independent row reads in flight, then the stores -- the random 384 B / 128 B row
    // reads of a big catalogue are HBM latency bound unless many of them are outstanding per wave (U = 4 -> 13:
    // 38.7 -> 32 us on the 100M-item catalogue).  The host picks U so that the chunks are evenly filled.
    const bool is_item = q < QI;
    const int* idx = is_item ? ii : ci;
    const float* tbl = is_item ? item_tbl + 4 * q : cate_tbl + 4 * (q - QI);
    const long C = is_item ? Di : Dc;
    for (int t0 = tslot; t0 < T; t0 += U * tpar) {
      int id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tpar;
        id[u] = idx[t < T ? t : tslot];
      }
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld4(tbl + (long)id[u] * C);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tpar;
        if (t < T) {
          st4(out + (long)t * D + 4 * q, v[u]);
          if (t < len) {
            msum += v[u];
            if (t >= rlo) rsum += v[u];
          }
        }
      }
    }
  }
  red[wave][0][lane] = msum;
  red[wave][1][lane] = rsum;
  __syncthreads();
  if (active && tslot == 0) {
    for (int s = 1; s < tpar; ++s) {
      msum += red[wave][0][lane + s * QD];
      rsum += red[wave][1][lane + s * QD];
    }
    // reference: sum / sum(mask)  and  sum / sum(recent_mask); len == 0 gives 0/0 = nan there too
    const float inv_len = 1.0f / (float)len;
    const float inv_rec = 1.0f / (float)(len < recent_k ? len : recent_k);
    st4(hist_mean + (long)h * D + 4 * q, msum * inv_len);
    st4(hist_recent + (long)h * D + 4 * q, rsum * inv_rec);
  }
}

extern "C" int clsr_gather_hist_fwd(const float* item_tbl, const float* cate_tbl,
                                    const int* item_idx, const int* cate_idx, long idx_row_stride,
                                    const int* seq_len, int len_stride, int Hn, int T, int Di,
                                    int Dc, int recent_k, float* hist, float* hist_mean,
                                    float* hist_recent, void* stream) {
  CLSR_CHECK_ARG(item_tbl && cate_tbl && item_idx && cate_idx && seq_len && hist && hist_mean && hist_recent);
  CLSR_CHECK_ARG(Hn >= 0 && T > 0 && recent_k > 0);
  CLSR_CHECK_SUPPORTED(Di % 4 == 0 && Dc % 4 == 0 && Di > 0 && Dc > 0 && (Di + Dc) <= 256);
  if (Hn == 0) return CLSR_OK;
  const int tpar = 64 / ((Di + Dc) / 4);
  const int niter = clsr_cdiv(T, tpar > 0 ? tpar : 1);         // steps per lane
  const int per_chunk = clsr_cdiv(niter, clsr_cdiv(niter, 13));  // evenly filled chunks of at most 13
  const dim3 grid(clsr_cdiv(Hn, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define CLSR_GATHER(U)                                                                                        \
  hipLaunchKernelGGL(gather_hist_fwd_kernel<U>, grid, block, 0, s, item_tbl, cate_tbl, item_idx, cate_idx,   \
                     idx_row_stride, seq_len, len_stride, Hn, T, Di, Dc, recent_k, hist, hist_mean, hist_recent)
  if (per_chunk <= 4) CLSR_GATHER(4);
  else if (per_chunk <= 9) CLSR_GATHER(9);
  else CLSR_GATHER(13);
#undef CLSR_GATHER
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- bf16 embedding tables ("bf16 tables + activations").  The same gather with 16-byte pieces
// of EIGHT bf16 values: rows of 2 * D bytes (192 B + 64 B at the 100M-item catalogue), hist written as bf16 (a pure copy:
// 26 000 algorithmic bytes per 50-step history at D = 128) or widened to fp32 for consumers that read fp32; the masked
// mean / recent-k mean are accumulated in fp32 from the widened values, exactly what an fp32 gather of the widened
// table would give.
typedef __bf16 emb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emb_f32x8 __attribute__((ext_vector_type(8)));
template <int U, bool OUT_H>
__global__ void __launch_bounds__(256) gather_hist_fwd_h_kernel(
    const __bf16* __restrict__ item_tbl, const __bf16* __restrict__ cate_tbl,
    const int* __restrict__ item_idx, const int* __restrict__ cate_idx, long idx_row_stride,
    const int* __restrict__ seq_len, int len_stride, int Hn, int T, int Di, int Dc, int recent_k,
    void* __restrict__ hist_v, float* __restrict__ hist_mean, float* __restrict__ hist_recent) {
  __shared__ emb_f32x8 red[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + wave;
  const int D = Di + Dc, QD = D >> 3, QI = Di >> 3;
  const int tpar = 64 / QD;
  const int tslot = lane / QD, q = lane - tslot * QD;
  const bool active = (h < Hn) && (tslot < tpar);
  const emb_f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  emb_f32x8 msum = z8, rsum = z8;
  int len = 0;
  if (active) {
    len = seq_len[(long)h * len_stride];
    const int* ii = item_idx + (long)h * idx_row_stride;
    const int* ci = cate_idx + (long)h * idx_row_stride;
    const int rlo = len - recent_k;
    const bool is_item = q < QI;
    const int* idx = is_item ? ii : ci;
    const __bf16* tbl = is_item ? item_tbl + 8 * q : cate_tbl + 8 * (q - QI);
    const long C = is_item ? Di : Dc;
    for (int t0 = tslot; t0 < T; t0 += U * tpar) {
      int id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tpar;
        id[u] = idx[t < T ? t : tslot];
      }
      emb_bf16x8 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const emb_bf16x8*>(tbl + (long)id[u] * C);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tpar;
        if (t < T) {
          const emb_f32x8 w = __builtin_convertvector(v[u], emb_f32x8);
          const long o = ((long)h * T + t) * D + 8 * q;
          if (OUT_H) {
            *reinterpret_cast<emb_bf16x8*>(reinterpret_cast<__bf16*>(hist_v) + o) = v[u];
          } else {
            float* out = reinterpret_cast<float*>(hist_v) + o;
            st4(out, (f32x4){w[0], w[1], w[2], w[3]});
            st4(out + 4, (f32x4){w[4], w[5], w[6], w[7]});
          }
          if (t < len) {
            msum += w;
            if (t >= rlo) rsum += w;
          }
        }
      }
    }
  }
  red[wave][0][lane] = msum;
  red[wave][1][lane] = rsum;
  __syncthreads();
  if (active && tslot == 0) {
    for (int s = 1; s < tpar; ++s) {
      msum += red[wave][0][lane + s * QD];
      rsum += red[wave][1][lane + s * QD];
    }
    const float inv_len = 1.0f / (float)len;
    const float inv_rec = 1.0f / (float)(len < recent_k ? len : recent_k);
    float* pm = hist_mean + (long)h * D + 8 * q;
    float* pr = hist_recent + (long)h * D + 8 * q;
    st4(pm, (f32x4){msum[0], msum[1], msum[2], msum[3]} * inv_len);
    st4(pm + 4, (f32x4){msum[4], msum[5], msum[6], msum[7]} * inv_len);
    st4(pr, (f32x4){rsum[0], rsum[1], rsum[2], rsum[3]} * inv_rec);
    st4(pr + 4, (f32x4){rsum[4], rsum[5], rsum[6], rsum[7]} * inv_rec);
  }
}

// clsr_gather_hist_fwd with bf16 tables; hist is written as bf16 (hist_bf16 != 0) or fp32; Di, Dc multiples of 8
extern "C" int clsr_gather_hist_fwd_h(const void* item_tbl, const void* cate_tbl, const int* item_idx,
                                      const int* cate_idx, long idx_row_stride, const int* seq_len, int len_stride,
                                      int Hn, int T, int Di, int Dc, int recent_k, void* hist, int hist_bf16,
                                      float* hist_mean, float* hist_recent, void* stream) {
  CLSR_CHECK_ARG(item_tbl && cate_tbl && item_idx && cate_idx && seq_len && hist && hist_mean && hist_recent);
  CLSR_CHECK_ARG(Hn >= 0 && T > 0 && recent_k > 0);
  CLSR_CHECK_SUPPORTED(Di % 8 == 0 && Dc % 8 == 0 && Di > 0 && Dc > 0 && (Di + Dc) <= 512);
  CLSR_CHECK_SUPPORTED(((uintptr_t)item_tbl % 16) == 0 && ((uintptr_t)cate_tbl % 16) == 0 && ((uintptr_t)hist % 16) == 0);
  if (Hn == 0) return CLSR_OK;
  const int tpar = 64 / ((Di + Dc) / 8);
  const int niter = clsr_cdiv(T, tpar > 0 ? tpar : 1);
  const int per_chunk = clsr_cdiv(niter, clsr_cdiv(niter, 13));
  const dim3 grid(clsr_cdiv(Hn, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const __bf16* it = (const __bf16*)item_tbl;
  const __bf16* ct = (const __bf16*)cate_tbl;
#define CLSR_GATHER_H(U, OH)                                                                                     \
  hipLaunchKernelGGL((gather_hist_fwd_h_kernel<U, OH>), grid, block, 0, s, it, ct, item_idx, cate_idx,           \
                     idx_row_stride, seq_len, len_stride, Hn, T, Di, Dc, recent_k, hist, hist_mean, hist_recent)
  if (hist_bf16) {
    if (per_chunk <= 4) CLSR_GATHER_H(4, true);
    else if (per_chunk <= 9) CLSR_GATHER_H(9, true);
    else CLSR_GATHER_H(13, true);
  } else {
    if (per_chunk <= 4) CLSR_GATHER_H(4, false);
    else if (per_chunk <= 9) CLSR_GATHER_H(9, false);
    else CLSR_GATHER_H(13, false);
  }
#undef CLSR_GATHER_H
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[n, col0 : col0+C] = tbl[idx[n * idx_stride], :]   (C % 4 == 0)
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ tbl,
                                                          const int* __restrict__ idx,
                                                          long idx_stride, int N, int C,
                                                          float* __restrict__ out, int ldo, int col0) {
  const int QC = C >> 2;
  const long total = (long)N * QC;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / QC), q = (int)(e - (long)n * QC);
    const f32x4 v = ld4(tbl + (long)idx[(long)n * idx_stride] * C + 4 * q);
    st4(out + (long)n * ldo + col0 + 4 * q, v);
  }
}

extern "C" int clsr_gather_rows(const float* tbl, const int* idx, long idx_stride, int N, int C,
                                float* out, int ldo, int col0, void* stream) {
  CLSR_CHECK_ARG(tbl && idx && out && N >= 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0 && C > 0 && ldo % 4 == 0 && col0 % 4 == 0);
  if (N == 0) return CLSR_OK;
  const long total = (long)N * (C / 4);
  int blocks = clsr_cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tbl, idx,
                     idx_stride, N, C, out, ldo, col0);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// tbl_grad[idx[n*idx_stride], :] += src[n, col0 : col0+C];  sumsq += sum(src slice ^2)
// (the slice values are the reference's IndexedSlices values of this lookup site; their
//  squared norm feeds tf.clip_by_norm, models/base_model.py:290-296).  fp32 hardware atomics.
__global__ void __launch_bounds__(256) scatter_add_rows_kernel(
    const float* __restrict__ src, int lds_, int col0, const int* __restrict__ idx, long idx_stride,
    int N, int C, float* __restrict__ tbl_grad, double* __restrict__ sumsq) {
  const long total = (long)N * C;
  float local = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / C), c = (int)(e - (long)n * C);
    const float v = src[(long)n * lds_ + col0 + c];
    local += v * v;
    atomicAdd(tbl_grad + (long)idx[(long)n * idx_stride] * C + c, v);
  }
  if (sumsq) {  // one atomic per block (uniform branch: sumsq is a kernel argument)
    __shared__ double red[4];
    const double tot = block256_sum_d((double)local, red);
    if (threadIdx.x == 0) atomicAdd(sumsq, tot);
  }
}

extern "C" int clsr_scatter_add_rows(const float* src, int ld_src, int col0, const int* idx,
                                     long idx_stride, int N, int C, float* tbl_grad, double* sumsq,
                                     void* stream) {
  CLSR_CHECK_ARG(src && idx && tbl_grad && N >= 0 && C > 0);
  if (N == 0) return CLSR_OK;
  int blocks = clsr_cdiv((long)N * C, 256 * 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                     ld_src, col0, idx, idx_stride, N, C, tbl_grad, sumsq);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Backward of the history gather + hist_mean / hist_recent:
//   g[h,t,:] = dhist[h,t,:] + (t < len) * dmean[h,:] / len + recent(t) * drecent[h,:] / cnt
//   item_grad[item_idx[h,t], :] += g[:Di] ; cate_grad[cate_idx[h,t], :] += g[Di:]
// sumsq[0] / sumsq[1] receive sum(g^2) of the item / cate parts (clip norm of this site).
__global__ void __launch_bounds__(256) gather_hist_bwd_kernel(
    const float* __restrict__ dhist, const float* __restrict__ dmean,
    const float* __restrict__ drecent, const int* __restrict__ item_idx,
    const int* __restrict__ cate_idx, long idx_row_stride, const int* __restrict__ seq_len,
    int len_stride, int Hn, int T, int Di, int Dc, int recent_k, float* __restrict__ item_grad,
    float* __restrict__ cate_grad, double* __restrict__ sumsq) {
  // Row 0 (padding / out-of-vocabulary id) receives the gradient of EVERY padded step: its slices are
  // first summed per block in LDS (ds_add_f32) and reach the table as one atomic per column per block.
  extern __shared__ float row0_acc[];  // [D]
  const int D = Di + Dc;
  for (int d = threadIdx.x; d < D; d += blockDim.x) row0_acc[d] = 0.f;
  __syncthreads();
  const long total = (long)Hn * T * D;
  float s_item = 0.f, s_cate = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const long ht = e / D;
    const int t = (int)(ht % T);
    const int h = (int)(ht / T);
    const int len = seq_len[(long)h * len_stride];
    float g = dhist[e];
    if (t < len) {
      if (dmean) g += dmean[(long)h * D + d] / (float)len;
      if (drecent && t >= len - recent_k)
        g += drecent[(long)h * D + d] / (float)(len < recent_k ? len : recent_k);
    }
    if (d < Di) {
      s_item += g * g;
      const int id = item_idx[(long)h * idx_row_stride + t];
      if (id == 0) atomicAdd(row0_acc + d, g);
      else atomicAdd(item_grad + (long)id * Di + d, g);
    } else {
      s_cate += g * g;
      const int id = cate_idx[(long)h * idx_row_stride + t];
      if (id == 0) atomicAdd(row0_acc + d, g);
      else atomicAdd(cate_grad + (long)id * Dc + (d - Di), g);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float v = row0_acc[d];
    if (v != 0.f) {
      if (d < Di) atomicAdd(item_grad + d, v);
      else atomicAdd(cate_grad + (d - Di), v);
    }
  }
  if (sumsq) {
    __shared__ double red[2][4];
    const double ti = block256_sum_d((double)s_item, red[0]);
    const double tc = block256_sum_d((double)s_cate, red[1]);
    if (threadIdx.x == 0) {
      atomicAdd(sumsq + 0, ti);
      atomicAdd(sumsq + 1, tc);
    }
  }
}

extern "C" int clsr_gather_hist_bwd(const float* dhist, const float* dmean, const float* drecent,
                                    const int* item_idx, const int* cate_idx, long idx_row_stride,
                                    const int* seq_len, int len_stride, int Hn, int T, int Di, int Dc,
                                    int recent_k, float* item_grad, float* cate_grad, double* sumsq,
                                    void* stream) {
  CLSR_CHECK_ARG(dhist && item_idx && cate_idx && seq_len && item_grad && cate_grad);
  CLSR_CHECK_ARG(Hn >= 0 && T > 0 && Di > 0 && Dc > 0);
  if (Hn == 0) return CLSR_OK;
  int blocks = clsr_cdiv((long)Hn * T * (Di + Dc), 256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gather_hist_bwd_kernel, dim3(blocks), dim3(256), (Di + Dc) * sizeof(float),
                     (hipStream_t)stream, dhist,
                     dmean, drecent, item_idx, cate_idx, idx_row_stride, seq_len, len_stride, Hn, T,
                     Di, Dc, recent_k, item_grad, cate_grad, sumsq);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// flags[idx[r * row_stride + c]] = 1 for r < nrows, c < ncols: the "involved" (tf.unique) id sets
// of sequential_base_model.py:409-433 / clsr.py:118-127, kept as a dense byte map per table.
__global__ void mark_rows_kernel(const int* __restrict__ idx, long nrows, int ncols, long row_stride,
                                 unsigned char* __restrict__ flags) {
  const long n = nrows * ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long r = e / ncols;
    const int c = (int)(e - r * ncols);
    flags[idx[r * row_stride + c]] = 1;
  }
}

extern "C" int clsr_mark_rows(const int* idx, long nrows, int ncols, long row_stride,
                              unsigned char* flags, void* stream) {
  CLSR_CHECK_ARG(idx && flags && nrows >= 0 && ncols > 0);
  if (nrows == 0) return CLSR_OK;
  int blocks = clsr_cdiv(nrows * ncols, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mark_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, nrows,
                     ncols, row_stride, flags);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
