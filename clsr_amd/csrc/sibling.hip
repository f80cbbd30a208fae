// Kernels only the sibling models of the reference need (gfx950); everything else they run on is shared with CLSR.
//
//   SLi-Rec  long-term "A2SVD" attention  models/base_model.py:595-625 (_attention):
//            att_inputs = inputs . attention_mat;  logits = att_inputs . query;  w = softmax_T(logits)  -- NO mask:
//            the padded steps (embedding row 0) take part, as in the reference;  output = sum_t w[t] * inputs[t]
//            (the caller reduces over T, models/sequential/sli_rec.py:38-41).
//   DIN      hist_embedding_sum = sum_t mask[t] * hist[t]  (models/sequential/din.py:27) = hist_mean * len.
//
// All of it is history-level work (Hn rows, tens of microseconds): one wavefront per history, plain loops.
#include "common.h"

// LDS (floats): wl[T]
__global__ void __launch_bounds__(64) asvd_att_fwd_kernel(const float* __restrict__ ai, const float* __restrict__ query,
                                                         const float* __restrict__ x, long Hn, int T, int D,
                                                         float* __restrict__ wts, float* __restrict__ out) {
  extern __shared__ float wl[];
  const int lane = threadIdx.x;
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) {
      const float* ap = ai + (h * T + t) * D;
      float s = 0.f;
      for (int d = 0; d < D; ++d) s = fmaf(ap[d], query[d], s);
      wl[t] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float e = __expf(wl[t] - mx);
      wl[t] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int t = lane; t < T; t += 64) {
      const float w = wl[t] * inv;
      wl[t] = w;
      wts[h * T + t] = w;
    }
    __syncthreads();
    for (int d = lane; d < D; d += 64) {
      const float* xp = x + h * T * D + d;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(wl[t], xp[(long)t * D], acc);
      out[h * D + d] = acc;
    }
    __syncthreads();
  }
}

extern "C" int clsr_asvd_att_fwd(const float* att_inputs, const float* query, const float* inputs, long Hn, int T,
                                 int D, float* wts, float* out, void* stream) {
  CLSR_CHECK_ARG(att_inputs && query && inputs && wts && out && Hn >= 0 && T > 0 && D > 0);
  if (Hn == 0) return CLSR_OK;
  const int blocks = Hn > 4096 ? 4096 : (int)Hn;
  hipLaunchKernelGGL(asvd_att_fwd_kernel, dim3(blocks), dim3(64), T * sizeof(float), (hipStream_t)stream, att_inputs,
                     query, inputs, Hn, T, D, wts, out);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// d logits[t] = w[t] * (dw[t] - sum_t' w[t'] dw[t']) with dw[t] = dout . x[t];
// d att_inputs[t, :] = dl[t] * query;  dx[t, :] += w[t] * dout;  dquery (per-block partial) += sum_t dl[t] * ai[t, :]
// LDS (floats): dl[T] | wl[T] | dql[D]
__global__ void __launch_bounds__(64) asvd_att_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ wts,
                                                         const float* __restrict__ ai, const float* __restrict__ query,
                                                         const float* __restrict__ x, long Hn, int T, int D,
                                                         float* __restrict__ dai, float* __restrict__ dx,
                                                         float* __restrict__ dq_partial) {
  extern __shared__ float lds[];
  float* dl = lds;
  float* wl = lds + T;
  float* dql = lds + 2 * T;
  const int lane = threadIdx.x;
  for (int d = lane; d < D; d += 64) dql[d] = 0.f;
  __syncthreads();
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    const float* dop = dout + h * D;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float* xp = x + (h * T + t) * D;
      float dw = 0.f;
      for (int d = 0; d < D; ++d) dw = fmaf(dop[d], xp[d], dw);
      const float w = wts[h * T + t];
      wl[t] = w;
      dl[t] = dw;
      s = fmaf(w, dw, s);
    }
    s = wave_sum(s);
    for (int t = lane; t < T; t += 64) dl[t] = wl[t] * (dl[t] - s);
    __syncthreads();
    for (int d = lane; d < D; d += 64) {
      const float qd = query[d], dod = dop[d];
      float dq = 0.f;
      for (int t = 0; t < T; ++t) {
        const long o = (h * T + t) * D + d;
        dq = fmaf(dl[t], ai[o], dq);
        dai[o] = dl[t] * qd;
        dx[o] += wl[t] * dod;
      }
      dql[d] += dq;
    }
    __syncthreads();
  }
  for (int d = lane; d < D; d += 64) dq_partial[(long)blockIdx.x * D + d] = dql[d];
}

extern "C" int clsr_asvd_att_bwd_parts(long Hn) { return Hn > 1024 ? 1024 : (Hn > 0 ? (int)Hn : 1); }

extern "C" int clsr_asvd_att_bwd(const float* dout, const float* wts, const float* att_inputs, const float* query,
                                 const float* inputs, long Hn, int T, int D, float* d_att_inputs, float* d_inputs,
                                 float* dquery_partial, void* stream) {
  CLSR_CHECK_ARG(dout && wts && att_inputs && query && inputs && d_att_inputs && d_inputs && dquery_partial);
  CLSR_CHECK_ARG(Hn > 0 && T > 0 && D > 0);
  const int blocks = clsr_asvd_att_bwd_parts(Hn);
  hipLaunchKernelGGL(asvd_att_bwd_kernel, dim3(blocks), dim3(64), (2 * T + D) * sizeof(float), (hipStream_t)stream,
                     dout, wts, att_inputs, query, inputs, Hn, T, D, d_att_inputs, d_inputs, dquery_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[h, :] (=|+=) src[h, :] * seq_len[h]
__global__ void scale_rows_by_len_kernel(const float* __restrict__ src, const int* __restrict__ seq_len,
                                         int len_stride, long Hn, int C, float* __restrict__ out, int accumulate) {
  const long total = Hn * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long h = e / C;
    const int len = seq_len[h * len_stride];
    const float v = len > 0 ? src[e] * (float)len : 0.f;      // an empty history: the masked sum is 0 (mean is 0/0)
    out[e] = accumulate ? out[e] + v : v;
  }
}

extern "C" int clsr_scale_rows_by_len(const float* src, const int* seq_len, int len_stride, long Hn, int C,
                                      float* out, int accumulate, void* stream) {
  CLSR_CHECK_ARG(src && seq_len && out && Hn >= 0 && C > 0);
  if (Hn == 0) return CLSR_OK;
  int blocks = clsr_cdiv(Hn * C, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_rows_by_len_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, seq_len,
                     len_stride, Hn, C, out, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// DIEN uses the attention WEIGHTS of _attention_fcn (return_alpha=True, dien.py:47), not its weighted sum: the
// gradient arrives as d w[r, t].  Masked softmax backward  ds[t] = w[t] * (dw[t] - sum_t' w[t'] dw[t'])  for t < len,
// 0 past it; per-block partial sums of ds (= d b_nn_output) like clsr_att_score_bwd.  One wave per row.
__global__ void __launch_bounds__(64) softmax_weights_bwd_kernel(const float* __restrict__ dw,
                                                                const float* __restrict__ wts,
                                                                const int* __restrict__ seq_len, int len_stride,
                                                                long R, int G, int T, float* __restrict__ ds,
                                                                float* __restrict__ b_partial) {
  const int lane = threadIdx.x;
  float p_db = 0.f;
  for (long r = blockIdx.x; r < R; r += gridDim.x) {
    const int len = seq_len[(r / G) * len_stride];
    float dot = 0.f;
    for (int t = lane; t < T && t < len; t += 64) dot = fmaf(wts[r * T + t], dw[r * T + t], dot);
    dot = wave_sum(dot);
    for (int t = lane; t < T; t += 64) {
      const float v = t < len ? wts[r * T + t] * (dw[r * T + t] - dot) : 0.f;
      ds[r * T + t] = v;
      p_db += v;
    }
  }
  p_db = wave_sum(p_db);
  if (lane == 0) b_partial[blockIdx.x] = p_db;
}

extern "C" int clsr_softmax_weights_bwd_parts(long R) { return R > 4096 ? 4096 : (R > 0 ? (int)R : 1); }

extern "C" int clsr_softmax_weights_bwd(const float* dw, const float* wts, const int* seq_len, int len_stride,
                                        long Hn, int G, int T, float* ds, float* b_partial, void* stream) {
  CLSR_CHECK_ARG(dw && wts && seq_len && ds && b_partial && Hn > 0 && G > 0 && T > 0);
  const long R = Hn * G;
  hipLaunchKernelGGL(softmax_weights_bwd_kernel, dim3(clsr_softmax_weights_bwd_parts(R)), dim3(64), 0,
                     (hipStream_t)stream, dw, wts, seq_len, len_stride, R, G, T, ds, b_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[r, :] = a[r, :] * b[r / G, :]   (DIEN's target * hist_embedding_sum feature, dien.py:58); its backward is
// clsr_att_prod_bwd_ld with T = 1
__global__ void mul_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int G,
                                long R, int C, float* __restrict__ out, int ldo) {
  const long total = R * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C;
    const int c = (int)(e - r * C);
    out[r * ldo + c] = a[r * lda + c] * b[(r / G) * ldb + c];
  }
}

extern "C" int clsr_mul_rows(const float* a, int lda, const float* b, int ldb, int G, long R, int C, float* out,
                             int ldo, void* stream) {
  CLSR_CHECK_ARG(a && b && out && G > 0 && R >= 0 && C > 0);
  if (R == 0) return CLSR_OK;
  int blocks = clsr_cdiv(R * C, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mul_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, G, R, C, out, ldo);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
