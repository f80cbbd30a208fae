// Regularisers, per-tensor gradient clipping and Adam for the CLSR step (gfx950).
//
// Reference: models/base_model.py:118-159 (_l2_loss over "involved" embedding rows and every
// non-embedding trainable), models/sequential/clsr.py:73-82 (_compute_discrepancy_loss),
// base_model.py:281-297 (compute_gradients -> per-tensor tf.clip_by_norm -> apply_gradients),
// base_model.py:263-264 tf.train.AdamOptimizer(lr): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
// m <- b1 m + (1-b1) g, v <- b2 v + (1-b2) g^2, var <- var - lr_t m / (sqrt(v) + eps); for the
// embedding tables (IndexedSlices) TF decays m, v over the WHOLE table and updates every row,
// which is what the dense table sweep below does ("adam"); "lazyadam" (base_model.py:275-276)
// touches flagged rows only.
//
// Dense parameters live in ONE flat fp32 buffer (param, grad, m, v) with a per-element tensor id
// map, so the whole dense update is three launches regardless of the number of tensors.
#include <stdlib.h>
#include "common.h"
#include <cstdlib>

// state[0]=step, [1]=beta1^t, [2]=beta2^t, [3]=lr_t, [4]=abort flag of the step   (doubles, device resident so graph
// replay works; abort: raised by a collective / grid barrier that gave up -- csrc/p2p.hip -- the kernels below then return)
__global__ void adam_tick_kernel(double* st, double lr, double b1, double b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && st[4] == 0.0) {      // (an aborted step does not advance the Adam clock)
    st[0] += 1.0;
    st[1] *= b1;
    st[2] *= b2;
    st[3] = lr * sqrt(1.0 - st[2]) / (1.0 - st[1]);
  }
}

extern "C" int clsr_adam_tick(double* state, double lr, double beta1, double beta2, void* stream) {
  CLSR_CHECK_ARG(state);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// One block per dense tensor: grad += l2 * param + l1 * sign(param); sumsq[tensor] = ||grad||^2;
// reg_loss += l2 * 0.5 * ||param||^2 + l1 * |param|_1   (base_model.py:118-147: _l2_loss / _l1_loss of the layer params)
__global__ void __launch_bounds__(1024) dense_reg_norm_kernel(const float* __restrict__ param,
                                                             float* __restrict__ grad,
                                                             const int* __restrict__ seg_off, float l2, float l1,
                                                             double* __restrict__ sumsq,
                                                             double* __restrict__ reg_loss,
                                                             double* __restrict__ adam_state, double lr, double b1,
                                                             double b2) {
  if (adam_state && blockIdx.x == 0 && threadIdx.x == 0 && adam_state[4] == 0.0) {   // the Adam clock of this step (adam_tick_kernel) rides along; not in an aborted step
    adam_state[0] += 1.0;
    adam_state[1] *= b1;
    adam_state[2] *= b2;
    adam_state[3] = lr * sqrt(1.0 - adam_state[2]) / (1.0 - adam_state[1]);
  }
  // 1024 threads: the largest tensor (25.6 k elements) is 25 dependent iterations instead of 100; the 16 wave
  // partials are combined in a fixed order (deterministic)
  __shared__ double red[2][16];
  const int seg = blockIdx.x;
  const int lo = seg_off[seg], hi = seg_off[seg + 1];
  double ss = 0.0, pp = 0.0;
  // (eight strides of loads in flight from clamped addresses: the read-modify-write of one stride after the other was a
  // chain of dependent round trips -- 47 us for the 25.6 k-element tensor, one workgroup per CU; same per-thread order)
  const int BS = blockDim.x;
  for (int e0 = lo + threadIdx.x; e0 < hi; e0 += 8 * BS) {
    float pv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * BS, ec = e < hi ? e : lo;
      pv[u] = param[ec];
      gv[u] = grad[ec];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * BS;
      if (e < hi) {
        const float p = pv[u];
        const float g = gv[u] + l2 * p + l1 * (float)((p > 0.f) - (p < 0.f));
        grad[e] = g;
        ss += (double)g * g;
        pp += 0.5 * (double)l2 * p * p + (double)l1 * fabsf(p);
      }
    }
  }
  ss = wave_sum_d(ss);
  pp = wave_sum_d(pp);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ss; red[1][threadIdx.x >> 6] = pp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, r = 0.0;
    for (int w = 0; w < BS / 64; ++w) { a += red[0][w]; r += red[1][w]; }
    sumsq[seg] = a;
    if (reg_loss) atomicAdd(reg_loss, r);
  }
}

static int dense_reg_norm_any(const float* param, float* grad, const int* seg_off, int nseg,
                              float l2, float l1, double* sumsq, double* reg_loss, double* adam_state,
                              double lr, double beta1, double beta2, int threads_arg, void* stream) {
  CLSR_CHECK_ARG(param && grad && seg_off && sumsq && nseg > 0);
  // (workgroups of 1 024 threads wait for a CU with sixteen free wave slots while the table sweeps of the other stream fill
  // the chip: CLSR_DENSE_REG_THREADS picks the size, A/B)
  constexpr int threads = 256;
  const int th = (threads_arg == 256 || threads_arg == 512 || threads_arg == 1024) ? threads_arg : threads;
  hipLaunchKernelGGL(dense_reg_norm_kernel, dim3(nseg), dim3(th), 0, (hipStream_t)stream, param, grad,
                     seg_off, l2, l1, sumsq, reg_loss, adam_state, lr, beta1, beta2);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_dense_reg_norm_tick(const float* param, float* grad, const int* seg_off, int nseg,
                                        float l2, float l1, double* sumsq, double* reg_loss, double* adam_state,
                                        double lr, double beta1, double beta2, void* stream) {
  return dense_reg_norm_any(param, grad, seg_off, nseg, l2, l1, sumsq, reg_loss, adam_state, lr, beta1, beta2, 0, stream);
}
// ... with the workgroup size chosen by the caller (256 | 512 | 1024; 0: the default): one workgroup walks one tensor, and the
// 128 x 1 536 kernels of 128-wide encoders are 768 elements per thread at 256 threads (109 us, the last kernel of that step)
extern "C" int clsr_dense_reg_norm_tick_t(const float* param, float* grad, const int* seg_off, int nseg,
                                          float l2, float l1, double* sumsq, double* reg_loss, double* adam_state,
                                          double lr, double beta1, double beta2, int threads, void* stream) {
  return dense_reg_norm_any(param, grad, seg_off, nseg, l2, l1, sumsq, reg_loss, adam_state, lr, beta1, beta2, threads, stream);
}

extern "C" int clsr_dense_reg_norm(const float* param, float* grad, const int* seg_off, int nseg,
                                   float l2, float l1, double* sumsq, double* reg_loss, void* stream) {
  return clsr_dense_reg_norm_tick(param, grad, seg_off, nseg, l2, l1, sumsq, reg_loss, nullptr, 0.0, 0.0, 0.0, stream);
}

__device__ __forceinline__ float clip_factor(double sumsq, float clip_norm) {
  if (clip_norm <= 0.f) return 1.0f;
  const float nrm = (float)sqrt(sumsq);
  return clip_norm / fmaxf(nrm, clip_norm);
}

// Flat dense Adam: g = grad[e] * clip_factor(sumsq[seg_of[e]]); grad is zeroed for the next step.
__global__ void __launch_bounds__(256) dense_adam_kernel(float* __restrict__ param, float* __restrict__ grad,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const int* __restrict__ seg_of,
                                                         const double* __restrict__ sumsq, float clip_norm,
                                                         const double* __restrict__ adam_state, float b1,
                                                         float b2, float eps, int n) {
  if (adam_state[4] != 0.0) return;      // the step was aborted (a collective / grid barrier gave up): touch nothing
  const float lr_t = (float)adam_state[3];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const float g = grad[e] * clip_factor(sumsq[seg_of[e]], clip_norm);
    const float mm = b1 * m[e] + (1.0f - b1) * g;
    const float vv = b2 * v[e] + (1.0f - b2) * g * g;
    m[e] = mm;
    v[e] = vv;
    param[e] -= lr_t * mm / (sqrtf(vv) + eps);
    grad[e] = 0.f;
  }
}

extern "C" int clsr_dense_adam(float* param, float* grad, float* m, float* v, const int* seg_of,
                               const double* sumsq, float clip_norm, const double* adam_state,
                               float beta1, float beta2, float eps, int n, void* stream) {
  CLSR_CHECK_ARG(param && grad && m && v && seg_of && sumsq && adam_state && n > 0);
  int blocks = clsr_cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dense_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                     seg_of, sumsq, clip_norm, adam_state, beta1, beta2, eps, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---------------------------------------------------------------------------- embedding tables
// count[0] = number of set flags (float, consumed by the discrepancy coefficient)
__global__ void count_flags_kernel(const unsigned char* __restrict__ flags, long V, float* __restrict__ count,
                                   double* __restrict__ adam_state, double lr, double b1, double b2) {
  if (adam_state && blockIdx.x == 0 && threadIdx.x == 0 && adam_state[4] == 0.0) {   // the Adam clock of the step (adam_tick_kernel) rides along; not in an aborted step
    adam_state[0] += 1.0;
    adam_state[1] *= b1;
    adam_state[2] *= b2;
    adam_state[3] = lr * sqrt(1.0 - adam_state[2]) / (1.0 - adam_state[1]);
  }
  float local = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < V; e += (long)gridDim.x * blockDim.x)
    local += flags[e] ? 1.f : 0.f;
  __shared__ double red[4];
  const double tot = block256_sum_d((double)local, red);
  if (threadIdx.x == 0 && tot != 0.0) atomicAdd(count, (float)tot);
}

extern "C" int clsr_count_flags_tick(const unsigned char* flags, long V, float* count, double* adam_state, double lr,
                                     double beta1, double beta2, void* stream) {
  CLSR_CHECK_ARG(flags && count && V > 0);
  int blocks = clsr_cdiv(V, 256 * 16);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flags, V, count, adam_state,
                     lr, beta1, beta2);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_count_flags(const unsigned char* flags, long V, float* count, void* stream) {
  return clsr_count_flags_tick(flags, V, count, nullptr, 0.0, 0.0, 0.0, stream);
}

// Pass A over the involved (flagged) rows of one table:
//   g_reg = l2 * row + cd * (row - partner_row)        cd = disc_scale / (count * C)  (user tables)
//   grad_table[row] += g_reg ; sumsq += ||g_reg||^2 ; reg_loss += l2/2 ||row||^2 ;
//   disc_loss += disc_loss_scale * ||row - partner_row||^2 / (count * C)   (only when disc_loss != null)
// These are the values of the reference's third IndexedSlices (the tf.unique "involved" lookup).
template <bool H>
__global__ void __launch_bounds__(256) table_reg_kernel(
    const void* __restrict__ table, const void* __restrict__ partner,
    const unsigned char* __restrict__ flags, long V, int C, float l2, float l1, float disc_scale,
    float disc_loss_scale, const float* __restrict__ count, float* __restrict__ grad_table,
    double* __restrict__ sumsq, double* __restrict__ reg_loss, double* __restrict__ disc_loss) {
  const float cd = partner ? disc_scale / (count[0] * (float)C) : 0.f;
  const float cl = (partner && disc_loss) ? disc_loss_scale / (count[0] * (float)C) : 0.f;
  double ss = 0.0, rl = 0.0, dl = 0.0;
  const long total = V * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / C;
    if (!flags[row]) continue;
    const float p = tbl_ld<H>(table, e);
    float g = l2 * p + l1 * (float)((p > 0.f) - (p < 0.f));
    if (partner) {
      const float d = p - tbl_ld<H>(partner, e);
      g += cd * d;
      dl += (double)d * d;
    }
    grad_table[e] += g;
    ss += (double)g * g;
    rl += 0.5 * (double)l2 * p * p + (double)l1 * fabsf(p);
  }
  __shared__ double red[3][4];
  ss = block256_sum_d(ss, red[0]); rl = block256_sum_d(rl, red[1]); dl = block256_sum_d(dl, red[2]);
  if (threadIdx.x == 0) {
    if (ss != 0.0) atomicAdd(sumsq, ss);
    if (reg_loss && rl != 0.0) atomicAdd(reg_loss, rl);
    if (disc_loss && dl != 0.0) atomicAdd(disc_loss, (double)cl * dl);
  }
}

static int table_reg_launch(const void* table, const void* partner, int bf16, const unsigned char* flags, long V, int C,
                            float l2, float l1, float disc_scale, float disc_loss_scale, const float* count,
                            float* grad_table, double* sumsq, double* reg_loss, double* disc_loss, void* stream) {
  CLSR_CHECK_ARG(table && flags && grad_table && sumsq && V > 0 && C > 0);
  CLSR_CHECK_ARG(!partner || count);
  int blocks = clsr_cdiv(V * C, 256 * 8);
  if (blocks > 512) blocks = 512;
  if (bf16)
    hipLaunchKernelGGL(table_reg_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, partner, flags, V, C,
                       l2, l1, disc_scale, disc_loss_scale, count, grad_table, sumsq, reg_loss, disc_loss);
  else
    hipLaunchKernelGGL(table_reg_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, partner, flags, V, C,
                       l2, l1, disc_scale, disc_loss_scale, count, grad_table, sumsq, reg_loss, disc_loss);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_table_reg(const float* table, const float* partner, const unsigned char* flags,
                              long V, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                              const float* count, float* grad_table, double* sumsq, double* reg_loss,
                              double* disc_loss, void* stream) {
  return table_reg_launch(table, partner, 0, flags, V, C, l2, l1, disc_scale, disc_loss_scale, count, grad_table, sumsq,
                          reg_loss, disc_loss, stream);
}
// bf16 tables (table / partner: bf16 [V, C]); gradients, norms and losses fp32 / fp64 as above
extern "C" int clsr_table_reg_h(const void* table, const void* partner, const unsigned char* flags,
                                long V, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                                const float* count, float* grad_table, double* sumsq, double* reg_loss,
                                double* disc_loss, void* stream) {
  return table_reg_launch(table, partner, 1, flags, V, C, l2, l1, disc_scale, disc_loss_scale, count, grad_table, sumsq,
                          reg_loss, disc_loss, stream);
}

// Pass B: Adam sweep of one table.  sumsq[i * sumsq_stride], i < nsum, are the squared norms of this table's
// IndexedSlices pieces (lookup sites + involved rows); lazy != 0 restricts the update to rows whose
// flag is set or that received gradient through a lookup (LazyAdam).  Clears grad rows and flags.
template <bool H>
__global__ void __launch_bounds__(256) table_adam_kernel(
    void* __restrict__ table, float* __restrict__ grad_table, float* __restrict__ m,
    float* __restrict__ v, unsigned char* __restrict__ flags, long V, int C,
    const double* __restrict__ sumsq, int sumsq_stride, int nsum, float clip_norm,
    const double* __restrict__ adam_state, float b1, float b2, float eps, int lazy) {
  double tot = 0.0;
  for (int i = 0; i < nsum; ++i) tot += sumsq[(long)i * sumsq_stride];
  const float factor = clip_factor(tot, clip_norm);
  if (adam_state[4] != 0.0) return;      // the step was aborted (a collective / grid barrier gave up): touch nothing
  const float lr_t = (float)adam_state[3];
  const long total = V * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / C;
    const float g = grad_table[e] * factor;
    if (lazy && !flags[row]) continue;  // every row that got gradient is also flagged as involved
    const float mm = b1 * m[e] + (1.0f - b1) * g;
    const float vv = b2 * v[e] + (1.0f - b2) * g * g;
    m[e] = mm;
    v[e] = vv;
    tbl_st<H>(table, e, tbl_ld<H>(table, e) - lr_t * mm / (sqrtf(vv) + eps));
    grad_table[e] = 0.f;
  }
}

__global__ void clear_bytes_kernel(unsigned char* p, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) p[e] = 0;
}

static int table_adam_launch(void* table, int bf16, float* grad_table, float* m, float* v, unsigned char* flags, long V,
                             int C, const double* sumsq, int sumsq_stride, int nsum, float clip_norm,
                             const double* adam_state, float beta1, float beta2, float eps, int lazy, void* stream) {
  CLSR_CHECK_ARG(table && grad_table && m && v && flags && sumsq && adam_state && V > 0 && C > 0 && nsum > 0);
  int blocks = clsr_cdiv(V * C, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (bf16)
    hipLaunchKernelGGL(table_adam_kernel<true>, dim3(blocks), dim3(256), 0, s, table, grad_table, m, v, flags, V, C,
                       sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1, beta2, eps, lazy);
  else
    hipLaunchKernelGGL(table_adam_kernel<false>, dim3(blocks), dim3(256), 0, s, table, grad_table, m, v, flags, V, C,
                       sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1, beta2, eps, lazy);
  CLSR_CHECK_LAUNCH();
  int cb = clsr_cdiv(V, 256);
  if (cb > 1024) cb = 1024;
  hipLaunchKernelGGL(clear_bytes_kernel, dim3(cb), dim3(256), 0, s, flags, V);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_table_adam(float* table, float* grad_table, float* m, float* v, unsigned char* flags,
                               long V, int C, const double* sumsq, int sumsq_stride, int nsum,
                               float clip_norm, const double* adam_state, float beta1, float beta2,
                               float eps, int lazy, void* stream) {
  return table_adam_launch(table, 0, grad_table, m, v, flags, V, C, sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1,
                           beta2, eps, lazy, stream);
}
// bf16 table: widened, updated in fp32 (fp32 moments / gradients), rounded to nearest-even
extern "C" int clsr_table_adam_h(void* table_bf16, float* grad_table, float* m, float* v, unsigned char* flags,
                                 long V, int C, const double* sumsq, int sumsq_stride, int nsum,
                                 float clip_norm, const double* adam_state, float beta1, float beta2,
                                 float eps, int lazy, void* stream) {
  return table_adam_launch(table_bf16, 1, grad_table, m, v, flags, V, C, sumsq, sumsq_stride, nsum, clip_norm, adam_state,
                           beta1, beta2, eps, lazy, stream);
}

// ---- row-list variants for huge vocabularies (100M-item catalogues): the same math as table_reg / lazy
// table_adam, but driven by the compacted id list of the involved rows (clsr_flags_compact) instead of a sweep
// over all V*C elements.  count[0] = number of listed rows (device memory).
// VW = floats per lane and access (4 when C % 4 == 0: 16-byte pieces of the 128-512 B rows, 4x fewer dependent
// id loads and address computations; random rows are HBM latency bound, so bytes in flight per lane matter)
template <int VW, bool H>
__global__ void __launch_bounds__(256) table_reg_rows_kernel(
    const void* __restrict__ table, const void* __restrict__ partner, const int* __restrict__ ids,
    const int* __restrict__ count, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
    const float* __restrict__ ucount, float* __restrict__ grad_table, double* __restrict__ sumsq,
    double* __restrict__ reg_loss, double* __restrict__ disc_loss) {
  const float cd = partner ? disc_scale / (ucount[0] * (float)C) : 0.f;
  const float cl = (partner && disc_loss) ? disc_loss_scale / (ucount[0] * (float)C) : 0.f;
  double ss = 0.0, rl = 0.0, dl = 0.0;
  const int QC = C / VW;
  const long total = (long)count[0] * QC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / QC;
    const long e = (long)ids[r] * C + (i - r * QC) * VW;
    float p[VW], pp[VW], gt[VW];
    if (VW == 4) {
      *reinterpret_cast<f32x4*>(p) = load4e<H>(table, e);
      if (partner) *reinterpret_cast<f32x4*>(pp) = load4e<H>(partner, e);
      *reinterpret_cast<f32x4*>(gt) = ld4(grad_table + e);
    } else {
      p[0] = tbl_ld<H>(table, e);
      if (partner) pp[0] = tbl_ld<H>(partner, e);
      gt[0] = grad_table[e];
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      float g = l2 * p[k] + l1 * (float)((p[k] > 0.f) - (p[k] < 0.f));
      if (partner) {
        const float d = p[k] - pp[k];
        g += cd * d;
        dl += (double)d * d;
      }
      gt[k] += g;
      ss += (double)g * g;
      rl += 0.5 * (double)l2 * p[k] * p[k] + (double)l1 * fabsf(p[k]);
    }
    if (VW == 4) st4(grad_table + e, *reinterpret_cast<f32x4*>(gt));
    else grad_table[e] = gt[0];
  }
  __shared__ double red[3][4];
  ss = block256_sum_d(ss, red[0]); rl = block256_sum_d(rl, red[1]); dl = block256_sum_d(dl, red[2]);
  if (threadIdx.x == 0) {
    if (ss != 0.0) atomicAdd(sumsq, ss);
    if (reg_loss && rl != 0.0) atomicAdd(reg_loss, rl);
    if (disc_loss && dl != 0.0) atomicAdd(disc_loss, (double)cl * dl);
  }
}

static int table_reg_rows_launch(const void* table, const void* partner, int bf16, const int* ids, const int* count, int cap,
                                 int C, float l2, float l1, float disc_scale, float disc_loss_scale, const float* ucount,
                                 float* grad_table, double* sumsq, double* reg_loss, double* disc_loss, void* stream) {
  CLSR_CHECK_ARG(table && ids && count && grad_table && sumsq && cap > 0 && C > 0);
  CLSR_CHECK_ARG(!partner || ucount);
  const bool vec = C % 4 == 0;
  int blocks = clsr_cdiv((long)cap * C, 256 * 4 * (vec ? 2 : 1));
  if (blocks > 2048) blocks = 2048;
#define TRR(VWV, HV)                                                                                                  \
  hipLaunchKernelGGL((table_reg_rows_kernel<VWV, HV>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, partner, \
                     ids, count, C, l2, l1, disc_scale, disc_loss_scale, ucount, grad_table, sumsq, reg_loss, disc_loss)
  if (vec) { if (bf16) TRR(4, true); else TRR(4, false); }
  else { if (bf16) TRR(1, true); else TRR(1, false); }
#undef TRR
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_table_reg_rows(const float* table, const float* partner, const int* ids, const int* count,
                                   int cap, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                                   const float* ucount, float* grad_table, double* sumsq, double* reg_loss,
                                   double* disc_loss, void* stream) {
  return table_reg_rows_launch(table, partner, 0, ids, count, cap, C, l2, l1, disc_scale, disc_loss_scale, ucount, grad_table,
                               sumsq, reg_loss, disc_loss, stream);
}
extern "C" int clsr_table_reg_rows_h(const void* table, const void* partner, const int* ids, const int* count,
                                     int cap, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                                     const float* ucount, float* grad_table, double* sumsq, double* reg_loss,
                                     double* disc_loss, void* stream) {
  return table_reg_rows_launch(table, partner, 1, ids, count, cap, C, l2, l1, disc_scale, disc_loss_scale, ucount, grad_table,
                               sumsq, reg_loss, disc_loss, stream);
}

// LazyAdam over the listed rows; clears their gradient rows and flags.
// UN pieces per lane and trip: their 4 * UN loads are issued before the first one is used (random 384-byte rows of a
// 38 GB table: the update is bound by how many row reads are in flight; UN = 1 ran at 4.4-4.7 TB/s)
template <int VW, int UN>
__global__ void __launch_bounds__(256) table_adam_rows_kernel(
    float* __restrict__ table, float* __restrict__ grad_table, float* __restrict__ m, float* __restrict__ v,
    unsigned char* __restrict__ flags, const int* __restrict__ ids, const int* __restrict__ count, int C,
    const double* __restrict__ sumsq, int sumsq_stride, int nsum, float clip_norm,
    const double* __restrict__ adam_state, float b1, float b2, float eps) {
  double tot = 0.0;
  for (int i = 0; i < nsum; ++i) tot += sumsq[(long)i * sumsq_stride];
  const float factor = clip_factor(tot, clip_norm);
  if (adam_state[4] != 0.0) return;      // the step was aborted (a collective / grid barrier gave up): touch nothing
  const float lr_t = (float)adam_state[3];
  const int QC = C / VW;
  const long total = (long)count[0] * QC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * UN) {
    long e[UN], row[UN];
    int q[UN];
    bool ok[UN];
    float g[UN][VW], mo[UN][VW], vo[UN][VW], po[UN][VW];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = i0 + u * stride;
      ok[u] = i < total;
      const long ic = ok[u] ? i : i0;
      const long r = ic / QC;
      q[u] = (int)(ic - r * QC);
      row[u] = ids[r];
      e[u] = row[u] * C + (long)q[u] * VW;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (VW == 4) {
        *reinterpret_cast<f32x4*>(g[u]) = ld4(grad_table + e[u]);
        *reinterpret_cast<f32x4*>(mo[u]) = ld4(m + e[u]);
        *reinterpret_cast<f32x4*>(vo[u]) = ld4(v + e[u]);
        *reinterpret_cast<f32x4*>(po[u]) = ld4(table + e[u]);
      } else {
        g[u][0] = grad_table[e[u]]; mo[u][0] = m[e[u]]; vo[u][0] = v[e[u]]; po[u][0] = table[e[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const float gg = g[u][k] * factor;
        const float mm = b1 * mo[u][k] + (1.0f - b1) * gg;
        const float vv = b2 * vo[u][k] + (1.0f - b2) * gg * gg;
        mo[u][k] = mm;
        vo[u][k] = vv;
        po[u][k] -= lr_t * mm / (sqrtf(vv) + eps);
      }
      if (ok[u]) {
        if (VW == 4) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          st4(m + e[u], *reinterpret_cast<f32x4*>(mo[u]));
          st4(v + e[u], *reinterpret_cast<f32x4*>(vo[u]));
          st4(table + e[u], *reinterpret_cast<f32x4*>(po[u]));
          st4(grad_table + e[u], z);
        } else {
          m[e[u]] = mo[u][0]; v[e[u]] = vo[u][0]; table[e[u]] = po[u][0]; grad_table[e[u]] = 0.f;
        }
        if (q[u] == 0) flags[row[u]] = 0;
      }
    }
  }
}

// The same update for a table stored as bf16 (SURVEY 8d "bf16 tables"): the row is widened, updated in fp32 with fp32
// moments and gradients, and rounded to nearest-even when written back (6 + 2 x 2 bytes per element instead of 8 x 4:
// 28 against 32).  Without an fp32 master copy an update smaller than half a bf16 ulp of the weight (2^-9 relative) is
// lost; the parity test pins exactly this arithmetic.
template <int UN>
__global__ void __launch_bounds__(256) table_adam_rows_h_kernel(
    __bf16* __restrict__ table, float* __restrict__ grad_table, float* __restrict__ m, float* __restrict__ v,
    unsigned char* __restrict__ flags, const int* __restrict__ ids, const int* __restrict__ count, int C,
    const double* __restrict__ sumsq, int sumsq_stride, int nsum, float clip_norm,
    const double* __restrict__ adam_state, float b1, float b2, float eps) {
  double tot = 0.0;
  for (int i = 0; i < nsum; ++i) tot += sumsq[(long)i * sumsq_stride];
  const float factor = clip_factor(tot, clip_norm);
  if (adam_state[4] != 0.0) return;      // the step was aborted (a collective / grid barrier gave up): touch nothing
  const float lr_t = (float)adam_state[3];
  const int QC = C / 4;
  const long total = (long)count[0] * QC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * UN) {
    long e[UN], row[UN];
    int q[UN];
    bool ok[UN];
    f32x4 g[UN], mo[UN], vo[UN];
    bf16x4_t ph[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = i0 + u * stride;
      ok[u] = i < total;
      const long ic = ok[u] ? i : i0;
      const long r = ic / QC;
      q[u] = (int)(ic - r * QC);
      row[u] = ids[r];
      e[u] = row[u] * C + (long)q[u] * 4;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      g[u] = ld4(grad_table + e[u]);
      mo[u] = ld4(m + e[u]);
      vo[u] = ld4(v + e[u]);
      ph[u] = *reinterpret_cast<const bf16x4_t*>(table + e[u]);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      f32x4 po = __builtin_convertvector(ph[u], f32x4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = g[u][k] * factor;
        const float mm = b1 * mo[u][k] + (1.0f - b1) * gg;
        const float vv = b2 * vo[u][k] + (1.0f - b2) * gg * gg;
        mo[u][k] = mm;
        vo[u][k] = vv;
        po[k] -= lr_t * mm / (sqrtf(vv) + eps);
      }
      if (ok[u]) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        st4(m + e[u], mo[u]);
        st4(v + e[u], vo[u]);
        *reinterpret_cast<bf16x4_t*>(table + e[u]) = __builtin_convertvector(po, bf16x4_t);
        st4(grad_table + e[u], z);
        if (q[u] == 0) flags[row[u]] = 0;
      }
    }
  }
}

extern "C" int clsr_table_adam_rows_h(void* table_bf16, float* grad_table, float* m, float* v, unsigned char* flags,
                                      const int* ids, const int* count, int cap, int C, const double* sumsq,
                                      int sumsq_stride, int nsum, float clip_norm, const double* adam_state,
                                      float beta1, float beta2, float eps, void* stream) {
  CLSR_CHECK_ARG(table_bf16 && grad_table && m && v && flags && ids && count && sumsq && adam_state && cap > 0 && C > 0);
  CLSR_CHECK_ARG(nsum > 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0 && ((uintptr_t)table_bf16 % 8) == 0);
  int blocks = clsr_cdiv((long)cap * C, 256 * 4 * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(table_adam_rows_h_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (__bf16*)table_bf16,
                     grad_table, m, v, flags, ids, count, C, sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1, beta2,
                     eps);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_table_adam_rows(float* table, float* grad_table, float* m, float* v, unsigned char* flags,
                                    const int* ids, const int* count, int cap, int C, const double* sumsq,
                                    int sumsq_stride, int nsum, float clip_norm, const double* adam_state,
                                    float beta1, float beta2, float eps, void* stream) {
  CLSR_CHECK_ARG(table && grad_table && m && v && flags && ids && count && sumsq && adam_state && cap > 0 && C > 0);
  CLSR_CHECK_ARG(nsum > 0);
  const bool vec = C % 4 == 0;
  int blocks = clsr_cdiv((long)cap * C, 256 * 4 * (vec ? 2 : 1));
  constexpr int cap_blocks = 4096;
  if (blocks > cap_blocks) blocks = cap_blocks;
  if (vec)
    hipLaunchKernelGGL((table_adam_rows_kernel<4, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, grad_table,
                       m, v, flags, ids, count, C, sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1, beta2, eps);
  else
    hipLaunchKernelGGL((table_adam_rows_kernel<1, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, grad_table,
                       m, v, flags, ids, count, C, sumsq, sumsq_stride, nsum, clip_norm, adam_state, beta1, beta2, eps);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// p[0..n) = 0 (doubles)
__global__ void zero_d_kernel(double* p, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = 0.0;
}
extern "C" int clsr_zero_doubles(double* p, int n, void* stream) {
  CLSR_CHECK_ARG(p && n > 0);
  hipLaunchKernelGGL(zero_d_kernel, dim3(clsr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

__global__ void zero_f_kernel(float* p, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) p[e] = 0.f;
}
extern "C" int clsr_zero_floats(float* p, long n, void* stream) {
  CLSR_CHECK_ARG(p && n > 0);
  int blocks = clsr_cdiv(n, 1024);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_f_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// dst[i] += src[i] (doubles): running sums of the per-step loss terms kept on the device so that an
// epoch loop does not have to synchronise after every step
__global__ void add_d_kernel(double* dst, const double* src, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] += src[e];
}
extern "C" int clsr_add_doubles(double* dst, const double* src, int n, void* stream) {
  CLSR_CHECK_ARG(dst && src && n > 0);
  hipLaunchKernelGGL(add_d_kernel, dim3(clsr_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, dst, src, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// dst[0..n16) = src[0..n16) in 16-byte words.  ``src`` is PINNED HOST memory (hipHostMalloc: mapped into
// the device address space), read by the kernel straight over PCIe: the feed upload becomes an ordinary
// kernel launch on the step's stream (one launch for the whole feed instead of ten hipMemcpyAsync).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void stage_words_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long n16) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (long)gridDim.x * blockDim.x)
    dst[e] = __builtin_nontemporal_load(src + e);
}
extern "C" int clsr_stage_feed(void* dst, const void* src_host, long nbytes, void* stream) {
  CLSR_CHECK_ARG(dst && src_host && nbytes > 0 && nbytes % 16 == 0);
  CLSR_CHECK_ARG(((uintptr_t)dst % 16) == 0 && ((uintptr_t)src_host % 16) == 0);
  const long n16 = nbytes / 16;
  int blocks = clsr_cdiv(n16, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(stage_words_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)dst,
                     (const u32x4*)src_host, n16);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Device-to-device copy in 16-byte words, four independent loads in flight per lane, streamed past the caches: the
// bandwidth probe of bench.py ("measured-peak ... on the same GPU" next to the 8 TB/s datasheet figure) --
// the float4 copy the MI355X guide quotes at 6.29 TB/s.  Not on the training path.
template <bool NT, int U>
__global__ void __launch_bounds__(256) copy_words_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long n16) {
  const long stride = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  for (; e + (U - 1) * stride < n16; e += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + e + u * stride) : src[e + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + e + u * stride);
      else dst[e + u * stride] = v[u];
    }
  }
  for (; e < n16; e += stride) dst[e] = src[e];
}
extern "C" int clsr_copy_words(void* dst, const void* src, long nbytes, void* stream) {
  CLSR_CHECK_ARG(dst && src && nbytes > 0 && nbytes % 16 == 0);
  CLSR_CHECK_ARG(((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0);
  const long n16 = nbytes / 16;
  // (probe variants, for the record: CLSR_COPY_NT = 0 | 1, CLSR_COPY_U = 4 | 8, CLSR_COPY_BLOCKS)
  constexpr int nt = 1, un = 4;
  constexpr long cap = 1L << 20;   // (swept: 2 Ki .. 1 Mi blocks, 64 MB .. 4 GB, plain / non-temporal, 4 / 8 in flight: scripts/copy_sweep.py)
  long blocks = (n16 + 256L * un - 1) / (256L * un);
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  if (nt && un == 8) hipLaunchKernelGGL((copy_words_kernel<true, 8>), dim3((int)blocks), dim3(256), 0, st, (u32x4*)dst, (const u32x4*)src, n16);
  else if (nt) hipLaunchKernelGGL((copy_words_kernel<true, 4>), dim3((int)blocks), dim3(256), 0, st, (u32x4*)dst, (const u32x4*)src, n16);
  else if (un == 8) hipLaunchKernelGGL((copy_words_kernel<false, 8>), dim3((int)blocks), dim3(256), 0, st, (u32x4*)dst, (const u32x4*)src, n16);
  else hipLaunchKernelGGL((copy_words_kernel<false, 4>), dim3((int)blocks), dim3(256), 0, st, (u32x4*)dst, (const u32x4*)src, n16);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
