// Batch-normalisation pieces of _fcn_net (gfx950).
//
// Reference: tf.layers.batch_normalization(momentum=0.95, epsilon=1e-4, training=is_train_stage)
// at models/base_model.py:673-679 -- non-fused path for rank-2/3 inputs: moments over every axis
// but the last (padded history positions included), biased variance for normalisation AND for
// the moving average, moving <- 0.95*moving + 0.05*batch, inference uses the moving statistics.
//
// The forward normalise + ReLU is never materialised: pgemm writes the pre-BN linear output z
// (and its per-block column sums), clsr_bn_finalize turns the sums into a per-feature affine
// (scale, shift) and the NEXT consumer applies relu(z*scale + shift) while loading.
// Backward:  dy = dh * (z*scale+shift > 0)      [clsr_bn_relu_bwd_reduce, + sums of dy, dy*xhat]
//            dz = a1*dy + a2*z + a3              [clsr_bn_bwd_coef + clsr_bn_bwd_apply]
// with a1 = gamma*invstd, a2 = -gamma*invstd^2*mean(dy*xhat), a3 = -gamma*invstd*mean(dy) - a2*mean.
#include "common.h"

// The two one-workgroup-per-feature reductions below are links of the step's dependent chain that run BESIDE the
// MFMA-saturated kernels of the other streams (40-80 workgroups, a few loads and adds each): with equal priority their
// waves queue for issue slots behind waves that hold the SIMD for 32 cycles per MFMA -- 28.6 us in the step for 5.7 us of
// work alone (profiles/r04_step_timeline_fp32.txt).  -DCLSR_BN_NO_PRIO: the equal-priority form (A/B builds).
#ifdef CLSR_BN_NO_PRIO
#define BN_CHAIN_PRIO()
#else
#define BN_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// column c of the [nparts][2][C] partial sums, over the partials: one wave, four partial rows per lane in flight, the lanes'
// sums combined by a wave shuffle reduction (a fixed order: deterministic)
__device__ __forceinline__ void bn_wave_sums(const double* __restrict__ partial, int nparts, int C, int c, double& s0,
                                             double& s1) {
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  for (int p0 = threadIdx.x; p0 < nparts; p0 += 4 * 64) {
    double x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + 64 * u, pc = p < nparts ? p : 0;
      x[u] = partial[((long)pc * 2 + 0) * C + c];
      y[u] = partial[((long)pc * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + 64 * u < nparts) { a[u] += x[u]; b[u] += y[u]; }
  }
  s0 = wave_sum_d((a[0] + a[1]) + (a[2] + a[3]));
  s1 = wave_sum_d((b[0] + b[1]) + (b[2] + b[3]));
}

// stats_partial: [nparts][2][C] doubles (column sums, column sums of squares).
// outputs: scale, shift (always); mean, invstd (saved for backward); moving stats updated in place
// when training.
__global__ void bn_finalize_kernel(const double* __restrict__ stats_partial, int nparts, int C,
                                   double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ moving_mean,
                                   float* __restrict__ moving_var, float momentum, float eps,
                                   int training, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  // ONE WAVE per feature and NO LDS (on the critical path of every BN layer: ~1000 partials, 16 per lane, four loads in
  // flight per sum): the launches run beside kernels whose workgroups own their CU's whole LDS (weight images of the
  // attention kernels, the fused encoder tail) -- a workgroup that asks for 64 bytes of LDS waits for one of THOSE to end
  // (40-49 us in the step for two of the eight launches of the chain, profiles/r06_fp32_timeline.txt)
  BN_CHAIN_PRIO();
  const int c = blockIdx.x;
  float mean, var;
  if (training) {
    double s = 0.0, q = 0.0;
    bn_wave_sums(stats_partial, nparts, C, c, s, q);
    if (threadIdx.x != 0) return;
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    moving_mean[c] = moving_mean[c] * momentum + mean * (1.0f - momentum);
    moving_var[c] = moving_var[c] * momentum + var * (1.0f - momentum);
  } else {
    if (threadIdx.x != 0) return;
    mean = moving_mean[c];
    var = moving_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (mean_out) mean_out[c] = mean;
  if (invstd_out) invstd_out[c] = invstd;
}

extern "C" int clsr_bn_finalize(const double* stats_partial, int nparts, int C, double count,
                                const float* gamma, const float* beta, float* moving_mean,
                                float* moving_var, float momentum, float eps, int training,
                                float* scale, float* shift, float* mean_out, float* invstd_out,
                                void* stream) {
  CLSR_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0);
  CLSR_CHECK_ARG(!training || (stats_partial && nparts > 0 && count > 0));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream,
                     stats_partial, nparts, C, count, gamma, beta, moving_mean, moving_var, momentum,
                     eps, training, scale, shift, mean_out, invstd_out);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Thread layout for [M, C] column reductions: thread -> (ty, q) with q a fixed float4 column.
#define COLRED_MAX_BLOCKS 512

// dh (in) -> dy (out, in place): dy = dh * (z*scale + shift > 0)
// partial[blockIdx.x][0][c] = sum dy ; partial[blockIdx.x][1][c] = sum dy * xhat,  xhat = (z-mean)*invstd
__global__ void __launch_bounds__(256) bn_relu_bwd_reduce_kernel(
    float* __restrict__ dh, const float* __restrict__ z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    int M, int C, double* __restrict__ partial) {
  __shared__ double red[2][256][4];
  const int QC = C >> 2;
  const int rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (ty < rpb) {
    const f32x4 sc = ld4(scale + 4 * q), sh = ld4(shift + 4 * q);
    const f32x4 mu = ld4(mean + 4 * q), is = ld4(invstd + 4 * q);
    for (long row = (long)blockIdx.x * rpb + ty; row < M; row += (long)gridDim.x * rpb) {
      const long off = row * C + 4 * q;
      const f32x4 zz = ld4(z + off);
      f32x4 d = ld4(dh + off);
      const f32x4 y = zz * sc + sh;
      d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
      d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
      st4(dh + off, d);
      const f32x4 xh = (zz - mu) * is;
      s1[0] += d.x; s1[1] += d.y; s1[2] += d.z; s1[3] += d.w;
      s2[0] += (double)d.x * xh.x; s2[1] += (double)d.y * xh.y;
      s2[2] += (double)d.z * xh.z; s2[3] += (double)d.w * xh.w;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { red[0][threadIdx.x][r] = s1[r]; red[1][threadIdx.x][r] = s2[r]; }
  __syncthreads();
  if (threadIdx.x < QC) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = 0.0, b = 0.0;
      for (int y = 0; y < rpb; ++y) { a += red[0][y * QC + threadIdx.x][r]; b += red[1][y * QC + threadIdx.x][r]; }
      partial[((long)blockIdx.x * 2 + 0) * C + 4 * threadIdx.x + r] = a;
      partial[((long)blockIdx.x * 2 + 1) * C + 4 * threadIdx.x + r] = b;
    }
  }
}

static int colred_blocks(int M, int C) {
  const int rpb = 256 / (C / 4);
  int b = clsr_cdiv(M, rpb * 4);
  if (b > COLRED_MAX_BLOCKS) b = COLRED_MAX_BLOCKS;
  if (b < 1) b = 1;
  return b;
}

extern "C" int clsr_colred_parts(int M, int C) { return colred_blocks(M, C); }

extern "C" int clsr_bn_relu_bwd_reduce(float* dh, const float* z, const float* scale,
                                       const float* shift, const float* mean, const float* invstd,
                                       int M, int C, double* partial, void* stream) {
  CLSR_CHECK_ARG(dh && z && scale && shift && mean && invstd && partial && M > 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0 && C >= 4 && C <= 1024);
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(colred_blocks(M, C)), dim3(256), 0,
                     (hipStream_t)stream, dh, z, scale, shift, mean, invstd, M, C, partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// coef[0..2][C] = a1, a2, a3 ; dgamma[c] (=|+=) sum dy*xhat ; dbeta[c] (=|+=) sum dy
__global__ void bn_bwd_coef_kernel(const double* __restrict__ partial, int nparts, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, float* __restrict__ coef,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                   int accumulate, double grad_scale) {
  BN_CHAIN_PRIO();
  const int c = blockIdx.x;  // one wave per feature, no LDS (see bn_finalize_kernel)
  double s1 = 0.0, s2 = 0.0;
  bn_wave_sums(partial, nparts, C, c, s1, s2);
  if (threadIdx.x != 0) return;
  const float g = gamma[c], is = invstd[c], mu = mean[c];
  const float c1 = (float)(s1 / count), c2 = (float)(s2 / count);
  const float a1 = g * is;
  const float a2 = -g * is * is * c2;
  coef[c] = a1;
  coef[C + c] = a2;
  coef[2 * C + c] = -a1 * c1 - a2 * mu;
  // grad_scale: data-parallel runs with synchronised statistics hold GLOBAL sums here and pre-divide by the world size,
  // so that the gradient all-reduce (SUM over ranks) restores them
  if (accumulate) {
    dgamma[c] += (float)(s2 * grad_scale);
    dbeta[c] += (float)(s1 * grad_scale);
  } else {
    dgamma[c] = (float)(s2 * grad_scale);
    dbeta[c] = (float)(s1 * grad_scale);
  }
}

extern "C" int clsr_bn_bwd_coef_scaled(const double* partial, int nparts, int C, double count,
                                       const float* gamma, const float* mean, const float* invstd,
                                       float* coef, float* dgamma, float* dbeta, int accumulate,
                                       double grad_scale, void* stream) {
  CLSR_CHECK_ARG(partial && gamma && mean && invstd && coef && dgamma && dbeta && nparts > 0 && C > 0);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream,
                     partial, nparts, C, count, gamma, mean, invstd, coef, dgamma, dbeta, accumulate, grad_scale);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_bn_bwd_coef(const double* partial, int nparts, int C, double count,
                                const float* gamma, const float* mean, const float* invstd,
                                float* coef, float* dgamma, float* dbeta, int accumulate,
                                void* stream) {
  return clsr_bn_bwd_coef_scaled(partial, nparts, C, count, gamma, mean, invstd, coef, dgamma, dbeta, accumulate, 1.0,
                                 stream);
}

// dy (in place) -> dz = a1*dy + a2*z + a3
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(float* __restrict__ dy,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ coef, long M,
                                                           int C) {
  const int QC = C >> 2;
  const long total = M * QC;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % QC);
    const f32x4 a1 = ld4(coef + 4 * q), a2 = ld4(coef + C + 4 * q), a3 = ld4(coef + 2 * C + 4 * q);
    const f32x4 d = ld4(dy + e * 4), zz = ld4(z + e * 4);
    st4(dy + e * 4, a1 * d + a2 * zz + a3);
  }
}

extern "C" int clsr_bn_bwd_apply(float* dy, const float* z, const float* coef, long M, int C,
                                 void* stream) {
  CLSR_CHECK_ARG(dy && z && coef && M > 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0);
  int blocks = clsr_cdiv(M * (C / 4), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, z, coef,
                     M, C);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// clsr_bn_bwd_coef + clsr_bn_bwd_apply in ONE launch for the small (row-level) layers: every block folds the partial
// sums itself (nparts x 2C doubles out of L2, ~100 KB), keeps the 3C coefficients in LDS and applies them to its share
// of dy; block 0 also writes coef / dgamma / dbeta.  One dispatch and one dependent-launch gap less per layer.
__global__ void __launch_bounds__(1024) bn_bwd_coef_apply_kernel(const double* __restrict__ partial, int nparts, int C,
                                                                 double count, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 float* __restrict__ coef, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ dy,
                                                                 const float* __restrict__ z, long M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bn_raw[];
  const int W = 2 * C;                                       // sums per part row
  const int NG = 1024 / W > 0 ? 1024 / W : 1;                // part groups folding side by side
  double* sm = reinterpret_cast<double*>(bn_raw);            // [NG][2C]
  float* cf = reinterpret_cast<float*>(sm + (size_t)NG * W); // [3][C]
  for (int t = threadIdx.x; t < NG * W; t += 1024) {
    const int pg = t / W, idx = t - pg * W;
    const double* p = partial + idx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int q = pg;
    for (; q + 3 * NG < nparts; q += 4 * NG) {
      s0 += p[(long)q * W];
      s1 += p[(long)(q + NG) * W];
      s2 += p[(long)(q + 2 * NG) * W];
      s3 += p[(long)(q + 3 * NG) * W];
    }
    for (; q < nparts; q += NG) s0 += p[(long)q * W];
    sm[t] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 1024) {
    double s1 = 0.0, s2 = 0.0;
    for (int pg = 0; pg < NG; ++pg) { s1 += sm[pg * W + c]; s2 += sm[pg * W + C + c]; }
    const float g = gamma[c], is = invstd[c], mu = mean[c];
    const float c1 = (float)(s1 / count), c2 = (float)(s2 / count);
    const float a1 = g * is;
    const float a2 = -g * is * is * c2;
    const float a3 = -a1 * c1 - a2 * mu;
    cf[c] = a1; cf[C + c] = a2; cf[2 * C + c] = a3;
    if (blockIdx.x == 0) {
      coef[c] = a1; coef[C + c] = a2; coef[2 * C + c] = a3;
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  __syncthreads();
  const int QC = C >> 2;
  const long total = M * QC;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % QC);
    const f32x4 a1 = ld4(cf + 4 * q), a2 = ld4(cf + C + 4 * q), a3 = ld4(cf + 2 * C + 4 * q);
    const f32x4 d = ld4(dy + e * 4), zz = ld4(z + e * 4);
    st4(dy + e * 4, a1 * d + a2 * zz + a3);
  }
}

extern "C" int clsr_bn_bwd_coef_apply(const double* partial, int nparts, int C, double count, const float* gamma,
                                      const float* mean, const float* invstd, float* coef, float* dgamma,
                                      float* dbeta, float* dy, const float* z, long M, void* stream) {
  CLSR_CHECK_ARG(partial && gamma && mean && invstd && coef && dgamma && dbeta && dy && z && nparts > 0 && C > 0 && M > 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0 && C <= 1024);
  int blocks = clsr_cdiv(M * (C / 4), 1024 * 8);
  if (blocks > 64) blocks = 64;
  const int W = 2 * C, NG = 1024 / W > 0 ? 1024 / W : 1;
  const size_t shmem = (size_t)NG * W * sizeof(double) + (size_t)3 * C * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_coef_apply_kernel, dim3(blocks), dim3(1024), shmem, (hipStream_t)stream, partial, nparts,
                     C, count, gamma, mean, invstd, coef, dgamma, dbeta, dy, z, M);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[e] = sum_p partial[p*n + e]  (e < n doubles): the per-block partial statistics of a batch-norm layer folded
// into ONE row, so that a data-parallel run all-reduces 2*C doubles instead of parts*2*C (clsr_amd/dp.py, sync-BN)
__global__ void __launch_bounds__(256) sum_parts_d_kernel(const double* __restrict__ partial, int nparts, int n,
                                                          double* __restrict__ out) {
  __shared__ double red[4];
  const int e = blockIdx.x;
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(long)p * n + e];
  s = block256_sum_d(s, red);
  if (threadIdx.x == 0) out[e] = s;
}

extern "C" int clsr_sum_parts_d(const double* partial, int nparts, int n, double* out, void* stream) {
  CLSR_CHECK_ARG(partial && out && nparts > 0 && n > 0);
  hipLaunchKernelGGL(sum_parts_d_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, partial, nparts, n, out);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
