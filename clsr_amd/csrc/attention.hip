// Score -> masked softmax -> weighted sum tail of _attention_fcn, forward and backward (gfx950).
//
// Reference: models/sequential/clsr.py:371-381 (squeeze, -(2**32)+1 padding mask, tf.nn.softmax
// over T, keys * weights) + the reduce_sum over T at clsr.py:154,221, and the last linear layer
// (w_nn_output, b_nn_output) of the att_fcn MLP (models/base_model.py:686-706) whose input is
// relu(BN(z1)), applied here on the fly from the pre-BN z1 and the BN affine.
//
// One wavefront owns one history group (the G rows that share a history); wave-level
// reductions only (max / sum for the softmax, dot products for the scores).  Thread layouts:
//   "column" layout  lane -> (tslot = lane / Q, q = lane % Q): 16-byte column chunk q of steps
//                    t = tslot, tslot + tpar, ... (coalesced row-major reads)
//   "step" layout    lane -> t (64 steps per chunk, up to 4 chunks => T <= 256)
#include "common.h"

#define ATT_MAXCH 4

struct AttOutArgs {
  const float* z1; const float* scale1; const float* shift1; const float* mean1; const float* invstd1;
  const float* w_out; const float* b_out;
  const int* seq_len; int len_stride;
  const float* keys;
  int Hn, G, T, C1, Dk;
  float* wts;            // [R, T] softmax weights (saved for backward)
  float* out;            // [R, Dk]
  // backward only
  const float* dout;     // [R, Dk]
  float* dy1;            // [R*T, C1]  (dh1 * relu mask)
  float* dkeys;          // [Hn, T, Dk]  +=
  double* bn_partial;    // [nblocks][2][C1]  sum dy1, sum dy1*xhat1
  float* w_partial;      // [nblocks][C1 + 4]  d w_out, then d b_out
};

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

// LDS (floats): sbuf[T*QC] | wl[T] | red[64*4]
__global__ void __launch_bounds__(64) att_out_fwd_kernel(AttOutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int T = a.T, C1 = a.C1, Dk = a.Dk;
  const int QC = C1 >> 2, tparC = 64 / QC, tsC = lane / QC, qC = lane - tsC * QC;
  const int QD = Dk >> 2, tparD = 64 / QD, tsD = lane / QD, qD = lane - tsD * QD;
  float* sbuf = lds;
  float* wl = sbuf + ((T * QC + 3) & ~3);
  f32x4* red = reinterpret_cast<f32x4*>(wl + ((T + 3) & ~3));
  f32x4 sc = {0, 0, 0, 0}, sh = sc, wo = sc;
  if (tsC < tparC) { sc = ld4(a.scale1 + 4 * qC); sh = ld4(a.shift1 + 4 * qC); wo = ld4(a.w_out + 4 * qC); }
  const float b_out = a.b_out[0];
  const long R = (long)a.Hn * a.G;
  for (long r = blockIdx.x; r < R; r += gridDim.x) {
    const long h = r / a.G;
    const int len = a.seq_len[h * a.len_stride];
    // (1) partial scores in column layout
    if (tsC < tparC) {
      const float* zp = a.z1 + r * T * C1 + 4 * qC;
      for (int t = tsC; t < T; t += tparC) {
        const f32x4 y = relu4(ld4(zp + (long)t * C1) * sc + sh);
        sbuf[t * QC + qC] = dot4(y, wo);
      }
    }
    __syncthreads();
    // (2) masked softmax in step layout
    float s[ATT_MAXCH];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < ATT_MAXCH; ++c) {
      const int t = c * 64 + lane;
      s[c] = -INFINITY;
      if (t < T && t < len) {
        float v = b_out;
        for (int q = 0; q < QC; ++q) v += sbuf[t * QC + q];
        s[c] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < ATT_MAXCH; ++c) {
      const int t = c * 64 + lane;
      s[c] = (t < T && t < len) ? __expf(s[c] - mx) : 0.f;
      sum += s[c];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < ATT_MAXCH; ++c) {
      const int t = c * 64 + lane;
      if (t < T) {
        // len == 0: every score equals the padding constant -> uniform weights (reference behaviour)
        const float w = len > 0 ? s[c] * inv : 1.0f / (float)T;
        wl[t] = w;
        if (a.wts) a.wts[r * T + t] = w;
      }
    }
    __syncthreads();
    // (3) weighted sum of the keys in column layout
    f32x4 acc = {0, 0, 0, 0};
    if (tsD < tparD) {
      const float* kp = a.keys + h * T * Dk + 4 * qD;
      const int tend = len > 0 ? len : T;
      for (int t = tsD; t < tend; t += tparD) acc += ld4(kp + (long)t * Dk) * wl[t];
    }
    red[lane] = acc;
    __syncthreads();
    if (tsD == 0) {
      for (int sidx = 1; sidx < tparD; ++sidx) acc += red[lane + sidx * QD];
      st4(a.out + r * Dk + 4 * qD, acc);
    }
    __syncthreads();
  }
}

static size_t att_fwd_lds(int T, int C1) {
  return ((size_t)((T * (C1 / 4) + 3) & ~3) + ((T + 3) & ~3) + 256) * sizeof(float);
}

extern "C" int clsr_att_out_fwd(const float* z1, const float* scale1, const float* shift1,
                                const float* w_out, const float* b_out, const int* seq_len,
                                int len_stride, const float* keys, int Hn, int G, int T, int C1, int Dk,
                                float* wts, float* out, void* stream) {
  CLSR_CHECK_ARG(z1 && scale1 && shift1 && w_out && b_out && seq_len && keys && out);
  CLSR_CHECK_ARG(Hn >= 0 && G > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(T <= 64 * ATT_MAXCH && C1 % 4 == 0 && Dk % 4 == 0 && C1 <= 256 && Dk <= 256);
  if (Hn == 0) return CLSR_OK;
  AttOutArgs a = {};
  a.z1 = z1; a.scale1 = scale1; a.shift1 = shift1; a.w_out = w_out; a.b_out = b_out;
  a.seq_len = seq_len; a.len_stride = len_stride; a.keys = keys;
  a.Hn = Hn; a.G = G; a.T = T; a.C1 = C1; a.Dk = Dk; a.wts = wts; a.out = out;
  long R = (long)Hn * G;
  int blocks = R > 8192 ? 8192 : (int)R;
  hipLaunchKernelGGL(att_out_fwd_kernel, dim3(blocks), dim3(64), att_fwd_lds(T, C1),
                     (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// Backward.  One wave per history group h; loops over its G rows.
// LDS (floats): dsl[T] | acc[T*Dk] | dol[Dk] | red[64*?]
#define ATT_BWD_MAX_BLOCKS 2048
__global__ void __launch_bounds__(64) att_out_bwd_kernel(AttOutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int T = a.T, C1 = a.C1, Dk = a.Dk;
  const int QC = C1 >> 2, tparC = 64 / QC, tsC = lane / QC, qC = lane - tsC * QC;
  const int QD = Dk >> 2, tparD = 64 / QD, tsD = lane / QD, qD = lane - tsD * QD;
  float* dsl = lds;                          // [T] d score
  float* dol = dsl + ((T + 3) & ~3);         // [Dk] dout row
  float* acc = dol + Dk;                     // [T*Dk] dkeys accumulator of this history
  f32x4 sc = {0, 0, 0, 0}, sh = sc, wo = sc, mu = sc, is = sc;
  if (tsC < tparC) {
    sc = ld4(a.scale1 + 4 * qC); sh = ld4(a.shift1 + 4 * qC); wo = ld4(a.w_out + 4 * qC);
    mu = ld4(a.mean1 + 4 * qC); is = ld4(a.invstd1 + 4 * qC);
  }
  f32x4 p_dy = {0, 0, 0, 0}, p_dyx = p_dy, p_dw = p_dy;  // column partial sums of this lane
  float p_db = 0.f;
  for (long h = blockIdx.x; h < a.Hn; h += gridDim.x) {
    const int len = a.seq_len[h * a.len_stride];
    const float* kp = a.keys + h * T * Dk;
    for (int e = lane; e < T * QD; e += 64) reinterpret_cast<f32x4*>(acc)[e] = (f32x4){0, 0, 0, 0};
    for (int gi = 0; gi < a.G; ++gi) {
      const long r = h * a.G + gi;
      __syncthreads();
      for (int d = lane; d < Dk; d += 64) dol[d] = a.dout[r * Dk + d];
      __syncthreads();
      // d weight_t = dout . keys[h,t,:]  (step layout), softmax backward
      float w[ATT_MAXCH], dw[ATT_MAXCH];
      float dotsum = 0.f;
#pragma unroll
      for (int c = 0; c < ATT_MAXCH; ++c) {
        const int t = c * 64 + lane;
        w[c] = 0.f; dw[c] = 0.f;
        if (t < T && t < len) {
          w[c] = a.wts[r * T + t];
          const float* row = kp + (long)t * Dk;
          float v = 0.f;
          for (int q = 0; q < QD; ++q) v += dot4(ld4(row + 4 * q), ld4(dol + 4 * q));
          dw[c] = v;
          dotsum += w[c] * v;
        }
      }
      dotsum = wave_sum(dotsum);
#pragma unroll
      for (int c = 0; c < ATT_MAXCH; ++c) {
        const int t = c * 64 + lane;
        if (t < T) {
          const float ds = (t < len) ? w[c] * (dw[c] - dotsum) : 0.f;
          dsl[t] = ds;
          p_db += ds;
        }
      }
      __syncthreads();
      // dkeys accumulation (column layout over Dk); for len == 0 the weights are the constant 1/T
      if (tsD < tparD) {
        const f32x4 dv = ld4(dol + 4 * qD);
        const int tend = len > 0 ? len : T;
        for (int t = tsD; t < tend; t += tparD) {
          const float wt = len > 0 ? a.wts[r * T + t] : 1.0f / (float)T;
          f32x4* ap = reinterpret_cast<f32x4*>(acc + t * Dk + 4 * qD);
          *ap += dv * wt;
        }
      }
      // dy1 = ds * w_out * (y1 > 0), column layout over C1; BN / w_out column sums
      if (tsC < tparC) {
        const float* zp = a.z1 + r * T * C1 + 4 * qC;
        float* dp = a.dy1 + r * T * C1 + 4 * qC;
        for (int t = tsC; t < T; t += tparC) {
          const f32x4 zz = ld4(zp + (long)t * C1);
          const f32x4 y = zz * sc + sh;
          const float ds = dsl[t];
          f32x4 d = wo * ds;
          d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
          d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
          st4(dp + (long)t * C1, d);
          p_dy += d;
          p_dyx += d * ((zz - mu) * is);
          p_dw += relu4(y) * ds;
        }
      }
    }
    __syncthreads();
    // dkeys[h] += acc
    float* dk = a.dkeys + h * T * Dk;
    for (int e = lane; e < T * QD; e += 64) {
      f32x4* gp = reinterpret_cast<f32x4*>(dk) + e;
      *gp += reinterpret_cast<f32x4*>(acc)[e];
    }
    __syncthreads();
  }
  // block partials: reduce the tslot copies through LDS (reuse acc region)
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(acc);
  red[lane] = p_dy; red[64 + lane] = p_dyx; red[128 + lane] = p_dw;
  __syncthreads();
  if (tsC == 0) {
    for (int s = 1; s < tparC; ++s) {
      p_dy += red[lane + s * QC]; p_dyx += red[64 + lane + s * QC]; p_dw += red[128 + lane + s * QC];
    }
    double* bp = a.bn_partial + (long)blockIdx.x * 2 * C1;
    bp[4 * qC + 0] = p_dy.x; bp[4 * qC + 1] = p_dy.y; bp[4 * qC + 2] = p_dy.z; bp[4 * qC + 3] = p_dy.w;
    bp[C1 + 4 * qC + 0] = p_dyx.x; bp[C1 + 4 * qC + 1] = p_dyx.y;
    bp[C1 + 4 * qC + 2] = p_dyx.z; bp[C1 + 4 * qC + 3] = p_dyx.w;
    st4(a.w_partial + (long)blockIdx.x * (C1 + 4) + 4 * qC, p_dw);
  }
  p_db = wave_sum(p_db);
  if (lane == 0) a.w_partial[(long)blockIdx.x * (C1 + 4) + C1] = p_db;
}

static int att_bwd_blocks(int Hn) { return Hn > ATT_BWD_MAX_BLOCKS ? ATT_BWD_MAX_BLOCKS : Hn; }
extern "C" int clsr_att_out_bwd_parts(int Hn) { return att_bwd_blocks(Hn); }

extern "C" int clsr_att_out_bwd(const float* dout, const float* wts, const float* z1,
                                const float* scale1, const float* shift1, const float* mean1,
                                const float* invstd1, const float* w_out, const int* seq_len,
                                int len_stride, const float* keys, int Hn, int G, int T, int C1, int Dk,
                                float* dy1, float* dkeys, double* bn_partial, float* w_partial,
                                void* stream) {
  CLSR_CHECK_ARG(dout && wts && z1 && scale1 && shift1 && mean1 && invstd1 && w_out && seq_len && keys);
  CLSR_CHECK_ARG(dy1 && dkeys && bn_partial && w_partial && Hn > 0 && G > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(T <= 64 * ATT_MAXCH && C1 % 4 == 0 && Dk % 4 == 0 && C1 <= 256 && Dk <= 256);
  AttOutArgs a = {};
  a.z1 = z1; a.scale1 = scale1; a.shift1 = shift1; a.mean1 = mean1; a.invstd1 = invstd1;
  a.w_out = w_out; a.seq_len = seq_len; a.len_stride = len_stride; a.keys = keys;
  a.Hn = Hn; a.G = G; a.T = T; a.C1 = C1; a.Dk = Dk; a.wts = const_cast<float*>(wts);
  a.dout = dout; a.dy1 = dy1; a.dkeys = dkeys; a.bn_partial = bn_partial; a.w_partial = w_partial;
  size_t accf = (size_t)T * Dk;
  if (accf < 192 * 4) accf = 192 * 4;
  size_t shmem = (((size_t)T + 3) / 4 * 4 + Dk + accf) * sizeof(float);
  CLSR_CHECK_SUPPORTED(shmem <= 160 * 1024);
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)att_out_bwd_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(att_out_bwd_kernel, dim3(att_bwd_blocks(Hn)), dim3(64), shmem,
                     (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[e] (=|+=) scale * sum_p partial[p*stride + e]   (float partials; e < n).  One block per output
// element: 256 threads stride over the partials, wave + LDS reduction.
__global__ void __launch_bounds__(256) reduce_parts_f_kernel(const float* __restrict__ partial, int nparts,
                                                             int stride, int n, float scale,
                                                             float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  const int e = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(long)p * stride + e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
    out[e] = accumulate ? out[e] + s : s;
  }
}

extern "C" int clsr_reduce_parts(const float* partial, int nparts, int stride, int n, float scale,
                                 float* out, int accumulate, void* stream) {
  CLSR_CHECK_ARG(partial && out && nparts > 0 && n > 0 && stride >= n);
  hipLaunchKernelGGL(reduce_parts_f_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, partial, nparts,
                     stride, n, scale, out, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
