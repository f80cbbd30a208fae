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
  const void* z1; const float* scale1; const float* shift1; const float* mean1; const float* invstd1;
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
template <bool ZH>
__global__ void __launch_bounds__(64) att_out_fwd_kernel(AttOutArgs a) {
  CLSR_CHAIN_PRIO();
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
  // NU row pieces in flight per lane and trip (one load per trip made this launch a chain of ~T / tparC dependent round trips per
  // row -- 64-82 us in the step for 164 MB).  Measured in round 5 (scripts: build_variant.sh -DATT_OUT_NU=9
  // -DATT_OUT_EARLY_KEYS=1): with nine pieces ONE trip covers a row at T = 50 and the key rows of the weighted sum -- which do
  // not depend on the scores -- can be requested before the scores are computed: the launch drops from 89 to 52 us, the step
  // does not move (2.670 / 2.680 / 2.667 against 2.677 / 2.672 / 2.652 ms, same box): four pieces, keys behind the softmax stay.
#ifndef ATT_OUT_NU
#define ATT_OUT_NU 4
#endif
#ifndef ATT_OUT_EARLY_KEYS
#define ATT_OUT_EARLY_KEYS 0
#endif
  constexpr int NU = ATT_OUT_NU;
  for (long r = blockIdx.x; r < R; r += gridDim.x) {
    const long h = r / a.G;
    const int len = a.seq_len[h * a.len_stride];
    const float* kp = a.keys + h * T * Dk + 4 * qD;
    const int tend = len > 0 ? len : T;
    f32x4 kv[NU];
    if (ATT_OUT_EARLY_KEYS && tsD < tparD) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = tsD + u * tparD;
        kv[u] = ld4(kp + (long)(t < tend ? t : tend - 1) * Dk);
      }
    }
    // (1) partial scores in column layout
    if (tsC < tparC) {
      const long zo = r * T * C1 + 4 * qC;
      for (int t0 = tsC; t0 < T; t0 += NU * tparC) {
        f32x4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int t = t0 + u * tparC;
          v[u] = load4e<ZH>(a.z1, zo + (long)(t < T ? t : T - 1) * C1);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int t = t0 + u * tparC;
          if (t < T) sbuf[t * QC + qC] = dot4(relu4(v[u] * sc + sh), wo);
        }
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
      for (int t0 = tsD; t0 < tend; t0 += NU * tparD) {
        if (!ATT_OUT_EARLY_KEYS || t0 != tsD) {       // (the first trip's rows were requested at the top)
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int t = t0 + u * tparD;
            kv[u] = ld4(kp + (long)(t < tend ? t : tend - 1) * Dk);
          }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int t = t0 + u * tparD;
          if (t < tend) acc += kv[u] * wl[t];
        }
      }
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

static int att_out_fwd_impl(const void* z1, int z1_bf16, const float* scale1, const float* shift1,
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
  if (z1_bf16)
    hipLaunchKernelGGL(att_out_fwd_kernel<true>, dim3(blocks), dim3(64), att_fwd_lds(T, C1), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(att_out_fwd_kernel<false>, dim3(blocks), dim3(64), att_fwd_lds(T, C1), (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_out_fwd(const float* z1, const float* scale1, const float* shift1,
                                const float* w_out, const float* b_out, const int* seq_len,
                                int len_stride, const float* keys, int Hn, int G, int T, int C1, int Dk,
                                float* wts, float* out, void* stream) {
  return att_out_fwd_impl(z1, 0, scale1, shift1, w_out, b_out, seq_len, len_stride, keys, Hn, G, T, C1, Dk, wts, out,
                          stream);
}

// the same with z1 stored as bf16 (speed mode, csrc/hgemm.hip)
extern "C" int clsr_att_out_fwd_h(const void* z1, const float* scale1, const float* shift1,
                                  const float* w_out, const float* b_out, const int* seq_len,
                                  int len_stride, const float* keys, int Hn, int G, int T, int C1, int Dk,
                                  float* wts, float* out, void* stream) {
  return att_out_fwd_impl(z1, 1, scale1, shift1, w_out, b_out, seq_len, len_stride, keys, Hn, G, T, C1, Dk, wts, out,
                          stream);
}

// Backward, split by access pattern:
//   att_score_bwd  (per history group, latency bound, tiny traffic): softmax backward -> d score ds[r, t],
//                  dkeys[h, t, :] += sum_g w[g, t] * dout[g, :], partial sums of d b_out
//   att_dy1_stats  (streaming over the R*T x C1 activations): with dy1 = ds * w_out * (y1 > 0), column sums
//                  of dy1, dy1 * xhat1 (the BN-1 backward reduction) and relu(y1) * ds (d w_out) -- dy1 itself
//                  is NOT written
//   att_dy1_apply  (streaming): dz1 = a1 * dy1 + a2 * z1 + a3 recomputing dy1 from (z1, ds): the only pass
//                  that writes an [R*T, C1] tensor.
// Traffic per launch chain at R*T = 1M, C1 = 40: 2 reads of z1 + 1 write of dz1 (0.49 GB) instead of the
// 0.82 GB of a materialised dy1 followed by a separate BN apply pass.
#define ATT_BWD_MAX_BLOCKS 4096
#define ATT_BWD_MAXG 8

// One workgroup per history group, one wave per row of the group (groups of more than ATT_BWD_MAXG rows: the
// waves take the rows round robin).
// LDS (floats): dol[G][Dk] | wl[G][Tp] | pdb[waves]
template <int NCH>
__global__ void __launch_bounds__(64 * ATT_BWD_MAXG) att_score_bwd_kernel(AttOutArgs a, float* __restrict__ ds_out,
                                                                          float* __restrict__ b_partial) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int T = a.T, Dk = a.Dk, G = a.G;
  const int Tp = (T + 3) & ~3;
  const int QD = Dk >> 2;
  float* dol = lds;                 // [G][Dk] dout rows
  float* wl = dol + G * Dk;         // [G][Tp] softmax weights used by the dkeys sum
  float* pdb = wl + G * Tp;
  float p_db = 0.f;
  for (long h = blockIdx.x; h < a.Hn; h += gridDim.x) {
    const int len = a.seq_len[h * a.len_stride];
    const float* kp = a.keys + h * T * Dk;
    __syncthreads();  // the previous group's dkeys pass is done with dol / wl
    for (int gi = wave; gi < G; gi += nw)
      for (int d = lane; d < Dk; d += 64) dol[gi * Dk + d] = a.dout[(h * G + gi) * Dk + d];
    __syncthreads();
    for (int gi = wave; gi < G; gi += nw) {
      const long r = h * G + gi;
      float w[NCH], dw[NCH];
      float dotsum = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int t = c * 64 + lane;
        w[c] = 0.f; dw[c] = 0.f;
        if (t < T && t < len) {
          w[c] = a.wts[r * T + t];
          const float* row = kp + (long)t * Dk;
          float v = 0.f;
#pragma unroll 5
          for (int q = 0; q < QD; ++q) v += dot4(ld4(row + 4 * q), ld4(dol + gi * Dk + 4 * q));
          dw[c] = v;
          dotsum += w[c] * v;
        }
      }
      dotsum = wave_sum(dotsum);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int t = c * 64 + lane;
        if (t < T) {
          const float ds = (t < len) ? w[c] * (dw[c] - dotsum) : 0.f;
          ds_out[r * T + t] = ds;
          wl[gi * Tp + t] = len > 0 ? w[c] : 1.0f / (float)T;  // len == 0: the weights are the constant 1/T
          p_db += ds;
        }
      }
    }
    __syncthreads();
    // dkeys[h, t, :] += sum_g w[g, t] * dout[g, :]   (all waves of the group)
    float* dk = a.dkeys + h * T * Dk;
    for (int e = tid; e < T * QD; e += blockDim.x) {
      const int t = e / QD, q = e - t * QD;
      f32x4 sum = {0, 0, 0, 0};
      for (int gg = 0; gg < G; ++gg) sum += ld4(dol + gg * Dk + 4 * q) * wl[gg * Tp + t];
      f32x4* gp = reinterpret_cast<f32x4*>(dk) + e;
      *gp += sum;
    }
  }
  p_db = wave_sum(p_db);
  if (lane == 0) pdb[wave] = p_db;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int wv = 0; wv < nw; ++wv) s += pdb[wv];
    b_partial[blockIdx.x] = s;
  }
}

static int att_bwd_blocks(int Hn) { return Hn > ATT_BWD_MAX_BLOCKS ? ATT_BWD_MAX_BLOCKS : Hn; }
extern "C" int clsr_att_score_bwd_parts(int Hn) { return att_bwd_blocks(Hn); }

extern "C" int clsr_att_score_bwd(const float* dout, const float* wts, const int* seq_len, int len_stride,
                                  const float* keys, int Hn, int G, int T, int Dk, float* ds, float* dkeys,
                                  float* b_partial, void* stream) {
  CLSR_CHECK_ARG(dout && wts && seq_len && keys && ds && dkeys && b_partial && Hn > 0 && G > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(T <= 64 * ATT_MAXCH && Dk % 4 == 0 && Dk <= 256);
  AttOutArgs a = {};
  a.seq_len = seq_len; a.len_stride = len_stride; a.keys = keys;
  a.Hn = Hn; a.G = G; a.T = T; a.Dk = Dk; a.wts = const_cast<float*>(wts);
  a.dout = dout; a.dkeys = dkeys;
  const size_t Tp = ((size_t)T + 3) / 4 * 4;
  const size_t shmem = ((size_t)G * Dk + (size_t)G * Tp + ATT_BWD_MAXG) * sizeof(float);
  CLSR_CHECK_SUPPORTED(shmem <= 64 * 1024);
  const dim3 grid(att_bwd_blocks(Hn)), block(64 * (G < ATT_BWD_MAXG ? G : ATT_BWD_MAXG));
  hipStream_t s = (hipStream_t)stream;
  if (T <= 64) hipLaunchKernelGGL(att_score_bwd_kernel<1>, grid, block, shmem, s, a, ds, b_partial);
  else if (T <= 128) hipLaunchKernelGGL(att_score_bwd_kernel<2>, grid, block, shmem, s, a, ds, b_partial);
  else hipLaunchKernelGGL(att_score_bwd_kernel<4>, grid, block, shmem, s, a, ds, b_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// column layout: thread (ty = tid / QC, q = tid % QC) owns the 16-byte column chunk q of rows ty, ty + rpb, ...
#define DY1_MAX_BLOCKS 2048
__device__ __forceinline__ f32x4 dy1_of(f32x4 zz, float ds, f32x4 sc, f32x4 sh, f32x4 wo) {
  const f32x4 y = zz * sc + sh;
  f32x4 d = wo * ds;
  d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
  d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
  return d;
}

template <bool ZH>
__global__ void __launch_bounds__(256) att_dy1_stats_kernel(
    const void* __restrict__ z1, const float* __restrict__ ds, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ w_out, long M, int C, double* __restrict__ bn_partial,
    float* __restrict__ w_partial) {
  CLSR_CHAIN_PRIO();
  __shared__ f32x4 red[3][256];
  const int QC = C >> 2;
  const int rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  f32x4 s1 = {0, 0, 0, 0}, s2 = s1, s3 = s1;
  if (ty < rpb) {
    const f32x4 sc = ld4(scale + 4 * q), sh = ld4(shift + 4 * q), wo = ld4(w_out + 4 * q);
    const f32x4 mu = ld4(mean + 4 * q), is = ld4(invstd + 4 * q);
    // a block owns a contiguous range of rows: fp32 partial sums stay short (<= M / blocks / rpb terms)
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
    // (four rows in flight per thread, accumulated in row order: one load per trip left this streaming pass at 2.9 TB/s)
    for (long row0 = lo + ty; row0 < hi; row0 += 4 * rpb) {
      f32x4 zv[4];
      float dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long row = row0 + (long)u * rpb;
        const long rc = row < hi ? row : hi - 1;
        zv[u] = load4e<ZH>(z1, rc * C + 4 * q);
        dv[u] = ds[rc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row0 + (long)u * rpb < hi) {
          const f32x4 zz = zv[u];
          const float dsv = dv[u];
          const f32x4 y = zz * sc + sh;
          const f32x4 d = dy1_of(zz, dsv, sc, sh, wo);
          s1 += d;
          s2 += d * ((zz - mu) * is);
          s3 += relu4(y) * dsv;
        }
      }
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = s3;
  __syncthreads();
  if (threadIdx.x < QC) {
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    f32x4 a2 = {0, 0, 0, 0};
    for (int y = 0; y < rpb; ++y) {
      const f32x4 u = red[0][y * QC + threadIdx.x], v = red[1][y * QC + threadIdx.x];
      a0[0] += u.x; a0[1] += u.y; a0[2] += u.z; a0[3] += u.w;
      a1[0] += v.x; a1[1] += v.y; a1[2] += v.z; a1[3] += v.w;
      a2 += red[2][y * QC + threadIdx.x];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bn_partial[((long)blockIdx.x * 2 + 0) * C + 4 * threadIdx.x + r] = a0[r];
      bn_partial[((long)blockIdx.x * 2 + 1) * C + 4 * threadIdx.x + r] = a1[r];
    }
    st4(w_partial + (long)blockIdx.x * C + 4 * threadIdx.x, a2);
  }
}

static int dy1_blocks(long M, int C) {
  const int rpb = 256 / (C / 4);
  long b = (M + (long)rpb * 8 - 1) / ((long)rpb * 8);
  if (b > DY1_MAX_BLOCKS) b = DY1_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" int clsr_att_dy1_parts(long M, int C1) { return dy1_blocks(M, C1); }

static int att_dy1_stats_impl(const void* z1, int z1_bf16, const float* ds, const float* scale1, const float* shift1,
                              const float* mean1, const float* invstd1, const float* w_out, long M, int C1,
                              double* bn_partial, float* w_partial, void* stream) {
  CLSR_CHECK_ARG(z1 && ds && scale1 && shift1 && mean1 && invstd1 && w_out && bn_partial && w_partial && M > 0);
  CLSR_CHECK_SUPPORTED(C1 % 4 == 0 && C1 >= 4 && C1 <= 1024);
  if (z1_bf16)
    hipLaunchKernelGGL(att_dy1_stats_kernel<true>, dim3(dy1_blocks(M, C1)), dim3(256), 0, (hipStream_t)stream, z1, ds,
                       scale1, shift1, mean1, invstd1, w_out, M, C1, bn_partial, w_partial);
  else
    hipLaunchKernelGGL(att_dy1_stats_kernel<false>, dim3(dy1_blocks(M, C1)), dim3(256), 0, (hipStream_t)stream, z1, ds,
                       scale1, shift1, mean1, invstd1, w_out, M, C1, bn_partial, w_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_dy1_stats(const float* z1, const float* ds, const float* scale1, const float* shift1,
                                  const float* mean1, const float* invstd1, const float* w_out, long M, int C1,
                                  double* bn_partial, float* w_partial, void* stream) {
  return att_dy1_stats_impl(z1, 0, ds, scale1, shift1, mean1, invstd1, w_out, M, C1, bn_partial, w_partial, stream);
}

extern "C" int clsr_att_dy1_stats_h(const void* z1, const float* ds, const float* scale1, const float* shift1,
                                    const float* mean1, const float* invstd1, const float* w_out, long M, int C1,
                                    double* bn_partial, float* w_partial, void* stream) {
  return att_dy1_stats_impl(z1, 1, ds, scale1, shift1, mean1, invstd1, w_out, M, C1, bn_partial, w_partial, stream);
}

// dz1 = a1 * dy1 + a2 * z1 + a3 with dy1 recomputed from (z1, ds); coef = [a1 | a2 | a3] from clsr_bn_bwd_coef
__global__ void __launch_bounds__(256) att_dy1_apply_kernel(
    const float* __restrict__ z1, const float* __restrict__ ds, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ w_out, const float* __restrict__ coef, long M,
    int C, float* __restrict__ dz1) {
  const int QC = C >> 2;
  const long total = M * QC;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / QC;
    const int q = (int)(e - row * QC);
    const f32x4 a1 = ld4(coef + 4 * q), a2 = ld4(coef + C + 4 * q), a3 = ld4(coef + 2 * C + 4 * q);
    const f32x4 zz = ld4(z1 + e * 4);
    const f32x4 d = dy1_of(zz, ds[row], ld4(scale + 4 * q), ld4(shift + 4 * q), ld4(w_out + 4 * q));
    st4(dz1 + e * 4, a1 * d + a2 * zz + a3);
  }
}

extern "C" int clsr_att_dy1_apply(const float* z1, const float* ds, const float* scale1, const float* shift1,
                                  const float* w_out, const float* coef, long M, int C1, float* dz1,
                                  void* stream) {
  CLSR_CHECK_ARG(z1 && ds && scale1 && shift1 && w_out && coef && dz1 && M > 0);
  CLSR_CHECK_SUPPORTED(C1 % 4 == 0);
  int blocks = clsr_cdiv(M * (C1 / 4), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(att_dy1_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z1, ds, scale1,
                     shift1, w_out, coef, M, C1, dz1);
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
