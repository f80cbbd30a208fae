// Row-level heads of the CLSR step (gfx950): alpha-gate input/fusion, MLP output layers, the
// group-softmax data loss, the contrastive loss; forward and backward.
//
// Reference: models/sequential/clsr.py:239-275 (concat_all, sigmoid alpha, user_embed fusion,
// model_output), models/base_model.py:686-706 (w_nn_output/b_nn_output), base_model.py:215-235
// (softmax data loss), clsr.py:46-71 (_compute_contrastive_loss).
//
// Rows b = h*G + g: the G rows of a history group share the history-level tensors
// (final_state, att_fea_long, hist_mean, hist_recent); one wavefront owns one group so
// history-level gradients are summed over the group in registers (no atomics).
#include "common.h"

// ------------------------------------------------------------------ alpha-gate input concat
// out[b, :] = [fs[h] (nfs) | target[b] (D) | L[h] (D) | S[b] (D) | tnow_last[b] | 0 pad]   (ldo wide)
// thread -> (row slot, 16-byte chunk); 32-bit indices (the element-wise form spent 70 us in 64-bit divisions)
__global__ void __launch_bounds__(256) alpha_concat_kernel(
    const float* __restrict__ fs, int nfs, const float* __restrict__ target, const float* __restrict__ L,
    const float* __restrict__ S, const float* __restrict__ tnow, long tnow_stride, int tnow_col, int tnow_group, int B,
    int G, int D, float* __restrict__ out, int ldo) {
  const int QC = ldo >> 2, rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  if (ty >= rpb) return;
  const int c = 4 * q;
  for (int b = blockIdx.x * rpb + ty; b < B; b += gridDim.x * rpb) {
    const int h = b / G;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < nfs) v = ld4(fs + (long)h * nfs + c);
    else if (c < nfs + D) v = ld4(target + (long)b * D + (c - nfs));
    else if (c < nfs + 2 * D) v = ld4(L + (long)h * D + (c - nfs - D));
    else if (c < nfs + 3 * D) v = ld4(S + (long)b * D + (c - nfs - 2 * D));
    else if (c == nfs + 3 * D) v.x = tnow[(long)(b / tnow_group) * tnow_stride + tnow_col];
    st4(out + (long)b * ldo + c, v);
  }
}

extern "C" int clsr_alpha_concat(const float* fs, int nfs, const float* target, const float* L,
                                 const float* S, const float* tnow, long tnow_stride, int tnow_col,
                                 int tnow_group, long B, int G, int D, float* out, int ldo, void* stream) {
  CLSR_CHECK_ARG(target && L && S && tnow && out && B > 0 && G > 0 && tnow_group > 0 && (nfs == 0 || fs));
  CLSR_CHECK_ARG(ldo >= nfs + 3 * D + 1);
  CLSR_CHECK_SUPPORTED(nfs % 4 == 0 && D % 4 == 0 && ldo % 4 == 0 && ldo <= 1024 && B < (1L << 31));
  const int rpb = 256 / (ldo / 4);
  int blocks = clsr_cdiv(B, rpb * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(alpha_concat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, fs, nfs,
                     target, L, S, tnow, tnow_stride, tnow_col, tnow_group, (int)B, G, D, out, ldo);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// d(concat): dfs[h] += sum_g dA[b, 0:nfs]; dtarget[b] += dA[b, nfs:nfs+D]; dL[h] += sum_g ...; dS[b] += ...
__global__ void __launch_bounds__(64) alpha_concat_bwd_kernel(const float* __restrict__ dA, int ldo, int nfs,
                                                              long Hn, int G, int D,
                                                              float* __restrict__ dfs,
                                                              float* __restrict__ dtarget,
                                                              float* __restrict__ dL,
                                                              float* __restrict__ dS) {
  const int lane = threadIdx.x;
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    for (int c = lane; c < nfs + 3 * D; c += 64) {
      float acc = 0.f;
      for (int gi = 0; gi < G; ++gi) {
        const long b = h * G + gi;
        const float v = dA[b * ldo + c];
        if (c < nfs) acc += v;
        else if (c < nfs + D) dtarget[b * D + (c - nfs)] += v;
        else if (c < nfs + 2 * D) acc += v;
        else dS[b * D + (c - nfs - 2 * D)] += v;
      }
      if (c < nfs) dfs[h * nfs + c] += acc;
      else if (c >= nfs + D && c < nfs + 2 * D) dL[h * D + (c - nfs - D)] += acc;
    }
  }
}

extern "C" int clsr_alpha_concat_bwd(const float* dA, int ldo, int nfs, long Hn, int G, int D,
                                     float* dfs, float* dtarget, float* dL, float* dS, void* stream) {
  CLSR_CHECK_ARG(dA && dtarget && dL && dS && Hn > 0 && G > 0 && (nfs == 0 || dfs));
  int blocks = Hn > 8192 ? 8192 : (int)Hn;
  hipLaunchKernelGGL(alpha_concat_bwd_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, dA, ldo,
                     nfs, Hn, G, D, dfs, dtarget, dL, dS);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ MLP output layer
// logit[b] = relu(z1[b,:]*scale + shift) . w_out + b_out
__global__ void mlp_out_fwd_kernel(const float* __restrict__ z1, const float* __restrict__ scale,
                                   const float* __restrict__ shift, const float* __restrict__ w_out,
                                   const float* __restrict__ b_out, long B, int C1,
                                   float* __restrict__ logit) {
  for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long)gridDim.x * blockDim.x) {
    const float* zp = z1 + b * C1;
    float s = b_out[0];
    for (int c = 0; c < C1; c += 4) {
      const f32x4 y = ld4(zp + c) * ld4(scale + c) + ld4(shift + c);
      const f32x4 w = ld4(w_out + c);
      s += fmaxf(y.x, 0.f) * w.x + fmaxf(y.y, 0.f) * w.y + fmaxf(y.z, 0.f) * w.z + fmaxf(y.w, 0.f) * w.w;
    }
    logit[b] = s;
  }
}

extern "C" int clsr_mlp_out_fwd(const float* z1, const float* scale, const float* shift,
                                const float* w_out, const float* b_out, long B, int C1, float* logit,
                                void* stream) {
  CLSR_CHECK_ARG(z1 && scale && shift && w_out && b_out && logit && B > 0);
  CLSR_CHECK_SUPPORTED(C1 % 4 == 0);
  int blocks = clsr_cdiv(B, 128);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mlp_out_fwd_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, z1, scale, shift,
                     w_out, b_out, B, C1, logit);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// dy1[b,c] = dlogit[b] * w_out[c] * (y1 > 0);  partial sums for BN backward and for d w_out / d b_out.
// Thread layout (ty, q) as in bn.hip; bn_partial [nblocks][2][C1] doubles, w_partial [nblocks][C1+4] floats.
__global__ void __launch_bounds__(256) mlp_out_bwd_kernel(
    const float* __restrict__ dlogit, const float* __restrict__ z1, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ w_out, long B, int C1, float* __restrict__ dy1,
    double* __restrict__ bn_partial, float* __restrict__ w_partial) {
  __shared__ double red[2][256][4];
  __shared__ float redw[256][4];
  __shared__ float redb[256];
  const int QC = C1 >> 2, rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  f32x4 sw = {0, 0, 0, 0};
  float sb = 0.f;
  if (ty < rpb) {
    const f32x4 sc = ld4(scale + 4 * q), sh = ld4(shift + 4 * q), mu = ld4(mean + 4 * q);
    const f32x4 is = ld4(invstd + 4 * q), wo = ld4(w_out + 4 * q);
    for (long b = (long)blockIdx.x * rpb + ty; b < B; b += (long)gridDim.x * rpb) {
      const float dl = dlogit[b];
      const f32x4 zz = ld4(z1 + b * C1 + 4 * q);
      const f32x4 y = zz * sc + sh;
      f32x4 d = wo * dl;
      d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
      d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
      st4(dy1 + b * C1 + 4 * q, d);
      const f32x4 xh = (zz - mu) * is;
      s1[0] += d.x; s1[1] += d.y; s1[2] += d.z; s1[3] += d.w;
      s2[0] += (double)d.x * xh.x; s2[1] += (double)d.y * xh.y;
      s2[2] += (double)d.z * xh.z; s2[3] += (double)d.w * xh.w;
      sw.x += fmaxf(y.x, 0.f) * dl; sw.y += fmaxf(y.y, 0.f) * dl;
      sw.z += fmaxf(y.z, 0.f) * dl; sw.w += fmaxf(y.w, 0.f) * dl;
      if (q == 0) sb += dl;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { red[0][threadIdx.x][r] = s1[r]; red[1][threadIdx.x][r] = s2[r]; }
  redw[threadIdx.x][0] = sw.x; redw[threadIdx.x][1] = sw.y; redw[threadIdx.x][2] = sw.z; redw[threadIdx.x][3] = sw.w;
  redb[threadIdx.x] = sb;
  __syncthreads();
  if (threadIdx.x < QC) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = 0.0, b2 = 0.0;
      float w = 0.f;
      for (int y = 0; y < rpb; ++y) {
        a += red[0][y * QC + threadIdx.x][r];
        b2 += red[1][y * QC + threadIdx.x][r];
        w += redw[y * QC + threadIdx.x][r];
      }
      bn_partial[((long)blockIdx.x * 2 + 0) * C1 + 4 * threadIdx.x + r] = a;
      bn_partial[((long)blockIdx.x * 2 + 1) * C1 + 4 * threadIdx.x + r] = b2;
      w_partial[(long)blockIdx.x * (C1 + 4) + 4 * threadIdx.x + r] = w;
    }
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int y = 0; y < rpb; ++y) s += redb[y * QC];
    w_partial[(long)blockIdx.x * (C1 + 4) + C1] = s;
  }
}

static int mlp_bwd_blocks(long B, int C1) {
  const int rpb = 256 / (C1 / 4);
  long b = (B + rpb * 4 - 1) / (rpb * 4);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" int clsr_mlp_out_bwd_parts(long B, int C1) { return mlp_bwd_blocks(B, C1); }

extern "C" int clsr_mlp_out_bwd(const float* dlogit, const float* z1, const float* scale,
                                const float* shift, const float* mean, const float* invstd,
                                const float* w_out, long B, int C1, float* dy1, double* bn_partial,
                                float* w_partial, void* stream) {
  CLSR_CHECK_ARG(dlogit && z1 && scale && shift && mean && invstd && w_out && dy1 && bn_partial && w_partial);
  CLSR_CHECK_SUPPORTED(C1 % 4 == 0 && C1 >= 4 && C1 <= 1024 && B > 0);
  hipLaunchKernelGGL(mlp_out_bwd_kernel, dim3(mlp_bwd_blocks(B, C1)), dim3(256), 0, (hipStream_t)stream,
                     dlogit, z1, scale, shift, mean, invstd, w_out, B, C1, dy1, bn_partial, w_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ alpha fusion
// alpha[b] = sigmoid(alpha_logit[b]) (or the manual constant);
// mo[b, :] = [alpha*L[h] + (1-alpha)*S[b] | target[b]]
__global__ void alpha_fuse_fwd_kernel(const float* __restrict__ alpha_logit, float manual_alpha,
                                      const float* __restrict__ L, const float* __restrict__ S,
                                      const float* __restrict__ target, long B, int G, int D,
                                      float* __restrict__ alpha, float* __restrict__ mo) {
  const long total = B * 2 * D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / (2 * D);
    const int c = (int)(e - b * 2 * D);
    const float al = alpha_logit ? sigmoidf_(alpha_logit[b]) : manual_alpha;
    if (c == 0 && alpha) alpha[b] = al;
    float v;
    if (c < D) v = al * L[(b / G) * D + c] + (1.0f - al) * S[b * D + c];
    else v = target[b * D + (c - D)];
    mo[e] = v;
  }
}

extern "C" int clsr_alpha_fuse_fwd(const float* alpha_logit, float manual_alpha, const float* L,
                                   const float* S, const float* target, long B, int G, int D,
                                   float* alpha, float* mo, void* stream) {
  CLSR_CHECK_ARG(L && S && target && mo && B > 0 && G > 0);
  int blocks = clsr_cdiv(B * 2 * D, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(alpha_fuse_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, alpha_logit,
                     manual_alpha, L, S, target, B, G, D, alpha, mo);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// backward: dmo[B, 2D] -> dalpha_logit[b], dL[h] += sum_g alpha*due, dS[b] += (1-alpha)*due,
//           dtarget[b] += dmo[b, D:]
__global__ void __launch_bounds__(64) alpha_fuse_bwd_kernel(
    const float* __restrict__ dmo, const float* __restrict__ alpha, float manual_alpha,
    const float* __restrict__ L, const float* __restrict__ S, long Hn, int G, int D,
    float* __restrict__ dalpha_logit, float* __restrict__ dL, float* __restrict__ dS,
    float* __restrict__ dtarget) {
  const int lane = threadIdx.x;
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    for (int c0 = 0; c0 < D; c0 += 64) {  // D <= 64 -> one pass; larger D handled in column blocks
      const int c = c0 + lane;
      float accL = 0.f;
      for (int gi = 0; gi < G; ++gi) {
        const long b = h * G + gi;
        const float al = alpha ? alpha[b] : manual_alpha;
        float part = 0.f;
        if (c < D) {
          const float due = dmo[b * 2 * D + c];
          part = due * (L[h * D + c] - S[b * D + c]);
          accL += al * due;
          dS[b * D + c] += (1.0f - al) * due;
          dtarget[b * D + c] += dmo[b * 2 * D + D + c];
        }
        part = wave_sum(part);
        if (dalpha_logit && lane == 0) {
          const float da = part * al * (1.0f - al);
          if (c0 == 0) dalpha_logit[b] = da; else dalpha_logit[b] += da;
        }
      }
      if (c < D) dL[h * D + c] += accL;
    }
  }
}

extern "C" int clsr_alpha_fuse_bwd(const float* dmo, const float* alpha, float manual_alpha,
                                   const float* L, const float* S, long Hn, int G, int D,
                                   float* dalpha_logit, float* dL, float* dS, float* dtarget,
                                   void* stream) {
  CLSR_CHECK_ARG(dmo && L && S && dL && dS && dtarget && Hn > 0 && G > 0);
  int blocks = Hn > 8192 ? 8192 : (int)Hn;
  hipLaunchKernelGGL(alpha_fuse_bwd_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, dmo, alpha,
                     manual_alpha, L, S, Hn, G, D, dalpha_logit, dL, dS, dtarget);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ group-softmax data loss
// loss = -(G/B) * sum_{b: label==1} log softmax_group(logit)[b]
// dlogit[b] = (G/B) * (npos_group * softmax[b] - [label==1])
__global__ void softmax_loss_kernel(const float* __restrict__ logit, const float* __restrict__ labels,
                                    long P, int G, float scale, double* __restrict__ loss_out,
                                    float* __restrict__ dlogit) {
  float local = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const float* lg = logit + p * G;
    const float* lb = labels + p * G;
    float mx = -INFINITY;
    for (int g = 0; g < G; ++g) mx = fmaxf(mx, lg[g]);
    float sum = 0.f;
    for (int g = 0; g < G; ++g) sum += __expf(lg[g] - mx);
    const float lse = mx + __logf(sum);
    int npos = 0;
    for (int g = 0; g < G; ++g)
      if (lb[g] == 1.0f) { ++npos; local -= (lg[g] - lse); }
    if (dlogit)
      for (int g = 0; g < G; ++g) {
        const float sm = __expf(lg[g] - lse);
        dlogit[p * G + g] = scale * ((float)npos * sm - (lb[g] == 1.0f ? 1.0f : 0.0f));
      }
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0 && local != 0.f) atomicAdd(loss_out, (double)local * scale);
}

// scale = 1 / (number of groups the mean runs over): 1/P single process, 1/(P * world) data parallel
extern "C" int clsr_softmax_loss(const float* logit, const float* labels, long P, int G, float scale,
                                 double* loss_out, float* dlogit, void* stream) {
  CLSR_CHECK_ARG(logit && labels && loss_out && P > 0 && G > 0);
  int blocks = clsr_cdiv(P, 128);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(softmax_loss_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, logit, labels,
                     P, G, scale, loss_out, dlogit);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ logit head: output layer + loss + its backward
// The three row-local launches at the turn of the step -- clsr_mlp_out_fwd (logit), clsr_softmax_loss (loss, dlogit),
// clsr_mlp_out_bwd (dy1, batch-norm sums, d w_out / d b_out) -- as ONE: a wave owns whole softmax groups (G consecutive
// rows), lane c owns column c of z1 (C1 <= 64), the G logits are wave sums.  Each of the three was a ~5 us link of the
// dependent chain of the heads.  Same partial layouts as clsr_mlp_out_bwd.
#define MT_MAXG 8
// wave64 sum on the DPP path (row_shr 1/2/4/8 = inclusive scan inside each row of 16, row_bcast 15 / 31 across the rows,
// total in lane 63, read back as a scalar): ~12 VALU instructions instead of the six dependent ds_bpermute round trips
// of __shfl_xor (20 of those per trip were this kernel's whole critical path)
#define MT_DPP_STEP(x, ctrl, rm) \
  (x) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rm), 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  MT_DPP_STEP(v, 0x111, 0xf);
  MT_DPP_STEP(v, 0x112, 0xf);
  MT_DPP_STEP(v, 0x114, 0xf);
  MT_DPP_STEP(v, 0x118, 0xf);
  MT_DPP_STEP(v, 0x142, 0xa);
  MT_DPP_STEP(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__global__ void __launch_bounds__(256) mlp_tail_softmax_kernel(
    const float* __restrict__ z1, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ w_out,
    const float* __restrict__ b_out, const float* __restrict__ labels, long P, int G, int C1, float lscale,
    double* __restrict__ loss_out, float* __restrict__ logit, float* __restrict__ dlogit_out, float* __restrict__ dy1,
    double* __restrict__ bn_partial, float* __restrict__ w_partial) {
  __shared__ double red[2][4][64];
  __shared__ float redw[4][64];
  __shared__ float redb[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool cv = lane < C1;
  const int c = cv ? lane : 0;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c], wo = cv ? w_out[c] : 0.f, bo = b_out[0];
  double s1 = 0.0, s2 = 0.0;
  float sw = 0.f, sb = 0.f, local = 0.f;
  // MT_NG groups per trip, every phase unrolled over all of them: the loads of 4 x G rows are in flight together and the
  // 4 x G wave sums are independent chains (one group at a time, the kernel took 30 us: a serial chain of load -> sum)
  constexpr int MT_NG = 4;
  for (long p0 = ((long)blockIdx.x * 4 + wave) * MT_NG; p0 < P; p0 += (long)gridDim.x * 4 * MT_NG) {
    float zz[MT_NG][MT_MAXG], lg[MT_NG][MT_MAXG], lb[MT_NG][MT_MAXG];
#pragma unroll
    for (int k = 0; k < MT_NG; ++k) {
      const long p = min(p0 + k, P - 1);        // (a group past the end repeats the last one; masked below)
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G) { zz[k][g] = z1[(p * G + g) * C1 + c]; lb[k][g] = labels[p * G + g]; }
    }
#pragma unroll
    for (int k = 0; k < MT_NG; ++k)
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G) lg[k][g] = wave_sum_dpp(fmaxf(zz[k][g] * sc + sh, 0.f) * wo) + bo;
#pragma unroll
    for (int k = 0; k < MT_NG; ++k) {
      const long p = p0 + k;
      const bool pv = p < P;
      float mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G) mx = fmaxf(mx, lg[k][g]);
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G) sum += __expf(lg[k][g] - mx);
      const float lse = mx + __logf(sum);
      int npos = 0;
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G && lb[k][g] == 1.0f) { ++npos; if (pv) local -= (lg[k][g] - lse); }
#pragma unroll
      for (int g = 0; g < MT_MAXG; ++g)
        if (g < G && pv) {
          const long b = p * G + g;
          const float sm = __expf(lg[k][g] - lse);
          const float dl = lscale * ((float)npos * sm - (lb[k][g] == 1.0f ? 1.0f : 0.0f));
          if (lane == 0) { logit[b] = lg[k][g]; if (dlogit_out) dlogit_out[b] = dl; }
          const float y = zz[k][g] * sc + sh;
          const float d = (cv && y > 0.f) ? wo * dl : 0.f;
          if (cv) dy1[b * C1 + c] = d;
          const float xh = (zz[k][g] - mu) * is;
          s1 += d; s2 += (double)d * xh;
          sw += fmaxf(y, 0.f) * dl;
          sb += dl;
        }
    }
  }
  // ONE double atomic per workgroup for the loss (one per wave -- 1 024 on one address -- took 20 us)
  __shared__ float redl[4];
  red[0][wave][lane] = s1; red[1][wave][lane] = s2; redw[wave][lane] = sw;
  if (lane == 0) { redb[wave] = sb; redl[wave] = local; }
  __syncthreads();
  if (threadIdx.x == 64) {
    const float l = (redl[0] + redl[1]) + (redl[2] + redl[3]);
    if (l != 0.f) atomicAdd(loss_out, (double)l * lscale);
  }
  if (threadIdx.x < C1) {
    const int t = threadIdx.x;
    bn_partial[((long)blockIdx.x * 2 + 0) * C1 + t] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
    bn_partial[((long)blockIdx.x * 2 + 1) * C1 + t] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    w_partial[(long)blockIdx.x * (C1 + 4) + t] = (redw[0][t] + redw[1][t]) + (redw[2][t] + redw[3][t]);
  }
  if (threadIdx.x == 0) w_partial[(long)blockIdx.x * (C1 + 4) + C1] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

static int mlp_tail_blocks(long P) {
  long b = (P + 15) / 16;          // four groups per wave and trip
  constexpr int cap = 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
extern "C" int clsr_mlp_tail_softmax_supported(int G, int C1) { return G >= 1 && G <= MT_MAXG && C1 >= 4 && C1 <= 64; }
extern "C" int clsr_mlp_tail_softmax_parts(long P) { return mlp_tail_blocks(P); }
// reference: the output layer of _fcn_net (base_model.py:695-708), _compute_data_loss "softmax" (base_model.py:215-235)
// and their gradients.  loss_out += -lscale * sum over the positives of log softmax_group(logit); dlogit_out optional.
extern "C" int clsr_mlp_tail_softmax(const float* z1, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* w_out, const float* b_out, const float* labels,
                                     long P, int G, int C1, float lscale, double* loss_out, float* logit,
                                     float* dlogit, float* dy1, double* bn_partial, float* w_partial, void* stream) {
  CLSR_CHECK_ARG(z1 && scale && shift && mean && invstd && w_out && b_out && labels && loss_out && logit && dy1 &&
                 bn_partial && w_partial && P > 0);
  CLSR_CHECK_SUPPORTED(clsr_mlp_tail_softmax_supported(G, C1));
  hipLaunchKernelGGL(mlp_tail_softmax_kernel, dim3(mlp_tail_blocks(P)), dim3(256), 0, (hipStream_t)stream, z1, scale,
                     shift, mean, invstd, w_out, b_out, labels, P, G, C1, lscale, loss_out, logit, dlogit, dy1,
                     bn_partial, w_partial);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ contrastive loss
// mode 1 (triplet): four hinge terms on element-wise squared distances; mode 0 (bpr): four softplus
// terms on dot products.  Every term is sum_b mask_b * term_b / sum_b mask_b, mask = len > threshold,
// times `weight`.  denom = sum_b mask_b comes from the host (it only depends on the fed lengths).
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(__expf(x)); }

__global__ void __launch_bounds__(64) contrastive_kernel(
    const float* __restrict__ L, const float* __restrict__ S, const float* __restrict__ M,
    const float* __restrict__ R, const int* __restrict__ seq_len, int len_stride, long Hn, int G, int D,
    int threshold, int mode, float margin, float weight, const float* __restrict__ denom_ptr,
    double* __restrict__ loss_out, float* __restrict__ dL, float* __restrict__ dS,
    float* __restrict__ dM, float* __restrict__ dR) {
  const int lane = threadIdx.x;
  const float coef = weight / denom_ptr[0];
  float loss_local = 0.f;
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    if (seq_len[h * len_stride] <= threshold) continue;
    if (mode != 1 && D > 64) {
      // bpr on wide embeddings: the four dot products run over all of D before any gradient can be formed;
      // the history-level vectors of up to 4 x 64 features stay in registers
      float l[4], m[4], r[4], gl[4], gm[4], gr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 64 * k + lane;
        const bool ok = c < D;
        l[k] = ok ? L[h * D + c] : 0.f; m[k] = ok ? M[h * D + c] : 0.f; r[k] = ok ? R[h * D + c] : 0.f;
        gl[k] = gm[k] = gr[k] = 0.f;
      }
      for (int gi = 0; gi < G; ++gi) {
        const long b = h * G + gi;
        float sv[4], p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = 64 * k + lane;
          sv[k] = c < D ? S[b * D + c] : 0.f;
          p1 += l[k] * (r[k] - m[k]); p2 += sv[k] * (m[k] - r[k]);
          p3 += m[k] * (sv[k] - l[k]); p4 += r[k] * (l[k] - sv[k]);
        }
        const float a1 = wave_sum(p1), a2 = wave_sum(p2), a3 = wave_sum(p3), a4 = wave_sum(p4);
        if (lane == 0) loss_local += softplusf_(a1) + softplusf_(a2) + softplusf_(a3) + softplusf_(a4);
        const float s1 = sigmoidf_(a1), s2 = sigmoidf_(a2), s3 = sigmoidf_(a3), s4 = sigmoidf_(a4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = 64 * k + lane;
          const float s = sv[k];
          gl[k] += s1 * (r[k] - m[k]) - s3 * m[k] + s4 * r[k];
          gm[k] += -s1 * l[k] + s2 * s + s3 * (s - l[k]);
          gr[k] += s1 * l[k] - s2 * s + s4 * (l[k] - s);
          if (c < D && dS) dS[b * D + c] += coef * (s2 * (m[k] - r[k]) + s3 * m[k] - s4 * r[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 64 * k + lane;
        if (c < D && dL) { dL[h * D + c] += coef * gl[k]; dM[h * D + c] += coef * gm[k]; dR[h * D + c] += coef * gr[k]; }
      }
      continue;
    }
    for (int c0 = 0; c0 < D; c0 += 64) {
      const int c = c0 + lane;
      const bool ok = c < D;
      const float l = ok ? L[h * D + c] : 0.f, m = ok ? M[h * D + c] : 0.f, r = ok ? R[h * D + c] : 0.f;
      float gl = 0.f, gm = 0.f, gr = 0.f;
      for (int gi = 0; gi < G; ++gi) {
        const long b = h * G + gi;
        const float s = ok ? S[b * D + c] : 0.f;
        float gs = 0.f;
        if (mode == 1) {
          if (ok) {
            const float dLM = (l - m) * (l - m), dLR = (l - r) * (l - r);
            const float dSM = (s - m) * (s - m), dSR = (s - r) * (s - r);
            const float t1 = dLM - dLR + margin, t2 = dSR - dSM + margin;
            const float t3 = dLM - dSM + margin, t4 = dSR - dLR + margin;
            if (t1 > 0.f) { loss_local += t1; gl += 2.f * (r - m); gm -= 2.f * (l - m); gr += 2.f * (l - r); }
            if (t2 > 0.f) { loss_local += t2; gs += 2.f * (m - r); gr -= 2.f * (s - r); gm += 2.f * (s - m); }
            if (t3 > 0.f) { loss_local += t3; gl += 2.f * (l - m); gs -= 2.f * (s - m); gm += 2.f * (s - l); }
            if (t4 > 0.f) { loss_local += t4; gs += 2.f * (s - r); gl -= 2.f * (l - r); gr += 2.f * (l - s); }
          }
        } else {
          // softplus(sum L*(R-M)), softplus(sum S*(M-R)), softplus(sum M*(S-L)), softplus(sum R*(L-S))
          const float a1 = wave_sum(l * (r - m)), a2 = wave_sum(s * (m - r));
          const float a3 = wave_sum(m * (s - l)), a4 = wave_sum(r * (l - s));
          if (c0 == 0 && lane == 0) loss_local += softplusf_(a1) + softplusf_(a2) + softplusf_(a3) + softplusf_(a4);
          const float s1 = sigmoidf_(a1), s2 = sigmoidf_(a2), s3 = sigmoidf_(a3), s4 = sigmoidf_(a4);
          gl += s1 * (r - m) - s3 * m + s4 * r;
          gs += s2 * (m - r) + s3 * m - s4 * r;
          gm += -s1 * l + s2 * s + s3 * (s - l);
          gr += s1 * l - s2 * s + s4 * (l - s);
        }
        if (ok && dS) dS[b * D + c] += coef * gs;
      }
      if (ok && dL) { dL[h * D + c] += coef * gl; dM[h * D + c] += coef * gm; dR[h * D + c] += coef * gr; }
    }
  }
  loss_local = wave_sum(loss_local);
  if (lane == 0 && loss_local != 0.f) atomicAdd(loss_out, (double)loss_local * coef);
}

extern "C" int clsr_contrastive(const float* L, const float* S, const float* M, const float* R,
                                const int* seq_len, int len_stride, long Hn, int G, int D, int threshold,
                                int mode, float margin, float weight, const float* denom_ptr,
                                double* loss_out, float* dL, float* dS, float* dM, float* dR,
                                void* stream) {
  CLSR_CHECK_ARG(L && S && M && R && seq_len && denom_ptr && loss_out && Hn > 0 && G > 0);
  CLSR_CHECK_ARG((dL && dS && dM && dR) || (!dL && !dS && !dM && !dR));
  CLSR_CHECK_SUPPORTED(mode == 1 || D <= 256);  // bpr keeps the history-level vectors in registers
  // a SMALL grid on purpose: this kernel runs on a side stream beside the first alpha-gate GEMM of the main chain; with
  // 4096 one-wave blocks the dispatcher was busy issuing them and the GEMM's 160 workgroups (20 us alone) took 83 us
  constexpr int cap = 1024;
  int blocks = Hn > cap ? cap : (int)Hn;
  hipLaunchKernelGGL(contrastive_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, L, S, M, R,
                     seq_len, len_stride, Hn, G, D, threshold, mode, margin, weight, denom_ptr, loss_out,
                     dL, dS, dM, dR);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ small row/column movers
// dst[n, dst_col0 + c] (=|+=) src[(n / row_div) * ld_src + src_col0 + c],  c < C
__global__ void copy_cols_kernel(const float* __restrict__ src, int ld_src, int src_col0, int row_div,
                                 long N, int C, float* __restrict__ dst, int ldd, int dst_col0,
                                 int accumulate) {
  const long total = N * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long n = e / C;
    const int c = (int)(e - n * C);
    const float v = src[(n / row_div) * ld_src + src_col0 + c];
    float* d = dst + n * ldd + dst_col0 + c;
    *d = accumulate ? *d + v : v;
  }
}

extern "C" int clsr_copy_cols(const float* src, int ld_src, int src_col0, int row_div, long N, int C,
                              float* dst, int ldd, int dst_col0, int accumulate, void* stream) {
  CLSR_CHECK_ARG(src && dst && N > 0 && C > 0 && row_div > 0);
  int blocks = clsr_cdiv(N * C, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     src_col0, row_div, N, C, dst, ldd, dst_col0, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// dst[h, dst_col0 + c] (=|+=) sum_{g<G} src[(h*G + g) * ld_src + src_col0 + c]
__global__ void group_sum_cols_kernel(const float* __restrict__ src, int ld_src, int src_col0, int G,
                                      long Hn, int C, float* __restrict__ dst, int ldd, int dst_col0,
                                      int accumulate) {
  const long total = Hn * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long h = e / C;
    const int c = (int)(e - h * C);
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += src[(h * G + g) * ld_src + src_col0 + c];
    float* d = dst + h * ldd + dst_col0 + c;
    *d = accumulate ? *d + s : s;
  }
}

extern "C" int clsr_group_sum_cols(const float* src, int ld_src, int src_col0, int G, long Hn, int C,
                                   float* dst, int ldd, int dst_col0, int accumulate, void* stream) {
  CLSR_CHECK_ARG(src && dst && Hn > 0 && C > 0 && G > 0);
  int blocks = clsr_cdiv(Hn * C, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(group_sum_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     src_col0, G, Hn, C, dst, ldd, dst_col0, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- short-term attention query (clsr.py:219): q[r] = [A[r / G, :Da] | B[r, :Db]] and its backward
// dA[h] += sum_g dq[(h, g), :Da],  dB[r] += dq[r, Da:]  -- one launch each instead of two strided column movers
__global__ void __launch_bounds__(256) query_concat_kernel(const float* __restrict__ A, int lda, int G,
                                                           const float* __restrict__ B, int ldb, int R, int Da, int Db,
                                                           float* __restrict__ out, int ldo) {
  const int QC = (Da + Db) >> 2, rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  if (ty >= rpb) return;
  const int c = 4 * q;
  for (int r = blockIdx.x * rpb + ty; r < R; r += gridDim.x * rpb) {
    const f32x4 v = c < Da ? ld4(A + (long)(r / G) * lda + c) : ld4(B + (long)r * ldb + (c - Da));
    st4(out + (long)r * ldo + c, v);
  }
}

extern "C" int clsr_query_concat(const float* A, int lda, int G, const float* B, int ldb, long R, int Da, int Db,
                                 float* out, int ldo, void* stream) {
  CLSR_CHECK_ARG(A && B && out && R > 0 && G > 0 && Da > 0 && Db > 0 && ldo >= Da + Db);
  CLSR_CHECK_SUPPORTED(Da % 4 == 0 && Db % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && Da + Db <= 1024 &&
                       R < (1L << 31));
  const int rpb = 256 / ((Da + Db) / 4);
  int blocks = clsr_cdiv(R, rpb * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(query_concat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A, lda, G, B, ldb, (int)R, Da,
                     Db, out, ldo);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

__global__ void __launch_bounds__(256) query_split_bwd_kernel(const float* __restrict__ dq, int ldq, int G, int Hn, int Da,
                                                              int Db, float* __restrict__ dA, int ldda,
                                                              float* __restrict__ dB, int lddb) {
  const int QC = (Da + Db) >> 2, rpb = 256 / QC;
  const int ty = threadIdx.x / QC, q = threadIdx.x - ty * QC;
  if (ty >= rpb) return;
  const int c = 4 * q;
  for (int h = blockIdx.x * rpb + ty; h < Hn; h += gridDim.x * rpb) {
    if (c < Da) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < G; ++g) s += ld4(dq + (long)(h * G + g) * ldq + c);
      float* d = dA + (long)h * ldda + c;
      st4(d, ld4(d) + s);
    } else {
      for (int g = 0; g < G; ++g) {
        float* d = dB + (long)(h * G + g) * lddb + (c - Da);
        st4(d, ld4(d) + ld4(dq + (long)(h * G + g) * ldq + c));
      }
    }
  }
}

extern "C" int clsr_query_split_bwd(const float* dq, int ldq, int G, long Hn, int Da, int Db, float* dA, int ldda,
                                    float* dB, int lddb, void* stream) {
  CLSR_CHECK_ARG(dq && dA && dB && Hn > 0 && G > 0 && Da > 0 && Db > 0 && ldq >= Da + Db);
  CLSR_CHECK_SUPPORTED(Da % 4 == 0 && Db % 4 == 0 && ldq % 4 == 0 && ldda % 4 == 0 && lddb % 4 == 0 && Da + Db <= 1024 &&
                       Hn * G < (1L << 31));
  const int rpb = 256 / ((Da + Db) / 4);
  int blocks = clsr_cdiv(Hn, rpb);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(query_split_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dq, ldq, G, (int)Hn, Da, Db,
                     dA, ldda, dB, lddb);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// out[e] = sa * a[e] + sb * b[e]
__global__ void axpby_kernel(float* __restrict__ out, const float* __restrict__ a, float sa,
                             const float* __restrict__ b, float sb, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = sa * a[e] + (b ? sb * b[e] : 0.f);
}

extern "C" int clsr_axpby(float* out, const float* a, float sa, const float* b, float sb, long n,
                          void* stream) {
  CLSR_CHECK_ARG(out && a && n > 0);
  int blocks = clsr_cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, a, sa, b, sb, n);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ------------------------------------------------------------------ attention layer-0 backward helpers
// z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp    (the re-associated first att_fcn layer)
// (1) dU[h,t,:] = sum_g dz0[(h,g),t,:]   dV[r,:] = sum_t dz0[r,t,:]
// One wave per history group; lane -> (t slot, 16-byte column chunk).  For every step the G rows of the
// group are loaded back to back (independent loads), summed in registers for dU (ONE store per element,
// no read-modify-write) and accumulated per row for dV.
#define ATT_MAXG 8
template <bool BIGG, bool H>
__global__ void __launch_bounds__(64) att_z0_bwd_reduce_kernel(const void* __restrict__ dz0, long Hn,
                                                               int G, int T, int C,
                                                               float* __restrict__ dU,
                                                               float* __restrict__ dV) {
  const int lane = threadIdx.x;
  const int QC = C >> 2, tpar = 64 / QC, ts = lane / QC, q = lane - ts * QC;
  __shared__ f32x4 red[ATT_MAXG][64];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    for (int g0 = 0; g0 < G; g0 += ATT_MAXG) {   // groups larger than ATT_MAXG rows: several passes
      const int gc = min(ATT_MAXG, G - g0);
      f32x4 accv[ATT_MAXG];
#pragma unroll
      for (int g = 0; g < ATT_MAXG; ++g) accv[g] = z4;
      if (ts < tpar) {
        for (int t = ts; t < T; t += tpar) {
          f32x4 d[ATT_MAXG];
#pragma unroll
          for (int g = 0; g < ATT_MAXG; ++g)
            d[g] = g < gc ? load4e<H>(dz0, ((h * G + g0 + g) * T + t) * C + 4 * q) : z4;
          f32x4 su = z4;
#pragma unroll
          for (int g = 0; g < ATT_MAXG; ++g) { su += d[g]; accv[g] += d[g]; }
          if (dU) {
            float* up = dU + (h * T + t) * C + 4 * q;
            st4(up, g0 == 0 ? su : ld4(up) + su);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < ATT_MAXG; ++g) red[g][lane] = accv[g];
      __syncthreads();
      if (ts == 0) {
#pragma unroll
        for (int g = 0; g < ATT_MAXG; ++g)
          if (g < gc) {
            f32x4 v = accv[g];
            for (int s2 = 1; s2 < tpar; ++s2) v += red[g][lane + s2 * QC];
            st4(dV + (h * G + g0 + g) * C + 4 * q, v);
          }
      }
      __syncthreads();
    }
  }
}

static int att_z0_bwd_reduce_impl(const void* dz0, int bf16, long Hn, int G, int T, int C, float* dU, float* dV,
                                  void* stream) {
  CLSR_CHECK_ARG(dz0 && dV && Hn > 0 && G > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(C % 4 == 0 && C <= 256);
  int blocks = Hn > 8192 ? 8192 : (int)Hn;
  if (bf16)
    hipLaunchKernelGGL((att_z0_bwd_reduce_kernel<false, true>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, dz0,
                       Hn, G, T, C, dU, dV);
  else
    hipLaunchKernelGGL((att_z0_bwd_reduce_kernel<false, false>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, dz0,
                       Hn, G, T, C, dU, dV);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_z0_bwd_reduce(const float* dz0, long Hn, int G, int T, int C, float* dU,
                                      float* dV, void* stream) {
  return att_z0_bwd_reduce_impl(dz0, 0, Hn, G, T, C, dU, dV, stream);
}

extern "C" int clsr_att_z0_bwd_reduce_h(const void* dz0, long Hn, int G, int T, int C, float* dU,
                                        float* dV, void* stream) {
  return att_z0_bwd_reduce_impl(dz0, 1, Hn, G, T, C, dU, dV, stream);
}

// (2) given daq = dz0 . Wp^T  [R*T, Q]:  da[h,t,:] = sum_g daq[(h,g),t,:] * q[(h,g),:]
//                                        dq[r,:]   = sum_t daq[r,t,:] * a[h,t,:]
// Same loop structure as (1): the G rows of a step are loaded together, da gets one store.
template <bool H>
__global__ void __launch_bounds__(64) att_prod_bwd_kernel(const void* __restrict__ daq, int ldd,
                                                          const float* __restrict__ a, int lda,
                                                          const float* __restrict__ q, int ldq, long Hn, int G,
                                                          int T, int Q, float* __restrict__ da, int ldda,
                                                          float* __restrict__ dq, int lddq, int acc_dq) {
  CLSR_CHAIN_PRIO();
  const int lane = threadIdx.x;
  const int QQ = Q >> 2, tpar = 64 / QQ, ts = lane / QQ, qq = lane - ts * QQ;
  __shared__ f32x4 red[ATT_MAXG][64];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long h = blockIdx.x; h < Hn; h += gridDim.x) {
    for (int g0 = 0; g0 < G; g0 += ATT_MAXG) {
      const int gc = min(ATT_MAXG, G - g0);
      f32x4 accq[ATT_MAXG], qv[ATT_MAXG];
#pragma unroll
      for (int g = 0; g < ATT_MAXG; ++g) {
        accq[g] = z4;
        qv[g] = (g < gc && ts < tpar) ? ld4(q + (h * G + g0 + g) * ldq + 4 * qq) : z4;
      }
      if (ts < tpar) {
        for (int t = ts; t < T; t += tpar) {
          const f32x4 av = ld4(a + (h * T + t) * lda + 4 * qq);
          f32x4 d[ATT_MAXG];
#pragma unroll
          for (int g = 0; g < ATT_MAXG; ++g)
            d[g] = g < gc ? load4e<H>(daq, ((h * G + g0 + g) * T + t) * ldd + 4 * qq) : z4;
          f32x4 su = z4;
#pragma unroll
          for (int g = 0; g < ATT_MAXG; ++g) { su += d[g] * qv[g]; accq[g] += d[g] * av; }
          float* ap = da + (h * T + t) * ldda + 4 * qq;
          st4(ap, g0 == 0 ? su : ld4(ap) + su);
        }
      }
#pragma unroll
      for (int g = 0; g < ATT_MAXG; ++g) red[g][lane] = accq[g];
      __syncthreads();
      if (ts == 0) {
#pragma unroll
        for (int g = 0; g < ATT_MAXG; ++g)
          if (g < gc) {
            f32x4 v = accq[g];
            for (int s2 = 1; s2 < tpar; ++s2) v += red[g][lane + s2 * QQ];
            float* qp = dq + (h * G + g0 + g) * lddq + 4 * qq;
            st4(qp, acc_dq ? ld4(qp) + v : v);
          }
      }
      __syncthreads();
    }
  }
}

static int att_prod_bwd_impl(const void* daq, int bf16, int ldd, const float* a, int lda, const float* q, int ldq,
                             long Hn, int G, int T, int Q, float* da, int ldda, float* dq, int lddq,
                             int accumulate_dq, void* stream) {
  CLSR_CHECK_ARG(daq && a && q && da && dq && Hn > 0 && G > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(Q % 4 == 0 && Q <= 256 && ldd % 4 == 0 && lda % 4 == 0 && ldq % 4 == 0 && ldda % 4 == 0 &&
                       lddq % 4 == 0);
  CLSR_CHECK_ARG(ldd >= Q && lda >= Q && ldq >= Q && ldda >= Q && lddq >= Q);
  int blocks = Hn > 8192 ? 8192 : (int)Hn;
  if (bf16)
    hipLaunchKernelGGL(att_prod_bwd_kernel<true>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, daq, ldd, a, lda, q,
                       ldq, Hn, G, T, Q, da, ldda, dq, lddq, accumulate_dq);
  else
    hipLaunchKernelGGL(att_prod_bwd_kernel<false>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, daq, ldd, a, lda, q,
                       ldq, Hn, G, T, Q, da, ldda, dq, lddq, accumulate_dq);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_prod_bwd_ld(const float* daq, int ldd, const float* a, int lda, const float* q, int ldq,
                                    long Hn, int G, int T, int Q, float* da, int ldda, float* dq, int lddq,
                                    int accumulate_dq, void* stream) {
  return att_prod_bwd_impl(daq, 0, ldd, a, lda, q, ldq, Hn, G, T, Q, da, ldda, dq, lddq, accumulate_dq, stream);
}

extern "C" int clsr_att_prod_bwd_h(const void* daq, int ldd, const float* a, int lda, const float* q, int ldq,
                                   long Hn, int G, int T, int Q, float* da, int ldda, float* dq, int lddq,
                                   int accumulate_dq, void* stream) {
  return att_prod_bwd_impl(daq, 1, ldd, a, lda, q, ldq, Hn, G, T, Q, da, ldda, dq, lddq, accumulate_dq, stream);
}

extern "C" int clsr_att_prod_bwd(const float* daq, const float* a, const float* q, long Hn, int G, int T,
                                 int Q, float* da, float* dq, void* stream) {
  return clsr_att_prod_bwd_ld(daq, Q, a, Q, q, Q, Hn, G, T, Q, da, Q, dq, Q, 0, stream);
}
