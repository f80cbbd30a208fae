// Exact-mode backward through the second attention layer and the batch-norm + ReLU below it (gfx950, fp32 MFMA):
//   x = dz1[m, :C1] = a1*dy1 + a2*z1 + a3      BN-1 backward; dy1 = ds[m] * w_out where relu(bn1(z1)) > 0, recomputed
//                                               from z1 and the score gradient ds in the GEMM prologue
//   dh0 = x . W1^T,  dy0 = dh0 where relu(bn0(z0)) > 0
//   pass 1 (coef0 == NULL): per-block partial sums of dy0 and dy0 * xhat0 -> stats   (nothing else is written)
//   pass 2 (coef0 given):   dz0[m, :C0] = c1*dy0 + c2*z0 + c3 (the complete BN-0 backward); dz1 stored for the weight gradient
// (reference: tf.gradients through _fcn_net, base_model.py:664-706).  The fp32 twin of clsr_hgemm_att_l1_bwd
// (csrc/hgemm.hip): TWO passes over (z1, z0) instead of  dy1-apply sweep -> GEMM that stores dy0 -> bn-apply sweep:
// 1.48 GB instead of 2.13 GB of traffic at 1M positions and two launches fewer on the dependent chain; the 6.5 GFLOP
// GEMM that is done twice costs less than the 0.65 GB it saves.
//
// Same "features x positions" orientation as pgemm_fast_kernel: one v_mfma_f32_16x16x4_f32 tile is
// D[16 out-features][16 positions]; A = W1^T rows from LDS, B = the prologue result of the lane's position (MFMA #r of
// the 16-wide k-chunk kk uses feature 16kk + 4g + r: one float4 of z1 per chunk); a wave owns 32 positions; ALL operand
// loads of a wave-tile (z1 chunks, z0 tiles) are issued before its first MFMA.
#include "common.h"
#include "clsr_hip.h"

struct L1BwdArgs {
  const float* z1; int ldz1;
  const float* ds;
  const float* pv[4];            // scale1, shift1, w_out, coef1 (a1 | a2 | a3, stride C1)
  const float* Wt; int Kp;       // packed W1^T (clsr_pack_batch): row n = out feature (C0 rows), K = C1
  const float* z0; int ldz0;
  const float* ev[5];            // scale0, shift0, then mean0, invstd0 (pass 1) | coef0 c1|c2|c3 stride C0 (pass 2)
  float* dz1; int lddz1;
  float* dz0; int lddz0;
  double* stats;
  int M, C1, C0;
};

template <int OT, int KC, bool APPLY>
__global__ void __launch_bounds__(256, OT <= 5 ? 2 : 1) att_l1_bwd_kernel(L1BwdArgs a) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NR = 16 * OT, KCP = 16 * KC;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int Kp = a.Kp;
  float* Wl = lds;                               // [NR][Kp]
  float* ptab = Wl + (size_t)NR * Kp;            // [5][KCP]: sc1, sh1, p = a1 * w_out, a2, a3
  float* etab = ptab + 5 * KCP;                  // [5][NR]:  sc0, sh0, (mean0, invstd0, -) | (c1, c2, c3)
  double* red = reinterpret_cast<double*>(etab + 5 * NR);   // [4 waves][2][NR]  (pass 1)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  {
    const int Kq = Kp >> 2;
    const int nrows = 16 * ((a.C0 + 15) >> 4);
    for (int e = tid; e < NR * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      reinterpret_cast<f32x4*>(Wl)[e] = row < nrows ? ld4(a.Wt + (long)row * Kp + 4 * c) : zero4;
    }
    for (int e = tid; e < 5 * KCP; e += 256) {
      const int which = e / KCP, k = e - which * KCP;
      float v = 0.f;
      if (k < a.C1) {
        if (which < 2) v = a.pv[which][k];
        else if (which == 2) v = a.pv[3][k] * a.pv[2][k];
        else v = a.pv[3][(which - 2) * a.C1 + k];
      }
      ptab[e] = v;
    }
    for (int e = tid; e < 5 * NR; e += 256) {
      const int which = e / NR, n = e - which * NR;
      float v = 0.f;
      if (n < a.C0) {
        if (which < 2) v = a.ev[which][n];
        else if (!APPLY) v = which < 4 ? a.ev[which][n] : 0.f;
        else v = a.ev[2][(which - 2) * a.C0 + n];
      }
      etab[e] = v;
    }
    if (!APPLY) for (int e = tid; e < 4 * 2 * NR; e += 256) red[e] = 0.0;
  }
  __syncthreads();

  float fsum[APPLY ? 1 : OT][4], fsq[APPLY ? 1 : OT][4];
  int pending = 0;
  if (!APPLY) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) { fsum[ot][r] = 0.f; fsq[ot][r] = 0.f; }
  }
  auto flush = [&]() {
    if (APPLY) return;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s_ = row16_sum(fsum[ot][r]);
        const float q_ = row16_sum(fsq[ot][r]);
        if (j == 0) {
          red[(wave * 2 + 0) * NR + 16 * ot + 4 * g + r] += (double)s_;
          red[(wave * 2 + 1) * NR + 16 * ot + 4 * g + r] += (double)q_;
        }
        fsum[ot][r] = 0.f;
        fsq[ot][r] = 0.f;
      }
  };
  bool nok[OT];
  int ncl[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    nok[ot] = 16 * ot + 4 * g < a.C0;
    ncl[ot] = nok[ot] ? 16 * ot + 4 * g : 0;
  }

  // XCD-aware tile order (see pgemm_fast_kernel)
  const int ntiles = (a.M + 31) >> 5;
  const int nb = gridDim.x;
  const int nx = nb >= 8 ? 8 : 1;
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int nslots = (nb - xcd + nx - 1) / nx;
  const int chunk = (ntiles + nx - 1) / nx;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const float* ldsA = Wl + (long)j * Kp + 4 * g;   // + ot*16*Kp + kk*16

  for (int tile = xcd * chunk + slot * 4 + wave; tile < t_end; tile += nslots * 4) {
    long mrow[2];
    bool valid[2];
    float dsv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int m = tile * 32 + s * 16 + j;
      valid[s] = m < a.M;
      mrow[s] = valid[s] ? m : a.M - 1;
      dsv[s] = a.ds[mrow[s]];
    }
    f32x4 zr[KC][2], ezr[2][OT];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int kc = 16 * kk + 4 * g < a.C1 ? 16 * kk + 4 * g : 0;
#pragma unroll
      for (int s = 0; s < 2; ++s) zr[kk][s] = ld4(a.z1 + mrow[s] * a.ldz1 + kc);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) ezr[s][ot] = ld4(a.z0 + mrow[s] * a.ldz0 + ncl[ot]);
    __builtin_amdgcn_sched_barrier(0);   // every load of the tile is in flight before the first one is consumed

    f32x4 acc[2][OT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) acc[s][ot] = zero4;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int kcol = 16 * kk + 4 * g;
      const bool ink = kcol < a.C1;
      const f32x4 sc = ld4(ptab + kcol), sh = ld4(ptab + KCP + kcol), pp = ld4(ptab + 2 * KCP + kcol);
      const f32x4 a2 = ld4(ptab + 3 * KCP + kcol), a3 = ld4(ptab + 4 * KCP + kcol);
      f32x4 b[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 zz = zr[kk][s];
        const f32x4 y = zz * sc + sh;
        f32x4 x = a2 * zz + a3;
        x.x += y.x > 0.f ? pp.x * dsv[s] : 0.f;
        x.y += y.y > 0.f ? pp.y * dsv[s] : 0.f;
        x.z += y.z > 0.f ? pp.z * dsv[s] : 0.f;
        x.w += y.w > 0.f ? pp.w * dsv[s] : 0.f;
        b[s] = ink ? x : zero4;
        if (APPLY && a.dz1 && ink && valid[s]) st4(a.dz1 + mrow[s] * a.lddz1 + kcol, x);
      }
      f32x4 wt[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) wt[ot] = ld4(ldsA + (long)ot * 16 * Kp + kk * 16);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].x, b[0].x); MFMA4(acc[1][ot], wt[ot].x, b[1].x); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].y, b[0].y); MFMA4(acc[1][ot], wt[ot].y, b[1].y); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].z, b[0].z); MFMA4(acc[1][ot], wt[ot].z, b[1].z); }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { MFMA4(acc[0][ot], wt[ot].w, b[0].w); MFMA4(acc[1][ot], wt[ot].w, b[1].w); }
    }

#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int nl = 16 * ot + 4 * g;
        const bool ok = valid[s] && nok[ot];
        const f32x4 zz = ezr[s][ot];
        const f32x4 y = zz * ld4(etab + nl) + ld4(etab + NR + nl);
        f32x4 v = acc[s][ot];
        v.x = (y.x > 0.f && ok) ? v.x : 0.f;
        v.y = (y.y > 0.f && ok) ? v.y : 0.f;
        v.z = (y.z > 0.f && ok) ? v.z : 0.f;
        v.w = (y.w > 0.f && ok) ? v.w : 0.f;
        if (APPLY) {
          if (ok) st4(a.dz0 + mrow[s] * a.lddz0 + nl,
                      ld4(etab + 2 * NR + nl) * v + ld4(etab + 3 * NR + nl) * zz + ld4(etab + 4 * NR + nl));
        } else {
          const f32x4 w2 = (zz - ld4(etab + 2 * NR + nl)) * ld4(etab + 3 * NR + nl);
          fsum[ot][0] += v.x; fsq[ot][0] = fmaf(v.x, w2.x, fsq[ot][0]);
          fsum[ot][1] += v.y; fsq[ot][1] = fmaf(v.y, w2.y, fsq[ot][1]);
          fsum[ot][2] += v.z; fsq[ot][2] = fmaf(v.z, w2.z, fsq[ot][2]);
          fsum[ot][3] += v.w; fsq[ot][3] = fmaf(v.w, w2.w, fsq[ot][3]);
        }
      }
#ifndef CLSR_STAT_FLUSH_N
#define CLSR_STAT_FLUSH_N 8
#endif
    if (!APPLY && ++pending == CLSR_STAT_FLUSH_N) {   // fp32 per-lane partials of at most 16 values between flushes
      flush();
      pending = 0;
    }
  }

  if (!APPLY) {
    if (pending) flush();
    __syncthreads();
    for (int e = tid; e < 2 * NR; e += 256) {
      const int which = e / NR, c = e - which * NR;
      if (c < a.C0) {
        double t = 0.0;
        for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * NR + c];
        a.stats[((long)blockIdx.x * 2 + which) * a.C0 + c] = t;
      }
    }
  }
}

static int l1b_grid(int M) {
  int gx = clsr_cdiv(clsr_cdiv(M, 32), 4);
  if (gx > 1024) gx = 1024;
  return gx < 1 ? 1 : gx;
}

// 1 when clsr_att_l1_bwd handles this shape (otherwise: clsr_att_dy1_apply + clsr_pgemm_bnbwd + clsr_bn_bwd_apply)
extern "C" int clsr_att_l1_bwd_supported(int C1, int C0) {
  return C1 >= 4 && C1 <= 48 && C0 >= 4 && C0 <= 128 && C1 % 4 == 0 && C0 % 4 == 0;
}
extern "C" int clsr_att_l1_bwd_stats_parts(int M) { return l1b_grid(M); }

template <int OT, int KC>
static int l1b_launch(const L1BwdArgs& a, bool apply, hipStream_t stream) {
  size_t shmem = ((size_t)16 * OT * a.Kp + 5 * 16 * KC + 5 * 16 * OT) * sizeof(float) + (size_t)4 * 2 * 16 * OT * 8;
  dim3 grid(l1b_grid(a.M));
  if (apply) {
    auto kernel = att_l1_bwd_kernel<OT, KC, true>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  } else {
    auto kernel = att_l1_bwd_kernel<OT, KC, false>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  }
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_l1_bwd(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                               const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0, int ldz0,
                               const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                               const float* coef0, float* dz1, int lddz1, float* dz0, int lddz0, double* stats, int M,
                               int C1, int C0, void* stream) {
  CLSR_CHECK_ARG(z1 && ds && scale1 && shift1 && w_out && coef1 && Wt && z0 && scale0 && shift0 && M > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l1_bwd_supported(C1, C0));
  CLSR_CHECK_ARG(coef0 ? (dz0 && lddz0 >= C0) : (mean0 && invstd0 && stats));
  CLSR_CHECK_ARG(ldz1 >= C1 && ldz0 >= C0 && Kp >= 16 * clsr_cdiv(C1, 16) && (!dz1 || lddz1 >= C1));
  CLSR_CHECK_SUPPORTED(ldz1 % 4 == 0 && ldz0 % 4 == 0 && Kp % 4 == 0 && (!dz0 || lddz0 % 4 == 0) && (!dz1 || lddz1 % 4 == 0) &&
                       ((uintptr_t)z1 % 16) == 0 && ((uintptr_t)z0 % 16) == 0);
  L1BwdArgs a = {};
  a.z1 = z1; a.ldz1 = ldz1; a.ds = ds; a.pv[0] = scale1; a.pv[1] = shift1; a.pv[2] = w_out; a.pv[3] = coef1;
  a.Wt = Wt; a.Kp = Kp; a.z0 = z0; a.ldz0 = ldz0; a.ev[0] = scale0; a.ev[1] = shift0;
  a.M = M; a.C1 = C1; a.C0 = C0; a.stats = stats;
  const bool apply = coef0 != nullptr;
  if (apply) { a.ev[2] = coef0; a.dz1 = dz1; a.lddz1 = lddz1; a.dz0 = dz0; a.lddz0 = lddz0; }
  else { a.ev[2] = mean0; a.ev[3] = invstd0; }
  hipStream_t s = (hipStream_t)stream;
  const int kc = clsr_cdiv(C1, 16), ot = clsr_cdiv(C0, 16);
#define L1B_GO(O, K) if (ot <= O && kc == K) return l1b_launch<O, K>(a, apply, s)
  L1B_GO(3, 1); L1B_GO(3, 2); L1B_GO(3, 3);
  L1B_GO(5, 1); L1B_GO(5, 2); L1B_GO(5, 3);
  L1B_GO(8, 1); L1B_GO(8, 2); L1B_GO(8, 3);
#undef L1B_GO
  return CLSR_OK;
}
