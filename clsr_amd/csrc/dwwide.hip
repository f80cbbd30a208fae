// Weight gradients of WIDE layers: dW[k, n] = sum_m X[m, k] * dY[m, n], db[n] = sum_m dY[m, n]  (gfx950, fp32 MFMA).
//
// Reference: the gradients tf.gradients builds for the input-side and hidden-to-hidden kernels of GRUCell / Time4LSTMCell
// (models/sequential/rnn_cell_implement.py:129-298, tf.nn.rnn_cell.GRUCell at clsr.py:160-168, 229-237) at the layer sizes
// of BASELINE configs[4] (hidden 128, embedding 128: K, N in 128 .. 1 536 over M = 204 800 positions).
//
// Why a second dW kernel: pgemm_dw_kernel (csrc/linear.hip) gives every wave an 80 x 80 accumulator chunk of the output
// -- the reference's default widths are 40 .. 80 -- and walks wider outputs as a grid of such chunks, every chunk
// re-reading its X and dY columns: the 128 x 1 536 input-side gradient of the catalogue step read dPin twice and hist
// twenty times, and the one multi-job launch of the encoders' weight gradients took 4.4 ms, the last third of the
// 13 ms step (profiles/r04_catalogue_step_timeline.txt).  Here a workgroup owns a 128 x 128 output tile (its four waves
// 64 x 64 each: 16 accumulator tiles) over a contiguous range of positions; 32-position stages of both operands are
// double-buffered through LDS with the next stage's loads in flight behind the MFMAs, one barrier per stage.  X is read
// N / 128 times, dY K / 128 times.  The position ranges' partial tiles go to a workspace [S][K][N] (+ [S][N] column sums)
// and are added in range order by a second launch: no float atomics, bit-identical results run to run.
//
// MFMA tile D[16 k][16 n] += A[16 k][4 m] . B[4 m][16 n]: lane (i, g) supplies X[4s + g][k0 + i] and dY[4s + g][n0 + i]
// for the four positions 4s .. 4s + 3 of MFMA step s; the row stride of the LDS stages is 144 floats (= 16 mod 32 banks:
// the two position rows a half-wave reads sit in disjoint banks).
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

#define DWW_ST 144
#define DWW_STAGE (32 * DWW_ST)

struct DwwArgs {
  const float* Xmul; int ldmul;          // optional: X[m, k] * Xmul[m, k] (the candidate kernel of a GRU: r * h)
  const float* X; int ldx; const float* dY; int ldy; float* part; float* bpart; int M, K, N, mper, S;
};

// X3: the products as split-bf16 sums xh.yh + xh.yl + xl.yh on v_mfma_f32_16x16x32_bf16 (fp32 accumulation, 2^-16 relative per
// term, like every other weight gradient of the parity mode's default): a 32-position stage is ONE K = 32 chunk -- a lane
// gathers the eight positions 8g .. 8g + 7 of its feature from the LDS stage (the same 64 ds_read_b32 per wave and stage as
// the fp32 form), splits them, and issues 48 bf16 MFMAs of ~16 cycles where the fp32 form issues 128 of 32: the fp32 kernel
// ran the matrix pipe 70 % busy at 0.56 of its peak (profiles/r05_catalogue_pmc.md), this one is bound by its operand traffic.
// NP = 0: fp32-input MFMAs; NP = 2: the two-piece sums above (precision="fp32x3"); NP = 3: three pieces per operand, the six products
// whose piece indices sum to <= 2, smallest first (2^-23 relative per term: fp32 accuracy, the form of precision="fp32" -- 96 bf16
// MFMAs of 16 cycles per stage and wave where the fp32-input form issues 128 of 32)
template <int NP>
__global__ void __launch_bounds__(256, 2) dw_wide_kernel(DwwArgs a) {
  constexpr bool X3 = NP > 0;
  extern __shared__ __attribute__((aligned(16))) float dww_lds[];
  float* Xs = dww_lds;                       // [2][32][DWW_ST]
  float* Ys = dww_lds + 2 * DWW_STAGE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int k0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int wk = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const long m0 = (long)blockIdx.x * a.mper;
  const long m1 = m0 + a.mper < a.M ? m0 + a.mper : a.M;
  // staging: thread -> (position tid / 32 + 8 q, 16-byte chunk tid % 32) of both operands
  const int sr = tid >> 5, sc = (tid & 31) * 4;
  const bool kin = k0 + sc < a.K, nin = n0 + sc < a.N;       // (K % 4 == N % 4 == 0)
  f32x4 xv[4], yv[4];
  auto fetch = [&](long m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long r = m + sr + 8 * q;
      xv[q] = (kin && r < m1) ? ld4(a.X + r * a.ldx + k0 + sc) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.Xmul && kin && r < m1) xv[q] *= ld4(a.Xmul + r * a.ldmul + k0 + sc);
      yv[q] = (nin && r < m1) ? ld4(a.dY + r * a.ldy + n0 + sc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st4(Xs + b * DWW_STAGE + (sr + 8 * q) * DWW_ST + sc, xv[q]);
      st4(Ys + b * DWW_STAGE + (sr + 8 * q) * DWW_ST + sc, yv[q]);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_b = a.bpart && blockIdx.y == 0 && wk == 0;
  const int nst = (int)((m1 - m0 + 31) / 32);
  if (nst > 0) {
    fetch(m0);
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int b = s & 1;
    if (s + 1 < nst) fetch(m0 + 32L * (s + 1));
    const float* xs = Xs + b * DWW_STAGE + wk + i;
    const float* ys = Ys + b * DWW_STAGE + wn + i;
    if (X3) {
      constexpr int PC = NP > 0 ? NP : 1;
      bf16x8 ap[PC][4], bp[PC][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x8 xv, yv8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xv[e] = xs[(8 * g + e) * DWW_ST + 16 * t];
          yv8[e] = ys[(8 * g + e) * DWW_ST + 16 * t];
        }
        if (want_b) bs[t] += ((yv8[0] + yv8[1]) + (yv8[2] + yv8[3])) + ((yv8[4] + yv8[5]) + (yv8[6] + yv8[7]));
#pragma unroll
        for (int q = 0; q < PC; ++q) {
          ap[q][t] = to_h(xv);
          bp[q][t] = to_h(yv8);
          if (q + 1 < PC) { xv -= to_f(ap[q][t]); yv8 -= to_f(bp[q][t]); }
        }
      }
      // every product of pieces whose indices sum to <= PC - 1, the smallest terms first
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int sum = PC - 1; sum >= 0; --sum)
#pragma unroll
          for (int qa = sum; qa >= 0; --qa)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) HMFMA(acc[kt][nt], ap[qa][kt], bp[sum - qa][nt]);
      }
    } else
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      float xa[4], yb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        xa[t] = xs[(4 * st + g) * DWW_ST + 16 * t];
        yb[t] = ys[(4 * st + g) * DWW_ST + 16 * t];
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) MFMA4(acc[kt][nt], xa[kt], yb[nt]);
      if (want_b) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bs[nt] += yb[nt];
      }
    }
    if (s + 1 < nst) stash(b ^ 1);
    __syncthreads();
  }
  // D lane (j = i, g): rows k = 16 kt + 4 g + {0..3}, column n = 16 nt + j
  float* P = a.part + (long)blockIdx.x * a.K * a.N;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn + 16 * nt + i;
      const int k = k0 + wk + 16 * kt + 4 * g;
      if (n < a.N) {
        if (k + 0 < a.K) P[(long)(k + 0) * a.N + n] = acc[kt][nt].x;
        if (k + 1 < a.K) P[(long)(k + 1) * a.N + n] = acc[kt][nt].y;
        if (k + 2 < a.K) P[(long)(k + 2) * a.N + n] = acc[kt][nt].z;
        if (k + 3 < a.K) P[(long)(k + 3) * a.N + n] = acc[kt][nt].w;
      }
    }
  }
  if (want_b) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float v = bs[nt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn + 16 * nt + i;
      if (g == 0 && n < a.N) a.bpart[(long)blockIdx.x * a.N + n] = v;
    }
  }
}

// dW[k, n] (=|+=) sum over the S position ranges of part[s][k][n]; db likewise.  A workgroup owns 16 consecutive output quads;
// its 16 thread slices each add the ranges s = slice, slice + 16, ... (eight loads in flight), the slices are then combined in
// slice order through LDS: a fixed order for given S -- bit-identical results run to run.  (Round 4: one thread per quad walked
// all S ranges -- 16 workgroups for a 128 x 128 gradient, 200-320 us in the catalogue step's tail, as long as the product.)
__global__ void __launch_bounds__(256) dw_wide_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart,
                                                             int S, int K, int N, float* __restrict__ dW, int ldw,
                                                             float* __restrict__ db, int accumulate) {
  __shared__ f32x4 red[16][16];
  const long KN = (long)K * N;
  const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long e = ((long)blockIdx.x * 16 + ql) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (e < KN) {
    for (int s0 = sl; s0 < S; s0 += 16 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + 16 * u;
        v[u] = s < S ? ld4(part + (long)s * KN + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
  }
  red[sl][ql] = acc;
  __syncthreads();
  if (sl == 0 && e < KN) {
    f32x4 v = red[0][ql];
#pragma unroll
    for (int u = 1; u < 16; ++u) v += red[u][ql];
    const long k = e / N;
    const int n = (int)(e - k * N);
    float* o = dW + k * ldw + n;
    if (accumulate) { v.x += o[0]; v.y += o[1]; v.z += o[2]; v.w += o[3]; }
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;          // (views of the flat gradient buffer: no 16-byte alignment)
  }
  // bias sums: the first ceil(N / 16) workgroups take 16 columns each, the same slice scheme (one thread per column walked all
  // S ranges before: 280 us at S = 493)
  if (db && bpart && (long)blockIdx.x * 16 < N) {
    __syncthreads();
    float* redb = reinterpret_cast<float*>(red);      // [16][16]
    const int n = blockIdx.x * 16 + ql;
    float v = 0.f;
    if (n < N) {
      for (int s0 = sl; s0 < S; s0 += 16 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s_ = s0 + 16 * u;
          t[u] = s_ < S ? bpart[(long)s_ * N + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
    }
    redb[sl * 16 + ql] = v;
    __syncthreads();
    if (sl == 0 && n < N) {
      float t = redb[ql];
#pragma unroll
      for (int u = 1; u < 16; ++u) t += redb[u * 16 + ql];
      db[n] = accumulate ? db[n] + t : t;
    }
  }
}

static int dww_parts(long M, int K, int N) {
  const int tiles = clsr_cdiv(K, 128) * clsr_cdiv(N, 128);
  int S = 512 / tiles;        // (~512 workgroups: one round at two per CU; fewer partial tiles to add up afterwards)
  if (S < 8) S = 8;
  // (capped at 128 until round 5: a one-tile gradient -- hidden-side kernels of the 128-wide encoders -- then ran 128 workgroups
  //  of 50 stages each on 256 CUs, 200 us for 210 MB of operands, one after the other at the very end of the catalogue step)
  if (S > 512) S = 512;
  const long per = (M + 31) / 32;          // stages
  if (S > per) S = (int)per;
  return S < 1 ? 1 : S;
}
extern "C" int clsr_pgemm_dw_wide_supported(long M, int K, int N) {
  return M >= 32768 && M < (1L << 31) && K >= 96 && N >= 96 && K % 4 == 0 && N % 4 == 0 && (long)K * N < (1L << 28);
}
extern "C" int clsr_pgemm_dw_wide_parts(long M, int K, int N) { return dww_parts(M, K, N); }
extern "C" long clsr_pgemm_dw_wide_workspace_floats(long M, int K, int N) {
  return (long)dww_parts(M, K, N) * ((long)K * N + N);
}

// dW [K, N] (row stride ldw) and optionally db [N] from X [M, K] (row stride ldx; times Xmul [M, K] element-wise when
// given) and dY [M, N] (row stride ldy): two
// launches on ``stream`` (partial tiles, then their sum in range order).  accumulate != 0: added to dW / db.
static int dw_wide_any(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                       int N, float* workspace, float* dW, int ldw, float* db, int accumulate, int pieces, void* stream) {
  CLSR_CHECK_ARG(X && dY && workspace && dW && ldx >= K && ldy >= N && ldw >= N);
  CLSR_CHECK_SUPPORTED(!Xmul || (ldmul >= K && ldmul % 4 == 0 && ((uintptr_t)Xmul % 16) == 0));
  CLSR_CHECK_SUPPORTED(clsr_pgemm_dw_wide_supported(M, K, N));
  CLSR_CHECK_SUPPORTED(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0 &&
                       ((uintptr_t)workspace % 16) == 0);
  DwwArgs a;
  a.Xmul = Xmul; a.ldmul = ldmul;
  a.X = X; a.ldx = ldx; a.dY = dY; a.ldy = ldy; a.M = (int)M; a.K = K; a.N = N;
  a.S = dww_parts(M, K, N);
  a.mper = (int)(((M + a.S - 1) / a.S + 31) / 32 * 32);
  a.S = (int)((M + a.mper - 1) / a.mper);          // (ranges that exist after rounding the range length up to whole stages)
  a.part = workspace;
  a.bpart = db ? workspace + (long)dww_parts(M, K, N) * K * N : nullptr;
  const size_t shmem = (size_t)4 * DWW_STAGE * sizeof(float);
  const dim3 grid(a.S, clsr_cdiv(K, 128), clsr_cdiv(N, 128));
  if (pieces == 0) {
    CLSR_HIP(hipFuncSetAttribute((const void*)dw_wide_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(dw_wide_kernel<0>, grid, dim3(256), shmem, (hipStream_t)stream, a);
  } else if (pieces == 2) {
    CLSR_HIP(hipFuncSetAttribute((const void*)dw_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(dw_wide_kernel<2>, grid, dim3(256), shmem, (hipStream_t)stream, a);
  } else {
    CLSR_HIP(hipFuncSetAttribute((const void*)dw_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(dw_wide_kernel<3>, grid, dim3(256), shmem, (hipStream_t)stream, a);
  }
  CLSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(dw_wide_reduce_kernel, dim3(clsr_cdiv((long)K * N / 4, 16)), dim3(256), 0, (hipStream_t)stream,
                     a.part, a.bpart, a.S, K, N, dW, ldw, db, accumulate);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_pgemm_dw_wide(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                                  int N, float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream) {
  return dw_wide_any(X, ldx, Xmul, ldmul, dY, ldy, M, K, N, workspace, dW, ldw, db, accumulate, 0, stream);
}
// the same with the products as split-bf16 sums (2^-16 relative per term; bias sums stay exact fp32 sums)
extern "C" int clsr_pgemm_dw_wide_x3(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                                     int N, float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream) {
  return dw_wide_any(X, ldx, Xmul, ldmul, dY, ldy, M, K, N, workspace, dW, ldw, db, accumulate, 2, stream);
}
// ... as three-piece sums (2^-23 relative per term: fp32 accuracy)
extern "C" int clsr_pgemm_dw_wide_x6(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                                     int N, float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream) {
  return dw_wide_any(X, ldx, Xmul, ldmul, dY, ldy, M, K, N, workspace, dW, ldw, db, accumulate, 3, stream);
}
