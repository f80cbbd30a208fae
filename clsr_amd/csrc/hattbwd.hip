// Speed-mode backward of the re-associated first attention layer (gfx950, bf16 MFMA):
//   z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp          (reference clsr.py:368-370, base_model.py:664)
// given dz0 (bf16, [R*T, A0]) ONE kernel produces all four reductions of that layer
//   da[h,t,:] = sum_g (dz0[(h,g),t,:] . Wp^T) * q[(h,g),:]      dq[r,:] = sum_t (dz0[r,t,:] . Wp^T) * a[h,t,:]
//   dU[h,t,:] = sum_g dz0[(h,g),t,:]                            dV[r,:] = sum_t dz0[r,t,:]
// and, when the packed Wu^T is given, the U-path share of da as well: da += dU . Wu^T  (a second set of MFMAs that
// accumulates across the rows of a group; saves the separate back-propagating product over the Hn*T positions).
// The [R*T, Q] product daq = dz0 . Wp^T is never written: it used to cost a GEMM that stored it (bf16), a sweep that
// read it back (att_prod_bwd) and a second sweep over dz0 (att_z0_bwd_reduce) -- 0.66 GB of traffic at 1M positions
// against the 0.16 GB of dz0 that this kernel reads once.
//
// One WAVE per history; it walks the history's 16-step tiles and, inside a tile, its G rows.  The MFMA operands are
// swapped with respect to hgemm.hip: A = the dz0 tile (rows = positions; lane (i, g) holds k = 32kt + 8g + {0..7} of
// position i -- the same 16-byte load as hgemm's B operand), B = weights, so a lane of the result
// D[16 positions][16 features] holds FOUR POSITIONS of ONE feature.  Sums over positions (dq, dV) are then three adds
// inside a lane + one LDS add across the four lane groups; sums over the rows of a group (da, dU) accumulate in
// registers across the g loop.  dz0 itself is brought into that layout by the matrix pipe as well (a product with the
// identity, exact in fp32 accumulation): no cross-lane shuffles anywhere (the last step of the position sums is an fp32 MFMA with a matrix of ones).
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

#define AB_GMAX 8   // rows per history group handled by this kernel (larger groups: the three-kernel path)

struct AttL0BwdArgs {
  const __bf16* dz0; int lddz;
  const __bf16* Wt; int Kp;      // packed bf16 image of Wp^T (clsr_pack_batch_bf16): Q out rows (permuted), K = A0
  const __bf16* Wu;              // optional packed image of Wu^T (same shape): da += dU . Wu^T
  const float* a; int lda;       // [Hn*T, Q]
  const float* q; int ldq;       // [R, Q]
  float* da; int ldda;           // [Hn*T, Q]
  float* dq; int lddq;           // [R, Q]
  float* dU; int lddu;           // [Hn*T, A0]
  float* dV; int lddv;           // [R, A0]
  long Hn;
  int G, T, Q, A0;
};

// packed row of out-feature o (see pack_batch_bf16_kernel)
__device__ __forceinline__ int packed_row(int o) {
  const int w = o & 31;
  return (o & ~31) + 16 * ((w & 7) >> 2) + 4 * (w >> 3) + (w & 3);
}

// NF = 16-feature tiles of Q, NZ = 16-feature tiles of A0
template <int NF, int NZ, bool WU>
__global__ void __launch_bounds__(256, 2) att_l0_bwd_kernel(AttL0BwdArgs s) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int KT = (NZ + 1) / 2;          // 32-wide k-tiles of A0
  constexpr int QP = 16 * NF, ZP = 16 * NZ;
  constexpr int NRW = 32 * ((NF + 1) / 2);  // packed weight rows
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  const int Kp = s.Kp;
  __bf16* Wl = reinterpret_cast<__bf16*>(lds_raw);
  const size_t wbytes = ((size_t)NRW * Kp * 2 + 15) & ~(size_t)15;
  __bf16* Wul = reinterpret_cast<__bf16*>(lds_raw + wbytes);
  const size_t off = wbytes * (WU ? 2 : 1);
  float* wl = reinterpret_cast<float*>(lds_raw + off) + (size_t)wave * AB_GMAX * (2 * QP + ZP);
  float* qs = wl;                       // [G][QP] query rows of the group
  float* dqs = wl + AB_GMAX * QP;       // [G][QP] dq accumulators
  float* dvs = wl + 2 * AB_GMAX * QP;   // [G][ZP] dV accumulators
  {
    const int Kq = Kp >> 3;
    const int nrows = 32 * ((s.Q + 31) >> 5);
    const bf16x8 z8 = {};
    for (int e = tid; e < NRW * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      // LDS rows in NATURAL feature order (the 16 lanes of a B-operand read then hit 16 consecutive rows: with the
      // row stride of 52 dwords those are 16 distinct bank groups; the permuted order put rows r and r + 16 together)
      const long src = (long)packed_row(row) * Kp + 8 * c;
      reinterpret_cast<bf16x8*>(Wl)[e] = row < nrows ? ld8h(s.Wt + src) : z8;
      if (WU) reinterpret_cast<bf16x8*>(Wul)[e] = row < nrows ? ld8h(s.Wu + src) : z8;
    }
  }
  __syncthreads();

  // B operands: weights of feature tile f, lane (j, g4) reads the row of feature 16f + j, k = 32kt + 8g4 + {0..7}
  int wrow[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wrow[f] = (16 * f + j) * Kp + 8 * g4;
  // identity blocks: feature tile z lives in k-tile z/2, columns 16*(z&1) + j -> lane group (z&1)*2 + (j>>3), slot j&7
  bf16x8 sel[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) sel[o][e] = (g4 == 2 * o + (j >> 3) && e == (j & 7)) ? (__bf16)1.0f : (__bf16)0.0f;

  const int G = s.G, T = s.T;
  const int NTT = (T + 15) >> 4;
  const int n_it = NTT * G;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (long h = (long)blockIdx.x * 4 + wave; h < s.Hn; h += (long)gridDim.x * 4) {
    // query rows into LDS, accumulators cleared (wave-private region: no workgroup barrier)
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      qs[e] = n < s.Q ? s.q[(h * G + g) * s.ldq + n] : 0.f;
      dqs[e] = 0.f;
    }
    for (int e = lane; e < G * ZP; e += 64) dvs[e] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // dz0 tiles of iteration i = (tt, g): a ring of four in flight
    struct Raw { bf16x8 x[KT]; };
    int itt = 0, ig = 0;   // issue cursor
    auto issue = [&]() -> Raw {
      const int tc = min(16 * itt + j, T - 1);
      // (offset clamped: for A0 < 32 the lanes past the row would read beyond the END of dz0 at its last row)
      const __bf16* p = s.dz0 + ((h * G + ig) * T + tc) * s.lddz + (8 * g4 < s.A0 ? 8 * g4 : 0);
      Raw r;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) r.x[kt] = ld8h(p + (32 * kt + 8 * g4 < s.A0 ? 32 * kt : 0));
      if (++ig == G) { ig = 0; if (itt + 1 < NTT) ++itt; }   // clamps at the last tile (surplus loads are never used)
      return r;
    };
    Raw ring[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) ring[d] = issue();

    int ctt = 0, cg = 0;   // consume cursor
    f32x4 at[NF], dacc[NF], uacc[NZ], wacc[WU ? NF : 1];
    for (int i0 = 0; i0 < n_it; i0 += 4) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (i0 + d < n_it) {
          const int t0 = 16 * ctt;
          if (cg == 0) {   // new tile: a[h, t, :] in the result layout (4 positions of feature 16f + j), accumulators
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const int n = 16 * f + j;
              const int nc = n < s.Q ? n : 0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int tc = min(t0 + 4 * g4 + e, T - 1);
                at[f][e] = s.a[(h * T + tc) * s.lda + nc];
              }
              dacc[f] = z4;
              if (WU) wacc[f] = z4;
            }
#pragma unroll
            for (int z = 0; z < NZ; ++z) uacc[z] = z4;
          }
          // A operand: rows beyond T are duplicates of step T-1 -> zero them
          const bool pv = t0 + j < T;
          const bf16x8 z8 = {};
          bf16x8 x[KT];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) x[kt] = (pv && 32 * kt + 8 * g4 < s.A0) ? ring[d].x[kt] : z8;
          ring[d] = issue();
          // daq^T tiles
          f32x4 acc[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[f] = z4;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int f = 0; f < NF; ++f) HMFMA(acc[f], x[kt], ld8h(Wl + wrow[f] + 32 * kt));
          if (WU) {   // (sum_g dz0_g) . Wu^T = sum_g (dz0_g . Wu^T): accumulated by the matrix pipe across the g loop
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
              for (int f = 0; f < NF; ++f) HMFMA(wacc[f], x[kt], ld8h(Wul + wrow[f] + 32 * kt));
          }
          // dz0^T tiles (product with the identity)
          f32x4 dzt[NZ];
#pragma unroll
          for (int z = 0; z < NZ; ++z) {
            dzt[z] = z4;
            HMFMA(dzt[z], x[z >> 1], sel[z & 1]);
          }
          // sums over the 16 positions of the tile: three adds inside a lane, then the four lane groups are summed by
          // the fp32 matrix pipe (ones[16x4] . partial[4x16]: every row of the result holds the column totals; LDS
          // float atomics with their 4-way address conflicts cost 200 us here).  Lane group g4 then owns feature tile
          // 4c + g4 of the accumulators: one conflict-free read-add-write per four tiles.
          float sq[NF], sv[NZ];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const float qv = qs[cg * QP + 16 * f + j];
            dacc[f] += acc[f] * qv;
            const f32x4 pr = acc[f] * at[f];
            sq[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (pr.x + pr.y) + (pr.z + pr.w), z4, 0, 0, 0)[0];
          }
#pragma unroll
          for (int z = 0; z < NZ; ++z) {
            uacc[z] += dzt[z];
            sv[z] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (dzt[z].x + dzt[z].y) + (dzt[z].z + dzt[z].w), z4, 0, 0, 0)[0];
          }
#pragma unroll
          for (int c = 0; c < NF; c += 4) {
            float v = sq[c];
#pragma unroll
            for (int o = 1; o < 4; ++o)
              if (c + o < NF) v = g4 == o ? sq[c + o] : v;
            if (c + g4 < NF) dqs[cg * QP + 16 * (c + g4) + j] += v;
          }
#pragma unroll
          for (int c = 0; c < NZ; c += 4) {
            float v = sv[c];
#pragma unroll
            for (int o = 1; o < 4; ++o)
              if (c + o < NZ) v = g4 == o ? sv[c + o] : v;
            if (c + g4 < NZ) dvs[cg * ZP + 16 * (c + g4) + j] += v;
          }
          if (cg == G - 1) {   // tile finished: da, dU of its 16 steps
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int t = t0 + 4 * g4 + e;
              if (t < T) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                  if (16 * f + j < s.Q) s.da[(h * T + t) * s.ldda + 16 * f + j] = dacc[f][e] + (WU ? wacc[f][e] : 0.f);
#pragma unroll
                for (int z = 0; z < NZ; ++z)
                  if (16 * z + j < s.A0) s.dU[(h * T + t) * s.lddu + 16 * z + j] = uacc[z][e];
              }
            }
          }
          if (++cg == G) { cg = 0; ++ctt; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      if (n < s.Q) s.dq[(h * G + g) * s.lddq + n] = dqs[e];
    }
    for (int e = lane; e < G * ZP; e += 64) {
      const int g = e / ZP, n = e - g * ZP;
      if (n < s.A0) s.dV[(h * G + g) * s.lddv + n] = dvs[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------ fp32 (exact mode)
// The same kernel on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: bit-exact fp32 fma chains), dz0 / weights fp32.
// MFMA #r of the 16-wide k-chunk kk uses feature 16kk + 4g + r (the k-slot -> feature map is free): the A operand of
// a chunk is ONE float4 load of the lane's position, the B operand one ds_read_b128 of the packed fp32 weights
// (clsr_pack_batch layout, row stride Kp).  Replaces clsr_pgemm (daq) + clsr_att_prod_bwd + clsr_att_z0_bwd_reduce.
struct AttL0BwdArgsF {
  const float* dz0; int lddz;
  const float* Wt; int Kp;       // packed fp32 Wp^T: row n = out feature n (natural order), K = A0
  const float* a; int lda;
  const float* q; int ldq;
  float* da; int ldda;
  float* dq; int lddq;
  float* dU; int lddu;           // may be NULL (G == 1: dU is dz0 itself)
  float* dV; int lddv;
  long Hn;
  int G, T, Q, A0;
};

template <int NF, int NZ>
__global__ void __launch_bounds__(256, 2) att_l0_bwd_f32_kernel(AttL0BwdArgsF s) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int QP = 16 * NF, ZP = 16 * NZ;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  const int Kp = s.Kp;
  float* Wl = reinterpret_cast<float*>(lds_raw);
  float* wl = Wl + (size_t)QP * Kp + (size_t)wave * AB_GMAX * (2 * QP + ZP);
  float* qs = wl;
  float* dqs = wl + AB_GMAX * QP;
  float* dvs = wl + 2 * AB_GMAX * QP;
  {
    const int Kq = Kp >> 2;
    const int nrows = 16 * ((s.Q + 15) >> 4);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < QP * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      reinterpret_cast<f32x4*>(Wl)[e] = row < nrows ? ld4(s.Wt + (long)row * Kp + 4 * c) : z;
    }
  }
  __syncthreads();
  const float* ldsB = Wl + (long)j * Kp + 4 * g4;   // + 16 f Kp + 16 kk
  // identity blocks of the transposing products: slot (g4, r) of a chunk is feature 4 g4 + r of its tile
  float sel[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sel[r] = (j == 4 * g4 + r) ? 1.0f : 0.0f;

  const int G = s.G, T = s.T;
  const int NTT = (T + 15) >> 4;
  const int n_it = NTT * G;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (long h = (long)blockIdx.x * 4 + wave; h < s.Hn; h += (long)gridDim.x * 4) {
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      qs[e] = n < s.Q ? s.q[(h * G + g) * s.ldq + n] : 0.f;
      dqs[e] = 0.f;
    }
    for (int e = lane; e < G * ZP; e += 64) dvs[e] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    struct Raw { f32x4 x[NZ]; };
    int itt = 0, ig = 0;
    auto issue = [&]() -> Raw {
      const int tc = min(16 * itt + j, T - 1);
      // (offset clamped: for A0 < 16 the lanes past the row would read beyond the END of dz0 at its last row)
      const float* p = s.dz0 + ((h * G + ig) * T + tc) * s.lddz + (4 * g4 < s.A0 ? 4 * g4 : 0);
      Raw r;
#pragma unroll
      for (int kk = 0; kk < NZ; ++kk) r.x[kk] = ld4(p + (16 * kk + 4 * g4 < s.A0 ? 16 * kk : 0));
      if (++ig == G) { ig = 0; if (itt + 1 < NTT) ++itt; }
      return r;
    };
    Raw ring[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) ring[d] = issue();

    int ctt = 0, cg = 0;
    f32x4 at[NF], dacc[NF], uacc[NZ];
    for (int i0 = 0; i0 < n_it; i0 += 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (i0 + d < n_it) {
          const int t0 = 16 * ctt;
          if (cg == 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const int n = 16 * f + j;
              const int nc = n < s.Q ? n : 0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int tc = min(t0 + 4 * g4 + e, T - 1);
                at[f][e] = s.a[(h * T + tc) * s.lda + nc];
              }
              dacc[f] = z4;
            }
#pragma unroll
            for (int z = 0; z < NZ; ++z) uacc[z] = z4;
          }
          const bool pv = t0 + j < T;
          f32x4 x[NZ];
#pragma unroll
          for (int kk = 0; kk < NZ; ++kk) x[kk] = (pv && 16 * kk + 4 * g4 < s.A0) ? ring[d].x[kk] : z4;
          ring[d] = issue();
          f32x4 acc[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[f] = z4;
#pragma unroll
          for (int kk = 0; kk < NZ; ++kk) {
            f32x4 w[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) w[f] = ld4(ldsB + (long)f * 16 * Kp + 16 * kk);
#pragma unroll
            for (int f = 0; f < NF; ++f) MFMA4(acc[f], x[kk].x, w[f].x);
#pragma unroll
            for (int f = 0; f < NF; ++f) MFMA4(acc[f], x[kk].y, w[f].y);
#pragma unroll
            for (int f = 0; f < NF; ++f) MFMA4(acc[f], x[kk].z, w[f].z);
#pragma unroll
            for (int f = 0; f < NF; ++f) MFMA4(acc[f], x[kk].w, w[f].w);
          }
          f32x4 dzt[NZ];
#pragma unroll
          for (int z = 0; z < NZ; ++z) {
            dzt[z] = z4;
            MFMA4(dzt[z], x[z].x, sel[0]);
            MFMA4(dzt[z], x[z].y, sel[1]);
            MFMA4(dzt[z], x[z].z, sel[2]);
            MFMA4(dzt[z], x[z].w, sel[3]);
          }
          float sq[NF], sv[NZ];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const float qv = qs[cg * QP + 16 * f + j];
            dacc[f] += acc[f] * qv;
            const f32x4 pr = acc[f] * at[f];
            sq[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (pr.x + pr.y) + (pr.z + pr.w), z4, 0, 0, 0)[0];
          }
#pragma unroll
          for (int z = 0; z < NZ; ++z) {
            uacc[z] += dzt[z];
            sv[z] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (dzt[z].x + dzt[z].y) + (dzt[z].z + dzt[z].w), z4, 0, 0, 0)[0];
          }
#pragma unroll
          for (int c = 0; c < NF; c += 4) {
            float v = sq[c];
#pragma unroll
            for (int o = 1; o < 4; ++o)
              if (c + o < NF) v = g4 == o ? sq[c + o] : v;
            if (c + g4 < NF) dqs[cg * QP + 16 * (c + g4) + j] += v;
          }
#pragma unroll
          for (int c = 0; c < NZ; c += 4) {
            float v = sv[c];
#pragma unroll
            for (int o = 1; o < 4; ++o)
              if (c + o < NZ) v = g4 == o ? sv[c + o] : v;
            if (c + g4 < NZ) dvs[cg * ZP + 16 * (c + g4) + j] += v;
          }
          if (cg == G - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int t = t0 + 4 * g4 + e;
              if (t < T) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                  if (16 * f + j < s.Q) s.da[(h * T + t) * s.ldda + 16 * f + j] = dacc[f][e];
                if (s.dU) {
#pragma unroll
                  for (int z = 0; z < NZ; ++z)
                    if (16 * z + j < s.A0) s.dU[(h * T + t) * s.lddu + 16 * z + j] = uacc[z][e];
                }
              }
            }
          }
          if (++cg == G) { cg = 0; ++ctt; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      if (n < s.Q) s.dq[(h * G + g) * s.lddq + n] = dqs[e];
    }
    for (int e = lane; e < G * ZP; e += 64) {
      const int g = e / ZP, n = e - g * ZP;
      if (n < s.A0) s.dV[(h * G + g) * s.lddv + n] = dvs[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int NF, int NZ>
static int att_l0_bwd_f32_launch(const AttL0BwdArgsF& a, hipStream_t stream) {
  size_t shmem = (size_t)16 * NF * a.Kp * 4 + (size_t)4 * AB_GMAX * (2 * 16 * NF + 16 * NZ) * 4;
  int gx = (int)((a.Hn + 3) / 4);
  if (gx > 512) gx = 512;
  auto kernel = att_l0_bwd_f32_kernel<NF, NZ>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(gx), dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static int tiles_class(int n) { return n <= 48 ? 3 : (n <= 80 ? 5 : 6); }

// 1 when clsr_att_l0_bwd_h handles this shape (otherwise: clsr_hgemm + clsr_att_prod_bwd_h + clsr_att_z0_bwd_reduce_h)
extern "C" int clsr_att_l0_bwd_h_supported(int G, int Q, int A0) {
  return G >= 1 && G <= AB_GMAX && Q >= 8 && Q <= 96 && A0 >= 8 && A0 <= 96 && Q % 8 == 0 && A0 % 8 == 0;
}

template <int NF, int NZ, bool WU>
static int att_l0_bwd_launch2(const AttL0BwdArgs& a, hipStream_t stream) {
  constexpr int NRW = 32 * ((NF + 1) / 2);
  size_t shmem = (((size_t)NRW * a.Kp * 2 + 15) & ~(size_t)15) * (WU ? 2 : 1) +
                 (size_t)4 * AB_GMAX * (2 * 16 * NF + 16 * NZ) * 4;
  int gx = (int)((a.Hn + 3) / 4);
  if (gx > 512) gx = 512;
  auto kernel = att_l0_bwd_kernel<NF, NZ, WU>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(gx), dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

template <int NF, int NZ>
static int att_l0_bwd_launch(const AttL0BwdArgs& a, hipStream_t stream) {
  return a.Wu ? att_l0_bwd_launch2<NF, NZ, true>(a, stream) : att_l0_bwd_launch2<NF, NZ, false>(a, stream);
}

extern "C" int clsr_att_l0_bwd_h(const void* dz0, int lddz, const void* Wt, const void* Wu, int Kp, const float* a, int lda,
                                 const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                                 float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, void* stream) {
  CLSR_CHECK_ARG(dz0 && Wt && a && q && da && dq && dU && dV && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l0_bwd_h_supported(G, Q, A0));
  CLSR_CHECK_SUPPORTED(lddz % 8 == 0 && Kp % 8 == 0 && ((uintptr_t)dz0 % 16) == 0);
  CLSR_CHECK_ARG(lddz >= A0 && Kp >= 32 * clsr_cdiv(A0, 32) && lda >= Q && ldq >= Q && ldda >= Q && lddq >= Q &&
                 lddu >= A0 && lddv >= A0);
  AttL0BwdArgs s = {};
  s.dz0 = (const __bf16*)dz0; s.lddz = lddz; s.Wt = (const __bf16*)Wt; s.Wu = (const __bf16*)Wu; s.Kp = Kp; s.a = a; s.lda = lda; s.q = q;
  s.ldq = ldq; s.da = da; s.ldda = ldda; s.dq = dq; s.lddq = lddq; s.dU = dU; s.lddu = lddu; s.dV = dV; s.lddv = lddv;
  s.Hn = Hn; s.G = G; s.T = T; s.Q = Q; s.A0 = A0;
  hipStream_t st = (hipStream_t)stream;
  const int nf = tiles_class(Q), nz = tiles_class(A0);
#define AB_GO(F, Z) if (nf == F && nz == Z) return att_l0_bwd_launch<F, Z>(s, st)
  AB_GO(3, 3); AB_GO(3, 5); AB_GO(3, 6);
  AB_GO(5, 3); AB_GO(5, 5); AB_GO(5, 6);
  AB_GO(6, 3); AB_GO(6, 5); AB_GO(6, 6);
#undef AB_GO
  return CLSR_OK;
}

// 1 when clsr_att_l0_bwd handles this shape (otherwise: clsr_pgemm + clsr_att_prod_bwd + clsr_att_z0_bwd_reduce)
extern "C" int clsr_att_l0_bwd_supported(int G, int Q, int A0) {
  return G >= 1 && G <= AB_GMAX && Q >= 4 && Q <= 80 && A0 >= 4 && A0 <= 80 && Q % 4 == 0 && A0 % 4 == 0;
}

extern "C" int clsr_att_l0_bwd(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                               const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                               float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, void* stream) {
  CLSR_CHECK_ARG(dz0 && Wt && a && q && da && dq && dV && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l0_bwd_supported(G, Q, A0));
  CLSR_CHECK_SUPPORTED(lddz % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)dz0 % 16) == 0 && ((uintptr_t)Wt % 16) == 0);
  CLSR_CHECK_ARG(lddz >= A0 && Kp >= 16 * clsr_cdiv(A0, 16) && lda >= Q && ldq >= Q && ldda >= Q && lddq >= Q &&
                 (!dU || lddu >= A0) && lddv >= A0);
  AttL0BwdArgsF s = {};
  s.dz0 = dz0; s.lddz = lddz; s.Wt = Wt; s.Kp = Kp; s.a = a; s.lda = lda; s.q = q; s.ldq = ldq; s.da = da;
  s.ldda = ldda; s.dq = dq; s.lddq = lddq; s.dU = dU; s.lddu = lddu; s.dV = dV; s.lddv = lddv;
  s.Hn = Hn; s.G = G; s.T = T; s.Q = Q; s.A0 = A0;
  hipStream_t st = (hipStream_t)stream;
  const int nf = tiles_class(Q), nz = tiles_class(A0);
#define AB_GO(F, Z) if (nf == F && nz == Z) return att_l0_bwd_f32_launch<F, Z>(s, st)
  AB_GO(3, 3); AB_GO(3, 5); AB_GO(5, 3); AB_GO(5, 5);   // (six-tile variants would spill: widths above 80 stay unsupported)
#undef AB_GO
  return CLSR_OK;
}
