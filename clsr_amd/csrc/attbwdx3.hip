// Split-bf16 forms of the two big backward kernels of the attention MLP (gfx950), with the weight gradients of their
// layers folded in.  Reference: tf.gradients through _attention_fcn / _fcn_net (clsr.py:343-381, base_model.py:627-708).
//
// Why.  The exact kernels (csrc/hattbwd.hip: att_l0_bwd_f32_kernel, csrc/attl1bwd.hip: att_l1_bwd_kernel) run their
// products on v_mfma_f32_16x16x4_f32, which issues at the fp32 VECTOR rate (32 cycles per SIMD and instruction): 88 and
// 120 MFMAs per 16- / 32-position tile made them matrix-pipe bound at 1.3-2 x their HBM floor, and the weight gradients
// of the same layers (pgemm_dw_kernel: dWp = (a*q)^T dz0, dW1 = relu(bn0(z0))^T dz1) RE-READ z0 / dz1 / dz0 / a / q from HBM
// on a stream of their own -- 0.9 GB and ~1 ms of contending kernel time per step at configs[1]
// (profiles/r04_fp32_stats.md).  Here every product is  x.y ~ xh.yh + xh.yl + xl.yh  on v_mfma_f32_16x16x32_bf16
// (x = xh + xl, xh = RNE_bf16(x), xl = RNE_bf16(x - xh): 16 significand bits, the dropped xl.yl term is 2^-18
// relative; fp32 accumulation; ~17 cycles per instruction and a K = 32 chunk each), which leaves room on the matrix
// pipe for the weight-gradient products of the SAME tile while it is in registers: the separate launches, their
// re-reads and the stored dz1 disappear.
//
// The weight gradients contract over POSITIONS.  In the result layout of an MFMA whose A operand is the position tile
// (D[16 positions][16 features]: lane (j, g) holds positions 4g..4g+3 of feature j) a lane's four values are four k-slots
// of an operand whose ROWS are features: two such tiles (32 positions) side by side are exactly one bf16x8 A or B
// operand of a 16x16x32 MFMA with k = positions -- no LDS transposition.  Both kernels therefore keep / bring their
// operands in that "feature-lane" layout (products with the identity on the bf16 pipe are exact for the hi and the lo
// image separately).  Per workgroup ONE partial chunk in the layout of pgemm_dw_kernel (CLSR_DW_CHUNK), summed by the
// step's batched clsr_dw_reduce_batch in a fixed order: deterministic, no float atomics.
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

#define X3_GMAX 8   // rows per history group (as AB_GMAX in hattbwd.hip)
#ifndef X3_L1P1_OCC
#define X3_L1P1_OCC 1      // (2: two waves per SIMD without the register prefetch -- 115.7 vs 111.9 us alone, nothing in the step; pass 2 sizes its partial chunks with the same grid)
#endif
#ifndef X3_OCC
#define X3_OCC 1
#endif
// ablation switches (scripts/abl_att_l0x3.sh): 0 removes the piece from the layer-0 kernel
#ifndef X3_L0H_OCC2
#define X3_L0H_OCC2 1      // two waves per SIMD for the one-piece bf16 instance of the layer-0 backward (256 registers, 11 spilled:
                           // 128 -> 92 us alone, bf16 step 2.321 -> 2.308 ms; the layer-1 instances spill 44 / 126 registers at
                           // that size and stay at one wave per SIMD)
#endif
#ifndef X3A_DW
#define X3A_DW 1
#endif
#ifndef X3A_STORE
#define X3A_STORE 1
#endif

__device__ __forceinline__ bf16x4 to_h4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 to_f4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ void split4(f32x4 v, bf16x4& hi, bf16x4& lo) {
  hi = to_h4(v);
  lo = to_h4(v - to_f4(hi));
}
__device__ __forceinline__ void split8x(f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = to_h(v);
  lo = to_h(v - to_f(hi));
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// K = 16 form (four positions per lane): the three-piece layer-0 instance takes its weight-gradient products per iteration
// instead of pairing two iterations into a K = 32 operand -- the saved pieces of the pair's first half cost 32 registers that
// instance does not have (92 B of scratch, 404 us in the step)
typedef short x3_s16x4 __attribute__((ext_vector_type(4)));
#define HMFMA16(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(x3_s16x4, (a)), __builtin_bit_cast(x3_s16x4, (b)), (acc), 0, 0, 0)

typedef __amdgpu_buffer_rsrc_t x3_rsrc_t;
#define X3_OOB 0x80000000u
// keeps the scheduler from hoisting every tile's LDS reads / epilogue arithmetic to the top of the iteration (with 512
// registers to fill it does, and then spills)
#ifdef X3_NO_SCHED_FENCE
#define X3_SCHED_FENCE()
#else
#define X3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
__device__ __forceinline__ x3_rsrc_t x3_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, X3_OOB, 0x00020000);
}
// resource over exactly n bytes: any offset >= n (a skip marker, alone or summed with another one) is dropped / reads 0
__device__ __forceinline__ x3_rsrc_t x3_rsrc_n(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ f32x4 x3_ld4(x3_rsrc_t rs, unsigned off, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0));
}
__device__ __forceinline__ float x3_ld1(x3_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ void x3_st1(x3_rsrc_t rs, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
}
// ---- storage type of the (row, step)-level tensors z1 / z0 / dz0: float (parity mode) or __bf16 (speed mode: half the
// bytes; with NP = 1 -- ONE bf16 piece per operand -- the kernels below are the speed mode's chain kernels, replacing
// csrc/hgemm.hip's position-tiled products + their separate weight-gradient launches).  Loads stay RAW in the prefetch
// registers (a conversion at the load would wait for it).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <typename ST> struct X3Raw8;
template <> struct X3Raw8<float> { f32x4 lo, hi; };
template <> struct X3Raw8<__bf16> { u32x4_t v; };
__device__ __forceinline__ f32x8 x3_f8(const X3Raw8<float>& r) { return (f32x8){r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w}; }
__device__ __forceinline__ f32x8 x3_f8(const X3Raw8<__bf16>& r) { return to_f(__builtin_bit_cast(bf16x8, r.v)); }
template <typename ST> __device__ __forceinline__ X3Raw8<ST> x3_ld8raw(x3_rsrc_t rs, unsigned off, unsigned soff);
template <> __device__ __forceinline__ X3Raw8<float> x3_ld8raw<float>(x3_rsrc_t rs, unsigned off, unsigned soff) {
  X3Raw8<float> r;
  r.lo = x3_ld4(rs, off, soff); r.hi = x3_ld4(rs, off + 16u, soff);
  return r;
}
template <> __device__ __forceinline__ X3Raw8<__bf16> x3_ld8raw<__bf16>(x3_rsrc_t rs, unsigned off, unsigned soff) {
  X3Raw8<__bf16> r;
  r.v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0));
  return r;
}
template <typename ST> __device__ __forceinline__ X3Raw8<ST> x3_ld8raw_g(const ST* p);
template <> __device__ __forceinline__ X3Raw8<float> x3_ld8raw_g<float>(const float* p) {
  X3Raw8<float> r;
  r.lo = ld4(p); r.hi = ld4(p + 4);
  return r;
}
template <> __device__ __forceinline__ X3Raw8<__bf16> x3_ld8raw_g<__bf16>(const __bf16* p) {
  X3Raw8<__bf16> r;
  r.v = *reinterpret_cast<const u32x4_t*>(p);
  return r;
}
// one element: raw bits in a 32-bit register (bf16: zero-extended), converted at the use
template <typename ST> __device__ __forceinline__ unsigned x3_ld1raw(x3_rsrc_t rs, unsigned off);
template <> __device__ __forceinline__ unsigned x3_ld1raw<float>(x3_rsrc_t rs, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
}
template <> __device__ __forceinline__ unsigned x3_ld1raw<__bf16>(x3_rsrc_t rs, unsigned off) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0);
}
template <typename ST> __device__ __forceinline__ float x3_f1(unsigned r);
template <> __device__ __forceinline__ float x3_f1<float>(unsigned r) { return __builtin_bit_cast(float, r); }
template <> __device__ __forceinline__ float x3_f1<__bf16>(unsigned r) { return __builtin_bit_cast(float, r << 16); }
template <typename ST> __device__ __forceinline__ void x3_st1s(x3_rsrc_t rs, unsigned off, float v);
template <> __device__ __forceinline__ void x3_st1s<float>(x3_rsrc_t rs, unsigned off, float v) { x3_st1(rs, off, v); }
template <> __device__ __forceinline__ void x3_st1s<__bf16>(x3_rsrc_t rs, unsigned off, float v) {
  const __bf16 h = (__bf16)v;      // (round to nearest even)
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rs, off, 0, 0);
}

// ------------------------------------------------------------------------------------------------ layer 0
//   z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp
// from ONE pass over dz0 (fp32, [R*T, A0]):  da, dq, dU, dV as att_l0_bwd_f32_kernel, and the partial sums of
//   dWp[c, n] = sum_{r,t} a[h,t,c] q[r,c] dz0[r,t,n].
// One WAVE per history; iterations (16-step tile, row of the group) are processed in PAIRS so that the eight positions a
// lane holds of its feature (4 of each iteration) fill one bf16x8 operand of the weight-gradient MFMAs.
struct AttL0BwdArgsX {
  const void* dz0; int lddz;     // float or __bf16 (the kernel's ST)
  const float* Wt; int Kp;       // packed fp32 Wp^T (clsr_pack_batch): row c = query feature (natural order), K = A0
  const float* a; int lda;
  const float* q; int ldq;
  float* da; int ldda;
  float* dq; int lddq;
  float* dU; int lddu;           // may be NULL (G == 1: dU is dz0 itself)
  float* dV; int lddv;
  float* dwp;                    // [gridDim.x][CLSR_DW_CHUNK] partial chunks of dWp (tiles kt < NF, nt < NZ)
  long Hn;
  int G, T, Q, A0;
};

// NP = bf16 pieces per operand: 3 (precision="fp32": every piece product whose indices sum to <= 2 -- 2^-23 relative, the level
// of an fp32 product), 2 (hi + lo, "fp32x3": 2^-16 per term) or 1 (speed mode); ST = storage type of dz0
template <int NF, int NZ, int NP, typename ST>
__global__ void __launch_bounds__(256, (NP == 1 && X3_L0H_OCC2) ? 2 : X3_OCC) att_l0_bwd_x3_kernel(AttL0BwdArgsX s) {
  constexpr unsigned SB = sizeof(ST);
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int KT = (NZ + 1) / 2;          // 32-wide k chunks of A0
  constexpr int QP = 16 * NF, ZP = 16 * NZ;
  constexpr int WS = 32 * KT + 8;           // bf16 row stride of the weight images (52 / 36 dwords: conflict-free 16-byte reads)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  constexpr int NPW = NP > 2 ? 3 : 2;       // weight images in LDS
  __bf16* Wh = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Wl = Wh + QP * WS;
  __bf16* Wr = Wl + QP * WS;                // (NP == 3 only)
  float* wl = reinterpret_cast<float*>(lds_raw + (size_t)NPW * QP * WS * 2) + (size_t)wave * X3_GMAX * (2 * QP + ZP);
  float* qs = wl;                       // [G][QP] query rows of the group
  float* dqs = wl + X3_GMAX * QP;       // [G][QP] dq accumulators
  float* dvs = wl + 2 * X3_GMAX * QP;   // [G][ZP] dV accumulators
  {
    constexpr int C8 = WS / 8;
    for (int e = tid; e < QP * C8; e += 256) {
      const int row = e / C8, k = 8 * (e - row * C8);
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < s.Q && k < s.A0) v = ld8f(s.Wt + (long)row * s.Kp + k);
      bf16x8 hi, lo;
      split8x(v, hi, lo);
      reinterpret_cast<bf16x8*>(Wh)[e] = hi;
      reinterpret_cast<bf16x8*>(Wl)[e] = lo;
      if constexpr (NP > 2) reinterpret_cast<bf16x8*>(Wr)[e] = to_h(v - to_f(hi) - to_f(lo));
    }
  }
  __syncthreads();
  const ST* dz0p = reinterpret_cast<const ST*>(s.dz0);

  // B operands: weights of query-feature tile f, lane (j, g4) reads row 16f + j, k = 32kt + 8g4 + {0..7}
  const int wrow = j * WS + 8 * g4;          // + 16 f WS + 32 kt
  // identity blocks: feature tile z of dz0 lives in k chunk z/2, columns 16 (z&1) + j -> lane group 2 (z&1) + (j>>3), slot j&7
  bf16x8 sel[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) sel[o][e] = (g4 == 2 * o + (j >> 3) && e == (j & 7)) ? (__bf16)1.0f : (__bf16)0.0f;

  const int G = s.G, T = s.T;
  const int NTT = (T + 15) >> 4;
  const int n_it = NTT * G;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x4 zh4 = {};
  // Every global access of the main loop is a raw buffer access (resource = the rows of the wave's history, exact size):
  // a lane that must not read / write gets an offset component of X3_SKIP -- out of range alone and in any sum of two -- so
  // that no load or store sits under a branch (190 guarded stores cost ~740 exec-mask instructions per four iterations),
  // and the address arithmetic is one 32-bit add per access.
  constexpr unsigned X3_SKIP = 0x40000000u;
  unsigned kofs[KT];                         // dz0: byte offset of the lane's 8 features of chunk kt
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) kofs[kt] = 32 * kt + 8 * g4 < s.A0 ? (32 * kt + 8 * g4) * SB : X3_SKIP;   // (skipped: reads 0)
  unsigned aofs[NF], dao[NF], duo[NZ];       // a (read, clamped), da / dU (written: skipped beyond the width)
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    aofs[f] = (16 * f + j < s.Q ? 16 * f + j : 0) * 4u;
    dao[f] = 16 * f + j < s.Q ? (16 * f + j) * 4u : X3_SKIP;
  }
#pragma unroll
  for (int z = 0; z < NZ; ++z) duo[z] = 16 * z + j < s.A0 ? (16 * z + j) * 4u : X3_SKIP;

  f32x4 accW[NF][NZ];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int z = 0; z < NZ; ++z) accW[f][z] = z4;

  for (long h = (long)blockIdx.x * 4 + wave; h < s.Hn; h += (long)gridDim.x * 4) {
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      qs[e] = n < s.Q ? s.q[(h * G + g) * s.ldq + n] : 0.f;
      dqs[e] = 0.f;
    }
    for (int e = lane; e < G * ZP; e += 64) dvs[e] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    const x3_rsrc_t rdz = x3_rsrc_n(dz0p + h * G * T * s.lddz, (unsigned)(G * T * s.lddz) * SB);
    const x3_rsrc_t ra = x3_rsrc_n(s.a + h * T * s.lda, (unsigned)(T * s.lda) * 4u);
    const x3_rsrc_t rda = x3_rsrc_n(s.da + h * T * s.ldda, (unsigned)(T * s.ldda) * 4u);
    const x3_rsrc_t rdu = x3_rsrc_n(s.dU ? s.dU + h * T * s.lddu : nullptr, s.dU ? (unsigned)(T * s.lddu) * 4u : 0u);

    // dz0 tiles of iteration i = (tt, g) in A-operand order: lane (position j, g4) holds features 32kt + 8g4 + {0..7}
    struct Raw { X3Raw8<ST> x[KT]; };
    int itt = 0, ig = 0;
    unsigned rowoff = (unsigned)(min(j, T - 1) * s.lddz) * SB;          // lane's row of tile itt inside a [T, lddz] block
    auto issue = [&]() -> Raw {
      const unsigned blk = (unsigned)(ig * T * s.lddz) * SB;             // (uniform)
      Raw r;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) r.x[kt] = x3_ld8raw<ST>(rdz, rowoff + kofs[kt], blk);
      if (++ig == G) {
        ig = 0;
        if (itt + 1 < NTT) { ++itt; rowoff = (unsigned)(min(16 * itt + j, T - 1) * s.lddz) * SB; }   // (clamps at the last tile)
      }
      return r;
    };
    Raw ring[2];
    ring[0] = issue();
    ring[1] = issue();

    int ctt = 0, cg = 0;
    f32x4 at[NF], dacc[NF], uacc[NZ];
    bf16x4 sah[NF], sal[NF], sbh[NZ], sbl[NZ];     // first half of a pair: (a*q) and dz0 in feature-lane layout, hi / lo
    for (int i0 = 0; i0 < n_it; i0 += 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const bool live = i0 + d < n_it;            // (uniform)
        bf16x4 cah[NF], cal[NF], cbh[NZ], cbl[NZ], car[NP > 2 ? NF : 1], cbr[NP > 2 ? NZ : 1];
#pragma unroll
        for (int f = 0; f < NF; ++f) { cah[f] = zh4; cal[f] = zh4; if constexpr (NP > 2) car[f] = zh4; }
#pragma unroll
        for (int z = 0; z < NZ; ++z) { cbh[z] = zh4; cbl[z] = zh4; if constexpr (NP > 2) cbr[z] = zh4; }
        if (live) {
          const int t0 = 16 * ctt;
          if (cg == 0) {   // new tile: a[h, t, :] in the result layout (4 positions of feature 16f + j), accumulators
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned ro = (unsigned)(min(t0 + 4 * g4 + e, T - 1) * s.lda) * 4u;
#pragma unroll
              for (int f = 0; f < NF; ++f) at[f][e] = x3_ld1(ra, ro + aofs[f]);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) dacc[f] = z4;
#pragma unroll
            for (int z = 0; z < NZ; ++z) uacc[z] = z4;
          }
          // A operands (rows beyond T are duplicates of step T-1 -> zero them; features beyond A0 were not read: zeros)
          // (xr: the remainder x - xh - xl as a third piece, used by the transposing products only -- dU and dV are SUMS of
          // dz0 that cancel analytically under the batch-norm (the bias gradient of the layer is exactly 0): two pieces
          // left 2^-17 |dz0| sqrt(N) of noise there, scripts/fuzz_step.py case 4)
          bf16x8 xh[KT], xl[KT], xr[KT];
          {
            const bool pv = t0 + j < T;      // (false only in a ragged last tile)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              f32x8 v = (t0 + 16 <= T || pv) ? x3_f8(ring[d].x[kt]) : z8;
              xh[kt] = to_h(v);
              if constexpr (NP > 1) {
                v -= to_f(xh[kt]);
                xl[kt] = to_h(v);
                xr[kt] = to_h(v - to_f(xl[kt]));
              }
            }
          }
          ring[d] = issue();
          // daq^T tiles: acc[f] = dz0 . Wp^T, 4 positions of query feature 16f + j
          f32x4 acc[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[f] = z4;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            bf16x8 wh[NF], wlo[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              wh[f] = ld8h(Wh + wrow + 16 * f * WS + 32 * kt);
              if constexpr (NP > 1) wlo[f] = ld8h(Wl + wrow + 16 * f * WS + 32 * kt);
            }
            if constexpr (NP > 2) {       // the three products of index sum 2 first (smallest terms)
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                const bf16x8 wr = ld8h(Wr + wrow + 16 * f * WS + 32 * kt);
                HMFMA(acc[f], xh[kt], wr);
                HMFMA(acc[f], xl[kt], wlo[f]);
                HMFMA(acc[f], xr[kt], wh[f]);
              }
            }
            if constexpr (NP > 1) {
#pragma unroll
              for (int f = 0; f < NF; ++f) HMFMA(acc[f], xh[kt], wlo[f]);
#pragma unroll
              for (int f = 0; f < NF; ++f) HMFMA(acc[f], xl[kt], wh[f]);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) HMFMA(acc[f], xh[kt], wh[f]);
          }
          // dz0^T tiles (products with the identity: exact for the hi and the lo image)
          f32x4 dzt[NZ];
#pragma unroll
          for (int z = 0; z < NZ; ++z) {
            f32x4 th = z4;
            HMFMA(th, xh[z >> 1], sel[z & 1]);
            cbh[z] = to_h4(th);
            if constexpr (NP > 1) {
              f32x4 tl = z4, tr = z4;
              HMFMA(tl, xl[z >> 1], sel[z & 1]);
              HMFMA(tr, xr[z >> 1], sel[z & 1]);
              cbl[z] = to_h4(tl);
              if constexpr (NP > 2) cbr[z] = to_h4(tr);
              dzt[z] = th + (tl + tr);
            } else {
              dzt[z] = th;      // (exact for a bf16 dz0)
            }
          }
          // sums over the 16 positions of the tile: three adds inside a lane, then the four lane groups are summed by the
          // fp32 matrix pipe (ones[16x4] . partial[4x16]); lane group g4 then owns feature tile 4c + g4 of the accumulators.
          // (per-lane partial slots + ds_add_f32 instead measured 297 against 179 us for the kernel)
          float sq[NF], sv[NZ];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const float qv = qs[cg * QP + 16 * f + j];
            dacc[f] += acc[f] * qv;
            const f32x4 pr = acc[f] * at[f];
            sq[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (pr.x + pr.y) + (pr.z + pr.w), z4, 0, 0, 0)[0];
            if constexpr (NP > 1) split4(at[f] * qv, cah[f], cal[f]);
            else cah[f] = to_h4(at[f] * qv);
            if constexpr (NP > 2) car[f] = to_h4(at[f] * qv - to_f4(cah[f]) - to_f4(cal[f]));
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
          if (X3A_STORE && cg == G - 1) {   // tile finished: da, dU of its 16 steps
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int t = t0 + 4 * g4 + e;
              const unsigned ra_ = t < T ? (unsigned)(t * s.ldda) * 4u : X3_SKIP;
              const unsigned ru_ = t < T ? (unsigned)(t * s.lddu) * 4u : X3_SKIP;
#pragma unroll
              for (int f = 0; f < NF; ++f) x3_st1(rda, ra_ + dao[f], dacc[f][e]);
#pragma unroll
              for (int z = 0; z < NZ; ++z) x3_st1(rdu, ru_ + duo[z], uacc[z][e]);
            }
          }
          if constexpr (NP > 2) {
            if (X3A_DW) {       // weight gradient of THIS iteration: k = the 4 positions a lane holds of its feature
#pragma unroll
              for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int z = 0; z < NZ; ++z) {
                  HMFMA16(accW[f][z], cah[f], cbr[z]);
                  HMFMA16(accW[f][z], cal[f], cbl[z]);
                  HMFMA16(accW[f][z], car[f], cbh[z]);
                  HMFMA16(accW[f][z], cah[f], cbl[z]);
                  HMFMA16(accW[f][z], cal[f], cbh[z]);
                  HMFMA16(accW[f][z], cah[f], cbh[z]);
                }
            }
          }
          if (++cg == G) { cg = 0; ++ctt; }
        }
        if constexpr (NP > 2) {
          // (no pairing: see HMFMA16)
        } else if (d == 0) {
#pragma unroll
          for (int f = 0; f < NF; ++f) { sah[f] = cah[f]; sal[f] = cal[f]; }
#pragma unroll
          for (int z = 0; z < NZ; ++z) { sbh[z] = cbh[z]; sbl[z] = cbl[z]; }
        } else if (X3A_DW) {
          // weight gradient of the pair: k = the 8 positions a lane holds of its feature (4 of each iteration)
          bf16x8 bh[NZ], bl[NZ];
#pragma unroll
          for (int z = 0; z < NZ; ++z) { bh[z] = cat4(sbh[z], cbh[z]); bl[z] = cat4(sbl[z], cbl[z]); }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const bf16x8 ah = cat4(sah[f], cah[f]), al = cat4(sal[f], cal[f]);
            if constexpr (NP > 1) {
#pragma unroll
              for (int z = 0; z < NZ; ++z) HMFMA(accW[f][z], ah, bl[z]);
#pragma unroll
              for (int z = 0; z < NZ; ++z) HMFMA(accW[f][z], al, bh[z]);
            }
#pragma unroll
            for (int z = 0; z < NZ; ++z) HMFMA(accW[f][z], ah, bh[z]);
          }
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

  // workgroup sum of the four waves' accumulators through LDS (waves 2, 3 -> 0, 1; 1 -> 0), then ONE partial chunk
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds_raw);
  constexpr int RED = NF * NZ * 256;
  auto store_to = [&](float* dst, int tstride) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int z = 0; z < NZ; ++z) {
        float* t = dst + (f * tstride + z) * 256 + (4 * g4) * 16 + j;
        const f32x4 v = accW[f][z];
        t[0] = v.x; t[16] = v.y; t[32] = v.z; t[48] = v.w;
      }
  };
  auto add_from = [&](const float* src) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int z = 0; z < NZ; ++z) {
        const float* t = src + (f * NZ + z) * 256 + (4 * g4) * 16 + j;
        accW[f][z] += (f32x4){t[0], t[16], t[32], t[48]};
      }
  };
  if (wave >= 2) store_to(red + (wave - 2) * RED, NZ);
  __syncthreads();
  if (wave < 2) add_from(red + wave * RED);
  __syncthreads();
  if (wave == 1) store_to(red, NZ);
  __syncthreads();
  if (wave == 0) {
    add_from(red);
    store_to(s.dwp + (long)blockIdx.x * CLSR_DW_CHUNK, CLSR_DW_T);
  }
}

extern "C" int clsr_dw_chunk_floats(void) { return CLSR_DW_CHUNK; }

static int x3_tiles_class(int n) { return n <= 48 ? 3 : 5; }

static int l0x_grid(long Hn, bool half = false) {
  long gx = (Hn + 3) / 4;
  const long cap = (half && X3_L0H_OCC2) ? 512 : 256;      // one workgroup per CU (512-register waves; the bf16 instance: two)
  if (gx > cap) gx = cap;                                    // -- a wave walks several histories
  return (int)gx;
}

extern "C" int clsr_att_l0_bwd_x3_supported(int G, int Q, int A0) {
  return G >= 1 && G <= X3_GMAX && Q >= 4 && Q <= 80 && A0 >= 8 && A0 <= 80 && Q % 4 == 0 && A0 % 8 == 0;
}
extern "C" int clsr_att_l0_bwd_x3_parts(long Hn) { return l0x_grid(Hn); }
extern "C" int clsr_att_l0_bwd_x1_h_parts(long Hn) { return l0x_grid(Hn, true); }

template <int NF, int NZ, int NP, typename ST>
static int att_l0_bwd_x3_launch(const AttL0BwdArgsX& a, hipStream_t stream) {
  constexpr int KT = (NZ + 1) / 2, WS = 32 * KT + 8;
  size_t shmem = (size_t)(NP > 2 ? 3 : 2) * 16 * NF * WS * 2 + (size_t)4 * X3_GMAX * (2 * 16 * NF + 16 * NZ) * 4;
  const size_t red = (size_t)2 * NF * NZ * 256 * 4;
  if (shmem < red) shmem = red;
  auto kernel = att_l0_bwd_x3_kernel<NF, NZ, NP, ST>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(l0x_grid(a.Hn, sizeof(ST) == 2)), dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static int att_l0_bwd_x3_any(const void* dz0, bool half, int pieces, int lddz, const float* Wt, int Kp, const float* a, int lda,
                             const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                             float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                             void* stream) {
  CLSR_CHECK_ARG(dz0 && Wt && a && q && da && dq && dV && dwp_partial && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l0_bwd_x3_supported(G, Q, A0));
  CLSR_CHECK_SUPPORTED(lddz % (half ? 8 : 4) == 0 && Kp % 4 == 0 && ((uintptr_t)dz0 % 16) == 0 && ((uintptr_t)Wt % 16) == 0);
  CLSR_CHECK_ARG(lddz >= A0 && Kp >= 16 * clsr_cdiv(A0, 16) && lda >= Q && ldq >= Q && ldda >= Q && lddq >= Q &&
                 (!dU || lddu >= A0) && lddv >= A0);
  AttL0BwdArgsX s = {};
  s.dz0 = dz0; s.lddz = lddz; s.Wt = Wt; s.Kp = Kp; s.a = a; s.lda = lda; s.q = q; s.ldq = ldq; s.da = da;
  s.ldda = ldda; s.dq = dq; s.lddq = lddq; s.dU = dU; s.lddu = lddu; s.dV = dV; s.lddv = lddv; s.dwp = dwp_partial;
  s.Hn = Hn; s.G = G; s.T = T; s.Q = Q; s.A0 = A0;
  hipStream_t st = (hipStream_t)stream;
  const int nf = x3_tiles_class(Q), nz = x3_tiles_class(A0);
#define X3_GO(F, Z) \
  if (nf == F && nz == Z) return half ? att_l0_bwd_x3_launch<F, Z, 1, __bf16>(s, st) : pieces == 3 ? att_l0_bwd_x3_launch<F, Z, 3, float>(s, st) : att_l0_bwd_x3_launch<F, Z, 2, float>(s, st)
  X3_GO(3, 3); X3_GO(3, 5); X3_GO(5, 3); X3_GO(5, 5);
#undef X3_GO
  clsr_set_error("%s:%d: no instance for Q = %d, A0 = %d", __FILE__, __LINE__, Q, A0);
  return CLSR_EUNSUPPORTED;
}

extern "C" int clsr_att_l0_bwd_x3(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                                  const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                                  float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                                  void* stream) {
  return att_l0_bwd_x3_any(dz0, false, 2, lddz, Wt, Kp, a, lda, q, ldq, Hn, G, T, Q, A0, da, ldda, dq, lddq, dU, lddu, dV,
                           lddv, dwp_partial, stream);
}
// the same with THREE bf16 pieces per operand (fp32 accuracy: precision="fp32")
extern "C" int clsr_att_l0_bwd_x6(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                                  const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                                  float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                                  void* stream) {
  return att_l0_bwd_x3_any(dz0, false, 3, lddz, Wt, Kp, a, lda, q, ldq, Hn, G, T, Q, A0, da, ldda, dq, lddq, dU, lddu, dV,
                           lddv, dwp_partial, stream);
}
// speed mode: dz0 stored as bf16 (uint16 bit patterns), ONE bf16 piece per operand (bf16-exact on the dz0 side)
extern "C" int clsr_att_l0_bwd_x1_h(const void* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                                    const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                                    float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                                    void* stream) {
  return att_l0_bwd_x3_any(dz0, true, 1, lddz, Wt, Kp, a, lda, q, ldq, Hn, G, T, Q, A0, da, ldda, dq, lddq, dU, lddu, dV,
                           lddv, dwp_partial, stream);
}

// ------------------------------------------------------------------------------------------------ layer 1
//   x = dz1[m, :C1] = a1*dy1 + a2*z1 + a3      BN-1 backward; dy1 = ds[m] * w_out where relu(bn1(z1)) > 0 (recomputed)
//   dh0 = x . W1^T,  dy0 = dh0 where relu(bn0(z0)) > 0
//   pass 1 (APPLY = false): per-block partial sums of dy0 and dy0 * xhat0 -> stats
//   pass 2 (APPLY = true):  dz0 = c1*dy0 + c2*z0 + c3 (the complete BN-0 backward), and the partial sums of
//                           dW1[n, c] = sum_m relu(bn0(z0))[m, n] x[m, c],  db1[c] = sum_m x[m, c]
//                           (dz1 is NOT stored: nothing reads it any more)
// The split-bf16 twin of att_l1_bwd_kernel with the MFMA operands SWAPPED: A = the dz1 tile (rows = positions; lane
// (i, g) computes the prologue for features 32c + 8g + {0..7} of position i), B = W1 rows from LDS, so that a lane of the
// result D[16 positions][16 C0-features] holds FOUR POSITIONS of ONE feature: the epilogue constants (scale0, shift0,
// c1..c3 / mean0, invstd0) are per-lane registers instead of LDS table reads, the batch-norm sums of pass 1 are
// per-lane accumulators without a cross-lane step, and relu(bn0(z0)) is already the A operand of the weight-gradient
// MFMAs.  z0 is read and dz0 written in that layout as 4-byte accesses: 16 lanes cover 64 contiguous bytes of a row --
// the same sectors as the 16-byte accesses of the position-lane kernels, the same bytes per TA cycle.
// A wave owns tiles of 32 positions (two 16-position halves = one K = 32 chunk of the weight-gradient products); the loads
// of its next tile are in flight while it computes (one 512-register wave per SIMD).
struct L1BwdArgsX {
  const void* z1; int ldz1;      // z1 / z0 / dz0: float or __bf16 (the kernel's ST)
  const float* ds;
  const float* pv[4];            // scale1, shift1, w_out, coef1 (a1 | a2 | a3, stride C1)
  const float* Wt; int Kp;       // packed fp32 W1^T (clsr_pack_batch): row n = C0 feature, K = C1
  const void* z0; int ldz0;
  const float* ev[5];            // scale0, shift0, then mean0, invstd0 (pass 1) | coef0 c1|c2|c3 stride C0 (pass 2)
  void* dz0; int lddz0;
  float* dw1;                    // pass 2: [gridDim.x][CLSR_DW_CHUNK] partial chunks (tiles kt < OT, nt < NC, bias sums)
  double* stats;                 // pass 1: [gridDim.x][2][C0]
  int M, C1, C0;
};

template <int OT, int NC, bool APPLY, int NP, typename ST>
__global__ void __launch_bounds__(256, APPLY ? 1 : X3_L1P1_OCC) att_l1_bwd_x3_kernel(L1BwdArgsX a) {
  CLSR_CHAIN_PRIO();
  constexpr unsigned SB = sizeof(ST);
  const ST* z1p = reinterpret_cast<const ST*>(a.z1);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int KC = (NC + 1) / 2, KCP = 32 * KC, WS = KCP + 8, NR = 16 * OT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  constexpr int NPW = NP > 2 ? 3 : 2;       // weight images in LDS
  __bf16* Wh = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Wl = Wh + NR * WS;
  __bf16* Wr = Wl + NR * WS;                // (NP == 3 only)
  float* ptab = reinterpret_cast<float*>(lds_raw + (size_t)NPW * NR * WS * 2);   // [5][KCP]: sc1, sh1, p = a1 * w_out, a2, a3
  {
    constexpr int C8 = WS / 8;
    for (int e = tid; e < NR * C8; e += 256) {
      const int row = e / C8, k = 8 * (e - row * C8);
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < a.C0 && k < a.C1) v = ld8f(a.Wt + (long)row * a.Kp + k);
      bf16x8 hi, lo;
      split8x(v, hi, lo);
      reinterpret_cast<bf16x8*>(Wh)[e] = hi;
      reinterpret_cast<bf16x8*>(Wl)[e] = lo;
      if constexpr (NP > 2) reinterpret_cast<bf16x8*>(Wr)[e] = to_h(v - to_f(hi) - to_f(lo));
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
  }
  __syncthreads();

  // epilogue constants of this lane's feature 16 ot + j of every C0 tile
  float sc0[OT], sh0[OT], e2[OT], e3[OT], e4[OT];
  bool nok[OT];
  int wrow[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    const int n = 16 * ot + j;
    nok[ot] = n < a.C0;
    const int nc = nok[ot] ? n : 0;
    sc0[ot] = nok[ot] ? a.ev[0][nc] : 0.f;
    sh0[ot] = nok[ot] ? a.ev[1][nc] : 0.f;
    if (APPLY) {
      e2[ot] = nok[ot] ? a.ev[2][nc] : 0.f;
      e3[ot] = nok[ot] ? a.ev[2][a.C0 + nc] : 0.f;
      e4[ot] = nok[ot] ? a.ev[2][2 * a.C0 + nc] : 0.f;
    } else {
      e2[ot] = nok[ot] ? a.ev[2][nc] : 0.f;
      e3[ot] = 0.f; e4[ot] = 0.f;
    }
    wrow[ot] = n * WS + 8 * g;
  }
  bf16x8 sel[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) sel[o][e] = (g == 2 * o + (j >> 3) && e == (j & 7)) ? (__bf16)1.0f : (__bf16)0.0f;

  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x4 accW[APPLY ? OT : 1][APPLY ? NC : 1];
  float bsum[APPLY ? NC : 1];
  float fsum[APPLY ? 1 : OT], fsq[APPLY ? 1 : OT];
  double dsum[APPLY ? 1 : OT], dsq[APPLY ? 1 : OT];
  if (APPLY) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) accW[ot][ct] = z4;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) bsum[ct] = 0.f;
  } else {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { fsum[ot] = 0.f; fsq[ot] = 0.f; dsum[ot] = 0.0; dsq[ot] = 0.0; }
  }

  // (exact sizes: a lane of a ragged last feature tile -- C0 % 16 != 0 -- addresses up to 15 floats past its row, at the
  // last row past the END of the tensor; out of range it reads 0 / its store is dropped)
  const x3_rsrc_t rz0 = x3_rsrc_n(a.z0, ((unsigned)(a.M - 1) * (unsigned)a.ldz0 + (unsigned)a.C0) * SB);
  const x3_rsrc_t rdz = x3_rsrc_n(APPLY ? a.dz0 : nullptr, APPLY ? ((unsigned)(a.M - 1) * (unsigned)a.lddz0 + (unsigned)a.C0) * SB : 0u);
  const int ntiles = (a.M + 31) >> 5;
  struct RawT { X3Raw8<ST> z1[2][KC]; unsigned z0[2][OT][4]; float ds[2]; };
  auto fetch = [&](int tile) -> RawT {
    RawT r;
    const int m0 = tile * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int m = m0 + 16 * s + j;
      const long mr = m < a.M ? m : a.M - 1;
      r.ds[s] = a.ds[mr];
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int k0 = 32 * c + 8 * g;
        r.z1[s][c] = x3_ld8raw_g<ST>(z1p + mr * a.ldz1 + (k0 < a.C1 ? k0 : 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = m0 + 16 * s + 4 * g + e;
        const unsigned off = p < a.M ? ((unsigned)p * (unsigned)a.ldz0 + (unsigned)j) * SB : X3_OOB;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) r.z0[s][ot][e] = x3_ld1raw<ST>(rz0, off + 16u * SB * ot);
      }
    }
    return r;
  };

  const int tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  // pass 2 (one 512-register wave per SIMD) keeps the next tile's loads in flight while it computes; pass 1 has no
  // weight-gradient accumulators and runs two waves per SIMD without the register prefetch (X3_L1P1_OCC)
  constexpr bool PF = APPLY || X3_L1P1_OCC < 2;
  RawT cur = fetch(tile);
  int pending = 0;
  for (; tile < ntiles; tile += tstride) {
    RawT nxt;
    if (PF) nxt = fetch(tile + tstride);      // (past the end: clamped / out-of-range loads, never used)
    const int m0 = tile * 32;
    // ---- prologue: dz1 of the lane's position (A operand: features 32c + 8g + {0..7}), split
    constexpr bool DR = APPLY || NP > 2;        // third piece: the bias sums of pass 2; every product with NP == 3
    bf16x8 dh[2][KC], dl[2][KC], dr[DR ? 2 : 1][DR ? KC : 1];      // (dl, dr: NP > 1 only)
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k0 = 32 * c + 8 * g;
      const f32x8 sc = ld8f(ptab + k0), sh = ld8f(ptab + KCP + k0), pp = ld8f(ptab + 2 * KCP + k0);
      const f32x8 a2 = ld8f(ptab + 3 * KCP + k0), a3 = ld8f(ptab + 4 * KCP + k0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x8 zz = x3_f8(cur.z1[s][c]);
        const f32x8 y = zz * sc + sh;
        f32x8 x = a2 * zz + a3;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e] > 0.f ? pp[e] * cur.ds[s] : 0.f;
        const bool pvalid = m0 + 16 * s + j < a.M;     // (features beyond C1: every table entry is 0 -> x = 0)
        const f32x8 xm = pvalid ? x : z8;
        if constexpr (NP > 1) {
          split8x(xm, dh[s][c], dl[s][c]);
          if (DR) dr[s][c] = to_h(xm - to_f(dh[s][c]) - to_f(dl[s][c]));   // (third piece: the bias sums below; NP == 3: every product)
        } else {
          dh[s][c] = to_h(xm);
        }
      }
    }
    X3_SCHED_FENCE();
    // ---- pass 2: dz1^T (feature-lane layout) through products with the identity: B operands of the weight gradient
    bf16x8 bh[APPLY ? NC : 1], bl[APPLY ? NC : 1], br[(APPLY && NP > 2) ? NC : 1];
    if (APPLY) {
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        f32x4 th0 = z4, th1 = z4;
        HMFMA(th0, dh[0][ct >> 1], sel[ct & 1]);
        HMFMA(th1, dh[1][ct >> 1], sel[ct & 1]);
        bh[ct] = cat4(to_h4(th0), to_h4(th1));
        f32x4 t = th0 + th1;
        if constexpr (NP > 1) {
          f32x4 tl0 = z4, tl1 = z4;
          HMFMA(tl0, dl[0][ct >> 1], sel[ct & 1]);
          HMFMA(tl1, dl[1][ct >> 1], sel[ct & 1]);
          bl[ct] = cat4(to_h4(tl0), to_h4(tl1));
          // (db1 = sum of dz1 cancels analytically under the batch-norm: its transposed image takes the third piece too)
          f32x4 tr0 = z4, tr1 = z4;
          HMFMA(tr0, dr[0][ct >> 1], sel[ct & 1]);
          HMFMA(tr1, dr[1][ct >> 1], sel[ct & 1]);
          if constexpr (NP > 2) br[ct] = cat4(to_h4(tr0), to_h4(tr1));
          t = (th0 + (tl0 + tr0)) + (th1 + (tl1 + tr1));
        }
        bsum[ct] += (t.x + t.y) + (t.z + t.w);
      }
    }
    bool pval[2][4];
    unsigned soff[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = m0 + 16 * s + 4 * g + e;
        pval[s][e] = p < a.M;
        soff[s][e] = pval[s][e] ? ((unsigned)p * (unsigned)a.lddz0 + (unsigned)j) * SB : X3_OOB;
      }
    // ---- per C0 tile: dh0^T (4 positions of feature 16 ot + j for both halves), epilogue in the feature-lane layout,
    //      weight gradient of the tile (k = the 8 positions a lane holds of its feature, 4 of each half)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      X3_SCHED_FENCE();
      f32x4 acc0 = z4, acc1 = z4;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const bf16x8 wh = ld8h(Wh + wrow[ot] + 32 * c);
        if constexpr (NP > 2) {         // the three products of index sum 2 first (smallest terms)
          const bf16x8 wl = ld8h(Wl + wrow[ot] + 32 * c), wr = ld8h(Wr + wrow[ot] + 32 * c);
          HMFMA(acc0, dh[0][c], wr); HMFMA(acc1, dh[1][c], wr);
          HMFMA(acc0, dl[0][c], wl); HMFMA(acc1, dl[1][c], wl);
          HMFMA(acc0, dr[0][c], wh); HMFMA(acc1, dr[1][c], wh);
        }
        if constexpr (NP > 1) {
          const bf16x8 wl = ld8h(Wl + wrow[ot] + 32 * c);
          HMFMA(acc0, dh[0][c], wl); HMFMA(acc1, dh[1][c], wl);
          HMFMA(acc0, dl[0][c], wh); HMFMA(acc1, dl[1][c], wh);
        }
        HMFMA(acc0, dh[0][c], wh); HMFMA(acc1, dh[1][c], wh);
      }
      bf16x4 xh[2], xl[2], xr[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 zz = {x3_f1<ST>(cur.z0[s][ot][0]), x3_f1<ST>(cur.z0[s][ot][1]), x3_f1<ST>(cur.z0[s][ot][2]),
                          x3_f1<ST>(cur.z0[s][ot][3])};
        const f32x4 y = zz * sc0[ot] + sh0[ot];
        const f32x4 ac = s ? acc1 : acc0;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (y[e] > 0.f && pval[s][e]) ? ac[e] : 0.f;
        if (APPLY) {
          const f32x4 o = e2[ot] * v + e3[ot] * zz + e4[ot];
#pragma unroll
          for (int e = 0; e < 4; ++e) x3_st1s<ST>(rdz, nok[ot] ? soff[s][e] + 16u * SB * ot : X3_OOB, o[e]);
          f32x4 x1;
#pragma unroll
          for (int e = 0; e < 4; ++e) x1[e] = pval[s][e] ? fmaxf(y[e], 0.f) : 0.f;
          if constexpr (NP > 1) split4(x1, xh[s], xl[s]);
          else xh[s] = to_h4(x1);
          if constexpr (NP > 2) xr[s] = to_h4(x1 - to_f4(xh[s]) - to_f4(xl[s]));
        } else {
          // (sum of dy0 * (z0 - mean0): the factor invstd0 of xhat0 is applied to the finished sums)
          const f32x4 w2 = zz - e2[ot];
          fsum[ot] += (v.x + v.y) + (v.z + v.w);
          fsq[ot] += (v.x * w2.x + v.y * w2.y) + (v.z * w2.z + v.w * w2.w);
        }
      }
      if (APPLY) {
        const bf16x8 ah = cat4(xh[0], xh[1]);
        if constexpr (NP > 2) {
          const bf16x8 al = cat4(xl[0], xl[1]), ar = cat4(xr[0], xr[1]);
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) {
            HMFMA(accW[ot][ct], ah, br[ct]);
            HMFMA(accW[ot][ct], al, bl[ct]);
            HMFMA(accW[ot][ct], ar, bh[ct]);
          }
        }
        if constexpr (NP > 1) {
          const bf16x8 al = cat4(xl[0], xl[1]);
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) HMFMA(accW[ot][ct], ah, bl[ct]);
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) HMFMA(accW[ot][ct], al, bh[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) HMFMA(accW[ot][ct], ah, bh[ct]);
      }
    }
    if (!APPLY && ++pending == 2) {      // fp32 per-lane partials of at most 16 values between flushes
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        dsum[ot] += (double)fsum[ot]; dsq[ot] += (double)fsq[ot];
        fsum[ot] = 0.f; fsq[ot] = 0.f;
      }
      pending = 0;
    }
    if (PF) cur = nxt;
    else if (tile + tstride < ntiles) cur = fetch(tile + tstride);
  }

  __syncthreads();      // the weight images / tables are dead: the LDS becomes the reduction buffer
  if (APPLY) {
    float* red = reinterpret_cast<float*>(lds_raw);
    constexpr int RED = OT * NC * 256;
    float* bred = red + 2 * RED;                 // [4][NC * 16]
    auto store_to = [&](float* dst, int tstr) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          float* t = dst + (ot * tstr + ct) * 256 + (4 * g) * 16 + j;
          const f32x4 v = accW[ot][ct];
          t[0] = v.x; t[16] = v.y; t[32] = v.z; t[48] = v.w;
        }
    };
    auto add_from = [&](const float* src) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          const float* t = src + (ot * NC + ct) * 256 + (4 * g) * 16 + j;
          accW[ot][ct] += (f32x4){t[0], t[16], t[32], t[48]};
        }
    };
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const float b = col4_sum(bsum[ct]);
      if (g == 0) bred[wave * NC * 16 + ct * 16 + j] = b;
    }
    if (wave >= 2) store_to(red + (wave - 2) * RED, NC);
    __syncthreads();
    if (wave < 2) add_from(red + wave * RED);
    __syncthreads();
    if (wave == 1) store_to(red, NC);
    __syncthreads();
    if (wave == 0) {
      add_from(red);
      float* dst = a.dw1 + (long)blockIdx.x * CLSR_DW_CHUNK;
      store_to(dst, CLSR_DW_T);
      if (g == 0) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          const int o = ct * 16 + j;
          dst[CLSR_DW_T * CLSR_DW_T * 256 + o] = (bred[o] + bred[NC * 16 + o]) + (bred[2 * NC * 16 + o] + bred[3 * NC * 16 + o]);
        }
      }
    }
  } else {
    if (pending) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { dsum[ot] += (double)fsum[ot]; dsq[ot] += (double)fsq[ot]; }
    }
    double* red = reinterpret_cast<double*>(lds_raw);   // [16 = wave * 4 + g][2][NR]
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      red[((wave * 4 + g) * 2 + 0) * NR + 16 * ot + j] = dsum[ot];
      red[((wave * 4 + g) * 2 + 1) * NR + 16 * ot + j] = dsq[ot];
    }
    __syncthreads();
    for (int e = tid; e < 2 * NR; e += 256) {
      const int which = e / NR, c = e - which * NR;
      if (c < a.C0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[(w * 2 + which) * NR + c];
        if (which == 1) t *= (double)a.ev[3][c];      // sum dy0 * xhat0 = invstd0 * sum dy0 * (z0 - mean0)
        a.stats[((long)blockIdx.x * 2 + which) * a.C0 + c] = t;
      }
    }
  }
}

static int l1x_grid(int M, bool apply = true) {
  int gx = clsr_cdiv(clsr_cdiv(M, 32), 4);
  const int cap = apply ? 256 : 256 * X3_L1P1_OCC;      // one workgroup per CU (pass 1: X3_L1P1_OCC)
  if (gx > cap) gx = cap;
  return gx < 1 ? 1 : gx;
}

extern "C" int clsr_att_l1_bwd_x3_supported(int C1, int C0) {
  return C1 >= 8 && C1 <= 48 && C0 >= 4 && C0 <= 80 && C1 % 8 == 0 && C0 % 4 == 0;
}
// partial rows of pass 1 (stats) / partial chunks of pass 2 (dw1_partial): sized for the larger of the two
extern "C" int clsr_att_l1_bwd_x3_parts(int M) { return l1x_grid(M, false); }

template <int OT, int NC, int NP, typename ST>
static int l1x_launch(const L1BwdArgsX& a, bool apply, hipStream_t stream) {
  constexpr int KC = (NC + 1) / 2, KCP = 32 * KC, WS = KCP + 8, NR = 16 * OT;
  size_t shmem = (size_t)(NP > 2 ? 3 : 2) * NR * WS * 2 + (size_t)5 * KCP * 4;
  const size_t red = apply ? ((size_t)2 * OT * NC * 256 + 4 * NC * 16) * 4 : (size_t)16 * 2 * NR * 8;
  if (shmem < red) shmem = red;
  dim3 grid(l1x_grid(a.M, apply));
  if (apply) {
    auto kernel = att_l1_bwd_x3_kernel<OT, NC, true, NP, ST>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  } else {
    auto kernel = att_l1_bwd_x3_kernel<OT, NC, false, NP, ST>;
    if (shmem > 64 * 1024) CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, stream, a);
  }
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// coef0 == NULL: pass 1 (stats);  coef0 given: pass 2 (dz0 + the partial chunks of dW1 / db1)
static int att_l1_bwd_x3_any(const void* z1, bool half, int pieces, int ldz1, const float* ds, const float* scale1, const float* shift1,
                             const float* w_out, const float* coef1, const float* Wt, int Kp, const void* z0,
                             int ldz0, const float* scale0, const float* shift0, const float* mean0,
                             const float* invstd0, const float* coef0, void* dz0, int lddz0, float* dw1_partial,
                             double* stats, int M, int C1, int C0, void* stream) {
  CLSR_CHECK_ARG(z1 && ds && scale1 && shift1 && w_out && coef1 && Wt && z0 && scale0 && shift0 && M > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l1_bwd_x3_supported(C1, C0));
  CLSR_CHECK_ARG(coef0 ? (dz0 && lddz0 >= C0 && dw1_partial) : (mean0 && invstd0 && stats));
  CLSR_CHECK_ARG(ldz1 >= C1 && ldz0 >= C0 && Kp >= 16 * clsr_cdiv(C1, 16));
  CLSR_CHECK_SUPPORTED(ldz1 % (half ? 8 : 4) == 0 && Kp % 4 == 0 && ((uintptr_t)z1 % 16) == 0 && ((uintptr_t)Wt % 16) == 0 &&
                       ((uintptr_t)z0 % 4) == 0 && (!coef0 || ((uintptr_t)dz0 % 4) == 0));
  // 32-bit byte offsets into z0 / dz0 (raw buffer accesses; bit 31 marks a lane that must not touch memory)
  CLSR_CHECK_SUPPORTED(((long)M * ldz0 + 80) * 4 < 0x7fffffffL && (!coef0 || ((long)M * lddz0 + 80) * 4 < 0x7fffffffL));
  L1BwdArgsX a = {};
  a.z1 = z1; a.ldz1 = ldz1; a.ds = ds; a.pv[0] = scale1; a.pv[1] = shift1; a.pv[2] = w_out; a.pv[3] = coef1;
  a.Wt = Wt; a.Kp = Kp; a.z0 = z0; a.ldz0 = ldz0; a.ev[0] = scale0; a.ev[1] = shift0;
  a.M = M; a.C1 = C1; a.C0 = C0; a.stats = stats;
  const bool apply = coef0 != nullptr;
  if (apply) { a.ev[2] = coef0; a.dz0 = dz0; a.lddz0 = lddz0; a.dw1 = dw1_partial; }
  else { a.ev[2] = mean0; a.ev[3] = invstd0; }
  hipStream_t s = (hipStream_t)stream;
  const int ot = x3_tiles_class(C0), nc = C1 <= 32 ? 2 : 3;
#define L1X_GO(O, N) \
  if (ot == O && nc == N) return half ? l1x_launch<O, N, 1, __bf16>(a, apply, s) : pieces == 3 ? l1x_launch<O, N, 3, float>(a, apply, s) : l1x_launch<O, N, 2, float>(a, apply, s)
  L1X_GO(3, 2); L1X_GO(3, 3); L1X_GO(5, 2); L1X_GO(5, 3);
#undef L1X_GO
  clsr_set_error("%s:%d: no instance for C1 = %d, C0 = %d", __FILE__, __LINE__, C1, C0);
  return CLSR_EUNSUPPORTED;
}

extern "C" int clsr_att_l1_bwd_x3(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                                  const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0,
                                  int ldz0, const float* scale0, const float* shift0, const float* mean0,
                                  const float* invstd0, const float* coef0, float* dz0, int lddz0, float* dw1_partial,
                                  double* stats, int M, int C1, int C0, void* stream) {
  return att_l1_bwd_x3_any(z1, false, 2, ldz1, ds, scale1, shift1, w_out, coef1, Wt, Kp, z0, ldz0, scale0, shift0, mean0,
                           invstd0, coef0, dz0, lddz0, dw1_partial, stats, M, C1, C0, stream);
}
// the same with THREE bf16 pieces per operand (fp32 accuracy: precision="fp32")
extern "C" int clsr_att_l1_bwd_x6(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                                  const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0,
                                  int ldz0, const float* scale0, const float* shift0, const float* mean0,
                                  const float* invstd0, const float* coef0, float* dz0, int lddz0, float* dw1_partial,
                                  double* stats, int M, int C1, int C0, void* stream) {
  return att_l1_bwd_x3_any(z1, false, 3, ldz1, ds, scale1, shift1, w_out, coef1, Wt, Kp, z0, ldz0, scale0, shift0, mean0,
                           invstd0, coef0, dz0, lddz0, dw1_partial, stats, M, C1, C0, stream);
}
// speed mode: z1 / z0 / dz0 stored as bf16 (uint16 bit patterns), ONE bf16 piece per operand
extern "C" int clsr_att_l1_bwd_x1_h(const void* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                                    const float* w_out, const float* coef1, const float* Wt, int Kp, const void* z0,
                                    int ldz0, const float* scale0, const float* shift0, const float* mean0,
                                    const float* invstd0, const float* coef0, void* dz0, int lddz0, float* dw1_partial,
                                    double* stats, int M, int C1, int C0, void* stream) {
  return att_l1_bwd_x3_any(z1, true, 1, ldz1, ds, scale1, shift1, w_out, coef1, Wt, Kp, z0, ldz0, scale0, shift0, mean0,
                           invstd0, coef0, dz0, lddz0, dw1_partial, stats, M, C1, C0, stream);
}
