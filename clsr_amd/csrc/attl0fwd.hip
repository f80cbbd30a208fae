// Re-associated first attention layer, exact (fp32) mode, one WAVE per history (gfx950):
//   z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp      + per-feature sum / sum of squares for the batch-norm
// (reference clsr.py:368-370 + base_model.py:664-673).  Same results as clsr_pgemm with the Xmul prologue and the
// addU/addV epilogue; a different mapping onto the machine:
//  * the MFMA operands are swapped (A = the (a*q) tile, rows = positions; B = weights), so a lane of the result
//    D[16 positions][16 features] holds FOUR POSITIONS of ONE feature: the batch-norm column sums are plain per-lane
//    accumulators (no cross-lane work per tile), U is added from registers, V[r, feature] is one LDS scalar;
//  * the wave walks the 16-step tiles of ITS history and, inside a tile, the G rows of the group: a[h,t,:] and
//    U[h,t,:] are loaded once per tile and re-used for the G rows (the position-tiled kernel re-reads them per row),
//    q[r,:] and V[r,:] sit in LDS, and the only global memory instructions between two MFMA blocks are the stores;
//  * v_mfma_f32_16x16x4_f32: MFMA #r of the 16-wide k-chunk kk uses feature 16kk + 4g + r, the A operand of a chunk is
//    the float4 (a * q) of the lane's position, the B operand one ds_read_b128 of the packed weights.
// ~1 VALU instruction per MFMA inside the row loop, 2.1 per MFMA over the whole launch (profiles/r02_att_pmc.md;
// 3.4 in pgemm_fast_kernel); matrix pipe busy 63 % of the CU-busy time.
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

#define AF_GMAX 8

struct AttL0FwdArgs {
  const float* a; int lda;      // [Hn*T, Q]
  const float* q; int ldq;      // [R, Q]
  const float* Wt; int Kp;      // packed Wp (clsr_pack_batch): row n = out feature, K = Q
  const float* U; int ldu;      // [Hn*T, A0]
  const float* V; int ldv;      // [R, A0]
  void* z0; int ldz;            // [R*T, A0] float, or __bf16 (the kernel's ST: the speed mode's storage)
  double* stats;                // [gridDim.x][2][A0] per-block partial sums, or NULL
  long Hn;
  int G, T, Q, A0;
};

// NZ = 16-feature tiles of A0 (outputs), NK = 16-wide chunks of Q (reduction)
// X3: the product as split-bf16 sums (a*q)h.Wh + (a*q)h.Wl + (a*q)l.Wh on v_mfma_f32_16x16x32_bf16 (fp32 accumulators that
// start from U + V: 2^-16 relative per product term).  The fp32 kernel is bound by the matrix pipe (100 fp32 MFMAs of 32
// cycles per 16 x 80 tile, 0.57 of the fp32 matrix peak); with 30 bf16 MFMAs of ~17 cycles it is bound by its stores.  A
// K = 32 chunk c takes the two 16-wide chunks 2c, 2c + 1 of the fp32 kernel side by side (k slot (g4, e) = feature
// 32c + 16 (e >> 2) + 4 g4 + (e & 3): the loads of the A operand are the same float4 pairs), the bf16 hi / lo images of the
// weights are laid out in LDS in that slot order.
// store one 16-position tile row piece: 4 floats, or 8 values rounded to bf16 (16 bytes either way)
template <typename ST> __device__ __forceinline__ void af_store_piece(ST* dst, const float* src);
template <> __device__ __forceinline__ void af_store_piece<float>(float* dst, const float* src) {
  __builtin_nontemporal_store(ld4(src), reinterpret_cast<f32x4*>(dst));
}
template <> __device__ __forceinline__ void af_store_piece<__bf16>(__bf16* dst, const float* src) {
  __builtin_nontemporal_store(to_h(ld8f(src)), reinterpret_cast<bf16x8*>(dst));
}

template <int NZ, int NK, int NP, typename ST>      // NP = 0: fp32-input MFMAs; 1 / 2 / 3: bf16 pieces per operand (speed mode / x3 / x6 products)
__global__ void __launch_bounds__(256, (NK >= 8 && NP >= 2) ? 1 : 2) att_l0_fwd_kernel(AttL0FwdArgs s) {      // (K = 128 with two / three weight images: 110-151 KB of LDS, one workgroup per CU anyway)
  CLSR_CHAIN_PRIO();
  constexpr int PW = sizeof(ST) == 2 ? 8 : 4;      // values per 16-byte store piece
  ST* z0p = reinterpret_cast<ST*>(s.z0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int ZP = 16 * NZ, QP = 16 * NK;
  constexpr int NC = (NK + 1) / 2, WS = 32 * NC + 8;      // X3: K = 32 chunks, bf16 row stride (conflict-free 16-byte reads)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  const int Kp = s.Kp;
  float* Wl = reinterpret_cast<float*>(lds_raw);
  __bf16* Wh = reinterpret_cast<__bf16*>(lds_raw);
  constexpr bool X3 = NP > 0;
  constexpr int NPI = NP > 0 ? NP : 1;
  const size_t wfloats = X3 ? (size_t)NP * ZP * WS / 2 : (size_t)ZP * Kp;      // (NP images x ZP x WS bf16)
  float* wl = Wl + wfloats + (size_t)wave * AF_GMAX * (QP + ZP);
  float* qs = wl;                     // [G][QP]
  float* vs = wl + AF_GMAX * QP;      // [G][ZP]
  double* red = reinterpret_cast<double*>(Wl + wfloats + (size_t)4 * AF_GMAX * (QP + ZP));   // [4][2][ZP]
  constexpr int TS = ZP + 4;   // row stride of the store-transposition tile: (4 g4 + e) * TS + 16 z + j hits 64 distinct banks
  float* tb = reinterpret_cast<float*>(red + 4 * 2 * ZP) + (size_t)wave * 16 * TS;   // [16 positions][TS] per wave
  if (X3) {
    constexpr int C8 = WS / 8;
    for (int e = tid; e < ZP * C8; e += 256) {
      const int row = e / C8, k8 = e - row * C8;           // slots 8 k8 .. 8 k8 + 7 of the row: chunk c, lane group gg
      const int c = k8 >> 2, gg = k8 & 3;
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < s.A0 && c < NC) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int f0 = 32 * c + 16 * hf + 4 * gg;
          if (f0 < s.Q) {                                   // (Q % 4 == 0)
            const f32x4 w = ld4(s.Wt + (long)row * Kp + f0);
            v[4 * hf] = w.x; v[4 * hf + 1] = w.y; v[4 * hf + 2] = w.z; v[4 * hf + 3] = w.w;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const bf16x8 h = to_h(v);
        reinterpret_cast<bf16x8*>(Wh + (size_t)i * ZP * WS)[e] = h;
        v -= to_f(h);
      }
    }
  } else {
    const int Kq = Kp >> 2;
    const int nrows = 16 * ((s.A0 + 15) >> 4);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < ZP * Kq; e += 256) {
      const int row = e / Kq, c = e - row * Kq;
      reinterpret_cast<f32x4*>(Wl)[e] = row < nrows ? ld4(s.Wt + (long)row * Kp + 4 * c) : z;
    }
  }
  __syncthreads();
  const float* ldsB = Wl + (long)j * Kp + 4 * g4;   // + 16 z Kp + 16 kk
  const int ldsH = j * WS + 8 * g4;                  // X3: + 16 z WS + 32 c
  // acc[z] += x . W over all chunks (x[kk] = the (a * q) float4 of the lane's position for chunk kk)
  auto mac = [&](f32x4 (&acc)[NZ], const f32x4 (&x)[NK], int woff) {
    if (X3) {
      const f32x4 z4_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 lo4 = x[2 * c], hi4 = 2 * c + 1 < NK ? x[2 * c + 1 < NK ? 2 * c + 1 : 0] : z4_;
        f32x8 v = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        bf16x8 ap[NPI];
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          ap[i] = to_h(v);
          if (i + 1 < NPI) v -= to_f(ap[i]);
        }
        // every piece product whose indices sum to <= NP - 1; ONE weight piece in registers at a time, the smaller x pieces
        // first (all pieces of the five tiles at once were 60 VGPRs: the K = 128 instance spilled)
#pragma unroll
        for (int wp = NPI - 1; wp >= 0; --wp) {
          bf16x8 w[NZ];
#pragma unroll
          for (int z = 0; z < NZ; ++z) w[z] = ld8h(Wh + (size_t)wp * ZP * WS + ldsH + woff + z * 16 * WS + 32 * c);
#pragma unroll
          for (int i = NPI - 1 - wp; i >= 0; --i)
#pragma unroll
            for (int z = 0; z < NZ; ++z) HMFMA(acc[z], ap[i], w[z]);
        }
      }
    } else {
      const float* lb = ldsB + woff;
      // weights of chunk kk + 1 are read from LDS while chunk kk is multiplied
      f32x4 w[2][NZ];
#pragma unroll
      for (int z = 0; z < NZ; ++z) w[0][z] = ld4(lb + (long)z * 16 * Kp);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        if (kk + 1 < NK) {
#pragma unroll
          for (int z = 0; z < NZ; ++z) w[(kk + 1) & 1][z] = ld4(lb + (long)z * 16 * Kp + 16 * (kk + 1));
        }
#pragma unroll
        for (int z = 0; z < NZ; ++z) MFMA4(acc[z], x[kk].x, w[kk & 1][z].x);
#pragma unroll
        for (int z = 0; z < NZ; ++z) MFMA4(acc[z], x[kk].y, w[kk & 1][z].y);
#pragma unroll
        for (int z = 0; z < NZ; ++z) MFMA4(acc[z], x[kk].z, w[kk & 1][z].z);
#pragma unroll
        for (int z = 0; z < NZ; ++z) MFMA4(acc[z], x[kk].w, w[kk & 1][z].w);
      }
    }
  };

  const int G = s.G, T = s.T;
  // the ragged last tile of a history (T % 16 steps) is PACKED across the G rows of the group when they fit into one
  // tile: T = 50, G = 5 -> 3 full tiles per row + one tile with the 5 x 2 left-over positions = 16 MFMA tiles per
  // history instead of 20 (the matrix pipe is what this kernel waits for)
  const int rem = T & 15;
  const bool packed = rem > 0 && G * rem <= 16;
  const int NTT = packed ? T >> 4 : (T + 15) >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  double d1[NZ], d2[NZ];
#pragma unroll
  for (int z = 0; z < NZ; ++z) { d1[z] = 0.0; d2[z] = 0.0; }
  int ncl[NZ];
  bool nok[NZ];
#pragma unroll
  for (int z = 0; z < NZ; ++z) {
    nok[z] = 16 * z + j < s.A0;
    ncl[z] = nok[z] ? 16 * z + j : 0;
  }

  for (long h = (long)blockIdx.x * 4 + wave; h < s.Hn; h += (long)gridDim.x * 4) {
    for (int e = lane; e < G * QP; e += 64) {
      const int g = e / QP, n = e - g * QP;
      qs[e] = n < s.Q ? s.q[(h * G + g) * s.ldq + n] : 0.f;
    }
    for (int e = lane; e < G * ZP; e += 64) {
      const int g = e / ZP, n = e - g * ZP;
      vs[e] = n < s.A0 ? s.V[(h * G + g) * s.ldv + n] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    struct Tile { f32x4 a[NK]; f32x4 u[NZ]; };
    auto load_tile = [&](int tt) -> Tile {
      Tile r;
      const int t0 = 16 * tt;
      // (lanes whose 4 g4 offset is already past the row -- Q < 16 -- read the row start: their values are zeroed
      // below, and an unclamped address ran up to 48 bytes past the END of the tensor at its last row)
      const float* ap = s.a + (h * T + min(t0 + j, T - 1)) * s.lda + (4 * g4 < s.Q ? 4 * g4 : 0);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) r.a[kk] = ld4(ap + (16 * kk + 4 * g4 < s.Q ? 16 * kk : 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* up = s.U + (h * T + min(t0 + 4 * g4 + e, T - 1)) * s.ldu;
#pragma unroll
        for (int z = 0; z < NZ; ++z) r.u[z][e] = up[ncl[z]];
      }
      return r;
    };
    float s1[NZ], s2[NZ];
#pragma unroll
    for (int z = 0; z < NZ; ++z) { s1[z] = 0.f; s2[z] = 0.f; }
    // (a register prefetch of the next tile's operands was measured: no gain, 40 VGPRs)
    for (int tt = 0; tt < NTT; ++tt) {
      Tile cur = load_tile(tt);
      const int t0 = 16 * tt;
      const bool fullT = t0 + 16 <= T;
      const bool pv = t0 + j < T;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) cur.a[kk] = (pv && 16 * kk + 4 * g4 < s.Q) ? cur.a[kk] : z4;
      // make the tile's operands ARRIVE here: left pending, the counter pass puts a vmcnt(0) inside the g loop (it
      // cannot tell the first iteration from the later ones), which then also waits for every store of the loop
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) asm volatile("" ::"v"(cur.a[kk]));
#pragma unroll
      for (int z = 0; z < NZ; ++z) asm volatile("" ::"v"(cur.u[z]));
      for (int g = 0; g < G; ++g) {
        // (opaque per iteration: keeps the compiler from parking the loop-invariant weight reads in ~70 VGPRs)
        int woff = 0;
        asm volatile("" : "+v"(woff));   // (an opaque OFFSET: an opaque pointer would lose its LDS address space -> flat loads)
        f32x4 x[NK];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) x[kk] = cur.a[kk] * ld4(qs + g * QP + 16 * kk + 4 * g4);
        f32x4 acc[NZ];
#pragma unroll
        for (int z = 0; z < NZ; ++z) acc[z] = cur.u[z] + vs[g * ZP + 16 * z + j];
        mac(acc, x, woff);
        if constexpr (sizeof(ST) == 2) {
          // bf16 storage: the batch-norm statistics are those of the STORED values (what the next layer and the backward
          // pass read back), as in csrc/hgemm.hip and in the oracle's bf16 emulation
#pragma unroll
          for (int z = 0; z < NZ; ++z) acc[z] = __builtin_convertvector(__builtin_convertvector(acc[z], bf16x4), f32x4);
        }
        // the result tile goes through LDS once so that every position's A0 floats leave as consecutive 16-byte stores
        // (a lane holds 4 positions of one feature: stored directly that is 4 x NZ dword stores of 64-byte pieces)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int z = 0; z < NZ; ++z) tb[(4 * g4 + e) * TS + 16 * z + j] = acc[z][e];
        if (s.stats) {
          if (fullT) {
#pragma unroll
            for (int z = 0; z < NZ; ++z) {
              s1[z] += (acc[z].x + acc[z].y) + (acc[z].z + acc[z].w);
              s2[z] += (acc[z].x * acc[z].x + acc[z].y * acc[z].y) + (acc[z].z * acc[z].z + acc[z].w * acc[z].w);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool tv = t0 + 4 * g4 + e < T;
#pragma unroll
              for (int z = 0; z < NZ; ++z) {
                const float v = tv ? acc[z][e] : 0.f;
                s1[z] += v;
                s2[z] = fmaf(v, v, s2[z]);
              }
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        ST* zrow = z0p + ((h * G + g) * T + t0) * s.ldz;
        constexpr int C4 = ZP / PW;
#pragma unroll
        for (int i = 0; i < (16 * C4 + 63) / 64; ++i) {
          const int idx = lane + 64 * i;
          const int row = idx / C4, c4 = idx - row * C4;
          if (idx < 16 * C4 && t0 + row < T && PW * c4 < s.A0)
            af_store_piece<ST>(zrow + (long)row * s.ldz + PW * c4, tb + row * TS + PW * c4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (packed) {
      const int np = G * rem, T0 = T - rem;
      const bool pv = j < np;                         // A operand: lane j = packed position j = (row j / rem, step j % rem)
      const int gi = pv ? j / rem : 0, ti = T0 + (pv ? j - gi * rem : 0);
      f32x4 x[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int kc = 16 * kk + 4 * g4;
        x[kk] = (pv && kc < s.Q) ? ld4(s.a + (h * T + ti) * s.lda + kc) * ld4(qs + gi * QP + kc) : z4;
      }
      f32x4 acc[NZ];
      bool vp[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {                    // result layout: this lane's positions 4 g4 + e
        const int p = 4 * g4 + e;
        vp[e] = p < np;
        const int gp = vp[e] ? p / rem : 0, tp = T0 + (vp[e] ? p - gp * rem : 0);
        const float* up = s.U + (h * T + tp) * s.ldu;
#pragma unroll
        for (int z = 0; z < NZ; ++z) acc[z][e] = up[ncl[z]] + vs[gp * ZP + 16 * z + j];
      }
      mac(acc, x, 0);
      if constexpr (sizeof(ST) == 2) {
#pragma unroll
        for (int z = 0; z < NZ; ++z) acc[z] = __builtin_convertvector(__builtin_convertvector(acc[z], bf16x4), f32x4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
          tb[(4 * g4 + e) * TS + 16 * z + j] = acc[z][e];
          const float v = vp[e] ? acc[z][e] : 0.f;
          s1[z] += v;
          s2[z] = fmaf(v, v, s2[z]);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      constexpr int C4t = ZP / PW;
#pragma unroll
      for (int i = 0; i < (16 * C4t + 63) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int row = idx / C4t, c4 = idx - row * C4t;
        if (idx < 16 * C4t && row < np && PW * c4 < s.A0) {
          const int gr = row / rem, tr = T0 + row - gr * rem;
          af_store_piece<ST>(z0p + ((h * G + gr) * T + tr) * s.ldz + PW * c4, tb + row * TS + PW * c4);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    // at most 4 * NTT * G values per lane and feature since the last flush: fp32 partials -> double accumulators
#pragma unroll
    for (int z = 0; z < NZ; ++z) { d1[z] += (double)s1[z]; d2[z] += (double)s2[z]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }

  if (s.stats) {
#pragma unroll
    for (int z = 0; z < NZ; ++z) {
      double a1 = d1[z], a2 = d2[z];
      a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
      a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
      if (g4 == 0) {
        red[(wave * 2 + 0) * ZP + 16 * z + j] = a1;
        red[(wave * 2 + 1) * ZP + 16 * z + j] = a2;
      }
    }
    __syncthreads();
    for (int e = tid; e < 2 * ZP; e += 256) {
      const int which = e / ZP, n = e - which * ZP;
      if (n < s.A0) {
        double t = 0.0;
        for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * ZP + n];
        s.stats[((long)blockIdx.x * 2 + which) * s.A0 + n] = t;
      }
    }
  }
}

static int af_grid(long Hn) {
  long gx = (Hn + 3) / 4;
  return (int)(gx > 512 ? 512 : gx);
}
static int af_class(int n) { return n <= 48 ? 3 : (n <= 80 ? 5 : 8); }

// 1 when clsr_att_l0_fwd handles this shape (otherwise: clsr_pgemm with Xmul / addU / addV)
extern "C" int clsr_att_l0_fwd_supported(int G, int Q, int A0) {
  // (Q up to 128: the per-row half of the short-term query of 128-wide layers, BASELINE configs[4])
  return G >= 1 && G <= AF_GMAX && Q >= 4 && Q <= 128 && A0 >= 4 && A0 <= 80 && Q % 4 == 0 && A0 % 4 == 0;
}
// number of per-block partial rows the statistics buffer receives: [parts][2][A0] doubles
extern "C" int clsr_att_l0_fwd_stats_parts(long Hn) { return af_grid(Hn); }

template <int NZ, int NK, int NP, typename ST = float>
static int att_l0_fwd_launch(const AttL0FwdArgs& a, hipStream_t stream) {
  constexpr int WS = 32 * ((NK + 1) / 2) + 8;
  size_t shmem = (NP ? (size_t)NP * 16 * NZ * WS * 2 : (size_t)16 * NZ * a.Kp * 4) + (size_t)4 * AF_GMAX * (16 * NK + 16 * NZ) * 4 +
                 (size_t)4 * 2 * 16 * NZ * 8 + (size_t)4 * 16 * (16 * NZ + 4) * 4;
  auto kernel = att_l0_fwd_kernel<NZ, NK, NP, ST>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(af_grid(a.Hn)), dim3(256), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

static int att_l0_fwd_any(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                          const float* U, int ldu, const float* V, int ldv, void* z0, int ldz, double* stats,
                          long Hn, int G, int T, int Q, int A0, int pieces, void* stream) {
  CLSR_CHECK_ARG(a && q && Wt && U && V && z0 && Hn > 0 && T > 0);
  CLSR_CHECK_SUPPORTED(clsr_att_l0_fwd_supported(G, Q, A0));
  CLSR_CHECK_SUPPORTED(lda % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)Wt % 16) == 0);
  CLSR_CHECK_ARG(lda >= Q && ldq >= Q && Kp >= 16 * clsr_cdiv(Q, 16) && ldu >= A0 && ldv >= A0 && ldz >= A0);
  CLSR_CHECK_SUPPORTED(ldz % (pieces == 1 ? 8 : 4) == 0 && ((uintptr_t)z0 % 16) == 0 && (pieces != 1 || A0 % 8 == 0));
  AttL0FwdArgs s = {};
  s.a = a; s.lda = lda; s.q = q; s.ldq = ldq; s.Wt = Wt; s.Kp = Kp; s.U = U; s.ldu = ldu; s.V = V; s.ldv = ldv;
  s.z0 = z0; s.ldz = ldz; s.stats = stats; s.Hn = Hn; s.G = G; s.T = T; s.Q = Q; s.A0 = A0;
  hipStream_t st = (hipStream_t)stream;
  const int nz = af_class(A0), nk = af_class(Q);
#define AF_GO(Z, K) \
  if (nz == Z && nk == K) \
    return pieces == 3 ? att_l0_fwd_launch<Z, K, 3>(s, st) : pieces == 2 ? att_l0_fwd_launch<Z, K, 2>(s, st) \
         : pieces == 1 ? att_l0_fwd_launch<Z, K, 1, __bf16>(s, st) : att_l0_fwd_launch<Z, K, 0>(s, st)
  AF_GO(3, 3); AF_GO(3, 5); AF_GO(5, 3); AF_GO(5, 5); AF_GO(3, 8); AF_GO(5, 8);
#undef AF_GO
  return CLSR_OK;
}

extern "C" int clsr_att_l0_fwd(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                               const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                               long Hn, int G, int T, int Q, int A0, void* stream) {
  return att_l0_fwd_any(a, lda, q, ldq, Wt, Kp, U, ldu, V, ldv, z0, ldz, stats, Hn, G, T, Q, A0, 0, stream);
}
// the same with the product as split-bf16 sums (see att_l0_fwd_kernel<.., X3>)
extern "C" int clsr_att_l0_fwd_x3(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                                  const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                                  long Hn, int G, int T, int Q, int A0, void* stream) {
  return att_l0_fwd_any(a, lda, q, ldq, Wt, Kp, U, ldu, V, ldv, z0, ldz, stats, Hn, G, T, Q, A0, 2, stream);
}
// ... over three bf16 pieces per operand (2^-23 relative: the level of the fp32 fma chain; 60 bf16 MFMAs of ~20 cycles per
// 16 x 80 tile instead of 100 fp32 MFMAs of 32)
extern "C" int clsr_att_l0_fwd_x6(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                                  const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                                  long Hn, int G, int T, int Q, int A0, void* stream) {
  return att_l0_fwd_any(a, lda, q, ldq, Wt, Kp, U, ldu, V, ldv, z0, ldz, stats, Hn, G, T, Q, A0, 3, stream);
}
// speed mode: ONE bf16 piece per operand (fp32 accumulation from U + V), z0 stored as bf16 (uint16 bit patterns, A0 % 8 == 0);
// the batch-norm sums are those of the stored (rounded) values
extern "C" int clsr_att_l0_fwd_x1_h(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                                    const float* U, int ldu, const float* V, int ldv, void* z0, int ldz, double* stats,
                                    long Hn, int G, int T, int Q, int A0, void* stream) {
  return att_l0_fwd_any(a, lda, q, ldq, Wt, Kp, U, ldu, V, ldv, z0, ldz, stats, Hn, G, T, Q, A0, 1, stream);
}
