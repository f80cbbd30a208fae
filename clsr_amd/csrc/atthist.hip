// History-level prologue of an attention block as ONE launch of split-bf16 products (gfx950):
//   a[h,t,:]  = keys[h,t,:] . A                                   (reference clsr.py:351-356, attention_mat)
//   U[h,t,:]  = a[h,t,:] . Wu  (+ (a[h,t,:qh] * q_hist[h,:]) . Wp1)   Wu = W0a + W0d of the re-associated first layer
//                                                                 (clsr.py:368-370; the qh term: net.py _att_qh)
// It replaces three position-tiled fp32 GEMM launches (36 + 83 + 78 us on the step's dependent chain at configs[1],
// profiles/r05a_step_timeline_fp32.txt, for 5.2 GFLOP and 33 MB read / 131 MB written) by one pass over the keys.
// One wave per history, 16-step tiles.  Chaining two products needs the first result as an A operand (rows = positions),
// while an MFMA leaves it with four positions of ONE feature per lane: it is brought back by products with the identity
// on the bf16 pipe (exact for the hi and the lo image), which yield four consecutive FEATURES of one position per lane;
// two such tiles side by side are one bf16x8 operand with the k slot (g, e) = feature 32c + 16 (e >> 2) + 4g + (e & 3)
// -- the weight images of the second product are laid out in LDS in that slot order.
// Every product is xh.yh + xh.yl + xl.yh of bf16 pairs with fp32 accumulation (2^-16 relative per term).
#include "common.h"
#include "clsr_hip.h"
#include "hmma.h"

// keeps the scheduler from hoisting the LDS weight reads of every chunk to the top of the tile (it then spills)
#define AH_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef __amdgpu_buffer_rsrc_t ah_rsrc_t;
__device__ __forceinline__ ah_rsrc_t ah_rsrc(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ void ah_st1(ah_rsrc_t rs, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
}
__device__ __forceinline__ bf16x4 ah_h4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 ah_f4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x8 ah_cat(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

struct AttHistFwdArgs {
  const float* keys; int ldk;      // [Hn*T, Dk]
  const float* At; int Kpa;        // packed attention_mat (clsr_pack_batch): row n = a feature (Q rows), K = Dk
  const float* Wut; int Kpu;       // packed Wu: row n = A0 feature, K = Q
  const float* Wpt; int Kpp;       // packed Wp[:qh]: row n = A0 feature, K = qh (NULL: no history-level query)
  const float* qh; int ldqh;       // [Hn, qh]
  float* a; int lda;               // [Hn*T, Q]
  float* U; int ldu;               // [Hn*T, A0]
  long Hn;
  int T, Dk, Q, A0, nqh;
};

// Pieces of a split fp32 value: x = p0 + p1 (+ p2), p0 = RNE_bf16(x), p1 = RNE_bf16(x - p0), p2 = RNE_bf16(x - p0 - p1).
// NP = 2 ("x3": products p0.q0 + p0.q1 + p1.q0, 2^-16 relative per term), NP = 3 ("x6": every product whose piece indices
// sum to <= 2, 2^-23 relative -- the level of an fp32 fma chain).  The FORWARD products of the parity mode use x6: a
// 2^-16 perturbation of a pre-activation flips ~100 x more ReLU decisions than fp32 rounding does, which on the small
// golden batches shows up as 1e-3-of-scale steps in single gradient tensors (tests/test_step_gpu.py).
template <int NP> struct AhP8 { bf16x8 p[NP]; };
template <int NP> struct AhP4 { bf16x4 p[NP]; };
template <int NP> __device__ __forceinline__ AhP8<NP> ah_split8(f32x8 v) {
  AhP8<NP> r;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    r.p[i] = to_h(v);
    if (i + 1 < NP) v -= to_f(r.p[i]);
  }
  return r;
}
template <int NP> __device__ __forceinline__ AhP4<NP> ah_split4(f32x4 v) {
  AhP4<NP> r;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    r.p[i] = ah_h4(v);
    if (i + 1 < NP) v -= ah_f4(r.p[i]);
  }
  return r;
}
// acc[n] += A . W[n] for n < N: every piece product with index sum <= NP - 1, smallest terms first; W pieces from the LDS
// images (piece i of tile n at img + i * pstride + n * tstride + off)
template <int NP, int N>
__device__ __forceinline__ void ah_mac(f32x4 (&acc)[N], const AhP8<NP>& a, const __bf16* img, int pstride, int tstride, int off) {
  bf16x8 w[NP][N];
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int n = 0; n < N; ++n) w[i][n] = ld8h(img + i * pstride + n * tstride + off);
#pragma unroll
  for (int sidx = NP - 1; sidx >= 0; --sidx)
#pragma unroll
    for (int i = 0; i <= sidx; ++i)
#pragma unroll
      for (int n = 0; n < N; ++n) HMFMA(acc[n], a.p[i], w[sidx - i][n]);
}

// fp32 packed weights [rows][Kp] -> NP bf16 piece images [NP][rows_pad][WS]; PERM: slot (c, g, e) <- feature 32c + 16 (e >> 2) + 4g + (e & 3)
template <bool PERM, int NP>
__device__ __forceinline__ void ah_stage(__bf16* img, int WS, int rows_pad, const float* Wt, int Kp, int rows, int K,
                                         int tid, int nthreads) {
  const int C8 = WS / 8;
  for (int e = tid; e < rows_pad * C8; e += nthreads) {
    const int row = e / C8, k8 = e - row * C8;
    f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      if (PERM) {
        const int c = k8 >> 2, gg = k8 & 3;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int f0 = 32 * c + 16 * hf + 4 * gg;
          if (f0 < K) {                                   // (K % 4 == 0)
            const f32x4 w = ld4(Wt + (long)row * Kp + f0);
            v[4 * hf] = w.x; v[4 * hf + 1] = w.y; v[4 * hf + 2] = w.z; v[4 * hf + 3] = w.w;
          }
        }
      } else {
        const int k = 8 * k8;
        if (k < K) {                                       // (K % 8 == 0)
          const f32x4 w0 = ld4(Wt + (long)row * Kp + k), w1 = ld4(Wt + (long)row * Kp + k + 4);
          v = (f32x8){w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        }
      }
    }
    const AhP8<NP> pc = ah_split8<NP>(v);
#pragma unroll
    for (int i = 0; i < NP; ++i) reinterpret_cast<bf16x8*>(img + (size_t)i * rows_pad * WS)[e] = pc.p[i];
  }
}

// NKC = 32-wide chunks of Dk, NQ = 16-feature tiles of Q, NZ = 16-feature tiles of A0, NH = 16-feature tiles of qh (0: none),
// NP = pieces per operand.  512 threads: eight waves share one copy of the weight images (one history each).
template <int NKC, int NQ, int NZ, int NH, int NP>
__global__ void __launch_bounds__(512, 1) att_hist_fwd_x3_kernel(AttHistFwdArgs s) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int QP = 16 * NQ, ZP = 16 * NZ;
  constexpr int WSA = 32 * NKC + 8;                     // attention_mat image: natural slot order
  constexpr int NQC = (NQ + 1) / 2, WSU = 32 * NQC + 8; // Wu image: permuted slots over the a features
  constexpr int NHC = (NH + 1) / 2, WSP = 32 * (NHC > 0 ? NHC : 1) + 8;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  __bf16* Ai = reinterpret_cast<__bf16*>(lds_raw);      // [NP][QP][WSA]
  __bf16* Ui = Ai + NP * QP * WSA;                      // [NP][ZP][WSU]
  __bf16* Pi = Ui + NP * ZP * WSU;                      // [NP][ZP][WSP]
  ah_stage<false, NP>(Ai, WSA, QP, s.At, s.Kpa, s.Q, s.Dk, tid, 512);
  ah_stage<true, NP>(Ui, WSU, ZP, s.Wut, s.Kpu, s.A0, s.Q, tid, 512);
  if (NH) ah_stage<true, NP>(Pi, WSP, ZP, s.Wpt, s.Kpp, s.A0, s.nqh, tid, 512);
  __syncthreads();

  const int T = s.T, NTT = (T + 15) >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x4 zh4 = {};
  // identity block of the transposing products: k slot (g4, e), e < 4 = position 4 g4 + e of the tile; column j selects it
  bf16x8 sel;
#pragma unroll
  for (int e = 0; e < 8; ++e) sel[e] = (e < 4 && 4 * g4 + e == j) ? (__bf16)1.0f : (__bf16)0.0f;
  constexpr unsigned SKIP = 0x40000000u;
  unsigned kofs[NKC], ao[NQ], uo[NZ];
#pragma unroll
  for (int c = 0; c < NKC; ++c) kofs[c] = 32 * c + 8 * g4 < s.Dk ? 32 * c + 8 * g4 : 0;
#pragma unroll
  for (int n = 0; n < NQ; ++n) ao[n] = 16 * n + j < s.Q ? (16 * n + j) * 4u : SKIP;
#pragma unroll
  for (int z = 0; z < NZ; ++z) uo[z] = 16 * z + j < s.A0 ? (16 * z + j) * 4u : SKIP;
  const int arow = j * WSA + 8 * g4, urow = j * WSU + 8 * g4, prow = j * WSP + 8 * g4;

  for (long h = (long)blockIdx.x * 8 + wave; h < s.Hn; h += (long)gridDim.x * 8) {
    const float* kp = s.keys + h * T * s.ldk;
    const ah_rsrc_t ra = ah_rsrc(s.a + h * T * s.lda, (unsigned)(T * s.lda) * 4u);
    const ah_rsrc_t ru = ah_rsrc(s.U + h * T * s.ldu, (unsigned)(T * s.ldu) * 4u);
    // history-level query in the position-lane layout of the transposed a: features 16n + 4 g4 + {0..3}
    f32x4 qv[NH ? NH : 1];
    if (NH) {
#pragma unroll
      for (int n = 0; n < NH; ++n) {
        const int f0 = 16 * n + 4 * g4;
        qv[n] = f0 < s.nqh ? ld4(s.qh + h * s.ldqh + f0) : z4;
      }
    }
    f32x8 kraw[NKC];
#pragma unroll
    for (int c = 0; c < NKC; ++c) kraw[c] = ld8f(kp + (long)min(j, T - 1) * s.ldk + kofs[c]);
    for (int tt = 0; tt < NTT; ++tt) {
      const int t0 = 16 * tt;
      // A operand of the first product: the keys of the lane's position, 8 features per chunk; next tile's loads in flight
      AhP8<NP> kx[NKC];
#pragma unroll
      for (int c = 0; c < NKC; ++c) kx[c] = ah_split8<NP>((t0 + j < T && 32 * c + 8 * g4 < s.Dk) ? kraw[c] : z8);
      {
        const long tn = min(t0 + 16 + j, T - 1);
#pragma unroll
        for (int c = 0; c < NKC; ++c) kraw[c] = ld8f(kp + tn * s.ldk + kofs[c]);
      }
      // a^T tiles: 4 positions of a feature 16n + j
      f32x4 aa[NQ];
#pragma unroll
      for (int n = 0; n < NQ; ++n) aa[n] = z4;
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        AH_FENCE();
        ah_mac<NP, NQ>(aa, kx[c], Ai, QP * WSA, 16 * WSA, arow + 32 * c);
      }
      unsigned ro_a[4], ro_u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = t0 + 4 * g4 + e;
        ro_a[e] = t < T ? (unsigned)(t * s.lda) * 4u : SKIP;
        ro_u[e] = t < T ? (unsigned)(t * s.ldu) * 4u : SKIP;
      }
#pragma unroll
      for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) ah_st1(ra, ro_a[e] + ao[n], aa[n][e]);
      AH_FENCE();
      // back to rows = positions: the pieces of a as A operands (rows = a features, k = positions) times the identity --
      // exact per piece; th[i][n] = piece i of four consecutive a features of the lane's position
      bf16x4 th[NP][NQ], ph[NP][NH ? NH : 1];
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        const AhP4<NP> pc = ah_split4<NP>(aa[n]);
        f32x4 full = z4;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          f32x4 d = z4;
          HMFMA(d, ah_cat(pc.p[i], zh4), sel);
          th[i][n] = ah_h4(d);               // (exact: the values are bf16 numbers)
          full += d;
        }
        if (NH && n < NH) {                  // (a * q_hist) for the product term, split again
          const AhP4<NP> pp = ah_split4<NP>(full * qv[n < NH ? n : 0]);
#pragma unroll
          for (int i = 0; i < NP; ++i) ph[i][n < NH ? n : 0] = pp.p[i];
        }
      }
      // U^T tiles: 4 positions of A0 feature 16z + j
      f32x4 uu[NZ];
#pragma unroll
      for (int z = 0; z < NZ; ++z) uu[z] = z4;
#pragma unroll
      for (int c = 0; c < NQC; ++c) {
        AH_FENCE();
        AhP8<NP> x;
#pragma unroll
        for (int i = 0; i < NP; ++i) x.p[i] = ah_cat(th[i][2 * c], 2 * c + 1 < NQ ? th[i][2 * c + 1 < NQ ? 2 * c + 1 : 0] : zh4);
        ah_mac<NP, NZ>(uu, x, Ui, ZP * WSU, 16 * WSU, urow + 32 * c);
      }
      if (NH) {
#pragma unroll
        for (int c = 0; c < NHC; ++c) {
          AH_FENCE();
          AhP8<NP> x;
#pragma unroll
          for (int i = 0; i < NP; ++i)
            x.p[i] = ah_cat(ph[i][2 * c < NH ? 2 * c : 0], 2 * c + 1 < NH ? ph[i][2 * c + 1 < NH ? 2 * c + 1 : 0] : zh4);
          ah_mac<NP, NZ>(uu, x, Pi, ZP * WSP, 16 * WSP, prow + 32 * c);
        }
      }
#pragma unroll
      for (int z = 0; z < NZ; ++z)
#pragma unroll
        for (int e = 0; e < 4; ++e) ah_st1(ru, ro_u[e] + uo[z], uu[z][e]);
    }
  }
}

static int ah_class(int n) { return n <= 48 ? 3 : 5; }

extern "C" int clsr_att_hist_fwd_x3_supported(int Dk, int Q, int A0, int qh) {
  return Dk >= 8 && Dk <= 64 && Dk % 8 == 0 && Q >= 4 && Q <= 80 && Q % 4 == 0 && A0 >= 4 && A0 <= 80 && A0 % 4 == 0 &&
         qh >= 0 && qh <= 48 && qh <= Q && qh % 4 == 0;
}

template <int NKC, int NQ, int NZ, int NH, int NP>
static int ah_launch(const AttHistFwdArgs& a, hipStream_t stream) {
  constexpr int QP = 16 * NQ, ZP = 16 * NZ, WSA = 32 * NKC + 8, WSU = 32 * ((NQ + 1) / 2) + 8;
  constexpr int NHC = (NH + 1) / 2, WSP = 32 * (NHC > 0 ? NHC : 1) + 8;
  const size_t shmem = (size_t)NP * 2 * (QP * WSA + ZP * WSU + (NH ? ZP * WSP : 0));
  long gx = (a.Hn + 7) / 8;
  if (gx > 256) gx = 256;
  auto kernel = att_hist_fwd_x3_kernel<NKC, NQ, NZ, NH, NP>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3((unsigned)gx), dim3(512), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// pieces = 2: split-bf16 products of 2^-16 relative accuracy ("x3"); pieces = 3: 2^-23 ("x6", the parity mode's forward)
extern "C" int clsr_att_hist_fwd_x3(const float* keys, int ldk, const float* At, int Kpa, const float* Wut, int Kpu,
                                    const float* Wpt, int Kpp, const float* q_hist, int ldqh, long Hn, int T, int Dk, int Q,
                                    int A0, int qh, int pieces, float* a, int lda, float* U, int ldu, void* stream) {
  CLSR_CHECK_ARG(keys && At && Wut && a && U && Hn > 0 && T > 0 && (pieces == 2 || pieces == 3));
  CLSR_CHECK_SUPPORTED(clsr_att_hist_fwd_x3_supported(Dk, Q, A0, qh));
  CLSR_CHECK_ARG(qh == 0 || (Wpt && q_hist && ldqh >= qh && Kpp >= 16 * clsr_cdiv(qh, 16)));
  CLSR_CHECK_ARG(ldk >= Dk && lda >= Q && ldu >= A0 && Kpa >= 16 * clsr_cdiv(Dk, 16) && Kpu >= 16 * clsr_cdiv(Q, 16));
  CLSR_CHECK_SUPPORTED(ldk % 4 == 0 && Kpa % 4 == 0 && Kpu % 4 == 0 && (qh == 0 || (Kpp % 4 == 0 && ldqh % 4 == 0)) &&
                       ((uintptr_t)keys % 16) == 0 && ((uintptr_t)At % 16) == 0 && ((uintptr_t)Wut % 16) == 0 &&
                       (qh == 0 || (((uintptr_t)Wpt % 16) == 0 && ((uintptr_t)q_hist % 16) == 0)));
  AttHistFwdArgs s = {};
  s.keys = keys; s.ldk = ldk; s.At = At; s.Kpa = Kpa; s.Wut = Wut; s.Kpu = Kpu; s.Wpt = qh ? Wpt : nullptr; s.Kpp = Kpp;
  s.qh = q_hist; s.ldqh = ldqh; s.a = a; s.lda = lda; s.U = U; s.ldu = ldu; s.Hn = Hn; s.T = T; s.Dk = Dk; s.Q = Q;
  s.A0 = A0; s.nqh = qh;
  hipStream_t st = (hipStream_t)stream;
  const int nkc = Dk <= 32 ? 1 : 2, nq = ah_class(Q), nz = ah_class(A0), nh = qh == 0 ? 0 : 3;
#define AH_GO(K, Qn, Z, H) \
  if (nkc == K && nq == Qn && nz == Z && nh == H) \
    return pieces == 2 ? ah_launch<K, Qn, Z, H, 2>(s, st) : ah_launch<K, Qn, Z, H, 3>(s, st)
  AH_GO(1, 3, 3, 0); AH_GO(1, 3, 5, 0); AH_GO(1, 5, 3, 0); AH_GO(1, 5, 5, 0);
  AH_GO(2, 3, 3, 0); AH_GO(2, 3, 5, 0); AH_GO(2, 5, 3, 0); AH_GO(2, 5, 5, 0);
  AH_GO(1, 3, 3, 3); AH_GO(1, 3, 5, 3); AH_GO(1, 5, 3, 3); AH_GO(1, 5, 5, 3);
  AH_GO(2, 3, 3, 3); AH_GO(2, 3, 5, 3); AH_GO(2, 5, 3, 3); AH_GO(2, 5, 5, 3);
#undef AH_GO
  clsr_set_error("%s:%d: no instance for Dk = %d, Q = %d, A0 = %d, qh = %d", __FILE__, __LINE__, Dk, Q, A0, qh);
  return CLSR_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ backward
// History-level tail of the attention backward in ONE launch (reference: tf.gradients through clsr.py:351-370):
//   da[h,t,:]   = (columns >= qh: da as the layer-0 kernel left it | columns < qh: (dU . Wp[:qh]^T) * q_hist[h,:]) + dU . Wu^T
//   dq_hist[h,:] += sum_t (dU[h,t,:] . Wp[:qh]^T) * a[h,t,:qh]
//   dkeys[h,t,:] += da[h,t,:] . A^T
// from one pass over dU: it replaces clsr_pgemm (daq1) + clsr_att_prod_bwd_ld + clsr_pgemm (dU . Wu^T, accumulate) +
// clsr_pgemm (da . A^T, accumulate) on the dependent chain between the layer-0 backward and the backward-through-time
// launch.  One wave per history, 16-step tiles, same operand conventions as the forward kernel above: the two products
// over dU leave feature-lane tiles (4 positions of one query feature per lane), in which da is read, finished and
// stored; for the last product it is brought back to rows = positions by products with the identity.
struct AttHistBwdArgs {
  const float* dU; int lddu;       // [Hn*T, A0]
  const float* WuT; int Kpu;       // packed Wu^T: row q = query feature (Q rows), K = A0
  const float* WpT; int Kpp;       // packed Wp[:qh]^T: qh rows, K = A0 (NULL: none)
  const float* AT; int Kpa;        // packed attention_mat^T: row d = key feature (Dk rows), K = Q
  const float* a; int lda;         // [Hn*T, Q]
  const float* qh; int ldqh;       // [Hn, qh]
  float* da; int ldda;             // [Hn*T, Q]
  float* dqh; int lddqh;           // [Hn, qh]   (+=)
  float* dkeys; int lddk;          // [Hn*T, Dk] (+=)
  long Hn;
  int T, Dk, Q, A0, nqh;
};

__device__ __forceinline__ float ah_ld1(ah_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

// NZC = 32-wide chunks of A0, NQ = tiles of Q, NH = tiles of qh (0: none), NK = tiles of Dk, NP = bf16 pieces per operand
// (2: the "fp32x3" / speed modes; 3: fp32 accuracy, precision="fp32").  NTH threads: with three pieces the weight images
// (110 KB at the default widths) leave room for ONE workgroup per CU -- eight waves share them instead of four (one wave
// per SIMD took 277 us in the step where the two-piece form takes 130: the walk is latency bound, it needs the waves)
template <int NZC, int NQ, int NH, int NK, int NP, int NTH>
__global__ void __launch_bounds__(NTH, NTH == 512 ? 1 : 2) att_hist_bwd_x3_kernel(AttHistBwdArgs s) {
  CLSR_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int QP = 16 * NQ, HP = 16 * (NH ? NH : 1), KP = 16 * NK;
  constexpr int WSZ = 32 * NZC + 8;                     // images over the A0 features: natural slot order
  constexpr int NQC = (NQ + 1) / 2, WSQ = 32 * NQC + 8; // attention_mat^T image: permuted slots over the query features
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g4 = lane >> 4;
  __bf16* Ui = reinterpret_cast<__bf16*>(lds_raw);        // [NP][QP][WSZ]
  __bf16* Pi = Ui + NP * QP * WSZ;                        // [NP][HP][WSZ]
  __bf16* Ai = Pi + (NH ? NP * HP * WSZ : 0);             // [NP][KP][WSQ]
  ah_stage<false, NP>(Ui, WSZ, QP, s.WuT, s.Kpu, s.Q, s.A0, tid, NTH);
  if (NH) ah_stage<false, NP>(Pi, WSZ, HP, s.WpT, s.Kpp, s.nqh, s.A0, tid, NTH);
  ah_stage<true, NP>(Ai, WSQ, KP, s.AT, s.Kpa, s.Dk, s.Q, tid, NTH);
  __syncthreads();

  const int T = s.T, NTT = (T + 15) >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x4 zh4 = {};
  bf16x8 sel;
#pragma unroll
  for (int e = 0; e < 8; ++e) sel[e] = (e < 4 && 4 * g4 + e == j) ? (__bf16)1.0f : (__bf16)0.0f;
  constexpr unsigned SKIP = 0x40000000u;
  unsigned zofs[NZC], qo[NQ], ko[NK];
  bool qold[NQ];                    // column of da that keeps what the layer-0 kernel wrote (>= qh)
#pragma unroll
  for (int c = 0; c < NZC; ++c) zofs[c] = 32 * c + 8 * g4 < s.A0 ? 32 * c + 8 * g4 : 0;
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    qo[n] = 16 * n + j < s.Q ? (16 * n + j) * 4u : SKIP;
    qold[n] = 16 * n + j >= s.nqh;
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) ko[k] = 16 * k + j < s.Dk ? (16 * k + j) * 4u : SKIP;
  const int urow = j * WSZ + 8 * g4, arow = j * WSQ + 8 * g4;

  for (long h = (long)blockIdx.x * (NTH / 64) + wave; h < s.Hn; h += (long)gridDim.x * (NTH / 64)) {
    const float* up = s.dU + h * T * s.lddu;
    const ah_rsrc_t ra = ah_rsrc(s.a + h * T * s.lda, (unsigned)(T * s.lda) * 4u);
    const ah_rsrc_t rda = ah_rsrc(s.da + h * T * s.ldda, (unsigned)(T * s.ldda) * 4u);
    const ah_rsrc_t rdk = ah_rsrc(s.dkeys + h * T * s.lddk, (unsigned)(T * s.lddk) * 4u);
    float qv[NH ? NH : 1], dqacc[NH ? NH : 1];      // q_hist / its gradient for the lane's feature 16n + j
    if (NH) {
#pragma unroll
      for (int n = 0; n < NH; ++n) {
        qv[n] = 16 * n + j < s.nqh ? s.qh[h * s.ldqh + 16 * n + j] : 0.f;
        dqacc[n] = 0.f;
      }
    }
    f32x8 uraw[NZC];
#pragma unroll
    for (int c = 0; c < NZC; ++c) uraw[c] = ld8f(up + (long)min(j, T - 1) * s.lddu + zofs[c]);
    for (int tt = 0; tt < NTT; ++tt) {
      const int t0 = 16 * tt;
      AhP8<NP> ux[NZC];
#pragma unroll
      for (int c = 0; c < NZC; ++c) ux[c] = ah_split8<NP>((t0 + j < T && 32 * c + 8 * g4 < s.A0) ? uraw[c] : z8);
      {
        const long tn = min(t0 + 16 + j, T - 1);
#pragma unroll
        for (int c = 0; c < NZC; ++c) uraw[c] = ld8f(up + tn * s.lddu + zofs[c]);
      }
      // feature-lane operands of this tile: da as it stands, a[:, :qh]
      unsigned ro_a[4], ro_da[4], ro_dk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = t0 + 4 * g4 + e;
        ro_a[e] = t < T ? (unsigned)(t * s.lda) * 4u : SKIP;
        ro_da[e] = t < T ? (unsigned)(t * s.ldda) * 4u : SKIP;
        ro_dk[e] = t < T ? (unsigned)(t * s.lddk) * 4u : SKIP;
      }
      f32x4 dold[NQ], af[NH ? NH : 1], dkold[NK];
#pragma unroll
      for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) dold[n][e] = ah_ld1(rda, (qold[n] ? ro_da[e] : SKIP) + qo[n]);     // (skipped: reads 0)
      if (NH) {
#pragma unroll
        for (int n = 0; n < NH; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) af[n][e] = ah_ld1(ra, ro_a[e] + qo[n < NQ ? n : 0]);
      }
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) dkold[k][e] = ah_ld1(rdk, ro_dk[e] + ko[k]);
      // dU . Wu^T and dU . Wp[:qh]^T: 4 positions of query feature 16n + j
      f32x4 r1[NQ], r2[NH ? NH : 1];
#pragma unroll
      for (int n = 0; n < NQ; ++n) r1[n] = z4;
#pragma unroll
      for (int n = 0; n < (NH ? NH : 1); ++n) r2[n] = z4;
#pragma unroll
      for (int c = 0; c < NZC; ++c) {
        AH_FENCE();
        ah_mac<NP, NQ>(r1, ux[c], Ui, QP * WSZ, 16 * WSZ, urow + 32 * c);
        if constexpr (NH > 0) ah_mac<NP, NH>(r2, ux[c], Pi, HP * WSZ, 16 * WSZ, urow + 32 * c);
      }
      AH_FENCE();
      // finish da, store it, split it, and bring it back to rows = positions for the last product
      bf16x4 th[NP][NQ];
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        f32x4 d = dold[n] + r1[n];
        if (NH && n < NH) {
          const f32x4 rr = r2[n < NH ? n : 0];
          d += rr * qv[n < NH ? n : 0];                     // (zero for columns >= qh: q_hist is 0 there)
          const f32x4 pa = rr * af[n < NH ? n : 0];
          dqacc[n < NH ? n : 0] += (pa.x + pa.y) + (pa.z + pa.w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ah_st1(rda, ro_da[e] + qo[n], d[e]);
        const AhP4<NP> pc = ah_split4<NP>(d);
#pragma unroll
        for (int i = 0; i < NP; ++i) {                      // (products with the identity: exact for each bf16 piece)
          f32x4 dt = z4;
          HMFMA(dt, ah_cat(pc.p[i], zh4), sel);
          th[i][n] = ah_h4(dt);
        }
      }
      // dkeys += da . A^T: 4 positions of key feature 16k + j
      f32x4 dk[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) dk[k] = dkold[k];
#pragma unroll
      for (int c = 0; c < NQC; ++c) {
        AH_FENCE();
        AhP8<NP> x;
#pragma unroll
        for (int i = 0; i < NP; ++i) x.p[i] = ah_cat(th[i][2 * c], 2 * c + 1 < NQ ? th[i][2 * c + 1 < NQ ? 2 * c + 1 : 0] : zh4);
        ah_mac<NP, NK>(dk, x, Ai, KP * WSQ, 16 * WSQ, arow + 32 * c);
      }
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) ah_st1(rdk, ro_dk[e] + ko[k], dk[k][e]);
    }
    if (NH) {
#pragma unroll
      for (int n = 0; n < NH; ++n) {
        const float v = col4_sum(dqacc[n]);
        if (g4 == 0 && 16 * n + j < s.nqh) s.dqh[h * s.lddqh + 16 * n + j] += v;
      }
    }
  }
}

extern "C" int clsr_att_hist_bwd_x3_supported(int Dk, int Q, int A0, int qh) {
  return Dk >= 4 && Dk <= 48 && Dk % 4 == 0 && Q >= 4 && Q <= 80 && Q % 4 == 0 && A0 >= 8 && A0 <= 96 && A0 % 8 == 0 &&
         qh >= 0 && qh <= 48 && qh <= Q && qh % 4 == 0;
}

template <int NZC, int NQ, int NH, int NK, int NP>
static int ahb_launch(const AttHistBwdArgs& a, hipStream_t stream) {
  constexpr int NTH = NP > 2 ? 512 : 256;
  constexpr int QP = 16 * NQ, HP = 16 * (NH ? NH : 1), KP = 16 * NK, WSZ = 32 * NZC + 8, WSQ = 32 * ((NQ + 1) / 2) + 8;
  const size_t shmem = (size_t)NP * 2 * (QP * WSZ + (NH ? HP * WSZ : 0) + KP * WSQ);
  long gx = (a.Hn + NTH / 64 - 1) / (NTH / 64);
  if (gx > (NTH == 512 ? 256 : 512)) gx = NTH == 512 ? 256 : 512;
  auto kernel = att_hist_bwd_x3_kernel<NZC, NQ, NH, NK, NP, NTH>;
  if (shmem > 64 * 1024)
    CLSR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3((unsigned)gx), dim3(NTH), shmem, stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" int clsr_att_hist_bwd_x3(const float* dU, int lddu, const float* WuT, int Kpu, const float* WpT, int Kpp,
                                    const float* AT, int Kpa, const float* a, int lda, const float* q_hist, int ldqh,
                                    long Hn, int T, int Dk, int Q, int A0, int qh, int pieces, float* da, int ldda,
                                    float* dq_hist, int lddqh, float* dkeys, int lddk, void* stream) {
  CLSR_CHECK_ARG(dU && WuT && AT && da && dkeys && Hn > 0 && T > 0 && (pieces == 2 || pieces == 3));
  CLSR_CHECK_SUPPORTED(clsr_att_hist_bwd_x3_supported(Dk, Q, A0, qh));
  CLSR_CHECK_ARG(qh == 0 || (WpT && a && q_hist && dq_hist && ldqh >= qh && lddqh >= qh && lda >= qh &&
                             Kpp >= 16 * clsr_cdiv(A0, 16)));
  CLSR_CHECK_ARG(lddu >= A0 && ldda >= Q && lddk >= Dk && Kpu >= 16 * clsr_cdiv(A0, 16) && Kpa >= 16 * clsr_cdiv(Q, 16));
  CLSR_CHECK_SUPPORTED(lddu % 4 == 0 && Kpu % 4 == 0 && Kpa % 4 == 0 && (qh == 0 || Kpp % 4 == 0) &&
                       ((uintptr_t)dU % 16) == 0 && ((uintptr_t)WuT % 16) == 0 && ((uintptr_t)AT % 16) == 0 &&
                       (qh == 0 || ((uintptr_t)WpT % 16) == 0));
  AttHistBwdArgs s = {};
  s.dU = dU; s.lddu = lddu; s.WuT = WuT; s.Kpu = Kpu; s.WpT = qh ? WpT : nullptr; s.Kpp = Kpp; s.AT = AT; s.Kpa = Kpa;
  s.a = a; s.lda = lda; s.qh = q_hist; s.ldqh = ldqh; s.da = da; s.ldda = ldda; s.dqh = dq_hist; s.lddqh = lddqh;
  s.dkeys = dkeys; s.lddk = lddk; s.Hn = Hn; s.T = T; s.Dk = Dk; s.Q = Q; s.A0 = A0; s.nqh = qh;
  hipStream_t st = (hipStream_t)stream;
  const int nzc = clsr_cdiv(A0, 32), nq = ah_class(Q), nh = qh == 0 ? 0 : 3, nk = 3;
#define AHB_GO(Z, Qn, H) \
  if (nzc == Z && nq == Qn && nh == H) return pieces == 3 ? ahb_launch<Z, Qn, H, 3, 3>(s, st) : ahb_launch<Z, Qn, H, 3, 2>(s, st)
  AHB_GO(1, 3, 0); AHB_GO(2, 3, 0); AHB_GO(3, 3, 0); AHB_GO(1, 5, 0); AHB_GO(2, 5, 0); AHB_GO(3, 5, 0);
  AHB_GO(1, 3, 3); AHB_GO(2, 3, 3); AHB_GO(3, 3, 3); AHB_GO(1, 5, 3); AHB_GO(2, 5, 3); AHB_GO(3, 5, 3);
#undef AHB_GO
  (void)nk;
  clsr_set_error("%s:%d: no instance for Q = %d, A0 = %d, qh = %d", __FILE__, __LINE__, Q, A0, qh);
  return CLSR_EUNSUPPORTED;
}
