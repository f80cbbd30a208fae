// Recurrent encoders of CLSR, ONE WAVE per encoder and 16 histories (gfx950): GRU and Time4LSTM, forward and
// backward-through-time.  Same arithmetic and the same argument blocks as csrc/rnn.hip (reference:
// tf.nn.rnn_cell.GRUCell under dynamic_rnn, clsr.py:160-168,201-208,229-237; Time4LSTMCell.call,
// rnn_cell_implement.py:129-298, at clsr.py:179-200).
//
// Why a second kernel.  csrc/rnn.hip splits the hidden features of one encoder over RNT = 3 waves: every time step then
// needs one or two LDS exchanges + workgroup barriers, and the waves of the 2-3 encoders of a step share SIMDs -- their
// dependent MFMA chains interleave on one matrix pipe.  Measured at configs[1]: 1.5 us per step for one GRU alone, 4.1 us
// per step for the two main encoders side by side, 205 + 289 us for the forward recurrences of a step.  Here a wave owns
// ALL feature tiles of its encoder: the MFMA output layout (lane (j, g): features 4g..4g+3 of a tile, history j) IS the
// B-operand layout of the next matvec, so the state never leaves the wave's registers -- no LDS, no barrier -- and the
// workgroup is the 2-4 encoders of the launch over the same 16 histories, one wave each, i.e. one SIMD (one matrix pipe)
// per encoder.
//
// MEASURED OUTCOME (round 3, profiles/r03_rnn_pmc.md): correct (tests/test_kernels_gpu.py), but SLOWER than the split
// kernels -- Time4LSTM alone 229 us against 122 us, the three forward encoders of a step 288 against 225 us -- and
// therefore OPT-IN (CLSR_RNN1=1).  The assumption behind it was that a wave's gate arithmetic (~560 VALU instructions per
// Time4LSTM step) hides under its own 120 MFMAs.  On gfx950 it cannot: v_mfma_f32_16x16x4_f32 (fp32 inputs) runs at the
// fp32 VECTOR rate and does not overlap VALU work of the same wave (scripts/mfma_valu_overlap.hip: one MFMA 14.2 ns, one
// MFMA + 4 v_fma 24.9 ns, + 6 v_fma 31.3 ns; the bf16 MFMA hides them).  SQ counters of the kernel agree: 3 840 cycles
// of matrix-pipe time + 3 900 cycles of VALU issue + 1 800 cycles of s_waitcnt per wave-step, added up, not overlapped.
// One wave per encoder therefore serialises three waves' worth of MFMA + VALU; the split kernels do a third of both per
// wave.  Kept: as the record of that measurement, and for the two defects of the tool chain found on the way (below:
// __builtin_bit_cast on a vector element, the store-data hazard behind buffer stores with an SGPR offset).
//
// Last tile.  With n = 40 the third feature tile holds 8 features.  The k-slot -> feature map of an MFMA is free as long
// as A and B agree, and so is the row -> feature map of the output tile (a permutation of the weight rows): the last
// tile assigns output row 4g + r (r < 2) to feature 32 + g + 4r, so that a lane's components 0 / 1 are exactly the
// operands of the TWO full MFMAs that reduce over those 8 features (10 instead of 12 MFMAs per gate and tile) -- the
// compact form csrc/rnn.hip reaches through its exchange buffer, here without moving anything.
#include <cstdlib>
#include <algorithm>
#include "common.h"
#include "clsr_hip.h"
#include "rnn_args.h"

#define R1Z4 ((f32x4){0.f, 0.f, 0.f, 0.f})
#ifdef R1_ABL_NOMFMA
#undef MFMA4
#define MFMA4(acc, a, b) (acc) += (a) * (b)
#endif
typedef __bf16 r1_bf16x4 __attribute__((ext_vector_type(4)));

// (R1_ABL_*: compile-time ablations for timing experiments -- scripts/build_variant.sh; never defined in the product build)
__device__ __forceinline__ f32x4 r1_sig4(f32x4 v) {
#ifdef R1_ABL_NOACT
  return v * 0.25f + 0.5f;
#else
  return (f32x4){sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
#endif
}
__device__ __forceinline__ f32x4 r1_tanh4(f32x4 v) {
#ifdef R1_ABL_NOACT
  return v * 0.5f;
#else
  return (f32x4){tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)};
#endif
}
__device__ __forceinline__ f32x4 r1_sel(bool c, f32x4 a, f32x4 b) { return c ? a : b; }
__device__ __forceinline__ int r1_wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// feature held by component r of lane group g in tile tl (-1: none); CMP: the last tile is compact (<= 8 features)
template <int NT, bool CMP>
__device__ __forceinline__ int r1_feat(int tl, int g, int r) {
  if (CMP && tl == NT - 1) return r < 2 ? 16 * tl + g + 4 * r : -1;
  return 16 * tl + 4 * g + r;
}

// Every global access of the time loop is a RAW BUFFER operation: a lane that must not load / store (history beyond
// Hn, feature beyond n, dead time step, tensor not given) gets an out-of-range offset -- the hardware returns zeros /
// drops the store.  No branch anywhere in the loop body: with a conditional load or store in it the compiler cannot count
// the outstanding memory operations and drains the whole queue (s_waitcnt vmcnt(0)) every step -- the first version of
// this kernel, written with guarded plain accesses, spent 8 us per step waiting for its own stores to reach HBM.
// Range: num_records = 2 GB for a tensor that is given, 0 for one that is not (every access of it is out of range); the
// out-of-range offset is 2 GB, so that adding an instruction offset to it can neither wrap nor come back into range.
// The range check covers the VGPR offset (+ instruction offset) only: the SGPR offset carries the per-step part.
#define R1_OOB 0x80000000u
typedef __amdgpu_buffer_rsrc_t r1_rsrc_t;
__device__ __forceinline__ r1_rsrc_t r1_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? 0x80000000u : 0u, 0x00020000);
}

// Per-lane byte offsets of the lane's tile pieces inside its history's block of a row-major tensor, computed ONCE: the
// loop adds the uniform per-step part as the SGPR offset of the instruction -- no per-access address arithmetic.
// a[tl]: the 16-byte piece of tile tl (first element of the compact last tile), b: second element of the compact tile.
template <int NT>
struct R1Off { unsigned a[NT]; unsigned b; };
template <int NT, bool CMP>
__device__ __forceinline__ R1Off<NT> r1_off(long row_elems, int g, int n, bool hv, int esize) {
  R1Off<NT> o;
  o.b = R1_OOB;
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    if (CMP && tl == NT - 1) {
      const int f0 = 16 * tl + g, f1 = f0 + 4;
      o.a[tl] = (hv && f0 < n) ? (unsigned)((row_elems + f0) * esize) : R1_OOB;
      o.b = (hv && f1 < n) ? (unsigned)((row_elems + f1) * esize) : R1_OOB;
    } else {
      const int f = 16 * tl + 4 * g;
      o.a[tl] = (hv && f < n) ? (unsigned)((row_elems + f) * esize) : R1_OOB;
    }
  }
  return o;
}
template <int NT, bool CMP>
__device__ __forceinline__ f32x4 r1_ldo(r1_rsrc_t rs, const R1Off<NT>& o, int tl, unsigned soff) {
  if (CMP && tl == NT - 1) {
    const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o.a[tl], soff, 0));
    const float y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o.b, soff, 0));
    return (f32x4){x, y, 0.f, 0.f};
  }
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o.a[tl], soff, 0));
}
template <int NT, bool CMP>
__device__ __forceinline__ void r1_sto(r1_rsrc_t rs, const R1Off<NT>& o, int tl, unsigned soff, f32x4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef R1_ABL_NOST
  if (v.x != 123.456f) return;
#endif
  if (CMP && tl == NT - 1) {
    const float vx = v.x, vy = v.y;      // (scalar copies: see r1_st)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rs, o.a[tl], soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), rs, o.b, soff, 0);
    return;
  }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, o.a[tl], soff, 0);
  // Store-data hazard: a VALU instruction must not overwrite the data registers of a 16-byte store in the wait state
  // right behind it.  The compiler inserts that wait state for plain / global stores, but takes a buffer store WITH AN
  // SGPR OFFSET for safe -- on gfx950 it is not: `buffer_store_dwordx4 v[12:15] ... s46 offen` followed directly by
  // `v_pk_fma_f32 v[14:15], ...` stored the new v15 for lanes 60..63 (found as one wrong gate value in four histories
  // of every workgroup).  The empty statement below READS v: whatever overwrites its registers is ordered behind it.
  asm volatile("s_nop 0" ::"v"(v));
}
// gradient-of-input-projection stores through precomputed offsets (o built with esize 4, or 2 for a bf16 tensor)
template <int NT, bool CMP>
__device__ __forceinline__ void r1_sto_dpin(r1_rsrc_t rs, const R1Off<NT>& o, int tl, unsigned soff, f32x4 v, int hbf) {
  if (!hbf) { r1_sto<NT, CMP>(rs, o, tl, soff, v); return; }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if (CMP && tl == NT - 1) {
    const __bf16 bx = (__bf16)v.x, by = (__bf16)v.y;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, bx), rs, o.a[tl], soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, by), rs, o.b, soff, 0);
    return;
  }
  const r1_bf16x4 hv4 = __builtin_convertvector(v, r1_bf16x4);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv4), rs, o.a[tl], soff, 0);
}

// the lane's components of tile tl of the row that starts `row` BYTES into the tensor (ok == false: zeros)
template <int NT, bool CMP>
__device__ __forceinline__ f32x4 r1_ld(r1_rsrc_t rs, unsigned row, int tl, int g, int n, bool ok) {
  if (CMP && tl == NT - 1) {
    const int f0 = 16 * tl + g, f1 = f0 + 4;
    const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && f0 < n) ? row + 4u * f0 : R1_OOB, 0, 0));
    const float y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && f1 < n) ? row + 4u * f1 : R1_OOB, 0, 0));
    return (f32x4){x, y, 0.f, 0.f};
  }
  const int f = 16 * tl + 4 * g;
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (ok && f < n) ? row + 4u * f : R1_OOB, 0, 0));
}
template <int NT, bool CMP>
__device__ __forceinline__ void r1_st(r1_rsrc_t rs, unsigned row, int tl, int g, int n, f32x4 v, bool ok) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef R1_ABL_NOST
  ok = ok && v.x == 123.456f;
#endif
  if (CMP && tl == NT - 1) {
    const int f0 = 16 * tl + g, f1 = f0 + 4;
    // (scalar copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this clang --
    //  both stores then carried v.x, found as wrong features 36..39 at hidden size 40)
    const float vx = v.x, vy = v.y;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rs, (ok && f0 < n) ? row + 4u * f0 : R1_OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), rs, (ok && f1 < n) ? row + 4u * f1 : R1_OOB, 0, 0);
    return;
  }
  const int f = 16 * tl + 4 * g;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (ok && f < n) ? row + 4u * f : R1_OOB, 0, 0);
}
// gradient-of-input-projection stores: fp32, or bf16 (speed mode).  col = ELEMENT offset of the block inside the row,
// rowe = element offset of the row (both in elements of the tensor's own type)
template <int NT, bool CMP>
__device__ __forceinline__ void r1_st_dpin(r1_rsrc_t rs, unsigned rowe, int tl, int g, int n, f32x4 v, int hbf, bool ok) {
  if (!hbf) { r1_st<NT, CMP>(rs, 4u * rowe, tl, g, n, v, ok); return; }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if (CMP && tl == NT - 1) {
    const int f0 = 16 * tl + g, f1 = f0 + 4;
    const __bf16 bx = (__bf16)v.x, by = (__bf16)v.y;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, bx), rs,
                                          (ok && f0 < n) ? 2u * (rowe + f0) : R1_OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, by), rs,
                                          (ok && f1 < n) ? 2u * (rowe + f1) : R1_OOB, 0, 0);
    return;
  }
  const int f = 16 * tl + 4 * g;
  const r1_bf16x4 hv4 = __builtin_convertvector(v, r1_bf16x4);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv4), rs, (ok && f < n) ? 2u * (rowe + f) : R1_OOB, 0, 0);
}

// MFMA A operand of output tile rt, k tile kt: lane (i, g), MFMA #c holds
//   forward  (tr = false): W[feat(kt, g, c)][colbase + feat_row(rt, i)]      out[o]  = sum_in W[in][o] state[in]
//   backward (tr = true):  W[feat_row(rt, i)][colbase + feat(kt, g, c)]      din[in] = sum_o  W[in][o] dgate[o]
// with feat_row(rt, i) = feat(rt, i >> 2, i & 3): output row 4g' + r of a tile is the feature that lane group g' holds
// as component r -- the output of one step is the B operand of the next without moving.
template <int NT, bool CMP>
__device__ __forceinline__ f32x4 r1_ld_w(r1_rsrc_t W, int ld, int colbase, int n, int rt, int kt, int i, int g, bool tr) {
  const int fo = r1_feat<NT, CMP>(rt, i >> 2, i & 3);
  f32x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int fk = r1_feat<NT, CMP>(kt, g, c);
    const bool ok = fo >= 0 && fo < n && fk >= 0 && fk < n;
    const unsigned idx = tr ? (unsigned)(fo * ld + colbase + fk) : (unsigned)(fk * ld + colbase + fo);
    v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(W, ok ? 4u * idx : R1_OOB, 0, 0));
  }
  return v;
}

// acc += W[kt] . b[kt] over one k tile (two MFMAs for the compact last tile)
template <int NT, bool CMP>
__device__ __forceinline__ void r1_mvt(f32x4& acc, const f32x4& w, const f32x4& b, int kt) {
  MFMA4(acc, w.x, b.x);
  MFMA4(acc, w.y, b.y);
  if (!(CMP && kt == NT - 1)) {
    MFMA4(acc, w.z, b.z);
    MFMA4(acc, w.w, b.w);
  }
}
// two independent chains side by side: a += Wa . b, c += Wc . d  (dependent MFMAs on one accumulator issue every 40
// cycles, independent ones every 32)
template <int NT, bool CMP>
__device__ __forceinline__ void r1_mv2(f32x4& a, const f32x4 (&wa)[NT], const f32x4 (&b)[NT], f32x4& c,
                                       const f32x4 (&wc)[NT], const f32x4 (&d)[NT]) {
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    MFMA4(a, wa[kt].x, b[kt].x); MFMA4(c, wc[kt].x, d[kt].x);
    MFMA4(a, wa[kt].y, b[kt].y); MFMA4(c, wc[kt].y, d[kt].y);
    if (!(CMP && kt == NT - 1)) {
      MFMA4(a, wa[kt].z, b[kt].z); MFMA4(c, wc[kt].z, d[kt].z);
      MFMA4(a, wa[kt].w, b[kt].w); MFMA4(c, wc[kt].w, d[kt].w);
    }
  }
}
// one product as two chains over alternating k tiles: a += W . b  (e: the second accumulator, added at the end)
template <int NT, bool CMP>
__device__ __forceinline__ void r1_mv1(f32x4& a, const f32x4 (&w)[NT], const f32x4 (&b)[NT]) {
  f32x4 e = R1Z4;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) r1_mvt<NT, CMP>((kt & 1) ? e : a, w[kt], b[kt], kt);
  if (NT > 1) a += e;
}

// ===================================================================================== GRU
// Dead steps (t >= the history's length, inside the wave's common range) are COMPUTED like live ones and their results
// stored: the state update is what the `live` select protects.  What lands in the saved tensors for such a step is
// finite and never used with a non-zero factor (the backward pass forms d = 0 for it first), out_seq is zero-filled
// behind the loop, and dPin comes out as exact zeros -- so no access of the loop depends on a per-lane predicate.
template <int NT, bool CMP>
__device__ __forceinline__ void gru1_fwd(const GruArgs& a, const int bx) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hv = h < a.Hn;
  const long hc = hv ? h : 0;
  const r1_rsrc_t bWg = r1_rsrc(a.Wgh), bWc = r1_rsrc(a.Wch), bPin = r1_rsrc(a.Pin), bh0 = r1_rsrc(a.h0);
  const r1_rsrc_t bhp = r1_rsrc(a.hprev), bga = r1_rsrc(a.gates), bos = r1_rsrc(a.out_seq), bhT = r1_rsrc(a.hT);
  f32x4 wr[NT][NT], wu[NT][NT], wc[NT][NT];      // [out tile][k tile]
#pragma unroll
  for (int ot = 0; ot < NT; ++ot)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      wr[ot][kt] = r1_ld_w<NT, CMP>(bWg, a.ldg, 0, n, ot, kt, j, g, false);
      wu[ot][kt] = r1_ld_w<NT, CMP>(bWg, a.ldg, n, n, ot, kt, j, g, false);
      wc[ot][kt] = r1_ld_w<NT, CMP>(bWc, a.ldc, 0, n, ot, kt, j, g, false);
    }
  f32x4 hs[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) hs[tl] = r1_ld<NT, CMP>(bh0, (unsigned)(hc * a.h0_stride * 4), tl, g, n, hv);
  const int len = hv ? min(a.seq_len[hc * a.len_stride], T) : 0;
  const int t0 = a.t0, tend = min(r1_wave_max(len), a.t1);
  const R1Off<NT> oP = r1_off<NT, CMP>(hc * (long)T * a.ldp, g, n, hv, 4);
  const R1Off<NT> oN = r1_off<NT, CMP>(hc * (long)T * n, g, n, hv, 4);
  const R1Off<NT> oG = r1_off<NT, CMP>(hc * (long)T * 3 * n, g, n, hv, 4);
  const unsigned ldp4 = 4u * a.ldp, n4 = 4u * n;
  f32x4 pn[3][NT];
#pragma unroll
  for (int gb = 0; gb < 3; ++gb)
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) pn[gb][tl] = r1_ldo<NT, CMP>(bPin, oP, tl, min(t0, T - 1) * ldp4 + gb * n4);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 accr[NT], accu[NT], accc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) { accr[tl] = pn[0][tl]; accu[tl] = pn[1][tl]; accc[tl] = pn[2][tl]; }
    {  // next step's input projections: in flight behind this step's MFMAs
      const unsigned tn = min(t + 1, T - 1) * ldp4;
#pragma unroll
      for (int gb = 0; gb < 3; ++gb)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) pn[gb][tl] = r1_ldo<NT, CMP>(bPin, oP, tl, tn + gb * n4);
    }
    f32x4 r[NT], u[NT], rh[NT], hn[NT], c[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {      // tile by tile: the gates of tile ot overlap the MFMAs of tile ot + 1
      r1_mv2<NT, CMP>(accr[ot], wr[ot], hs, accu[ot], wu[ot], hs);
      r[ot] = r1_sig4(accr[ot]);
      u[ot] = r1_sig4(accu[ot]);
      rh[ot] = r[ot] * hs[ot];
    }
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      r1_mv1<NT, CMP>(accc[ot], wc[ot], rh);
      c[ot] = r1_tanh4(accc[ot]);
      hn[ot] = u[ot] * hs[ot] + (1.0f - u[ot]) * c[ot];
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      r1_sto<NT, CMP>(bhp, oN, tl, t * n4, hs[tl]);
      r1_sto<NT, CMP>(bga, oG, tl, t * 3 * n4, r[tl]);
      r1_sto<NT, CMP>(bga, oG, tl, t * 3 * n4 + n4, u[tl]);
      r1_sto<NT, CMP>(bga, oG, tl, t * 3 * n4 + 2 * n4, c[tl]);
      r1_sto<NT, CMP>(bos, oN, tl, t * n4, hn[tl]);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) hs[tl] = r1_sel(live, hn[tl], hs[tl]);
  }
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    r1_st<NT, CMP>(bhT, (unsigned)(hc * n4), tl, g, n, hs[tl], hv);
    if (a.out_seq)
      for (int t = max(len, t0); t < a.t1; ++t) r1_st<NT, CMP>(bos, (unsigned)(hc * T + t) * n4, tl, g, n, R1Z4, hv);
  }
}

template <int NT, bool CMP>
__device__ __forceinline__ void gru1_bwd(const GruArgs& a, const int bx) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hv = h < a.Hn;
  const long hc = hv ? h : 0;
  const r1_rsrc_t bWg = r1_rsrc(a.Wgh), bWc = r1_rsrc(a.Wch), bhp = r1_rsrc(a.hprev), bga = r1_rsrc(a.gates);
  const r1_rsrc_t bdo = r1_rsrc(a.dout_seq), bdp = r1_rsrc(a.dPin), bdhT = r1_rsrc(a.dhT), bdh0 = r1_rsrc(a.dh0);
  f32x4 wr[NT][NT], wu[NT][NT], wc[NT][NT];      // transposed operands: [in tile][gate tile]
#pragma unroll
  for (int ot = 0; ot < NT; ++ot)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      wr[ot][kt] = r1_ld_w<NT, CMP>(bWg, a.ldg, 0, n, ot, kt, j, g, true);
      wu[ot][kt] = r1_ld_w<NT, CMP>(bWg, a.ldg, n, n, ot, kt, j, g, true);
      wc[ot][kt] = r1_ld_w<NT, CMP>(bWc, a.ldc, 0, n, ot, kt, j, g, true);
    }
  const unsigned n4 = 4u * n;
  f32x4 dh[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) dh[tl] = r1_ld<NT, CMP>(bdhT, (unsigned)(hc * n4), tl, g, n, hv);
  const int len = hv ? min(a.seq_len[hc * a.len_stride], T) : 0;
  const int t0 = a.t0, tlast = min(r1_wave_max(len), a.t1) - 1;
  const int hbf = a.dpin_bf16;
  const unsigned es = hbf ? 2u : 4u;
  for (int t = max(len, max(t0, tlast + 1)); t < a.t1; ++t) {      // zero dPin past the wave's common range
    const unsigned dp = (unsigned)((hc * T + t) * a.lddp);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
      for (int gb = 0; gb < 3; ++gb) r1_st_dpin<NT, CMP>(bdp, dp + gb * n, tl, g, n, R1Z4, hbf, hv);
  }
  const R1Off<NT> oN = r1_off<NT, CMP>(hc * (long)T * n, g, n, hv, 4);
  const R1Off<NT> oG = r1_off<NT, CMP>(hc * (long)T * 3 * n, g, n, hv, 4);
  const R1Off<NT> oD = r1_off<NT, CMP>(hc * (long)T * a.lddp, g, n, hv, (int)es);
  f32x4 cr[NT], cu[NT], cc[NT], chp[NT], cdo[NT];
  auto fetch = [&](int t) {     // saved activations of step t (finite for every lane of the wave: see gru1_fwd)
    const unsigned tc = (unsigned)(t > 0 ? t : 0);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      cr[tl] = r1_ldo<NT, CMP>(bga, oG, tl, tc * 3 * n4);
      cu[tl] = r1_ldo<NT, CMP>(bga, oG, tl, tc * 3 * n4 + n4);
      cc[tl] = r1_ldo<NT, CMP>(bga, oG, tl, tc * 3 * n4 + 2 * n4);
      chp[tl] = r1_ldo<NT, CMP>(bhp, oN, tl, tc * n4);
      cdo[tl] = r1_ldo<NT, CMP>(bdo, oN, tl, tc * n4);      // (zeros when the tensor is not given)
    }
  };
  fetch(tlast);
  for (int t = tlast; t >= t0; --t) {
    const bool live = t < len;
    f32x4 dcp[NT], dup[NT], dhn[NT], hr1[NT], rr[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const f32x4 d = r1_sel(live, dh[tl] + cdo[tl], R1Z4);
      const f32x4 due = d * (chp[tl] - cc[tl]);
      dup[tl] = due * cu[tl] * (1.0f - cu[tl]);
      dcp[tl] = d * (1.0f - cu[tl]) * (1.0f - cc[tl] * cc[tl]);
      dhn[tl] = d * cu[tl];
      hr1[tl] = chp[tl] * cr[tl] * (1.0f - cr[tl]);
      rr[tl] = cr[tl];
    }
    fetch(t - 1);     // the saved activations of the next (earlier) step arrive behind this step's MFMAs
    f32x4 drp[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      f32x4 drh = R1Z4;
      r1_mv1<NT, CMP>(drh, wc[ot], dcp);
      drp[ot] = drh * hr1[ot];
      dhn[ot] += drh * rr[ot];
    }
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      f32x4 e = R1Z4;
      r1_mv2<NT, CMP>(dhn[ot], wu[ot], dup, e, wr[ot], drp);
      dhn[ot] += e;
    }
    const unsigned dp = (unsigned)t * a.lddp * es;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp, drp[tl], hbf);
      r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp + n * es, dup[tl], hbf);
      r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp + 2 * n * es, dcp[tl], hbf);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) dh[tl] = r1_sel(live, dhn[tl], dh[tl]);
  }
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) r1_st<NT, CMP>(bdh0, (unsigned)(hc * n4), tl, g, n, dh[tl], hv);
}

// ================================================================================ Time4LSTM
// Pin blocks (each n wide): 0 i | 1 j | 2 f | 3 o (already holds Tn.Wo1 + Tl.Wo2) | 4 tns | 5 tls
//   c' = sig(f + 1) * sig(tls) * c + sig(i) * sig(tns) * tanh(j) ;  m' = sig(o) * tanh(c')
template <int NT, bool CMP>
__device__ __forceinline__ void t4_1_fwd(const T4Args& a, const int bx) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hv = h < a.Hn;
  const long hc = hv ? h : 0;
  const r1_rsrc_t bWm = r1_rsrc(a.Wm), bPin = r1_rsrc(a.Pin), bos = r1_rsrc(a.out_seq), bact = r1_rsrc(a.act);
  const r1_rsrc_t bcst = r1_rsrc(a.cst), bmp = r1_rsrc(a.mprev), bsi = r1_rsrc(a.st_in), bso = r1_rsrc(a.st_out);
  f32x4 wm[4][NT][NT];
#pragma unroll
  for (int gb = 0; gb < 4; ++gb)
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) wm[gb][ot][kt] = r1_ld_w<NT, CMP>(bWm, a.ldm, gb * n, n, ot, kt, j, g, false);
  const unsigned n4 = 4u * n;
  f32x4 cs[NT], ms[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    cs[tl] = r1_ld<NT, CMP>(bsi, (unsigned)(hc * 2 * n4), tl, g, n, hv);
    ms[tl] = r1_ld<NT, CMP>(bsi, (unsigned)(hc * 2 * n4) + n4, tl, g, n, hv);
  }
  const int len = hv ? min(a.seq_len[hc * a.len_stride], T) : 0;
  const int t0 = a.t0, tend = min(r1_wave_max(len), a.t1);
  const R1Off<NT> oP = r1_off<NT, CMP>(hc * (long)T * a.ldp, g, n, hv, 4);
  const R1Off<NT> oN = r1_off<NT, CMP>(hc * (long)T * n, g, n, hv, 4);
  const R1Off<NT> oA = r1_off<NT, CMP>(hc * (long)T * 6 * n, g, n, hv, 4);
  const unsigned ldp4 = 4u * a.ldp;
  f32x4 pn[6][NT];
#pragma unroll
  for (int gb = 0; gb < 6; ++gb)
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) pn[gb][tl] = r1_ldo<NT, CMP>(bPin, oP, tl, min(t0, T - 1) * ldp4 + gb * n4);
  for (int t = t0; t < tend; ++t) {
    const bool live = t < len;
    f32x4 acc[4][NT], tns[NT], tls[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) acc[gb][tl] = pn[gb][tl];
      tns[tl] = pn[4][tl];
      tls[tl] = pn[5][tl];
    }
    {
      const unsigned tn = min(t + 1, T - 1) * ldp4;
#pragma unroll
      for (int gb = 0; gb < 6; ++gb)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) pn[gb][tl] = r1_ldo<NT, CMP>(bPin, oP, tl, tn + gb * n4);
    }
    f32x4 cn[NT], mn[NT], ig[NT], jg[NT], fg[NT], og[NT], tng[NT], tlg[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      // gate by gate (each product as two half chains): the activation of one gate has the MFMAs of the next gate to
      // hide under; the time gates (input side only) hide under the first product of the tile
      r1_mv1<NT, CMP>(acc[0][ot], wm[0][ot], ms);
      tng[ot] = r1_sig4(tns[ot]);
      tlg[ot] = r1_sig4(tls[ot]);
      r1_mv1<NT, CMP>(acc[1][ot], wm[1][ot], ms);
      ig[ot] = r1_sig4(acc[0][ot]);
      r1_mv1<NT, CMP>(acc[2][ot], wm[2][ot], ms);
      jg[ot] = r1_tanh4(acc[1][ot]);
      r1_mv1<NT, CMP>(acc[3][ot], wm[3][ot], ms);
      fg[ot] = r1_sig4(acc[2][ot] + 1.0f);
      cn[ot] = fg[ot] * tlg[ot] * cs[ot] + ig[ot] * tng[ot] * jg[ot];
      og[ot] = r1_sig4(acc[3][ot]);
      mn[ot] = og[ot] * r1_tanh4(cn[ot]);
    }
    const unsigned tn4 = t * n4, ta = t * 6 * n4;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      r1_sto<NT, CMP>(bos, oN, tl, tn4, mn[tl]);
      r1_sto<NT, CMP>(bact, oA, tl, ta, ig[tl]); r1_sto<NT, CMP>(bact, oA, tl, ta + n4, jg[tl]);
      r1_sto<NT, CMP>(bact, oA, tl, ta + 2 * n4, fg[tl]); r1_sto<NT, CMP>(bact, oA, tl, ta + 3 * n4, og[tl]);
      r1_sto<NT, CMP>(bact, oA, tl, ta + 4 * n4, tng[tl]); r1_sto<NT, CMP>(bact, oA, tl, ta + 5 * n4, tlg[tl]);
      r1_sto<NT, CMP>(bcst, oN, tl, tn4, cn[tl]);
      r1_sto<NT, CMP>(bmp, oN, tl, tn4, ms[tl]);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      cs[tl] = r1_sel(live, cn[tl], cs[tl]);
      ms[tl] = r1_sel(live, mn[tl], ms[tl]);
    }
  }
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    for (int t = max(len, t0); t < a.t1; ++t) r1_st<NT, CMP>(bos, (unsigned)(hc * T + t) * n4, tl, g, n, R1Z4, hv);
    r1_st<NT, CMP>(bso, (unsigned)(hc * 2 * n4), tl, g, n, cs[tl], hv);
    r1_st<NT, CMP>(bso, (unsigned)(hc * 2 * n4) + n4, tl, g, n, ms[tl], hv);
  }
}

template <int NT, bool CMP>
__device__ __forceinline__ void t4_1_bwd(const T4Args& a, const int bx) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int n = a.n, T = a.T;
  const long h = (long)bx * 16 + j;
  const bool hv = h < a.Hn;
  const long hc = hv ? h : 0;
  const r1_rsrc_t bWm = r1_rsrc(a.Wm), bact = r1_rsrc(a.act), bcst = r1_rsrc(a.cst), bdo = r1_rsrc(a.dout_seq);
  const r1_rsrc_t bdp = r1_rsrc(a.dPin), bdi = r1_rsrc(a.dst_in), bdo2 = r1_rsrc(a.dst_out);
  f32x4 wm[4][NT][NT];
#pragma unroll
  for (int gb = 0; gb < 4; ++gb)
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) wm[gb][ot][kt] = r1_ld_w<NT, CMP>(bWm, a.ldm, gb * n, n, ot, kt, j, g, true);
  const unsigned n4 = 4u * n;
  f32x4 dc[NT], dm[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    dc[tl] = r1_ld<NT, CMP>(bdi, (unsigned)(hc * 2 * n4), tl, g, n, hv);
    dm[tl] = r1_ld<NT, CMP>(bdi, (unsigned)(hc * 2 * n4) + n4, tl, g, n, hv);
  }
  const int len = hv ? min(a.seq_len[hc * a.len_stride], T) : 0;
  const int t0 = a.t0, tlast = min(r1_wave_max(len), a.t1) - 1;
  const int hbf = a.dpin_bf16;
  const unsigned es = hbf ? 2u : 4u;
  for (int t = max(len, max(t0, tlast + 1)); t < a.t1; ++t) {
    const unsigned dp = (unsigned)((hc * T + t) * a.lddp);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
      for (int gb = 0; gb < 6; ++gb) r1_st_dpin<NT, CMP>(bdp, dp + gb * n, tl, g, n, R1Z4, hbf, hv);
  }
  const R1Off<NT> oN = r1_off<NT, CMP>(hc * (long)T * n, g, n, hv, 4);
  const R1Off<NT> oA = r1_off<NT, CMP>(hc * (long)T * 6 * n, g, n, hv, 4);
  const R1Off<NT> oD = r1_off<NT, CMP>(hc * (long)T * a.lddp, g, n, hv, (int)es);
  f32x4 ca[6][NT], ccn[NT], ccp[NT], cdo[NT];
  auto fetch = [&](int t) {
    const unsigned tc = (unsigned)(t > 0 ? t : 0), tp = (unsigned)(t > 1 ? t - 1 : 0);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
      for (int gb = 0; gb < 6; ++gb) ca[gb][tl] = r1_ldo<NT, CMP>(bact, oA, tl, tc * 6 * n4 + gb * n4);
      ccn[tl] = r1_ldo<NT, CMP>(bcst, oN, tl, tc * n4);
      ccp[tl] = r1_sel(t > 0, r1_ldo<NT, CMP>(bcst, oN, tl, tp * n4), R1Z4);      // c entering the step (zero at t = 0)
      cdo[tl] = r1_ldo<NT, CMP>(bdo, oN, tl, tc * n4);
    }
  };
  fetch(tlast);
  for (int t = tlast; t >= t0; --t) {
    const bool live = t < len;
    f32x4 dg[4][NT], dtn[NT], dtl[NT], dcn[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const f32x4 ig = ca[0][tl], jg = ca[1][tl], fg = ca[2][tl], og = ca[3][tl], tn = ca[4][tl], tlg = ca[5][tl];
      const f32x4 cp = ccp[tl];
      const f32x4 d = r1_sel(live, dm[tl] + cdo[tl], R1Z4);
      const f32x4 tc = r1_tanh4(ccn[tl]);
      const f32x4 dcc = r1_sel(live, dc[tl] + d * og * (1.0f - tc * tc), R1Z4);
      dg[3][tl] = d * tc * og * (1.0f - og);                          // d o_pre
      dg[2][tl] = dcc * tlg * cp * fg * (1.0f - fg);                  // d f_pre
      dg[0][tl] = dcc * tn * jg * ig * (1.0f - ig);                   // d i_pre
      dg[1][tl] = dcc * ig * tn * (1.0f - jg * jg);                   // d j_pre
      dtn[tl] = dcc * ig * jg * tn * (1.0f - tn);                     // d tns_pre
      dtl[tl] = dcc * fg * cp * tlg * (1.0f - tlg);                   // d tls_pre
      dcn[tl] = dcc * fg * tlg;
    }
    fetch(t - 1);
    f32x4 dmn[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {     // d m_prev[tile ot] = sum over gates and k tiles: two chains
      f32x4 da = R1Z4, db = R1Z4;
      r1_mv2<NT, CMP>(da, wm[0][ot], dg[0], db, wm[1][ot], dg[1]);
      r1_mv2<NT, CMP>(da, wm[2][ot], dg[2], db, wm[3][ot], dg[3]);
      dmn[ot] = da + db;
    }
    const unsigned dp = (unsigned)t * a.lddp * es, ne = n * es;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp + gb * ne, dg[gb][tl], hbf);
      r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp + 4 * ne, dtn[tl], hbf);
      r1_sto_dpin<NT, CMP>(bdp, oD, tl, dp + 5 * ne, dtl[tl], hbf);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      dc[tl] = r1_sel(live, dcn[tl], dc[tl]);
      dm[tl] = r1_sel(live, dmn[tl], dm[tl]);
    }
  }
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    r1_st<NT, CMP>(bdo2, (unsigned)(hc * 2 * n4), tl, g, n, dc[tl], hv);
    r1_st<NT, CMP>(bdo2, (unsigned)(hc * 2 * n4) + n4, tl, g, n, dm[tl], hv);
  }
}

// workgroup = the encoders of the launch over the same 16 histories: wave w runs encoder w (GRUs first, then the
// Time4LSTM); the waves never meet (no LDS, no barrier) and sit on different SIMDs of the CU
template <int NT, bool CMP, bool BWD>
__global__ void __launch_bounds__(256) rnn1_kernel(RnnMultiArgs a) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_amdgcn_s_setprio(3);      // T-serial chain: ahead of whatever throughput kernel shares the SIMD
  if (w < a.ngru) {
    if (BWD) gru1_bwd<NT, CMP>(a.gru[w], blockIdx.x);
    else gru1_fwd<NT, CMP>(a.gru[w], blockIdx.x);
  } else {
    if (BWD) t4_1_bwd<NT, CMP>(a.t4, blockIdx.x);
    else t4_1_fwd<NT, CMP>(a.t4, blockIdx.x);
  }
}

static bool rnn1_class(int n, int* nt, bool* cmp) {
  if (n < 4 || n > 48 || n % 4) return false;
  *nt = (n + 15) / 16;
  *cmp = (*nt == 3) && (n - 32 <= 8);     // built compact form: the third tile of hidden sizes 36 / 40
  return true;
}

bool rnn1_supported(const RnnMultiArgs& m) {
  const char* on = getenv("CLSR_RNN1");      // opt-in (see the header); read per call: tests switch it inside one process
  if (!on || on[0] != '1') return false;
  const int cnt = m.ngru + (m.has_t4 ? 1 : 0);
  if (cnt < 1 || cnt > 4) return false;
  int n0 = m.has_t4 ? m.t4.n : m.gru[0].n;
  int nt; bool cmp;
  if (!rnn1_class(n0, &nt, &cmp)) return false;
  for (int i = 0; i < m.ngru; ++i)
    if (m.gru[i].n != n0 || m.gru[i].att) return false;
  // raw buffer addressing: 32-bit byte offsets into every tensor of the launch
  const GruArgs& g0 = m.gru[0];
  const long rows = m.has_t4 ? (long)m.t4.Hn * m.t4.T : (long)g0.Hn * g0.T;
  long ldmax = 6L * n0;
  for (int i = 0; i < m.ngru; ++i) ldmax = std::max(ldmax, (long)std::max(m.gru[i].ldp, m.gru[i].lddp));
  if (m.has_t4) ldmax = std::max(ldmax, (long)std::max(m.t4.ldp, m.t4.lddp));
  if (rows * ldmax * 4 >= 0x7FFF0000L) return false;
  for (int i = 0; i < m.ngru; ++i)
    if (g0.Hn * (long)m.gru[i].h0_stride * 4 >= 0x7FFF0000L) return false;
  return true;
}

int rnn1_launch(const RnnMultiArgs& m, int Hn, bool backward, hipStream_t stream) {
  const int n0 = m.has_t4 ? m.t4.n : m.gru[0].n;
  int nt; bool cmp;
  rnn1_class(n0, &nt, &cmp);
  const dim3 grid(clsr_cdiv(Hn, 16)), block(64 * (m.ngru + (m.has_t4 ? 1 : 0)));
#define R1_GO(NT_, CMP_)                                                                          \
  if (nt == NT_ && cmp == CMP_) {                                                                 \
    if (backward) hipLaunchKernelGGL((rnn1_kernel<NT_, CMP_, true>), grid, block, 0, stream, m);  \
    else hipLaunchKernelGGL((rnn1_kernel<NT_, CMP_, false>), grid, block, 0, stream, m);          \
  }
  R1_GO(3, true) R1_GO(3, false) R1_GO(2, false) R1_GO(1, false)
#undef R1_GO
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// 1 when clsr_rnn_*_multi runs the one-wave-per-encoder kernels for recurrences of hidden size n (all encoders of the
// launch alike): the host then puts every recurrence of a pass into ONE launch (they do not compete for a matrix pipe)
extern "C" int clsr_rnn_one_wave(int n) {
  RnnMultiArgs m = {};
  m.ngru = 1;
  m.gru[0].n = n;
  return rnn1_supported(m) ? 1 : 0;
}
