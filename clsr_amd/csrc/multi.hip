// Multi-launch forms of the small kernels of the CLSR step (gfx950): several independent jobs per launch,
// blockIdx.y = job, descriptors passed to the kernel BY VALUE (read from host memory at call time).
// Same arithmetic as the single-job kernels in embedding.hip / attention.hip / optim.hip.
#include <stdlib.h>
#include "common.h"
#include "clsr_hip.h"

template <typename D>
struct MultiArgs {
  D d[CLSR_MULTI_MAX];
};

extern "C" int clsr_sizeof_zero_desc(void) { return (int)sizeof(clsr_zero_desc); }
extern "C" int clsr_sizeof_scatter_desc(void) { return (int)sizeof(clsr_scatter_desc); }
extern "C" int clsr_sizeof_multi_descs(int* mark, int* gather, int* rp, int* table) {
  if (mark) *mark = (int)sizeof(clsr_mark_desc);
  if (gather) *gather = (int)sizeof(clsr_gather_desc);
  if (rp) *rp = (int)sizeof(clsr_rp_desc);
  if (table) *table = (int)sizeof(clsr_table_desc);
  return CLSR_OK;
}

// ---- involved-row flags (embedding.hip: mark_rows_kernel)
__global__ void __launch_bounds__(256) mark_rows_multi_kernel(MultiArgs<clsr_mark_desc> a) {
  const clsr_mark_desc d = a.d[blockIdx.y];
  const long n = d.nrows * d.ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long r = e / d.ncols;
    const int c = (int)(e - r * d.ncols);
    d.flags[d.idx[r * d.row_stride + c]] = 1;
  }
}

extern "C" int clsr_mark_rows_multi(const clsr_mark_desc* descs, int n, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_MULTI_MAX);
  MultiArgs<clsr_mark_desc> a;
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    CLSR_CHECK_ARG(descs[i].idx && descs[i].flags && descs[i].nrows >= 0 && descs[i].ncols > 0);
    a.d[i] = descs[i];
    const long e = descs[i].nrows * descs[i].ncols;
    mx = e > mx ? e : mx;
  }
  int blocks = clsr_cdiv(mx, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mark_rows_multi_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- zero fills: the per-step accumulators (norm / loss slots, gradient pool, sort counters, user counter) in ONE launch
__global__ void __launch_bounds__(256) zero_multi_kernel(MultiArgs<clsr_zero_desc> a) {
  const clsr_zero_desc d = a.d[blockIdx.y];
  const long words = d.nbytes >> 2;
  unsigned* w = reinterpret_cast<unsigned*>(d.p);
  long head = ((16 - ((uintptr_t)d.p & 15)) & 15) >> 2;     // words up to the first 16-byte boundary
  head = head < words ? head : words;
  const long n16 = (words - head) >> 2;
  uint4* v = reinterpret_cast<uint4*>(w + head);
  const uint4 z = {0u, 0u, 0u, 0u};
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (long)gridDim.x * blockDim.x) v[e] = z;
  if (blockIdx.x == 0) {
    for (long e = threadIdx.x; e < head; e += blockDim.x) w[e] = 0u;
    for (long e = head + 4 * n16 + threadIdx.x; e < words; e += blockDim.x) w[e] = 0u;
  }
}

extern "C" int clsr_zero_multi(const clsr_zero_desc* descs, int n, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_MULTI_MAX);
  MultiArgs<clsr_zero_desc> a;
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    CLSR_CHECK_ARG(descs[i].p && descs[i].nbytes >= 0 && descs[i].nbytes % 4 == 0 && ((uintptr_t)descs[i].p % 4) == 0);
    a.d[i] = descs[i];
    mx = descs[i].nbytes > mx ? descs[i].nbytes : mx;
  }
  int blocks = clsr_cdiv(mx, 256 * 16 * 4);     // four 16-byte stores per thread
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_multi_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- row scatters of the embedding gradients (embedding.hip: scatter_add_rows_kernel), one job per lookup site
__global__ void __launch_bounds__(256) scatter_add_rows_multi_kernel(MultiArgs<clsr_scatter_desc> a) {
  __shared__ double red[4];
  const clsr_scatter_desc d = a.d[blockIdx.y];
  const long total = (long)d.N * d.C;
  float local = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / d.C), c = (int)(e - (long)n * d.C);
    const float v = d.src[(long)n * d.ld_src + d.col0 + c];
    local += v * v;
    atomicAdd(d.tbl_grad + (long)d.idx[(long)n * d.idx_stride] * d.C + c, v);
  }
  if (d.sumsq) {   // block-uniform
    const double tot = block256_sum_d((double)local, red);
    if (threadIdx.x == 0 && tot != 0.0) atomicAdd(d.sumsq, tot);
  }
}

extern "C" int clsr_scatter_add_rows_multi(const clsr_scatter_desc* descs, int n, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_MULTI_MAX);
  MultiArgs<clsr_scatter_desc> a;
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    CLSR_CHECK_ARG(descs[i].src && descs[i].idx && descs[i].tbl_grad && descs[i].N >= 0 && descs[i].C > 0);
    a.d[i] = descs[i];
    const long e = (long)descs[i].N * descs[i].C;
    mx = e > mx ? e : mx;
  }
  int blocks = clsr_cdiv(mx, 256 * 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(scatter_add_rows_multi_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- row gathers (embedding.hip: gather_rows_kernel)
template <bool H>
__global__ void __launch_bounds__(256) gather_rows_multi_kernel(MultiArgs<clsr_gather_desc> a) {
  const clsr_gather_desc d = a.d[blockIdx.y];
  const int QC = d.C >> 2;
  const long total = (long)d.N * QC;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / QC), q = (int)(e - (long)n * QC);
    const f32x4 v = load4e<H>(d.table, (long)d.idx[(long)n * d.idx_stride] * d.C + 4 * q);
    st4(d.out + (long)n * d.ldo + d.col0 + 4 * q, v);
  }
}

static int gather_rows_multi_launch(const clsr_gather_desc* descs, int n, int bf16, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_MULTI_MAX);
  MultiArgs<clsr_gather_desc> a;
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    const clsr_gather_desc& d = descs[i];
    CLSR_CHECK_ARG(d.table && d.idx && d.out && d.N >= 0);
    CLSR_CHECK_SUPPORTED(d.C % 4 == 0 && d.C > 0 && d.ldo % 4 == 0 && d.col0 % 4 == 0);
    a.d[i] = d;
    const long e = (long)d.N * (d.C / 4);
    mx = e > mx ? e : mx;
  }
  int blocks = clsr_cdiv(mx, 256);
  if (blocks > 2048) blocks = 2048;
  if (bf16) hipLaunchKernelGGL(gather_rows_multi_kernel<true>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gather_rows_multi_kernel<false>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_gather_rows_multi(const clsr_gather_desc* descs, int n, void* stream) {
  return gather_rows_multi_launch(descs, n, 0, stream);
}
// the tables of the descriptors are bf16 [V, C] (the pointer field is read as such); outputs fp32
extern "C" int clsr_gather_rows_multi_h(const clsr_gather_desc* descs, int n, void* stream) {
  return gather_rows_multi_launch(descs, n, 1, stream);
}

// ---- out[e] (=|+=) scale * sum_p partial[p*stride + e]  (attention.hip: reduce_parts_f_kernel)
__global__ void __launch_bounds__(256) reduce_parts_multi_kernel(MultiArgs<clsr_rp_desc> a) {
  __shared__ float red[4];
  const clsr_rp_desc d = a.d[blockIdx.y];
  const int e = blockIdx.x;
  if (e >= d.n) return;  // block-uniform
  float s = 0.f;
  for (int p = threadIdx.x; p < d.nparts; p += 256) s += d.partial[(long)p * d.stride + e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = ((red[0] + red[1]) + (red[2] + red[3])) * d.scale;
    d.out[e] = d.accumulate ? d.out[e] + s : s;
  }
}

extern "C" int clsr_reduce_parts_multi(const clsr_rp_desc* descs, int n, void* stream) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= CLSR_MULTI_MAX);
  MultiArgs<clsr_rp_desc> a;
  int mx = 1;
  for (int i = 0; i < n; ++i) {
    const clsr_rp_desc& d = descs[i];
    CLSR_CHECK_ARG(d.partial && d.out && d.nparts > 0 && d.n > 0 && d.stride >= d.n);
    a.d[i] = d;
    mx = d.n > mx ? d.n : mx;
  }
  hipLaunchKernelGGL(reduce_parts_multi_kernel, dim3(mx, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

// ---- regulariser pass over the involved rows of several tables (optim.hip: table_reg_kernel)
struct TablesArgs {
  clsr_table_desc t[4];
  float l2;
  const float* ucount;
  double* reg_loss;
  float l1;
  float clip_norm;
  const double* adam_state;
  float b1, b2, eps;
  int lazy;
};

template <bool H>
__global__ void __launch_bounds__(256) tables_reg_multi_kernel(TablesArgs a) {
  const clsr_table_desc d = a.t[blockIdx.y];
  const int C = d.C;
  const float cd = d.partner ? d.disc_scale / (a.ucount[0] * (float)C) : 0.f;
  const float cl = (d.partner && d.disc_loss) ? d.disc_loss_scale / (a.ucount[0] * (float)C) : 0.f;
  double ss = 0.0, rl = 0.0, dl = 0.0;
  const long total = d.V * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / C;
    if (!d.flags[row]) continue;
    const float p = tbl_ld<H>(d.table, e);
    float g = a.l2 * p + a.l1 * (float)((p > 0.f) - (p < 0.f));
    if (d.partner) {
      const float df = p - tbl_ld<H>(d.partner, e);
      g += cd * df;
      dl += (double)df * df;
    }
    d.grad[e] += g;
    ss += (double)g * g;
    rl += 0.5 * (double)a.l2 * p * p + (double)a.l1 * fabsf(p);
  }
  __shared__ double red[3][4];
  ss = block256_sum_d(ss, red[0]); rl = block256_sum_d(rl, red[1]); dl = block256_sum_d(dl, red[2]);
  if (threadIdx.x == 0) {
    if (ss != 0.0) atomicAdd(d.sumsq_reg, ss);
    if (a.reg_loss && rl != 0.0) atomicAdd(a.reg_loss, rl);
    if (d.disc_loss && dl != 0.0) atomicAdd(d.disc_loss, (double)cl * dl);
  }
}

// The same sweep for tables whose rows are multiples of four values (all of the reference's): one 16-byte access per
// thread and operand, 32-bit index arithmetic (the scalar form divides a 64-bit element index by C per element), and the
// loads of four grid strides issued together from clamped addresses, selected by the row flags afterwards -- a load
// behind `if (flag)` leaves only once the flag has arrived, one dependent round trip per element of the stride loop.
// Measured at configs[1] (5 M values in four tables, on the tail of the step): 40 -> 36 us, the Adam sweep 32 -> 27 us --
// both were closer to their traffic (~70 MB through the L2 / Infinity Cache) than to their latency chains.
template <bool H>
__global__ void __launch_bounds__(256) tables_reg_multi_v4_kernel(TablesArgs a) {
  // (memory-bound sweep that may run BESIDE the fused encoder tail -- the early update of the user tables: without a wave
  //  priority above that kernel's it only got the issue slots the MFMA chain left over, 329 us for 25 us of work)
  __builtin_amdgcn_s_setprio(3);
  const clsr_table_desc d = a.t[blockIdx.y];
  const unsigned QC = (unsigned)d.C >> 2, total = (unsigned)d.V * QC;
  const float cd = d.partner ? d.disc_scale / (a.ucount[0] * (float)d.C) : 0.f;
  const float cl = (d.partner && d.disc_loss) ? d.disc_loss_scale / (a.ucount[0] * (float)d.C) : 0.f;
  double ss = 0.0, rl = 0.0, dl = 0.0;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned q0 = blockIdx.x * 256u + threadIdx.x; q0 < total; q0 += 4u * stride) {
    bool f[4];
    f32x4 p[4], pp[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned q = q0 + (unsigned)u * stride, qs = q < total ? q : 0u;
      f[u] = q < total && d.flags[qs / QC];
      p[u] = load4e<H>(d.table, 4L * qs);
      pp[u] = d.partner ? load4e<H>(d.partner, 4L * qs) : f32x4{0.f, 0.f, 0.f, 0.f};
      g[u] = ld4(d.grad + 4L * qs);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!f[u]) continue;
      const unsigned q = q0 + (unsigned)u * stride;
      f32x4 r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pv = p[u][c];
        float gv = a.l2 * pv + a.l1 * (float)((pv > 0.f) - (pv < 0.f));
        if (d.partner) {
          const float df = pv - pp[u][c];
          gv += cd * df;
          dl += (double)df * df;
        }
        r[c] = g[u][c] + gv;
        ss += (double)gv * gv;
        rl += 0.5 * (double)a.l2 * pv * pv + (double)a.l1 * fabsf(pv);
      }
      st4(d.grad + 4L * q, r);
    }
  }
  // NO LDS: a workgroup of the fused encoder tail (csrc/encbwd.hip) owns a CU's whole LDS, and a launch that asks for as much
  // as 96 bytes of it waits for that kernel to end -- the early update of the user tables ran 330 us beside it.  One atomic
  // per wave instead of one per workgroup (the sums were unordered across workgroups already).
  ss = wave_sum_d(ss); rl = wave_sum_d(rl); dl = wave_sum_d(dl);
  if ((threadIdx.x & 63) == 0) {
    if (ss != 0.0) atomicAdd(d.sumsq_reg, ss);
    if (a.reg_loss && rl != 0.0) atomicAdd(a.reg_loss, rl);
    if (d.disc_loss && dl != 0.0) atomicAdd(d.disc_loss, (double)cl * dl);
  }
}
// every table of the launch: rows of 4 k values, 16-byte aligned operands, fewer than 2^31 values
static bool tables_v4_ok(const clsr_table_desc* descs, int n, bool with_moments) {
  for (int i = 0; i < n; ++i) {
    const clsr_table_desc& d = descs[i];
    if (d.C % 4 || d.V * d.C >= (1L << 31)) return false;
    if (((uintptr_t)d.table | (uintptr_t)d.grad | (uintptr_t)d.partner) & 15) return false;
    if (with_moments && (((uintptr_t)d.m | (uintptr_t)d.v) & 15)) return false;
  }
  return true;
}

static int fill_tables(TablesArgs& a, const clsr_table_desc* descs, int n, long* max_elems) {
  CLSR_CHECK_ARG(descs && n > 0 && n <= 4);
  long mx = 1;
  for (int i = 0; i < n; ++i) {
    const clsr_table_desc& d = descs[i];
    CLSR_CHECK_ARG(d.table && d.grad && d.flags && d.V > 0 && d.C > 0);
    a.t[i] = d;
    mx = d.V * d.C > mx ? d.V * d.C : mx;
  }
  *max_elems = mx;
  return CLSR_OK;
}

static int tables_reg_multi_launch(const clsr_table_desc* descs, int n, int bf16, float l2, float l1, const float* ucount,
                                   double* reg_loss, void* stream) {
  TablesArgs a = {};
  long mx = 0;
  int rc = fill_tables(a, descs, n, &mx);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) CLSR_CHECK_ARG(descs[i].sumsq_reg && (!descs[i].partner || ucount));
  a.l2 = l2; a.l1 = l1; a.ucount = ucount; a.reg_loss = reg_loss;
  int blocks = clsr_cdiv(mx, 256 * 8);
  if (blocks > 512) blocks = 512;
  if (tables_v4_ok(descs, n, false)) {
    blocks = clsr_cdiv(mx / 4, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    if (bf16) hipLaunchKernelGGL(tables_reg_multi_v4_kernel<true>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(tables_reg_multi_v4_kernel<false>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  } else if (bf16) hipLaunchKernelGGL(tables_reg_multi_kernel<true>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(tables_reg_multi_kernel<false>, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_tables_reg_multi(const clsr_table_desc* descs, int n, float l2, float l1, const float* ucount,
                                     double* reg_loss, void* stream) {
  return tables_reg_multi_launch(descs, n, 0, l2, l1, ucount, reg_loss, stream);
}
// table / partner of every descriptor are bf16 [V, C]
extern "C" int clsr_tables_reg_multi_h(const clsr_table_desc* descs, int n, float l2, float l1, const float* ucount,
                                       double* reg_loss, void* stream) {
  return tables_reg_multi_launch(descs, n, 1, l2, l1, ucount, reg_loss, stream);
}

// ---- Adam sweep of several tables (optim.hip: table_adam_kernel); the flags are cleared by the same launch
//      chain (second kernel: every element of the sweep has read its flag by then)
__device__ __forceinline__ float clipf(double sumsq, float clip_norm) {
  if (clip_norm <= 0.f) return 1.0f;
  const float nrm = (float)sqrt(sumsq);
  return clip_norm / fmaxf(nrm, clip_norm);
}

template <bool H>
__global__ void __launch_bounds__(256) tables_adam_multi_kernel(TablesArgs a) {
  const clsr_table_desc d = a.t[blockIdx.y];
  double tot = 0.0;
  for (int i = 0; i < d.nsum; ++i) tot += d.sumsq_adam[(long)i * d.sumsq_stride];
  const float factor = clipf(tot, a.clip_norm);
  if (a.adam_state[4] != 0.0) return;    // aborted step (csrc/p2p.hip, csrc/headsfused.hip): touch nothing
  const float lr_t = (float)a.adam_state[3];
  const float b1 = a.b1, b2 = a.b2;
  const long total = d.V * d.C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / d.C;
    if (a.lazy && !d.flags[row]) continue;  // every row that got gradient is also flagged as involved
    const float g = d.grad[e] * factor;
    const float mm = b1 * d.m[e] + (1.0f - b1) * g;
    const float vv = b2 * d.v[e] + (1.0f - b2) * g * g;
    d.m[e] = mm;
    d.v[e] = vv;
    tbl_st<H>(d.table, e, tbl_ld<H>(d.table, e) - lr_t * mm / (sqrtf(vv) + a.eps));
    d.grad[e] = 0.f;
  }
}

template <bool H>
__global__ void __launch_bounds__(256) tables_adam_multi_v4_kernel(TablesArgs a) {
  __builtin_amdgcn_s_setprio(3);
  const clsr_table_desc d = a.t[blockIdx.y];
  double tot = 0.0;
  for (int i = 0; i < d.nsum; ++i) tot += d.sumsq_adam[(long)i * d.sumsq_stride];
  const float factor = clipf(tot, a.clip_norm);
  if (a.adam_state[4] != 0.0) return;    // aborted step (csrc/p2p.hip, csrc/headsfused.hip): touch nothing
  const float lr_t = (float)a.adam_state[3];
  const float b1 = a.b1, b2 = a.b2;
  const unsigned QC = (unsigned)d.C >> 2, total = (unsigned)d.V * QC;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned q0 = blockIdx.x * 256u + threadIdx.x; q0 < total; q0 += 2u * stride) {
    bool f[2];
    f32x4 g[2], m[2], v[2], w[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned q = q0 + (unsigned)u * stride, qs = q < total ? q : 0u;
      f[u] = q < total && (!a.lazy || d.flags[qs / QC]);
      g[u] = ld4(d.grad + 4L * qs); m[u] = ld4(d.m + 4L * qs); v[u] = ld4(d.v + 4L * qs);
      w[u] = load4e<H>(d.table, 4L * qs);
      // dense Adam does not read the row flags: the thread of a row's first chunk clears the row's flag here and the
      // separate clearing launch is dropped (lazy Adam: every chunk of a row reads the flag first -- the second launch stays)
      if (!a.lazy && q < total && q % QC == 0) d.flags[q / QC] = 0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!f[u]) continue;
      const long e = 4L * (q0 + (unsigned)u * stride);
      f32x4 mm, vv, ww;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gc = g[u][c] * factor;
        mm[c] = b1 * m[u][c] + (1.0f - b1) * gc;
        vv[c] = b2 * v[u][c] + (1.0f - b2) * gc * gc;
        ww[c] = w[u][c] - lr_t * mm[c] / (sqrtf(vv[c]) + a.eps);
      }
      st4(d.m + e, mm);
      st4(d.v + e, vv);
      tbl_st4<H>(d.table, e, ww);
      st4(d.grad + e, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
}

__global__ void __launch_bounds__(256) tables_clear_flags_kernel(TablesArgs a) {
  const clsr_table_desc d = a.t[blockIdx.y];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < d.V; e += (long)gridDim.x * blockDim.x) d.flags[e] = 0;
}

static int tables_adam_multi_launch(const clsr_table_desc* descs, int n, int bf16, float clip_norm,
                                    const double* adam_state, float beta1, float beta2, float eps, int lazy,
                                    void* stream) {
  TablesArgs a = {};
  long mx = 0;
  int rc = fill_tables(a, descs, n, &mx);
  if (rc) return rc;
  CLSR_CHECK_ARG(adam_state);
  long mxv = 1;
  for (int i = 0; i < n; ++i) {
    CLSR_CHECK_ARG(descs[i].m && descs[i].v && descs[i].sumsq_adam && descs[i].nsum > 0);
    mxv = descs[i].V > mxv ? descs[i].V : mxv;
  }
  a.clip_norm = clip_norm; a.adam_state = adam_state; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.lazy = lazy;
  int blocks = clsr_cdiv(mx, 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = tables_v4_ok(descs, n, true);
  if (v4) {
    blocks = clsr_cdiv(mx / 4, 256 * 2);
    if (blocks > 2048) blocks = 2048;
    if (bf16) hipLaunchKernelGGL(tables_adam_multi_v4_kernel<true>, dim3(blocks, n), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tables_adam_multi_v4_kernel<false>, dim3(blocks, n), dim3(256), 0, s, a);
  } else if (bf16) hipLaunchKernelGGL(tables_adam_multi_kernel<true>, dim3(blocks, n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(tables_adam_multi_kernel<false>, dim3(blocks, n), dim3(256), 0, s, a);
  CLSR_CHECK_LAUNCH();
  if (v4 && !lazy) return CLSR_OK;          // (the sweep cleared the flags itself)
  int cb = clsr_cdiv(mxv, 256);
  if (cb > 512) cb = 512;
  hipLaunchKernelGGL(tables_clear_flags_kernel, dim3(cb, n), dim3(256), 0, s, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
extern "C" int clsr_tables_adam_multi(const clsr_table_desc* descs, int n, float clip_norm,
                                      const double* adam_state, float beta1, float beta2, float eps, int lazy,
                                      void* stream) {
  return tables_adam_multi_launch(descs, n, 0, clip_norm, adam_state, beta1, beta2, eps, lazy, stream);
}
// the tables of the descriptors are bf16 [V, C]: widened, updated in fp32, rounded to nearest-even
extern "C" int clsr_tables_adam_multi_h(const clsr_table_desc* descs, int n, float clip_norm,
                                        const double* adam_state, float beta1, float beta2, float eps, int lazy,
                                        void* stream) {
  return tables_adam_multi_launch(descs, n, 1, clip_norm, adam_state, beta1, beta2, eps, lazy, stream);
}
