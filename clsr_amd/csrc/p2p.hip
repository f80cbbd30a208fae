// Small-message all-reduce over peer-mapped buffers (gfx950, xGMI / same-device IPC) -- SURVEY 8(b)(ii): "communicators
// are created by the host and passed in as opaque void* to the collective wrappers".
//
// Why: synchronised batch-norm (the statistics of tf.layers.batch_normalization, models/base_model.py:673-679,
// must be those of the GLOBAL batch for single-device parity) needs 16 all-reduces of 2 C doubles (<= 2 KB) per step,
// each in the middle of a dependent chain of kernels.  Through torch.distributed every one costs ~20 us of host call plus
// two stream hops into and out of RCCL's stream, even with one rank (+6.6 % step time).  Here the sum is ONE
// single-workgroup kernel on the issuing stream:
//   1. push   every rank writes its n doubles into ITS slot of every peer's exchange buffer, fences, then publishes the
//             sequence number of this all-reduce in its flag of every peer's buffer (release, system scope);
//   2. wait   until the flags of all ranks in the OWN buffer have reached the sequence number (acquire);
//   3. sum    the world slots of the own buffer in RANK ORDER (every rank computes the same fp64 sum: replicas stay
//             bit-identical), in place.
// Two slot sets alternate with the sequence number: a rank can be at most one all-reduce ahead of a peer (it cannot
// finish all-reduce k + 1 before that peer has pushed k + 1, which the peer does after it has finished reading k) -- as
// long as the all-reduces of a sequence execute IN ORDER, which holds per HIP stream only: the step issues them from two
// streams (main chain, long-term attention chain), and with one sequence for both a later all-reduce that ran first
// had its flag overwritten by an earlier one that ran second (the peers then waited for a number that never came back).
// Every stream therefore gets a CHANNEL of its own (slots, flags, sequence), assigned in order of first use -- the same
// order on every rank, because every rank issues the same sequence of calls.
// The exchange buffers are fine-grained device allocations (no stale lines in a reader's L2) mapped into the peers by
// hipIpc handles that the host exchanges once.
// A peer that never arrives must not hang the device: the wait is bounded (clsr_p2p_timeout_ticks below: CLSR_P2P_TIMEOUT_S
// when set; otherwise 60 s while the job warms up -- the ranks' first steps drift apart by seconds -- and 10 s once the
// stepper has armed the bound: the ranks of a step meet at the process-group collective that opens it, so inside a step they
// are milliseconds apart).  A wait that gives up (1) raises the communicator's STICKY error word, (2) raises the step's
// abort flag (clsr_comm_set_abort: adam_state[4] of the net -- every optimiser kernel returns without touching a
// parameter or a moment and the Adam clock stops while it is set) and (3) returns NaN instead of a partial sum, so that
// nothing downstream can mistake the statistics for valid ones.  The batch-norm moving statistics of that step ARE
// poisoned by it: the host treats the net as invalid until a checkpoint is restored (CLSRNet.check_abort -- sticky; raised
// at every loss read, by state_dict and so by every checkpoint, at the end of an epoch, and one step late by the
// data-parallel stepper in train_step and in the run() of capture()).
#include "common.h"
#include "clsr_hip.h"
#include <string.h>
#include <stdlib.h>

#define P2P_MAXW 8            // ranks (one node)
#define P2P_MAXN 256          // doubles per all-reduce
#define P2P_NCH 4             // channels = streams that issue all-reduces

struct P2PBuf {
  double data[P2P_NCH][2][P2P_MAXW][P2P_MAXN];
  unsigned long long flag[P2P_NCH][2][P2P_MAXW];
  unsigned long long err;
  // fused sync-BN kernels (bn_sync_kernel): this rank's folded sums of a layer and the count of its blocks that have arrived
  double stage[P2P_NCH][P2P_MAXN];
  unsigned arrived[P2P_NCH];
};

struct P2PComm {
  int rank, world;
  P2PBuf* bufs[P2P_MAXW];     // own buffer at [rank], peers' mapped buffers elsewhere
  unsigned long long seq[P2P_NCH];     // host-side count of issued all-reduces per channel
  void* stream_of[P2P_NCH];            // channel -> the stream that owns it (assigned at first use)
  int nch;
  double* abort_flag;                  // optional device double raised when a wait gives up (clsr_comm_set_abort)
};

struct P2PArgs {
  P2PBuf* bufs[P2P_MAXW];
  double* x;
  int n, rank, world, ch;
  unsigned long long seq;
  long long timeout_ticks;      // of the 100 MHz wall clock
  double* abort_flag;
};

__global__ void __launch_bounds__(256) allreduce_small_kernel(P2PArgs a) {
  __shared__ int gave_up;
  const int tid = threadIdx.x;
  if (tid == 0) gave_up = 0;
  __syncthreads();
  const int slot = (int)(a.seq & 1ull);
  P2PBuf* mine = a.bufs[a.rank];
  if (tid < a.n) {
    const double v = a.x[tid];
    for (int p = 0; p < a.world; ++p)
      __hip_atomic_store(&a.bufs[p]->data[a.ch][slot][a.rank][tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < a.world) {
    __hip_atomic_store(&a.bufs[tid]->flag[a.ch][slot][a.rank], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();            // constant 100 MHz counter
    while (__hip_atomic_load(&mine->flag[a.ch][slot][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
      if (wall_clock64() - t0 > a.timeout_ticks) {   // a peer never arrived
        __hip_atomic_store(&mine->err, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.abort_flag) __hip_atomic_store(a.abort_flag, (double)a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gave_up = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < a.n) {
    double s = 0.0;
    for (int r = 0; r < a.world; ++r)
      s += __hip_atomic_load(&mine->data[a.ch][slot][r][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.x[tid] = gave_up ? __builtin_nan("") : s;     // never a partial sum that looks like statistics
  }
}

// ---- synchronised batch-norm statistics of ONE layer in ONE launch (VERDICT r4/r5 #3c): block c folds the per-block partial
// sums of feature c (as bn_finalize_kernel / bn_bwd_coef_kernel do), the LAST block of the rank to arrive pushes the 2 C folded
// doubles to every peer (the protocol of allreduce_small_kernel: slot = seq & 1, flags with system-scope release / acquire,
// sum in rank order -- bit-identical on every rank) and finishes the layer for all C features.  Replaces clsr_sum_parts_d +
// clsr_allreduce_small + clsr_bn_finalize / clsr_bn_bwd_coef_scaled: 32 launches of the data-parallel step's chain.
struct BnSyncArgs {
  P2PArgs p;                      // (x unused)
  const double* partial; int nparts; int C; double count;
  int mode;                       // 0: forward finalize, 1: backward coefficients
  const float* gamma; const float* beta; float* moving_mean; float* moving_var; float momentum, eps;
  float* scale; float* shift; float* mean_out; float* invstd_out;           // mode 0
  const float* mean; const float* invstd; float* coef; float* dgamma; float* dbeta; double grad_scale;   // mode 1
};

__global__ void __launch_bounds__(256) bn_sync_kernel(BnSyncArgs a) {
  __shared__ double red[2][4];
  __shared__ int last_s, gave_up;
  __shared__ double tot[P2P_MAXN];
  const int tid = threadIdx.x, c = blockIdx.x, C = a.C, n = 2 * C;
  P2PBuf* mine = a.p.bufs[a.p.rank];
  const int ch = a.p.ch;
  double s = 0.0, q = 0.0;
  for (int p = tid; p < a.nparts; p += 256) {
    s += a.partial[((long)p * 2 + 0) * C + c];
    q += a.partial[((long)p * 2 + 1) * C + c];
  }
  s = block256_sum_d(s, red[0]);
  q = block256_sum_d(q, red[1]);
  if (tid == 0) {
    __hip_atomic_store(&mine->stage[ch][c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine->stage[ch][C + c], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&mine->arrived[ch], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = old == (unsigned)C - 1u;
    gave_up = 0;
  }
  __syncthreads();
  if (!last_s) return;
  if (tid == 0) __hip_atomic_store(&mine->arrived[ch], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (the next layer of this channel)
  const int slot = (int)(a.p.seq & 1ull);
  double v = 0.0;
  if (tid < n) v = __hip_atomic_load(&mine->stage[ch][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.p.world > 1) {
    if (tid < n)
      for (int p = 0; p < a.p.world; ++p)
        __hip_atomic_store(&a.p.bufs[p]->data[ch][slot][a.p.rank][tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (tid < a.p.world) {
      __hip_atomic_store(&a.p.bufs[tid]->flag[ch][slot][a.p.rank], a.p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(&mine->flag[ch][slot][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.p.seq) {
        if (wall_clock64() - t0 > a.p.timeout_ticks) {   // a peer never arrived
          __hip_atomic_store(&mine->err, a.p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (a.p.abort_flag) __hip_atomic_store(a.p.abort_flag, (double)a.p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gave_up = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < n) {
      double t = 0.0;
      for (int r = 0; r < a.p.world; ++r)
        t += __hip_atomic_load(&mine->data[ch][slot][r][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = gave_up ? __builtin_nan("") : t;      // never a partial sum that looks like statistics
    }
  }
  if (tid < n) tot[tid] = v;
  __syncthreads();
  if (tid >= C) return;
  const double s1 = tot[tid], s2 = tot[C + tid];
  if (a.mode == 0) {
    const double m = s1 / a.count;
    double var_d = s2 / a.count - m * m;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)m, var = (float)var_d;
    a.moving_mean[tid] = a.moving_mean[tid] * a.momentum + mean * (1.0f - a.momentum);
    a.moving_var[tid] = a.moving_var[tid] * a.momentum + var * (1.0f - a.momentum);
    const float invstd = 1.0f / sqrtf(var + a.eps);
    const float sc = a.gamma[tid] * invstd;
    a.scale[tid] = sc;
    a.shift[tid] = a.beta[tid] - mean * sc;
    if (a.mean_out) a.mean_out[tid] = mean;
    if (a.invstd_out) a.invstd_out[tid] = invstd;
  } else {
    const float g = a.gamma[tid], is = a.invstd[tid], mu = a.mean[tid];
    const float c1 = (float)(s1 / a.count), c2 = (float)(s2 / a.count);
    const float a1 = g * is;
    const float a2 = -g * is * is * c2;
    a.coef[tid] = a1;
    a.coef[C + tid] = a2;
    a.coef[2 * C + tid] = -a1 * c1 - a2 * mu;
    a.dgamma[tid] = (float)(s2 * a.grad_scale);
    a.dbeta[tid] = (float)(s1 * a.grad_scale);
  }
}

long long clsr_p2p_timeout_ticks(void);
static int bn_sync_fill(void* comm, void* stream, P2PArgs& a) {
  P2PComm* c = (P2PComm*)comm;
  for (int r = 0; r < P2P_MAXW; ++r) a.bufs[r] = r < c->world ? c->bufs[r] : nullptr;
  a.x = nullptr; a.n = 0; a.rank = c->rank; a.world = c->world;
  int ch = 0;
  while (ch < c->nch && c->stream_of[ch] != stream) ++ch;
  if (ch == c->nch) {
    CLSR_CHECK_SUPPORTED(c->nch < P2P_NCH);      // more streams than channels
    c->stream_of[c->nch++] = stream;
  }
  a.ch = ch;
  a.seq = ++c->seq[ch];
  a.timeout_ticks = clsr_p2p_timeout_ticks();
  a.abort_flag = c->abort_flag;
  return CLSR_OK;
}
// clsr_bn_finalize (training) with the statistics summed over the ranks of ``comm`` inside the launch; ``count`` = rows of ALL ranks
extern "C" int clsr_bn_finalize_sync(void* comm, const double* stats_partial, int nparts, int C, double count,
                                     const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                                     float momentum, float eps, float* scale, float* shift, float* mean_out,
                                     float* invstd_out, void* stream) {
  CLSR_CHECK_ARG(comm && stats_partial && gamma && beta && moving_mean && moving_var && scale && shift && nparts > 0 && count > 0);
  CLSR_CHECK_SUPPORTED(C > 0 && 2 * C <= P2P_MAXN);
  BnSyncArgs a = {};
  int rc = bn_sync_fill(comm, stream, a.p);
  if (rc) return rc;
  a.partial = stats_partial; a.nparts = nparts; a.C = C; a.count = count; a.mode = 0;
  a.gamma = gamma; a.beta = beta; a.moving_mean = moving_mean; a.moving_var = moving_var; a.momentum = momentum; a.eps = eps;
  a.scale = scale; a.shift = shift; a.mean_out = mean_out; a.invstd_out = invstd_out;
  hipLaunchKernelGGL(bn_sync_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
// clsr_bn_bwd_coef_scaled (no accumulate) with the backward sums summed over the ranks inside the launch
extern "C" int clsr_bn_bwd_coef_sync(void* comm, const double* partial, int nparts, int C, double count, const float* gamma,
                                     const float* mean, const float* invstd, float* coef, float* dgamma, float* dbeta,
                                     double grad_scale, void* stream) {
  CLSR_CHECK_ARG(comm && partial && gamma && mean && invstd && coef && dgamma && dbeta && nparts > 0 && count > 0);
  CLSR_CHECK_SUPPORTED(C > 0 && 2 * C <= P2P_MAXN);
  BnSyncArgs a = {};
  int rc = bn_sync_fill(comm, stream, a.p);
  if (rc) return rc;
  a.partial = partial; a.nparts = nparts; a.C = C; a.count = count; a.mode = 1;
  a.gamma = gamma; a.mean = mean; a.invstd = invstd; a.coef = coef; a.dgamma = dgamma; a.dbeta = dbeta; a.grad_scale = grad_scale;
  hipLaunchKernelGGL(bn_sync_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}

extern "C" long clsr_comm_buffer_bytes(void) { return (long)sizeof(P2PBuf); }
extern "C" int clsr_comm_max_doubles(void) { return P2P_MAXN; }
extern "C" int clsr_comm_max_world(void) { return P2P_MAXW; }

// fine-grained (uncached) device allocation of the exchange buffer, zeroed
extern "C" int clsr_comm_alloc(void** buf_out) {
  CLSR_CHECK_ARG(buf_out);
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, sizeof(P2PBuf), hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, sizeof(P2PBuf), hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    clsr_set_error("%s:%d: exchange buffer allocation failed: %s", __FILE__, __LINE__, hipGetErrorString(e));
    return CLSR_ELAUNCH;
  }
  CLSR_HIP(hipMemset(p, 0, sizeof(P2PBuf)));
  CLSR_HIP(hipDeviceSynchronize());
  *buf_out = p;
  return CLSR_OK;
}
extern "C" int clsr_comm_free(void* buf) {
  if (buf) CLSR_HIP(hipFree(buf));
  return CLSR_OK;
}
extern "C" int clsr_comm_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
extern "C" int clsr_comm_ipc_handle(void* buf, void* handle_out) {
  CLSR_CHECK_ARG(buf && handle_out);
  hipIpcMemHandle_t h;
  CLSR_HIP(hipIpcGetMemHandle(&h, buf));
  memcpy(handle_out, &h, sizeof(h));
  return CLSR_OK;
}
extern "C" int clsr_comm_ipc_open(const void* handle, void** peer_out) {
  CLSR_CHECK_ARG(handle && peer_out);
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  CLSR_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *peer_out = p;
  return CLSR_OK;
}
extern "C" int clsr_comm_ipc_close(void* peer) {
  if (peer) CLSR_HIP(hipIpcCloseMemHandle(peer));
  return CLSR_OK;
}

// bufs[r] = the exchange buffer of rank r as THIS process addresses it (own allocation at [rank], opened handles elsewhere)
extern "C" int clsr_comm_create(int rank, int world, void* const* bufs, void** comm_out) {
  CLSR_CHECK_ARG(bufs && comm_out && world >= 1 && world <= P2P_MAXW && rank >= 0 && rank < world);
  P2PComm* c = new P2PComm();
  c->rank = rank; c->world = world; c->nch = 0; c->abort_flag = nullptr;
  for (int i = 0; i < P2P_NCH; ++i) { c->seq[i] = 0; c->stream_of[i] = nullptr; }
  for (int r = 0; r < world; ++r) {
    if (!bufs[r]) { delete c; clsr_set_error("%s:%d: exchange buffer of rank %d missing", __FILE__, __LINE__, r); return CLSR_EINVAL; }
    c->bufs[r] = (P2PBuf*)bufs[r];
  }
  *comm_out = c;
  return CLSR_OK;
}
extern "C" int clsr_comm_destroy(void* comm) {
  delete (P2PComm*)comm;
  return CLSR_OK;
}
// Forget which stream owns which channel (the sequence numbers of the channels go on): the next all-reduces assign the
// channels afresh in order of first use.  Every rank must call it at the same point of its call sequence (after a
// self-test on temporary streams: a later stream that happens to reuse a destroyed stream's address on ONE rank only
// would otherwise land on a different channel there).
extern "C" int clsr_comm_reset_channels(void* comm) {
  CLSR_CHECK_ARG(comm);
  P2PComm* c = (P2PComm*)comm;
  CLSR_HIP(hipDeviceSynchronize());
  c->nch = 0;
  for (int i = 0; i < P2P_NCH; ++i) c->stream_of[i] = nullptr;
  return CLSR_OK;
}
// The step's abort flag: a device double (the net's adam_state[4]) that a wait which gives up sets to a non-zero value; NULL
// detaches it.  The optimiser kernels of csrc/optim.hip / csrc/multi.hip do nothing while it is non-zero.
extern "C" int clsr_comm_set_abort(void* comm, double* flag) {
  CLSR_CHECK_ARG(comm);
  ((P2PComm*)comm)->abort_flag = flag;
  return CLSR_OK;
}
// bounded waits of the peer-to-peer kernels, in ticks of the 100 MHz wall clock.  CLSR_P2P_TIMEOUT_S seconds (fractions
// allowed, at least 10 ms) when the variable is set -- honoured from the first call.  Otherwise: 60 s while the job warms up
// (the ranks' first steps drift apart by seconds: allocations, launch-plan recording, lazy module loads -- a slow first
// step on one rank must WAIT, not abort the step), 10 s once the stepper has armed the bound after its first good steps
// (clsr_p2p_arm_timeout, clsr_amd/dp.py).
static int g_p2p_armed = 0;
extern "C" int clsr_p2p_arm_timeout(int armed) {
  __atomic_store_n(&g_p2p_armed, armed ? 1 : 0, __ATOMIC_RELAXED);
  return CLSR_OK;
}
long long clsr_p2p_timeout_ticks(void) {
  static const double fixed = []() {
    const char* e = getenv("CLSR_P2P_TIMEOUT_S");
    if (!e || !*e) return -1.0;
    double sec = atof(e);
    if (!(sec >= 0.01)) sec = 0.01;
    if (sec > 3600.0) sec = 3600.0;
    return sec;
  }();
  const double sec = fixed > 0.0 ? fixed : (__atomic_load_n(&g_p2p_armed, __ATOMIC_RELAXED) ? 10.0 : 60.0);
  return (long long)(sec * 1e8);
}
extern "C" long clsr_p2p_timeout_ms(void) { return (long)(clsr_p2p_timeout_ticks() / 100000); }
// sequence number of the last all-reduce in which this rank gave up waiting for a peer (0: none); synchronises the device
extern "C" long clsr_comm_error(void* comm) {
  if (!comm) return -1;
  P2PComm* c = (P2PComm*)comm;
  unsigned long long e = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(&e, &c->bufs[c->rank]->err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)e;
}

// data[0:n] <- sum over the ranks of data[0:n] (doubles, n <= 256), in place, asynchronous on ``stream``; every rank of
// the communicator must issue the same sequence of calls
extern "C" int clsr_allreduce_small(void* comm, double* data, int n, void* stream) {
  CLSR_CHECK_ARG(comm && data && n > 0);
  CLSR_CHECK_SUPPORTED(n <= P2P_MAXN);
  P2PComm* c = (P2PComm*)comm;
  P2PArgs a;
  for (int r = 0; r < P2P_MAXW; ++r) a.bufs[r] = r < c->world ? c->bufs[r] : nullptr;
  a.x = data; a.n = n; a.rank = c->rank; a.world = c->world;
  int ch = 0;
  while (ch < c->nch && c->stream_of[ch] != stream) ++ch;
  if (ch == c->nch) {
    CLSR_CHECK_SUPPORTED(c->nch < P2P_NCH);      // more streams than channels
    c->stream_of[c->nch++] = stream;
  }
  a.ch = ch;
  a.seq = ++c->seq[ch];
  a.timeout_ticks = clsr_p2p_timeout_ticks();
  a.abort_flag = c->abort_flag;
  hipLaunchKernelGGL(allreduce_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  CLSR_CHECK_LAUNCH();
  return CLSR_OK;
}
