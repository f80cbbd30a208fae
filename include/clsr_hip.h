/* C ABI of libclsr_hip.so -- the MI355X (gfx950) kernels of the CLSR training / scoring step.
 *
 * The reference (tsinghua-fib-lab/CLSR) has no native code and no FFI: its hot path is the single
 * sess.run() inside CLSRModel.train / eval_with_user / infer
 * (reco_utils/recommender/deeprec/models/sequential/clsr.py:383-408,
 *  sequential_base_model.py:294-352).  Every entry point below replaces the TF-1.15 ops that one
 * part of that graph expands to; the reference lines are cited per function.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - all pointers are DEVICE pointers into caller-owned buffers (no ownership transfer, no hidden
 *    allocation); fp32 row-major unless stated; `stream` is a hipStream_t passed as void*;
 *  - every call is asynchronous on `stream`, performs no host synchronisation and is re-entrant;
 *  - return 0 on success, <0 on error (-1 invalid argument, -2 HIP launch failure, -3 unsupported
 *    shape); clsr_last_error() returns the calling thread's last message; nothing throws or exits;
 *  - "history-level" tensors have one row per history group; the G = 1 + train_num_ngs rows of a
 *    training group (row b = h*G + g) share them.  Row-level index / feature arrays are read with a
 *    row stride instead of being de-duplicated on the host (idx_row_stride = G*T, len_stride = G).
 */
#ifndef CLSR_HIP_H
#define CLSR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int clsr_version(void);
int clsr_last_error(char* buf, size_t n);

/* hipGraph capture of everything enqueued on `stream` between begin and end (one replay = one step) */
int clsr_graph_begin(void* stream);
int clsr_graph_end(void* stream, void** graph_exec_out);
int clsr_graph_launch(void* graph_exec, void* stream);
int clsr_graph_destroy(void* graph_exec);

/* ---- embedding lookups: tf.nn.embedding_lookup at sequential_base_model.py:384-437, clsr.py:108-116;
 *      history prologue (concat, hist_mean, hist_recent) clsr.py:145-150,157,173-177 */
int clsr_gather_hist_fwd(const float* item_tbl, const float* cate_tbl, const int* item_idx,
                         const int* cate_idx, long idx_row_stride, const int* seq_len, int len_stride,
                         int Hn, int T, int Di, int Dc, int recent_k, float* hist, float* hist_mean,
                         float* hist_recent, void* stream);
/* bf16 embedding tables (16-byte pieces of eight values; Di, Dc multiples of 8); hist written as bf16 (hist_bf16) or
 * widened to fp32; the masked means are accumulated in fp32 from the widened values */
int clsr_gather_hist_fwd_h(const void* item_tbl, const void* cate_tbl, const int* item_idx,
                           const int* cate_idx, long idx_row_stride, const int* seq_len, int len_stride,
                           int Hn, int T, int Di, int Dc, int recent_k, void* hist, int hist_bf16,
                           float* hist_mean, float* hist_recent, void* stream);
int clsr_gather_hist_bwd(const float* dhist, const float* dmean, const float* drecent,
                         const int* item_idx, const int* cate_idx, long idx_row_stride,
                         const int* seq_len, int len_stride, int Hn, int T, int Di, int Dc, int recent_k,
                         float* item_grad, float* cate_grad, double* sumsq, void* stream);
int clsr_gather_rows(const float* tbl, const int* idx, long idx_stride, int N, int C, float* out,
                     int ldo, int col0, void* stream);
int clsr_scatter_add_rows(const float* src, int ld_src, int col0, const int* idx, long idx_stride, int N,
                          int C, float* tbl_grad, double* sumsq, void* stream);
/* Sorted segmented reduction of the history-lookup gradient: (id, position) pairs grouped by id on the device with a
 * hand-written counting sort (LDS-aggregated histogram, scan, scatter: three launches for all tables of a step), then
 * runs of equal ids summed in registers (dedup of IndexedSlices).  Output: ascending ids when
 * vocab <= 2^18, grouped by (id mod 2^18) beyond; order inside a run of equal ids undefined. */
#define CLSR_SORTIDS_MAX 8
typedef struct clsr_sortids_desc {
  const int* ids; int* keys_out; int* perm_out; int* counts;   /* counts: (1 << bits) ints, ZERO on entry */
  long nrows; long row_stride; int ncols; int bits;
  /* clsr_sort_ids_stable_multi only: a SECOND id source appended to the first -- entries nrows*ncols + r (r < nrows2) are
   * ids2[r * row_stride2] (the target rows' ids behind a history lookup's: one list per embedding table) */
  const int* ids2; long nrows2; long row_stride2;
} clsr_sortids_desc;
int clsr_sizeof_sortids_desc(void);
int clsr_sort_ids_bits(long vocab);
int clsr_sort_ids_multi(const clsr_sortids_desc* descs_host, int n, void* stream);
/* ---- small-message all-reduce over peer-mapped exchange buffers (csrc/p2p.hip): the 2 C doubles of
 *      a synchronised batch-norm layer, summed by ONE single-workgroup kernel on the issuing stream (push into every
 *      peer's buffer, flags with system-scope release / acquire, sum in rank order: bit-identical on every rank).  The
 *      host allocates one exchange buffer per rank (clsr_comm_alloc: fine-grained device memory), exchanges the hipIpc
 *      handles (clsr_comm_ipc_handle / clsr_comm_ipc_open) and creates the communicator from the world pointers; the
 *      communicator is an opaque void* for every later call.  Not capturable into a hipGraph (the sequence number is a
 *      launch argument). */
long clsr_comm_buffer_bytes(void);
int clsr_comm_max_doubles(void);
int clsr_comm_max_world(void);
int clsr_comm_alloc(void** buf_out);
int clsr_comm_free(void* buf);
int clsr_comm_ipc_handle_bytes(void);
int clsr_comm_ipc_handle(void* buf, void* handle_out);
int clsr_comm_ipc_open(const void* handle, void** peer_out);
int clsr_comm_ipc_close(void* peer);
int clsr_comm_create(int rank, int world, void* const* bufs, void** comm_out);
int clsr_comm_destroy(void* comm);
int clsr_comm_reset_channels(void* comm);   /* forget the stream -> channel map (same point on every rank) */
/* device double raised (non-zero) when an all-reduce of this communicator gives up waiting for a peer; NULL detaches */
int clsr_comm_set_abort(void* comm, double* flag);
long clsr_comm_error(void* comm);    /* sequence number of the last all-reduce that timed out waiting for a peer (0: none) */
/* bound of the cross-rank waits (csrc/p2p.hip, csrc/headsfused.hip): CLSR_P2P_TIMEOUT_S when set; otherwise 60 s until the
 * stepper arms the production bound (10 s) after its first good steps -- a slow FIRST step must wait, not abort */
int clsr_p2p_arm_timeout(int armed);
long clsr_p2p_timeout_ms(void);
int clsr_allreduce_small(void* comm, double* data, int n, void* stream);
/* synchronised batch-norm statistics of one layer in ONE launch: the per-block partial sums are folded per feature, the last block
 * of the rank exchanges the 2 C sums with the peers of ``comm`` (the all-reduce above, inside the launch) and finishes the layer --
 * clsr_bn_finalize (training; count = rows of all ranks) / clsr_bn_bwd_coef_scaled (no accumulate).  2 C <= clsr_comm_max_doubles. */
int clsr_bn_finalize_sync(void* comm, const double* stats_partial, int nparts, int C, double count, const float* gamma,
                          const float* beta, float* moving_mean, float* moving_var, float momentum, float eps, float* scale,
                          float* shift, float* mean_out, float* invstd_out, void* stream);
int clsr_bn_bwd_coef_sync(void* comm, const double* partial, int nparts, int C, double count, const float* gamma,
                          const float* mean, const float* invstd, float* coef, float* dgamma, float* dbeta, double grad_scale,
                          void* stream);

/* ---- deterministic embedding gradients (csrc/segsum.hip): STABLE radix sort (ascending ids, equal ids in position order;
 *      desc.bits = significant bits of the ids, desc.counts unused; 1 + 2 * ceil(bits / 8) launches for all tables) and
 *      segmented sums that add every row's total to its gradient table once, in a fixed order -- no float atomics: two
 *      runs of a step give bit-identical tables and clip norms. */
long clsr_sort_ids_stable_workspace_bytes(long total_entries, int n_tables);
int clsr_sort_ids_stable_multi(const clsr_sortids_desc* descs_host, int n, void* workspace, long workspace_bytes,
                               void* stream);
/* one lookup site: grad[key, gcol0 : gcol0 + C] += sum over the sorted entries e with keys[e] == key of
 * src[perm[e], col0 : col0 + C] (+ src2; + dmean[h] / len + recent-k term of the history prologue when given, perm = h*T+t);
 * sumsq += sum of the squared slice values.  src / src2: fp32, or bf16 (src_bf16). */
#define CLSR_SEGSUM_MAX 8
typedef struct clsr_segsum_desc {
  const void* src; const void* src2; const float* dmean; const float* drecent;
  const int* keys; const int* perm; const int* seq_len; float* grad; double* sumsq;
  long n; int src_bf16; int len_stride; int T; int D; int col0; int C; int recent_k; int ldg; int gcol0;
  /* assign != 0: every row total is STORED (the rows are known to be zero and no other site writes the table: no row
   * read).  Second source (n1 > 0): entries with perm >= n1 are rows of src_b -- slice = src_b[perm - n1, colb : colb + C]
   * (fp32, row stride ldb), their squared norms go to sumsq_b: the target rows of a table's second lookup site in the
   * same sorted list as the history lookup's (clsr_sortids_desc.ids2) */
  int assign;
  const float* src_b; double* sumsq_b; long n1; int ldb; int colb;
  /* (rounds 4-5: chunks per wave of the separate border launch; round 6 folded the borders into the one launch -- the field is
   * ignored and stays for the layout of existing callers) */
  int border_wch; int pad_;
} clsr_segsum_desc;
int clsr_sizeof_segsum_desc(void);
long clsr_segsum_workspace_bytes(const clsr_segsum_desc* descs_host, int n);
/* ONE launch.  The sites of one call must write different tables (or disjoint columns); nothing else may write them
 * meanwhile.  workspace: ZERO when first used (it keeps a launch epoch and a finished-workgroup counter between launches: no
 * clearing afterwards), not shared by two launches in flight.  clsr_segsum_error (synchronous): != 0 when a bounded wait
 * for an earlier chunk's partial gave up -- never expected. */
int clsr_segsum_error(const void* workspace);
int clsr_segsum_multi(const clsr_segsum_desc* descs_host, int n, void* workspace, long workspace_bytes, void* stream);
/* Weight gradient of a WIDE layer (K, N >= 96 over M >= 32 768 positions: the 128-wide layer sizes of BASELINE configs[4]):
 * dW[k, n] (=|+=) sum_m X[m, k] (* Xmul[m, k], optional) * dY[m, n], db[n] (=|+=) sum_m dY[m, n] (db may be NULL); 128 x 128 tiles over position
 * ranges, partial tiles summed in range order by a second launch (csrc/dwwide.hip; reference: the kernel gradients of
 * GRUCell / Time4LSTMCell, rnn_cell_implement.py:129-298).  workspace: clsr_pgemm_dw_wide_workspace_floats floats. */
int clsr_pgemm_dw_wide_supported(long M, int K, int N);
int clsr_pgemm_dw_wide_parts(long M, int K, int N);
long clsr_pgemm_dw_wide_workspace_floats(long M, int K, int N);
int clsr_pgemm_dw_wide(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K, int N,
                       float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream);
/* ... with the products as split-bf16 sums on v_mfma_f32_16x16x32_bf16 (2^-16 relative per product term, fp32 accumulation;
 * the fp32 form runs the matrix pipe 70 % busy at 0.56 of its peak, profiles/r05_catalogue_pmc.md) */
int clsr_pgemm_dw_wide_x3(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                          int N, float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream);
/* ... as three-piece sums (2^-23 relative per term: the accuracy of an fp32 product; precision="fp32"). */
int clsr_pgemm_dw_wide_x6(const float* X, int ldx, const float* Xmul, int ldmul, const float* dY, int ldy, long M, int K,
                          int N, float* workspace, float* dW, int ldw, float* db, int accumulate, void* stream);

/* ---- the row-level heads of the CLSR training step as two persistent launches (csrc/headsfused.hip) -------------------
 * Replaces, for the reference's default widths, the chain clsr_alpha_concat -> 2 x (clsr_pgemm + clsr_bn_finalize) ->
 * clsr_mlp_out_fwd -> clsr_alpha_fuse_fwd -> 2 x (clsr_pgemm + clsr_bn_finalize) -> clsr_mlp_tail_softmax ->
 * clsr_bn_bwd_coef_apply / clsr_pgemm_bnbwd / clsr_pgemm ... -> clsr_alpha_fuse_bwd -> ... -> clsr_alpha_concat_bwd
 * (reference: models/sequential/clsr.py:239-275, models/base_model.py:653-708, 215-235 and their gradients): workgroups keep
 * their rows in LDS through every layer and meet at grid barriers for the batch-norm statistics.  The weight gradients of
 * the four dense layers are NOT computed here (the caller's batched clsr_pgemm_dw launch reads ain / al_z0 / mo / lg_z0 and
 * the dz tensors written here); the output-layer gradients leave as [parts][C1 + 4] partial rows (clsr_mlp_out_bwd's
 * layout, parts = clsr_heads_fused_parts).  bn[i]: alpha layer 0, alpha layer 1, logit layer 0, logit layer 1. */
typedef struct clsr_bn_ptrs {
  const float* gamma; const float* beta; float* moving_mean; float* moving_var;
  float* scale; float* shift; float* mean; float* invstd; float* coef; float* dgamma; float* dbeta;
} clsr_bn_ptrs;
typedef struct clsr_heads_desc {
  /* inputs: final state of the causal GRU [B/G, nfs], target embedding [B, D], long-term interest [B/G, D], short-term
   * interest [B, D], time_to_now (row b / tnow_group, column tnow_col, row stride tnow_stride), labels [B] */
  const float* fs; const float* target; const float* att_long; const float* att_short; const float* tnow; const float* labels;
  long tnow_stride; int tnow_col; int tnow_group;
  long B; int G; int D; int nfs; int a_in; int ld; int A0; int A1; int L0; int L1;
  /* packed transposed weights (clsr_pack_batch layout, row strides kp_*): forward and transposed packs */
  const float* al_w0; const float* al_w1; const float* lg_w0; const float* lg_w1;
  const float* al_w0T; const float* al_w1T; const float* lg_w0T; const float* lg_w1T;
  int kp_al_w0; int kp_al_w1; int kp_lg_w0; int kp_lg_w1; int kp_al_w0T; int kp_al_w1T; int kp_lg_w0T; int kp_lg_w1T;
  const float* al_b0; const float* al_b1; const float* al_wout; const float* al_bout;
  const float* lg_b0; const float* lg_b1; const float* lg_wout; const float* lg_bout;
  clsr_bn_ptrs bn[4];
  float momentum; float eps; float lscale; int pad_;
  /* forward outputs */
  float* ain; float* al_z0; float* al_z1; float* alpha; float* mo; float* lg_z0; float* lg_z1; float* logit;
  float* dlogit;            /* optional */
  double* loss;             /* += data loss */
  /* backward outputs (dL / dS / dtarget / dfs are added to) */
  float* lg_dz1; float* lg_dz0; float* dmo; float* al_dz1; float* al_dz0; float* lg_wp; float* al_wp;
  float* dL; float* dS; float* dtarget; float* dfs;
  void* workspace; long workspace_bytes;
  /* data-parallel runs with synchronised batch-norm statistics: a communicator from clsr_heads_comm_create (every rank
   * then issues the same sequence of step1 / step2 calls with the same B); NULL: the statistics of this process's rows */
  void* comm;
  /* the step's abort flag (a device double, the net's adam_state[4]; may be NULL): set non-zero when a grid barrier of a
   * launch gives up (a launch wider than the device can hold at once, a peer rank that never pushed); while it is set the
   * optimiser kernels leave parameters and moments untouched */
  double* abort_flag;
} clsr_heads_desc;
int clsr_sizeof_heads_desc(void);
int clsr_heads_fused_supported(long B, int G, int D, int nfs, int a_in, int A0, int A1, int L0, int L1);
int clsr_heads_fused_parts(long B, int G);
long clsr_heads_fused_workspace_bytes(void);
long clsr_heads_fused_counter_bytes(void);   /* leading bytes of the workspace that must be zero before step1 */
/* communicator for the cross-rank sums: every rank allocates an exchange buffer (clsr_heads_comm_alloc: uncached device
 * memory; handles / mapping / freeing through clsr_comm_ipc_handle, clsr_comm_ipc_open, clsr_comm_ipc_close, clsr_comm_free)
 * and passes all ranks' buffers as it addresses them */
long clsr_heads_comm_buffer_bytes(void);
int clsr_heads_comm_max_world(void);
int clsr_heads_comm_alloc(void** buf_out);
int clsr_heads_comm_create(int rank, int world, void* const* bufs, void** comm_out);
int clsr_heads_comm_destroy(void* comm);
/* start-up check: every rank calls it at the same point; *ok_out (device) = 1 iff all eight stages gave the expected sums */
int clsr_heads_comm_self_test(void* comm, int nblocks, void* workspace, long workspace_bytes, int* ok_out, float timeout_s,
                              void* stream);
int clsr_heads_fused_error(const void* workspace);   /* synchronous; != 0: a grid barrier timed out in some launch since the
                                                       * sticky error words of the workspace were last cleared */
int clsr_heads_fused_clear_error(void* workspace);
int clsr_heads_fused_step1(const clsr_heads_desc* d_host, void* stream);
int clsr_heads_fused_step2(const clsr_heads_desc* d_host, void* stream);

long clsr_sort_ids_workspace_bytes(long n, long vocab);
int clsr_sort_ids(const int* ids, long nrows, int ncols, long row_stride, long vocab, int* keys_out,
                  int* perm_out, void* workspace, long workspace_bytes, void* stream);
/* ...sorted2: g = dhist + dhist2 (+ mean / recent terms): the long-term branch's share of d(hist) is added on the fly
 * instead of by an axpby pass over both tensors (dhist2 may be NULL) */
int clsr_gather_bwd_sorted2(const float* dhist, const float* dhist2, const float* dmean, const float* drecent,
                            const int* keys, const int* perm, const int* seq_len, int len_stride, long n, int T, int D,
                            int col0, int C, int recent_k, float* grad, int ldg, int gcol0, double* sumsq, void* stream);
int clsr_gather_bwd_sorted_max_cols(int D, int col0, int C, int ldg, int gcol0);   /* widest column block per launch */
/* d(hist) (and its optional second addend) as bf16 tensors */
int clsr_gather_bwd_sorted2_h(const void* dhist_bf16, const void* dhist2_bf16, const float* dmean,
                              const float* drecent, const int* keys, const int* perm, const int* seq_len,
                              int len_stride, long n, int T, int D, int col0, int C, int recent_k,
                              float* grad, int ldg, int gcol0, double* sumsq, void* stream);
int clsr_gather_bwd_sorted(const float* dhist, const float* dmean, const float* drecent, const int* keys,
                           const int* perm, const int* seq_len, int len_stride, long n, int T, int D, int col0,
                           int C, int recent_k, float* grad, int ldg, int gcol0, double* sumsq, void* stream);

/* touched-row exchange for multi-GPU runs (new surface; the reference is single-device):
 * byte map of involved rows -> ascending id list (count on the device, bounded by cap), pack / unpack of
 * the listed rows of a dense gradient table.  mode 0 clears the rows, mode 1 adds rows and sets flags. */
long clsr_flags_compact_workspace_bytes(long V);
int clsr_flags_compact(const unsigned char* flags, long V, int* ids_out, int cap, int* count_out,
                       void* workspace, long workspace_bytes, void* stream);
int clsr_flags_compact_off(const unsigned char* flags, long V, int id_offset, int* ids_out, int cap,
                           int* count_out, void* workspace, long workspace_bytes, void* stream);
int clsr_range_offsets(const int* ids, const int* count, const int* bounds, int W, int* offsets, void* stream);
int clsr_rows_pack(const float* table, const int* ids, const int* count, int cap, int C, float* rows_out,
                   void* stream);
int clsr_rows_unpack(const int* ids, const float* rows, const int* count, int cap, int C, int mode,
                     float* table, unsigned char* flags, void* stream);
/* "involved" id sets (tf.unique, sequential_base_model.py:409-433, clsr.py:118-127) as byte maps */
int clsr_mark_rows(const int* idx, long nrows, int ncols, long row_stride, unsigned char* flags,
                   void* stream);

/* ---- linear layers: tf.tensordot / MatMul + BiasAdd (base_model.py:664,704; clsr.py:363;
 *      rnn_cell_implement.py:207-231) and their gradients; fp32 MFMA */
int clsr_pack_weight(const float* src1, int ld1, float s1, const float* src2, int ld2, float s2,
                     int transposed, int O, int I, int Ip, float* Wt, void* stream);
/* Batched block packer / strided block copier: descriptor i writes
 *   dst[(o0+o)*Kp + i0+i] = s1*A(o,i) + s2*B(o,i),  o < O, i < I,  A(o,i) = transposed ? src1[o*ld1+i] : src1[i*ld1+o]
 * (B likewise from src2, optional).  The descriptor table lives in DEVICE memory; one launch packs every weight
 * block of the step (and, with transposed=1, scatters assembled gradient blocks back into the variables). */
typedef struct clsr_pack_desc {
  const float* src1; const float* src2; float* dst;
  float s1; float s2;
  int ld1; int ld2; int transposed; int O; int I; int Kp; int o0; int i0;
} clsr_pack_desc;
int clsr_sizeof_pack_desc(void);
int clsr_pack_batch(const clsr_pack_desc* descs_device, int n, int max_elems, void* stream);
/* ---- bf16 SPEED mode of the attention block (csrc/hgemm.hip; BASELINE.json configs[1] "bf16"): the (row, step)-level
 *      activations of _attention_fcn (clsr.py:343-381, base_model.py:627-708) are stored as bf16 (void* arguments) and
 *      multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; statistics, biases and coefficients stay fp32.
 *      Wt = bf16 image packed by clsr_pack_batch_bf16 (row stride Kp = clsr_hgemm_kp(K) bf16 elements, 32*ceil(N/32)
 *      rows, zero-initialised by the caller). */
int clsr_hgemm_stats_parts(int M);
int clsr_hgemm_kp(int K);
int clsr_pack_batch_bf16(const clsr_pack_desc* descs_device, int n, int max_elems, void* stream);
int clsr_hgemm_mul_uv(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul, const void* Wt,
                      int Kp, const float* addU, int ldu, const float* addV, int ldv, void* Y, int ldy,
                      double* stats, int M, int K, int N, void* stream);
int clsr_hgemm(const void* X, int ldx, const float* in_scale, const float* in_shift, int in_relu,
               const void* Wt, int Kp, const float* bias, void* Y, int ldy, double* stats, int M, int K,
               int N, void* stream);
int clsr_hgemm_f32(const float* X, int ldx, const void* Wt, int Kp, float* Y, int ldy, int accumulate,
                   int M, int K, int N, void* stream);
int clsr_hgemm_att_l1_bwd(const void* z1, int ldz1, const float* ds, const float* scale1,
                          const float* shift1, const float* w_out, const float* coef1, const void* Wt,
                          int Kp, const void* z0, int ldz0, const float* scale0, const float* shift0,
                          const float* mean0, const float* invstd0, const float* coef0, void* dz1,
                          int lddz1, void* dz0, int lddz0, double* stats, int M, int C1, int C0,
                          void* stream);
int clsr_pgemm_dw_partial_h(const void* X, int x_bf16, int ldx, int T, int G, const float* Xmul, int ldmul,
                            const float* in_scale, const float* in_shift, int in_relu,
                            const void* dY, int dy_bf16, int ldy, int M, int K, int N, float* workspace,
                            void* stream);
int clsr_hdw_partial(const void* X, int x_bf16, int ldx, int T, int G, const float* Xmul, int ldmul,
                     const float* in_scale, const float* in_shift, int in_relu, const void* dY,
                     int dy_bf16, int ldy, int M, int K, int N, float* workspace, void* stream);
/* ---- split-bf16 products (csrc/encbwd.hip; round 5: attbwdx3.hip, atthist.hip ...): fp32 operands in HBM,
 *      every value split into bf16 hi + bf16 lo in registers, products taken as hi*hi + lo*hi + hi*lo on
 *      v_mfma_f32_16x16x32_bf16 with fp32 accumulation (<= 2^-16 relative per product).  Same contracts as the fp32-MFMA
 *      entry points they replace. */
int clsr_enc_bwd_fused_x3_parts(long M);
long clsr_enc_bwd_fused_x3_workspace_floats(long M, int p);
int clsr_enc_bwd_fused_x3(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                          const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                          float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                          float* ws_hp2, float* ws_hp2r, long M, void* stream);
/* ... with three bf16 pieces per operand (2^-23 relative: the level of an fp32 product) */
int clsr_enc_bwd_fused_x6(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                          const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                          float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                          float* ws_hp2, float* ws_hp2r, long M, void* stream);
int clsr_att_out_fwd_h(const void* z1, const float* scale1, const float* shift1,
                       const float* w_out, const float* b_out, const int* seq_len,
                       int len_stride, const float* keys, int Hn, int G, int T, int C1, int Dk,
                       float* wts, float* out, void* stream);
int clsr_att_dy1_stats_h(const void* z1, const float* ds, const float* scale1, const float* shift1,
                         const float* mean1, const float* invstd1, const float* w_out, long M, int C1,
                         double* bn_partial, float* w_partial, void* stream);
int clsr_att_z0_bwd_reduce_h(const void* dz0, long Hn, int G, int T, int C, float* dU,
                             float* dV, void* stream);
int clsr_att_prod_bwd_h(const void* daq, int ldd, const float* a, int lda, const float* q, int ldq,
                        long Hn, int G, int T, int Q, float* da, int ldda, float* dq, int lddq,
                        int accumulate_dq, void* stream);
/* All four reductions of the re-associated first attention layer's backward in ONE pass over dz0 (bf16 [R*T, A0]; csrc/hattbwd.hip):
 *   da[h,t,:] = sum_g (dz0[(h,g),t,:] . Wp^T) * q[(h,g),:]     dq[r,:] = sum_t (dz0[r,t,:] . Wp^T) * a[h,t,:]
 *   dU[h,t,:] = sum_g dz0[(h,g),t,:]                           dV[r,:] = sum_t dz0[r,t,:]
 * (reference clsr.py:368-370 under tf.gradients).  Wt = packed bf16 image of Wp^T (clsr_pack_batch_bf16: Q rows, K = A0);
 * Wu (optional) = packed image of Wu^T, same shape and row stride: da additionally receives dU . Wu^T.
 * Replaces clsr_hgemm (daq) + clsr_att_prod_bwd_h + clsr_att_z0_bwd_reduce_h where clsr_att_l0_bwd_h_supported(G, Q, A0). */
int clsr_att_l0_bwd_h_supported(int G, int Q, int A0);
int clsr_att_l0_bwd_h(const void* dz0, int lddz, const void* Wt, const void* Wu, int Kp, const float* a, int lda,
                      const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                      float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, void* stream);
/* Exact-mode backward through the second attention layer + the batch-norm / ReLU below it in TWO passes over (z1, z0)
 * (csrc/attl1bwd.hip; the fp32 twin of clsr_hgemm_att_l1_bwd, same arguments with fp32 tensors): dz1 is recomputed from
 * (z1, ds) in the GEMM prologue; pass 1 (coef0 == NULL) writes the per-block BN-0 backward sums
 * [clsr_att_l1_bwd_stats_parts(M)][2][C0]; pass 2 (coef0 given) writes dz0 and dz1.  Replaces clsr_att_dy1_apply +
 * clsr_pgemm_bnbwd + clsr_bn_bwd_apply where clsr_att_l1_bwd_supported(C1, C0). */
int clsr_att_l1_bwd_supported(int C1, int C0);
int clsr_att_l1_bwd_stats_parts(int M);
int clsr_att_l1_bwd(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                    const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0, int ldz0,
                    const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                    const float* coef0, float* dz1, int lddz1, float* dz0, int lddz0, double* stats, int M,
                    int C1, int C0, void* stream);
/* Re-associated first attention layer, exact mode, one wave per history (csrc/attl0fwd.hip):
 *   z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:] * q[r,:]) . Wp,   stats = per-block partial column sums / sums of squares
 *   [clsr_att_l0_fwd_stats_parts(Hn)][2][A0] doubles (NULL: none).  Wt = packed Wp (clsr_pack_batch: A0 rows, K = Q).
 * Same results as clsr_pgemm(X = a, Xmul = q, addU = U, addV = V) (reference clsr.py:368-370, base_model.py:664-673). */
int clsr_att_l0_fwd_supported(int G, int Q, int A0);
int clsr_att_l0_fwd_stats_parts(long Hn);
int clsr_att_l0_fwd(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                    const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                    long Hn, int G, int T, int Q, int A0, void* stream);
/* History-level prologue of an attention block in one launch of split-bf16 products (csrc/atthist.hip):
 *   a = keys . A (At = packed attention_mat: Q rows, K = Dk),  U = a . Wu (Wut: A0 rows, K = Q)
 *   qh > 0: U += (a[:, :qh] * q_hist[h, :]) . Wp[:qh] (Wpt: A0 rows, K = qh; q_hist [Hn, qh]) -- the history-level share of
 *   the product term of the short-term query.  Replaces clsr_pgemm x 3 (reference clsr.py:351-356, 368-370).
 *   pieces = bf16 pieces per operand: 2 -> products of 2^-16 relative accuracy, 3 -> 2^-23 (the level of an fp32 fma chain:
 *   the parity mode's forward; a 2^-16 perturbation of the pre-activations flips ReLU decisions on small batches). */
int clsr_att_hist_fwd_x3_supported(int Dk, int Q, int A0, int qh);
int clsr_att_hist_fwd_x3(const float* keys, int ldk, const float* At, int Kpa, const float* Wut, int Kpu,
                         const float* Wpt, int Kpp, const float* q_hist, int ldqh, long Hn, int T, int Dk, int Q,
                         int A0, int qh, int pieces, float* a, int lda, float* U, int ldu, void* stream);
/* Plain projection Y[m, :N] = X[m, :K] . W + bias over M positions as split-bf16 products (csrc/projx3.hip; Wt = packed W: N
 * rows, K inputs; bias may be NULL; pieces = 2: 2^-16 relative per product term, 3: 2^-23).  Same result as clsr_pgemm
 * without prologue / epilogue; used for the K-fused time-gate projection of the Time4LSTM in front of the recurrences
 * (reference rnn_cell_implement.py:214-236). */
int clsr_proj_x3_supported(int M, int K, int N);
int clsr_proj_x3(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M, int K,
                 int N, int pieces, void* stream);
/* The Time4LSTM's time-gate projection WITHOUT its [hist | TT] input image: the 2n tanh time features of a position
 * (clsr_t4_time_inputs_fwd's arithmetic: tanh(t_now w1 + b1) | tanh(t_first w2 + b2)) are computed in the product's
 * prologue as columns [col0, col0 + 2n) of the input, columns [0, D) are the history embeddings, the rest zero.
 * K = col0 + 2n <= 128.  (reference rnn_cell_implement.py:200-236) */
int clsr_proj_x3_tt_supported(int M, int D, int col0, int n, int N);
int clsr_proj_x3_tt(const float* hist, int D, const float* tnow, const float* tfirst, long row_stride, int T,
                    const float* w1, const float* b1, const float* w2, const float* b2, int n, int col0,
                    const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M, int N, int pieces,
                    void* stream);
/* ... for K of any width (K % 8 == 0) in ONE launch: the accumulators of a wave's position tiles stay in registers while its
 * workgroup re-stages the weight block of each 128-wide K slab in LDS (csrc/projx3.hip: proj_x3_kloop_kernel);
 * accumulate != 0: Y += X . W + b.  The step routes every PLAIN position-level product of wide layers (K or N > 80:
 * BASELINE configs[4]) through it -- three bf16 pieces per operand in the parity mode (2^-23 relative: fp32 level) */
int clsr_proj_x3_wide_supported(int M, int K, int N);
int clsr_proj_x3_wide(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy, int M,
                      int K, int N, int pieces, int accumulate, void* stream);
/* Back-projections of the encoders' input-side gradients from ONE pass over dPin [M, NX] (split-bf16 products, csrc/projx3.hip):
 *   dhist[m, :D] += dPin[m, :NX] . W_x^T;   dTT[m, :H2] = dPin[m, tcol0 : tcol0 + H3] . W_t^T
 * WxT / WtT = packed transposed weights (D rows, K = NX / H2 rows, K = H3).  Replaces two clsr_pgemm(3) launches that each
 * read dPin (reference: tf.gradients through the input side of the cells, rnn_cell_implement.py:214-236). */
int clsr_enc_back_x3_supported(int M, int NX, int D, int H2, int H3, int tcol0);
int clsr_enc_back_x3(const float* dPin, int ldp, const float* WxT, int Kpx, const float* WtT, int Kpt, int tcol0,
                     float* dhist, int ldh, float* dTT, int ldt, int M, int NX, int D, int H2, int H3, void* stream);
/* Second attention layer, forward: z1 = relu(z0 * scale0 + shift0) . W1 + b1 with the product over three bf16 pieces per
 * operand on the bf16 matrix pipe (2^-23 relative: the level of an fp32 fma chain), stats = per-block partial column sums /
 * sums of squares of z1, [clsr_att_l1_fwd_stats_parts(M)][2][C1] doubles (NULL: none).  Wt = packed W1 (C1 rows, K = C0).
 * Same results as clsr_pgemm with the affine + ReLU prologue (reference base_model.py:664-679). */
int clsr_att_l1_fwd_supported(int C0, int C1);
int clsr_att_l1_fwd_stats_parts(int M);
int clsr_att_l1_fwd(const float* z0, int ldz0, const float* scale0, const float* shift0, const float* Wt, int Kp,
                    const float* bias, float* z1, int ldz1, double* stats, int M, int C0, int C1, void* stream);
/* History-level tail of the attention backward in one launch (csrc/atthist.hip), from one pass over dU [Hn*T, A0]:
 *   da = (columns >= qh: da as written by the layer-0 kernel | columns < qh: (dU . Wp[:qh]^T) * q_hist[h]) + dU . Wu^T
 *   dq_hist[h] += sum_t (dU . Wp[:qh]^T) * a[h,t,:qh];   dkeys += da . A^T
 * WuT / WpT / AT = packed transposed weights (rows = Q / qh / Dk, K = A0 / A0 / Q).  Replaces clsr_pgemm (daq1) +
 * clsr_att_prod_bwd_ld + two accumulating clsr_pgemm (reference: tf.gradients through clsr.py:351-370). */
int clsr_att_hist_bwd_x3_supported(int Dk, int Q, int A0, int qh);
int clsr_att_hist_bwd_x3(const float* dU, int lddu, const float* WuT, int Kpu, const float* WpT, int Kpp,
                         const float* AT, int Kpa, const float* a, int lda, const float* q_hist, int ldqh,
                         long Hn, int T, int Dk, int Q, int A0, int qh, int pieces, float* da, int ldda, float* dq_hist,
                         int lddqh, float* dkeys, int lddk, void* stream);   /* pieces = 2 (2^-16 per term) | 3 (2^-23) */
/* clsr_att_l0_fwd with the product term as split-bf16 sums on the bf16 matrix pipe (fp32 accumulators that start from
 * U + V; 2^-16 relative per product term): bound by its stores instead of the fp32 matrix pipe */
int clsr_att_l0_fwd_x3(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                       const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                       long Hn, int G, int T, int Q, int A0, void* stream);
/* ... over three bf16 pieces per operand (2^-23 relative per product term: fp32 accuracy on the bf16 matrix pipe) */
int clsr_att_l0_fwd_x6(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                       const float* U, int ldu, const float* V, int ldv, float* z0, int ldz, double* stats,
                       long Hn, int G, int T, int Q, int A0, void* stream);
/* The same four reductions in exact mode: dz0 and the packed Wp^T (clsr_pack_batch layout) fp32, fp32 matrix pipe; dU may
 * be NULL (G == 1: dU is dz0).  Replaces clsr_pgemm (daq) + clsr_att_prod_bwd + clsr_att_z0_bwd_reduce. */
int clsr_att_l0_bwd_supported(int G, int Q, int A0);
int clsr_att_l0_bwd(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                    const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                    float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, void* stream);
/* The same pass over dz0 with every product as a split-bf16 product (x.y ~ xh.yh + xh.yl + xl.yh on the bf16 matrix pipe,
 * fp32 accumulation: 2^-16 relative per term) AND the weight gradient of the product term folded in (csrc/attbwdx3.hip):
 *   dWp[c, n] = sum_{r,t} a[h,t,c] q[r,c] dz0[r,t,n]   as clsr_att_l0_bwd_x3_parts(Hn) partial chunks in the layout of
 * clsr_pgemm_dw_partial (summed by clsr_dw_reduce_batch with nparts = that count) -- the separate weight-gradient launch
 * and its re-read of dz0 / a / q are gone (reference: tf.gradients through clsr.py:368-370). */
int clsr_dw_chunk_floats(void);   /* floats per partial chunk of a weight gradient (5 x 5 tiles of 16 x 16 + 5 x 16 bias sums) */
int clsr_att_l0_bwd_x3_supported(int G, int Q, int A0);
int clsr_att_l0_bwd_x3_parts(long Hn);
int clsr_att_l0_bwd_x3(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                       const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                       float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                       void* stream);
/* Second attention layer + the batch-norm / ReLU below it, backward, as split-bf16 products with the weight gradient of
 * the layer folded into pass 2 (csrc/attbwdx3.hip; arguments as clsr_att_l1_bwd):
 *   coef0 == NULL: pass 1, stats = [clsr_att_l1_bwd_x3_parts(M)][2][C0] partial sums of dy0 and dy0 * xhat0;
 *   coef0 given:   pass 2, dz0 = c1*dy0 + c2*z0 + c3 and dw1_partial = clsr_att_l1_bwd_x3_parts(M) partial chunks (layout
 *                  of clsr_pgemm_dw_partial, bias sums included) of dW1 = relu(bn0(z0))^T dz1, db1 = sum dz1 -- dz1 itself
 *                  is not stored.  (reference: tf.gradients through _fcn_net, base_model.py:664-706) */
int clsr_att_l1_bwd_x3_supported(int C1, int C0);
int clsr_att_l1_bwd_x3_parts(int M);
int clsr_att_l1_bwd_x3(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                       const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0, int ldz0,
                       const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                       const float* coef0, float* dz0, int lddz0, float* dw1_partial, double* stats, int M, int C1,
                       int C0, void* stream);
/* ... with THREE bf16 pieces per operand (every piece product whose indices sum to <= 2: 2^-23 relative, the level of an fp32
 * product -- precision="fp32"); same contracts */
int clsr_att_l0_bwd_x6(const float* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                       const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                       float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                       void* stream);
int clsr_att_l1_bwd_x6(const float* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                       const float* w_out, const float* coef1, const float* Wt, int Kp, const float* z0, int ldz0,
                       const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                       const float* coef0, float* dz0, int lddz0, float* dw1_partial, double* stats, int M, int C1,
                       int C0, void* stream);
/* Speed mode (precision = "bf16") on the SAME chain kernels: the (row, step)-level tensors z0 / z1 / dz0 stored as bf16
 * (uint16 bit patterns, row strides in elements, % 8 == 0), ONE bf16 piece per operand on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation, batch-norm sums from the fp32 accumulators, the weight gradients dW1 / db1 / dWp folded in as in the x3
 * forms (fp32 partial chunks).  Replace clsr_hgemm_l0_group / clsr_hgemm / clsr_hgemm_att_l1_bwd / clsr_att_l0_bwd_h and
 * their separate clsr_hdw launches.  History- and row-level operands (a, q, U, V, da, dq, dU, dV), the packed weights
 * (clsr_pack_batch, fp32) and every statistic stay fp32.  Same arguments / *_supported / *_parts as the x3 / x6 entries.
 * (reference: clsr.py:343-381, base_model.py:627-708 -- at bf16 product precision) */
int clsr_att_l0_fwd_x1_h(const float* a, int lda, const float* q, int ldq, const float* Wt, int Kp,
                         const float* U, int ldu, const float* V, int ldv, void* z0, int ldz, double* stats,
                         long Hn, int G, int T, int Q, int A0, void* stream);
int clsr_att_l1_fwd_x1_h(const void* z0, int ldz0, const float* scale0, const float* shift0, const float* Wt, int Kp,
                         const float* bias, void* z1, int ldz1, double* stats, int M, int C0, int C1, void* stream);
int clsr_att_l1_bwd_x1_h(const void* z1, int ldz1, const float* ds, const float* scale1, const float* shift1,
                         const float* w_out, const float* coef1, const float* Wt, int Kp, const void* z0, int ldz0,
                         const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                         const float* coef0, void* dz0, int lddz0, float* dw1_partial, double* stats, int M, int C1,
                         int C0, void* stream);
int clsr_att_l0_bwd_x1_h_parts(long Hn);      /* partial chunks clsr_att_l0_bwd_x1_h writes (its workgroups run two per CU) */
int clsr_att_l0_bwd_x1_h(const void* dz0, int lddz, const float* Wt, int Kp, const float* a, int lda,
                         const float* q, int ldq, long Hn, int G, int T, int Q, int A0, float* da, int ldda,
                         float* dq, int lddq, float* dU, int lddu, float* dV, int lddv, float* dwp_partial,
                         void* stream);
/* clsr_hgemm_mul_uv with one wave per history group (a, U read once per 16 steps of a history and re-used for its G
 * rows; same packed bf16 weights image, same statistics layout with clsr_hgemm_l0_group_stats_parts(Hn) partial rows) */
int clsr_hgemm_l0_group_supported(int G, int Q, int A0);
int clsr_hgemm_l0_group_stats_parts(long Hn);
int clsr_hgemm_l0_group(const float* a, int lda, const float* q, int ldq, const void* Wt, int Kp,
                        const float* U, int ldu, const float* V, int ldv, void* z0, int ldz, double* stats,
                        long Hn, int G, int T, int Q, int A0, void* stream);
/* Y[m, :N] (fp32, =|+=) X[m, :K] . W with a bf16 X (the fp32-X form is clsr_hgemm_f32) */
int clsr_hgemm_hf32(const void* X, int ldx, const void* Wt, int Kp, float* Y, int ldy, int accumulate,
                    int M, int K, int N, void* stream);
int clsr_cvt_f32_to_bf16(const float* src, void* dst, long n, void* stream);
int clsr_cvt_bf16_to_f32(const void* src, float* dst, long n, void* stream);
int clsr_pgemm_stats_parts(int M);
int clsr_pgemm_range(const float* X, int ldx, const float* Wt, int Kp, const float* bias, float* Y, int ldy,
                     int accumulate, int Hn, int T, int t0, int t1, int K, int N, void* stream);
int clsr_pgemm(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
               const float* in_scale, const float* in_shift, int in_relu, const float* Wt, int Kp,
               const float* bias, const float* addU, int ldu, const float* addV, int ldv, float* Y,
               int ldy, int accumulate, double* stats, int M, int K, int N, void* stream);
/* back-propagating product fused with the ReLU + BN backward reduction of the layer below */
int clsr_pgemm_bnbwd(const float* X, int ldx, const float* Wt, int Kp, float* Y, int ldy, const float* z,
                     int ldz, const float* scale, const float* shift, const float* mean, const float* invstd,
                     double* stats, int M, int K, int N, void* stream);
long clsr_pgemm_dw_workspace_floats(int M, int K, int N);
int clsr_pgemm_dw(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                  const float* in_scale, const float* in_shift, int in_relu, const float* dY, int ldy,
                  int M, int K, int N, float scale, float* dW, int ldw, float* db, int accumulate,
                  float* workspace, void* stream);
/* deferred form: partial chunks only (own workspace per call), then ONE batched reduction of every weight
 * gradient of the backward pass.  Descriptor table in DEVICE memory. */
typedef struct clsr_dw_desc {
  const float* partial; float* dW; float* db;
  float scale;
  int nparts; int K; int N; int ldw; int accumulate;
  /* optional second product of the same K x N shape: dW = scale * sum(partial) + scale2 * sum(partial2)
   * (the W0d block of the re-associated attention layer = d(W0a+W0d) - d(W0q-W0d), without a launch of its own) */
  const float* partial2; float scale2; int nparts2;
} clsr_dw_desc;
int clsr_sizeof_dw_desc(void);
int clsr_pgemm_dw_parts(int M);
int clsr_hdw_parts(int M);      /* partial chunks written by clsr_hdw_partial(_multi): <= clsr_pgemm_dw_parts(M), same workspace */
int clsr_pgemm_dw_partial(const float* X, int ldx, int T, int G, const float* Xmul, int ldmul,
                          const float* in_scale, const float* in_shift, int in_relu, const float* dY, int ldy,
                          int M, int K, int N, float* workspace, void* stream);
/* max_outputs / 64 blocks per descriptor; one pass needs 64 * max(4 * ceil(K/16) * ceil(N/16) + ceil(N/64)) (a smaller
 * value is correct too: blocks stride over the remaining work) */
int clsr_dw_reduce_batch(const clsr_dw_desc* descs_device, int n, int max_outputs, void* stream);
int clsr_reduce_parts(const float* partial, int nparts, int stride, int n, float scale, float* out,
                      int accumulate, void* stream);

/* ---- batch normalisation of _fcn_net: tf.layers.batch_normalization(momentum=0.95, eps=1e-4)
 *      base_model.py:673-679 (non-fused: stats over all axes but the last, biased variance) */
int clsr_sum_parts_d(const double* partial, int nparts, int n, double* out, void* stream);
int clsr_bn_finalize(const double* stats_partial, int nparts, int C, double count, const float* gamma,
                     const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                     int training, float* scale, float* shift, float* mean_out, float* invstd_out,
                     void* stream);
int clsr_colred_parts(int M, int C);
int clsr_bn_relu_bwd_reduce(float* dh, const float* z, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int M, int C, double* partial,
                            void* stream);
int clsr_bn_bwd_coef(const double* partial, int nparts, int C, double count, const float* gamma,
                     const float* mean, const float* invstd, float* coef, float* dgamma, float* dbeta,
                     int accumulate, void* stream);
/* ...with dgamma / dbeta multiplied by grad_scale (data-parallel runs with synchronised statistics: 1 / world) */
int clsr_bn_bwd_coef_scaled(const double* partial, int nparts, int C, double count, const float* gamma,
                            const float* mean, const float* invstd, float* coef, float* dgamma, float* dbeta,
                            int accumulate, double grad_scale, void* stream);
int clsr_bn_bwd_apply(float* dy, const float* z, const float* coef, long M, int C, void* stream);
/* clsr_bn_bwd_coef + clsr_bn_bwd_apply in one launch (row-level layers: every block folds the partial sums itself) */
int clsr_bn_bwd_coef_apply(const double* partial, int nparts, int C, double count, const float* gamma,
                           const float* mean, const float* invstd, float* coef, float* dgamma, float* dbeta,
                           float* dy, const float* z, long M, void* stream);

/* ---- attention tail: score layer + padding mask + softmax over T + weighted sum, clsr.py:371-381 */
int clsr_att_out_fwd(const float* z1, const float* scale1, const float* shift1, const float* w_out,
                     const float* b_out, const int* seq_len, int len_stride, const float* keys, int Hn,
                     int G, int T, int C1, int Dk, float* wts, float* out, void* stream);
/* backward of the tail, split by access pattern: score/softmax backward per history group (d score ds[R*T],
 * dkeys +=, partial sums of d b_out), then two streaming passes over z1 that never materialise dy1 =
 * ds * w_out * relu'(y1): column sums for the BN-1 backward + d w_out, and dz1 = a1*dy1 + a2*z1 + a3. */
int clsr_att_score_bwd_parts(int Hn);
int clsr_att_score_bwd(const float* dout, const float* wts, const int* seq_len, int len_stride,
                       const float* keys, int Hn, int G, int T, int Dk, float* ds, float* dkeys,
                       float* b_partial, void* stream);
int clsr_att_dy1_parts(long M, int C1);
int clsr_att_dy1_stats(const float* z1, const float* ds, const float* scale1, const float* shift1,
                       const float* mean1, const float* invstd1, const float* w_out, long M, int C1,
                       double* bn_partial, float* w_partial, void* stream);
int clsr_att_dy1_apply(const float* z1, const float* ds, const float* scale1, const float* shift1,
                       const float* w_out, const float* coef, long M, int C1, float* dz1, void* stream);

/* ---- recurrent encoders under dynamic_rnn: GRUCell clsr.py:160-168,201-208,229-237;
 *      Time4LSTMCell rnn_cell_implement.py:129-298 at clsr.py:179-200 */
int clsr_gru_fwd(const float* Pin, int ldp, const float* Wgh, int ldg, const float* Wch, int ldc,
                 const float* h0, long h0_stride, const int* seq_len, int len_stride, int Hn, int T,
                 int n, float* hT, float* out_seq, float* hprev, float* gates, void* stream);
int clsr_gru_bwd(const float* gates, const float* hprev, const float* Wgh, int ldg, const float* Wch,
                 int ldc, const int* seq_len, int len_stride, int Hn, int T, int n, const float* dhT,
                 const float* dout_seq, float* dPin, float* dh0, void* stream);
int clsr_t4lstm_fwd(const float* Pin, int ldp, const float* Wm, int ldm, const int* seq_len,
                    int len_stride, int Hn, int T, int n, float* out_seq, float* act, float* cst,
                    float* mprev, void* stream);
int clsr_t4lstm_bwd(const float* act, const float* cst, const float* Wm, int ldm, const int* seq_len,
                    int len_stride, int Hn, int T, int n, const float* dout_seq, float* dPin,
                    void* stream);
int clsr_t4_time_inputs_fwd(const float* tnow, const float* tfirst, long row_stride, const float* w1,
                            const float* b1, const float* w2, const float* b2, long Hn, int T, int n,
                            float* TT, void* stream);
/* the same, and (XT != NULL) a second image of every row:  XT[row, 0:D] = hist[row, :],  XT[row, col0:col0+2n] = TT[row, :]
 * (row stride ldxt, the columns between stay as they are: zero) -- the input of ONE product for the three time-gate blocks
 * of the Time4LSTM input projection, [hist | TT] . [W_x ; W_t] instead of hist . W_x followed by += TT . W_t
 * (rnn_cell_implement.py:207-231). */
int clsr_t4_time_inputs_fwd2(const float* tnow, const float* tfirst, long row_stride, const float* w1, const float* b1,
                             const float* w2, const float* b2, long Hn, int T, int n, float* TT, const float* hist, int D,
                             float* XT, int ldxt, int col0, void* stream);
int clsr_t4_time_inputs_bwd_parts(long Hn, int T, int n);
int clsr_t4_time_inputs_bwd(const float* dTT, const float* TT, const float* tnow, const float* tfirst,
                            long row_stride, long Hn, int T, int n, float* partial, void* stream);
/* the same partial sums over the steps [t0, t1) only: clsr_t4_time_inputs_bwd_parts(Hn, t1 - t0, n) partial rows */
int clsr_t4_time_inputs_bwd_range(const float* dTT, const float* TT, const float* tnow, const float* tfirst,
                                  long row_stride, long Hn, int T, int t0, int t1, int n, float* partial, void* stream);

/* Fused launch of up to 3 GRUs + one Time4LSTM over the same histories (one grid, blockIdx.y = encoder).
 * Forward reads Pin, the weights and h0 and writes hT/out_seq (+ hprev/gates | act/cst/mprev when non-null);
 * backward reads the saved activations + dhT/dout_seq and writes dPin (+ dh0). */
typedef struct clsr_gru_desc {
  const float* Pin; const float* Wgh; const float* Wch; const float* h0;
  float* hT; float* out_seq; float* hprev; float* gates;
  const float* dhT; const float* dout_seq; float* dPin; float* dh0;
  long h0_stride; int ldp; int ldg; int ldc; int n; int lddp; int in_div;
  /* attentional update gate (DIEN, VecAttGRUCell rnn_cell_implement.py:594-623): u <- (1 - att[s, t]) * u with one
   * sequence per candidate ROW; sequence s reads Pin / seq_len of history s / in_div.  datt [Hn, T] (backward) is
   * accumulated with atomics: zero it first.  Both NULL / in_div <= 1: the plain GRU. */
  const float* att; float* datt;
  int dpin_bf16;             /* dPin is a bf16 tensor (lddp in elements): clsr_rnn_bwd_multi stores the input-projection
                              * gradients as bf16 for consumers that run on the bf16 matrix pipe (speed mode) */
  int products;              /* hidden-to-hidden products of the launch: 0 = process default (CLSR_RNN_PRODUCTS = fp32 | x3,
                              * x3 when unset), 1 = fp32-input MFMA (bit-exact fp32), 2 = split-bf16: W.h as
                              * Whi.hhi + Whi.hlo + Wlo.hhi on v_mfma_f32_16x16x32_bf16, fp32 accumulation (2^-17 relative).
                              * One form per launch: the last non-zero value among its descriptors decides. */
  /* Fused input projection (forward, split-bf16 launches, n <= 48, Dx % 8 == 0, Dx < 64; every encoder of the launch
   * must carry it): X != NULL -- Pin is not read; the input side of step t is X[h, t, 0:Dx] . [Wgx | Wcx] + [bg | bc]
   * computed in the recurrence (Wgx / Wcx: the x rows of gates/kernel and candidate/kernel, leading dimensions ldg / ldc;
   * reference: GRUCell's single matmul over [inputs, state], clsr.py:160-168). */
  const float* X; const float* Wgx; const float* Wcx; const float* bg; const float* bc;
  int ldx; int Dx;
} clsr_gru_desc;
typedef struct clsr_t4_desc {
  const float* Pin; const float* Wm;
  float* out_seq; float* act; float* cst; float* mprev;
  const float* dout_seq; float* dPin;
  int ldp; int ldm; int n; int lddp;
  int dpin_bf16; int products;   /* as in clsr_gru_desc */
  /* state carried between the launches of a recurrence that runs as a chain of time ranges (clsr_rnn_*_multi_range):
   * [Hn, 2n] rows  c | m  entering t0 / leaving t1 (forward),  dc | dm  entering t1 - 1 / leaving t0 (backward).
   * NULL: zeros in, nothing stored.  (The GRU carries its state through h0 / hT and dhT / dh0.) */
  const float* st_in; float* st_out; const float* dst_in; float* dst_out;
  /* act_tiled != 0: `act` is a private tile-major image of clsr_t4_act_tiled_floats(Hn, T, n) floats that also holds the
   * cell states (cst is ignored): written by the forward and read by the backward recurrence only, every access one
   * contiguous KB per wave instead of 64 pieces of 16 rows.  Split-bf16 launches only (products = 2). */
  int act_tiled; int pad_;
  /* Fused input projection of the blocks i | j | f (as in clsr_gru_desc): X != NULL -- Pin holds only the blocks
   * o | tns | tls (ldp >= 3n: the products that also take the time features); Wkx = the x rows of the lstm kernel
   * (leading dimension ldm), bk = its bias [4n] (Time4LSTMCell.call, rnn_cell_implement.py:229-244). */
  const float* X; const float* Wkx; const float* bk;
  int ldx; int Dx;
} clsr_t4_desc;
long clsr_t4_act_tiled_floats(long Hn, int T, int n);
int clsr_sizeof_gru_desc(void);
int clsr_sizeof_t4_desc(void);
int clsr_rnn_fwd_multi(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4, const int* seq_len,
                       int len_stride, int Hn, int T, void* stream);
int clsr_rnn_bwd_multi(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4, const int* seq_len,
                       int len_stride, int Hn, int T, void* stream);
/* The same recurrences restricted to the steps [t0, t1) (backward: descending from t1 - 1 to t0).  dynamic_rnn
 * (reference call sites clsr.py:161,194,202,210,230) is one T-step loop; run as a chain of ranges -- each launch starting
 * from the state the previous one left -- the results are identical, and the batched input projections of range k + 1 /
 * the weight gradients of range k - 1 run beside the recurrence of range k instead of before / after all of it. */
int clsr_rnn_fwd_multi_range(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4, const int* seq_len,
                             int len_stride, int Hn, int T, int t0, int t1, void* stream);
int clsr_rnn_bwd_multi_range(const clsr_gru_desc* grus, int ngru, const clsr_t4_desc* t4, const int* seq_len,
                             int len_stride, int Hn, int T, int t0, int t1, void* stream);

/* ---- heads: alpha gate + fusion clsr.py:239-275; MLP output layer base_model.py:686-706;
 *      softmax data loss base_model.py:215-235; contrastive loss clsr.py:46-71 */
int clsr_alpha_concat(const float* fs, int nfs, const float* target, const float* L, const float* S,
                      const float* tnow, long tnow_stride, int tnow_col, int tnow_group, long B, int G,
                      int D, float* out, int ldo, void* stream);
int clsr_alpha_concat_bwd(const float* dA, int ldo, int nfs, long Hn, int G, int D, float* dfs,
                          float* dtarget, float* dL, float* dS, void* stream);
int clsr_mlp_out_fwd(const float* z1, const float* scale, const float* shift, const float* w_out,
                     const float* b_out, long B, int C1, float* logit, void* stream);
int clsr_mlp_out_bwd_parts(long B, int C1);
int clsr_mlp_out_bwd(const float* dlogit, const float* z1, const float* scale, const float* shift,
                     const float* mean, const float* invstd, const float* w_out, long B, int C1,
                     float* dy1, double* bn_partial, float* w_partial, void* stream);
int clsr_alpha_fuse_fwd(const float* alpha_logit, float manual_alpha, const float* L, const float* S,
                        const float* target, long B, int G, int D, float* alpha, float* mo, void* stream);
int clsr_alpha_fuse_bwd(const float* dmo, const float* alpha, float manual_alpha, const float* L,
                        const float* S, long Hn, int G, int D, float* dalpha_logit, float* dL, float* dS,
                        float* dtarget, void* stream);
/* output layer of the logit MLP + group-softmax loss + the backward of both in ONE launch (csrc/heads.hip): replaces
 * clsr_mlp_out_fwd -> clsr_softmax_loss -> clsr_mlp_out_bwd of a training step (base_model.py:695-708, :215-235).  P
 * groups of G consecutive rows, z1 [P*G, C1]; partial layouts of clsr_mlp_out_bwd with clsr_mlp_tail_softmax_parts(P)
 * blocks; dlogit may be NULL. */
int clsr_mlp_tail_softmax_supported(int G, int C1);
int clsr_mlp_tail_softmax_parts(long P);
int clsr_mlp_tail_softmax(const float* z1, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* w_out, const float* b_out, const float* labels,
                          long P, int G, int C1, float lscale, double* loss_out, float* logit,
                          float* dlogit, float* dy1, double* bn_partial, float* w_partial, void* stream);
int clsr_softmax_loss(const float* logit, const float* labels, long P, int G, float scale,
                      double* loss_out, float* dlogit, void* stream);
int clsr_contrastive(const float* L, const float* S, const float* M, const float* R, const int* seq_len,
                     int len_stride, long Hn, int G, int D, int threshold, int mode, float margin,
                     float weight, const float* denom_ptr, double* loss_out, float* dL, float* dS,
                     float* dM, float* dR, void* stream);

/* small movers + backward helpers of the re-associated first att_fcn layer
 *   z0[r,t,:] = U[h,t,:] + V[r,:] + (a[h,t,:]*q[r,:]).Wp   ==  [a, q, a-q, a*q].W0 + b0  (clsr.py:368-370) */
int clsr_query_concat(const float* A, int lda, int G, const float* B, int ldb, long R, int Da, int Db,
                      float* out, int ldo, void* stream);
int clsr_query_split_bwd(const float* dq, int ldq, int G, long Hn, int Da, int Db, float* dA, int ldda,
                         float* dB, int lddb, void* stream);
int clsr_copy_cols(const float* src, int ld_src, int src_col0, int row_div, long N, int C, float* dst,
                   int ldd, int dst_col0, int accumulate, void* stream);
int clsr_group_sum_cols(const float* src, int ld_src, int src_col0, int G, long Hn, int C, float* dst,
                        int ldd, int dst_col0, int accumulate, void* stream);
int clsr_axpby(float* out, const float* a, float sa, const float* b, float sb, long n, void* stream);
int clsr_att_z0_bwd_reduce(const float* dz0, long Hn, int G, int T, int C, float* dU, float* dV,
                           void* stream);
int clsr_att_prod_bwd(const float* daq, const float* a, const float* q, long Hn, int G, int T, int Q,
                      float* da, float* dq, void* stream);
/* the same on column blocks of wider tensors (leading dimensions) and with dq += instead of dq =: used when the
 * history-level columns of the query (the short-term intention half of [h_T | target], clsr.py:219) are split
 * off the per-row product term */
int clsr_att_prod_bwd_ld(const float* daq, int ldd, const float* a, int lda, const float* q, int ldq, long Hn,
                         int G, int T, int Q, float* da, int ldda, float* dq, int lddq, int accumulate_dq,
                         void* stream);

/* ---- host side: tokenizer of the sequential TSV files (io/sequential_iterator.py:72-163, parse_file /
 * parser_one_line).  Plain C++: vocabulary look-ups (dict.get(token, 0)) through a hash of the pickled dict's keys,
 * numbers through strtol / strtod, histories flattened with per-line offsets; the time features and the padding
 * are computed from these arrays in numpy.  clsr_host_tsv_count / _parse return 1 for anything irregular (short
 * lines, ragged history columns, number syntax outside [0-9.+-eE]) -- the caller then uses the literal parser. */
void* clsr_host_vocab_create(const char* blob, const long* offsets, const int* ids, long n);
int clsr_host_vocab_destroy(void* vocab);
int clsr_host_tsv_count(const char* buf, long nbytes, long* n_lines, long* n_tokens);
int clsr_host_tsv_parse(const char* buf, long nbytes, const void* user_vocab, const void* item_vocab,
                        const void* cate_vocab, int* labels, int* users, int* items, int* cates, double* cur_time,
                        long* hist_off, int* hist_items, int* hist_cates, double* hist_ts);

/* ---- sibling models of the reference that run on the same kernels (clsr_amd/seqnet.py)
 * SLi-Rec's long-term "A2SVD" attention, models/base_model.py:595-625 (_attention): logits = (x.A).query,
 * softmax over ALL T steps (no mask: padded steps hold embedding row 0 and take part), out = sum_t w[t] x[t];
 * att_inputs = x.A comes from clsr_pgemm.  Backward: d att_inputs (written), d inputs (accumulated), per-block
 * partials of d query [clsr_asvd_att_bwd_parts(Hn)][D].
 * DIN's masked history sum (models/sequential/din.py:27) = hist_mean * len: clsr_scale_rows_by_len. */
int clsr_asvd_att_fwd(const float* att_inputs, const float* query, const float* inputs, long Hn, int T, int D,
                      float* wts, float* out, void* stream);
int clsr_asvd_att_bwd_parts(long Hn);
int clsr_asvd_att_bwd(const float* dout, const float* wts, const float* att_inputs, const float* query,
                      const float* inputs, long Hn, int T, int D, float* d_att_inputs, float* d_inputs,
                      float* dquery_partial, void* stream);
int clsr_scale_rows_by_len(const float* src, const int* seq_len, int len_stride, long Hn, int C, float* out,
                           int accumulate, void* stream);
/* DIEN (models/sequential/dien.py:13-64): the gradient of _attention_fcn arrives through its WEIGHTS
 * (return_alpha=True): masked softmax backward ds[r,t] from dw[r,t] + per-block partials of sum(ds);
 * out[r,:] = a[r,:] * b[r / G,:] for the target * hist_embedding_sum feature (backward: clsr_att_prod_bwd_ld, T = 1).
 * The attentional GRU itself is clsr_rnn_*_multi with clsr_gru_desc.att set. */
int clsr_softmax_weights_bwd_parts(long R);
int clsr_softmax_weights_bwd(const float* dw, const float* wts, const int* seq_len, int len_stride, long Hn, int G,
                             int T, float* ds, float* b_partial, void* stream);
int clsr_mul_rows(const float* a, int lda, const float* b, int ldb, int G, long R, int C, float* out, int ldo,
                  void* stream);

/* ---- regularisers, clip, Adam: base_model.py:118-159,249-297; clsr.py:73-82.  l2 / l1: embed_l2 / embed_l1 for the
 *      involved embedding rows, layer_l2 / layer_l1 for the dense variables (loss += l2/2 ||w||^2 + l1 |w|_1,
 *      grad += l2 w + l1 sign(w)) */
/* Adam state: FIVE device doubles  step | beta1^t | beta2^t | lr_t | abort.  abort != 0 (raised by clsr_allreduce_small /
 * the fused heads launches when a bounded wait gives up, see clsr_comm_set_abort) makes every optimiser kernel below return
 * without touching a parameter or a moment; the host clears it after reporting. */
int clsr_adam_tick(double* state, double lr, double beta1, double beta2, void* stream);
int clsr_dense_reg_norm(const float* param, float* grad, const int* seg_off, int nseg, float l2, float l1,
                        double* sumsq, double* reg_loss, void* stream);
/* the same launch also advances the Adam clock (clsr_adam_tick) when adam_state != NULL: one dispatch less per step;
 * every consumer of the clock is ordered after this launch */
int clsr_dense_reg_norm_tick(const float* param, float* grad, const int* seg_off, int nseg, float l2, float l1,
                             double* sumsq, double* reg_loss, double* adam_state, double lr, double beta1, double beta2,
                             void* stream);
/* ... with the workgroup size given (256 | 512 | 1024; 0 = default): one workgroup walks one tensor */
int clsr_dense_reg_norm_tick_t(const float* param, float* grad, const int* seg_off, int nseg, float l2, float l1,
                               double* sumsq, double* reg_loss, double* adam_state, double lr, double beta1, double beta2,
                               int threads, void* stream);
int clsr_dense_adam(float* param, float* grad, float* m, float* v, const int* seg_of,
                    const double* sumsq, float clip_norm, const double* adam_state, float beta1,
                    float beta2, float eps, int n, void* stream);
int clsr_count_flags(const unsigned char* flags, long V, float* count, void* stream);
/* ...and the Adam clock of the step in the same launch (adam_state may be NULL): the FIRST launch of the update phase,
 * so that the table path (regulariser, Adam) and the dense path read the clock without waiting for each other */
int clsr_count_flags_tick(const unsigned char* flags, long V, float* count, double* adam_state, double lr, double beta1,
                          double beta2, void* stream);
int clsr_table_reg(const float* table, const float* partner, const unsigned char* flags, long V, int C,
                   float l2, float l1, float disc_scale, float disc_loss_scale, const float* count,
                   float* grad_table, double* sumsq, double* reg_loss, double* disc_loss, void* stream);
/* bf16 embedding tables (table / partner: bf16 [V, C]; the `_h` forms of the table kernels: rows widened to fp32,
 * gradients / moments / norms as in the fp32 forms, parameters written back rounded to nearest-even) */
int clsr_table_reg_h(const void* table, const void* partner, const unsigned char* flags,
                     long V, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                     const float* count, float* grad_table, double* sumsq, double* reg_loss,
                     double* disc_loss, void* stream);
int clsr_table_adam(float* table, float* grad_table, float* m, float* v, unsigned char* flags, long V,
                    int C, const double* sumsq, int sumsq_stride, int nsum, float clip_norm,
                    const double* adam_state, float beta1, float beta2, float eps, int lazy, void* stream);
int clsr_table_adam_h(void* table_bf16, float* grad_table, float* m, float* v, unsigned char* flags,
                      long V, int C, const double* sumsq, int sumsq_stride, int nsum,
                      float clip_norm, const double* adam_state, float beta1, float beta2,
                      float eps, int lazy, void* stream);
/* row-list variants for huge vocabularies: ids/count from clsr_flags_compact (involved rows) */
int clsr_table_reg_rows(const float* table, const float* partner, const int* ids, const int* count, int cap,
                        int C, float l2, float l1, float disc_scale, float disc_loss_scale, const float* ucount,
                        float* grad_table, double* sumsq, double* reg_loss, double* disc_loss, void* stream);
int clsr_table_reg_rows_h(const void* table, const void* partner, const int* ids, const int* count,
                          int cap, int C, float l2, float l1, float disc_scale, float disc_loss_scale,
                          const float* ucount, float* grad_table, double* sumsq, double* reg_loss,
                          double* disc_loss, void* stream);
int clsr_table_adam_rows(float* table, float* grad_table, float* m, float* v, unsigned char* flags,
                         const int* ids, const int* count, int cap, int C, const double* sumsq,
                         int sumsq_stride, int nsum, float clip_norm, const double* adam_state, float beta1,
                         float beta2, float eps, void* stream);
/* the same for a table stored as bf16: widened, updated in fp32 (fp32 moments / gradients), rounded to nearest-even */
int clsr_table_adam_rows_h(void* table_bf16, float* grad_table, float* m, float* v, unsigned char* flags,
                           const int* ids, const int* count, int cap, int C, const double* sumsq,
                           int sumsq_stride, int nsum, float clip_norm, const double* adam_state,
                           float beta1, float beta2, float eps, void* stream);
int clsr_zero_doubles(double* p, int n, void* stream);
int clsr_add_doubles(double* dst, const double* src, int n, void* stream);
int clsr_zero_floats(float* p, long n, void* stream);
/* feed upload: device arena <- pinned host arena read by the kernel over PCIe (replaces the
   feed_dict host->device copies of session.run, base_model.py:345-357) */
int clsr_stage_feed(void* dst, const void* src_host, long nbytes, void* stream);
/* measurement probe (bench.py): device-to-device copy in 16-byte words, non-temporal, 4 loads in flight per lane */
int clsr_copy_words(void* dst, const void* src, long nbytes, void* stream);

/* ---- multi-launch forms of the small kernels: up to CLSR_MULTI_MAX independent jobs in ONE launch
 *      (blockIdx.y = job).  The descriptor array is read on the HOST and passed to the kernel by value, so these
 *      calls are hipGraph-capturable like any other launch.  Every tiny dependent launch costs ~5 us of device
 *      time on its own; the step had ~75 of them. */
#define CLSR_MULTI_MAX 16
typedef struct clsr_mark_desc { const int* idx; unsigned char* flags; long nrows; long row_stride; int ncols; int pad_; } clsr_mark_desc;
typedef struct clsr_gather_desc { const float* table; const int* idx; float* out; long idx_stride; int N; int C; int ldo; int col0; } clsr_gather_desc;
/* ---- fused tail of the backward pass through the encoders of the default graph (csrc/encbwd.hip): from ONE pass over
 * dPin [M, 480] (M = Hn * T; columns as net.py lays out the fused input projection: short_term_intention r,u | c @0,
 * causal2 r,u | c @120, time4lstm i,j,f,o | tns | tls @240) the partial sums of seven weight gradients -- hist^T dPin
 * (+ the bias sums of all 480 columns), hprev1^T dPin[:, 0:80], (hprev1 * r1)^T dPin[:, 80:120], hprev2^T dPin[:, 120:200],
 * (hprev2 * r2)^T dPin[:, 200:240], mprev^T dPin[:, 240:400], TT^T dPin[:, 360:480]; gates1 / gates2 are the saved
 * r | u | c activations [M, 120] -- and  dhist += dPin . W_x^T  (Wt: packed W_x^T, 40 rows, K = 480).
 * Reference: the gradients of GRUCell / Time4LSTMCell.call under dynamic_rnn, clsr.py:160-237,
 * rnn_cell_implement.py:129-298.  Every workspace ws_* receives clsr_enc_bwd_fused_parts(M) partial slots per 80-column
 * chunk in the layout of clsr_pgemm_dw_partial (bias sums of dPin ride in ws_hist), for clsr_dw_reduce_batch.
 * Replaces clsr_pgemm_dw_partial_multi (seven jobs) + clsr_pgemm(dPin, W_x^T) at the default widths (D = n = 40);
 * clsr_enc_bwd_fused_supported says whether a shape is covered, everything else keeps those two. */
int clsr_enc_bwd_fused_supported(int D, int n, int NX);
int clsr_enc_bwd_fused_parts(long M);
long clsr_enc_bwd_fused_workspace_floats(long M, int product);
/* speed mode: the seven weight gradients of the same tail from a bf16 dPin [M, 480] on the bf16 matrix pipe (left operands
 * fp32 in memory, rounded to bf16 when staged; fp32 accumulation; same column layout and partial layout;
 * clsr_enc_bwd_fused_h_parts(M) partial slots).  With Wt_bf16 (the clsr_pack_batch_bf16 image of W_x^T, row stride Kph)
 * it also accumulates dhist += dPin . W_x^T; NULL: weight gradients only.  Replaces the seven jobs of
 * clsr_hdw_partial_multi (and clsr_hgemm_hf32 for d(hist)). */
int clsr_enc_bwd_fused_h_parts(long M);
long clsr_enc_bwd_fused_h_workspace_floats(long M, int product);
int clsr_enc_bwd_fused_h(const void* dPin_bf16, const float* hist, const float* hprev1, const float* gates1,
                         const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                         float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt,
                         float* ws_hp2, float* ws_hp2r, const void* Wt_bf16, int Kph, float* dhist, long M, void* stream);
int clsr_enc_bwd_fused(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                       const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                       const float* Wt, int Kp, float* dhist, float* ws_hist, float* ws_hp1, float* ws_hp1r,
                       float* ws_mprev, float* ws_tt, float* ws_hp2, float* ws_hp2r, long M, void* stream);
/* ... and d(hist)[m, :] += dhist2[m, :] + the shares of the history prologue (clsr.py:145-150,157,173-177) of row m = (h, t):
 * dmean[h, :] / len_h when t < len_h, drecent[h, :] / min(len_h, recent_k) when len_h - recent_k <= t < len_h (dmean, drecent
 * may be NULL; len_h = seq_len[h * len_stride], M = Hn * T) -- the terms the segmented sums of the history lookups
 * (clsr_segsum_multi: src2, dmean, drecent) otherwise add per sorted entry; with them folded in here those run lean. */
int clsr_enc_bwd_fused_fold(const float* dPin, const float* hist, const float* hprev1, const float* gates1,
                            const float* mprev, const float* TT, const float* hprev2, const float* gates2,
                            const float* Wt, int Kp, float* dhist, const float* dhist2, const float* dmean,
                            const float* drecent, const int* seq_len, int len_stride, int T, int recent_k,
                            float* ws_hist, float* ws_hp1, float* ws_hp1r, float* ws_mprev, float* ws_tt, float* ws_hp2,
                            float* ws_hp2r, long M, void* stream);

/* one weight-gradient product of a multi-job launch (clsr_pgemm_dw_partial_multi / clsr_hdw_partial_multi): the arguments
 * of clsr_pgemm_dw_partial / clsr_hdw_partial */
typedef struct clsr_dwjob {
  const void* X; const float* Xmul; const float* in_scale; const float* in_shift; const void* dY; float* workspace;
  int x_bf16; int ldx; int T; int G; int ldmul; int in_relu; int dy_bf16; int ldy; int M; int K; int N; int pad_;
  /* time-range form (all zero: the plain product).  M = Hn * rm_tc virtual rows: row v is the physical row
   * (v / rm_tc) * rm_T + rm_t0 + v % rm_tc of X and dY (the steps [rm_t0, rm_t0 + rm_tc) of every history).  The job uses
   * pgx blocks along the rows (0: clsr_pgemm_dw_parts(M)) and writes the partial slots [poff, poff + pgx) of the pstride
   * slots per (K, N) chunk of its workspace: the launches over the ranges of one product fill one workspace, which ONE
   * clsr_dw_reduce_batch descriptor with nparts = pstride then sums. */
  int rm_tc; int rm_T; int rm_t0; int pgx; int pstride; int poff;
} clsr_dwjob;
int clsr_sizeof_dwjob(void);
int clsr_pgemm_dw_partial_multi(const clsr_dwjob* jobs_host, int n, void* stream);
int clsr_hdw_partial_multi(const clsr_dwjob* jobs_host, int n, void* stream);
typedef struct clsr_rp_desc { const float* partial; float* out; float scale; int nparts; int stride; int n; int accumulate; int pad_; } clsr_rp_desc;
typedef struct clsr_table_desc {
  float* table; const float* partner; float* grad; float* m; float* v; unsigned char* flags;
  double* sumsq_reg; double* disc_loss; const double* sumsq_adam;
  long V; int C; int nsum; int sumsq_stride; float disc_scale; float disc_loss_scale; int pad_;
} clsr_table_desc;
/* zero fills of several byte ranges (4-byte aligned, multiples of 4 bytes) in one launch */
typedef struct clsr_zero_desc { void* p; long nbytes; } clsr_zero_desc;
int clsr_sizeof_zero_desc(void);
int clsr_zero_multi(const clsr_zero_desc* descs_host, int n, void* stream);
/* clsr_scatter_add_rows for several lookup sites in one launch (sumsq may be NULL) */
typedef struct clsr_scatter_desc {
  const float* src; const int* idx; float* tbl_grad; double* sumsq; long idx_stride; int ld_src; int col0; int N; int C;
} clsr_scatter_desc;
int clsr_sizeof_scatter_desc(void);
int clsr_scatter_add_rows_multi(const clsr_scatter_desc* descs_host, int n, void* stream);
int clsr_sizeof_multi_descs(int* mark, int* gather, int* rp, int* table);
int clsr_mark_rows_multi(const clsr_mark_desc* descs_host, int n, void* stream);
int clsr_gather_rows_multi(const clsr_gather_desc* descs_host, int n, void* stream);
int clsr_gather_rows_multi_h(const clsr_gather_desc* descs_host, int n, void* stream);   /* bf16 tables, fp32 outputs */
int clsr_reduce_parts_multi(const clsr_rp_desc* descs_host, int n, void* stream);
int clsr_tables_reg_multi(const clsr_table_desc* descs_host, int n, float l2, float l1, const float* ucount,
                          double* reg_loss, void* stream);
int clsr_tables_reg_multi_h(const clsr_table_desc* descs_host, int n, float l2, float l1, const float* ucount,
                            double* reg_loss, void* stream);
int clsr_tables_adam_multi(const clsr_table_desc* descs_host, int n, float clip_norm, const double* adam_state,
                           float beta1, float beta2, float eps, int lazy, void* stream);
int clsr_tables_adam_multi_h(const clsr_table_desc* descs_host, int n, float clip_norm, const double* adam_state,
                             float beta1, float beta2, float eps, int lazy, void* stream);

/* ---- evaluation metrics on the device (csrc/metrics.hip): cal_metric / cal_weighted_metric of
 *      deeprec_utils.py:554-821 as SequentialBaseModel.run_eval / run_weighted_eval use them
 *      (sequential_base_model.py:204-292), by exact pair / rank COUNTING (no sort).  The double accumulators (out) must be
 *      ZERO on entry and hold the final sums on return: block partials are added as 64-bit fixed-point integers, so the
 *      result does not depend on the order in which blocks finish (bit-identical from run to run).  Rank ties inside a group: the later line ranks first (stable ascending sort read backwards). */
int clsr_eval_logloss(const float* pred, const float* labels, long N, double* out, void* stream);
int clsr_eval_compact_pos(const float* pred, const float* labels, long N, float* pos_out, int* count, void* stream);
int clsr_eval_auc_pairs(const float* pred, const float* labels, long N, const float* pos, const int* count,
                        void* out_u64x3, void* stream);
int clsr_eval_group_metrics(const float* pred, const float* labels, long n_groups, int G, const int* ks_host, int nk,
                            int want_auc, double* out, int* err, void* stream);
int clsr_eval_user_auc(const float* pred, const float* labels, const int* perm, const int* ends, int nb, long N,
                       double* out, int* err, void* stream);

/* ---- host-side (no GPU) replay of CPython's random module for the input pipeline: continue the MT19937 stream of
 *      random.getstate() through random.shuffle / the in-batch negative sampling of io/sequential_iterator.py:249-261,
 *      612-634, bit-identically, and hand the advanced state back (key[624], *pos). */
int clsr_host_mt_shuffle(unsigned* key, int* pos, long* perm, long n);
int clsr_host_mt_sample_negatives(unsigned* key, int* pos, const long* items, long n, int ngs, long* src);

#ifdef __cplusplus
}
#endif
#endif /* CLSR_HIP_H */
