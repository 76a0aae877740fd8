/* r3d.h -- C ABI of libr3d_hip.so, the MI355X (gfx950) hot path of R3DFSSeg.
 *
 * The reference (Pixie8888/R3DFSSeg) is pure Python/PyTorch and has no FFI of its own;
 * each entry point below names the reference code it replaces (file:line under
 * /root/reference).  The binding a maintainer of the reference would add is a ctypes
 * stub, shown in INTEGRATION.md (r3dfsseg_amd/_lib.py is that stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch caching allocator in practice), borrowed for the duration of the call;
 *   - kernels never allocate; scratch is passed in (sizes from the *_ws_words helpers); the entry points whose scratch is
 *     carved into many arrays (prototypes, label propagation, contrastive loss, reverse neighbour list) also take its
 *     size in words and return an error for a short buffer instead of writing past it;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     immediately: 0 = OK, non-zero = error, message in r3d_last_error_string();
 *   - no host synchronisation anywhere: data-dependent counts (points per class,
 *     prototypes, graph nodes) stay in a device-side descriptor;
 *   - activations are POINT-MAJOR fp32 matrices (one point per row, `ld*` = row stride in
 *     floats); the reference's channel-major (B, C, N) tensors are converted at the
 *     forward() boundary by r3d_cm_to_pm / r3d_pm_to_cm;
 *   - indices are int32;
 *   - BATCHES OF EPISODES (ABI version 3).  Episodes are independent units (the reference runs one per step,
 *     mpti_train_noise.py:57-98); here E of them go through ONE launch sequence.  Encoder side: the clouds of the batch
 *     are rows of one matrix, episode after episode, [S support clouds | Q query clouds] each; BatchNorm keeps the
 *     statistics of every getFeatures call apart (models/mpti.py:434,436), so the `_seg` entry points take the two
 *     alternating segment sizes (rows_a = S N, rows_b = Q N, or in clouds) -- segment 2 e + p is call p of episode e,
 *     rows_b == 0 means equal segments -- and address the BatchNorm vectors of segment s at (pointer + s * bn_stride).
 *     Head side: the `_batched` entry points take n_ep and the stride of every per-episode array (capacity sized).
 *     A segment's / an episode's results do not depend on the batch it runs in: reductions are partitioned by the
 *     segment's own size and summed relative to its first element.
 */
#ifndef R3D_H
#define R3D_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* r3d_last_error_string(void);
int r3d_abi_version(void); /* 4 (round 4: r3d_edge_stats1 + esum, r3d_edgeconv_bwd + zwin, esum) */
/* Arithmetic of the GEMM-shaped kernels that decide no index (self-attention forward / backward, r3d_pointwise_conv*,
 * r3d_gemm_tn): 0 = fp32 matrix core (v_mfma_f32_32x32x2_f32), 1 = every fp32 operand cut into three bf16 pieces, six
 * v_mfma_f32_32x32x16_bf16 per product block accumulated in fp32 (fp32-level accuracy at 2.67x the matrix rate;
 * csrc/common.h, csrc/gemm_bx3.hip).  kNN scores and the EdgeConv edge GEMM, which decide indices, always run on the
 * fp32 core.  Default 1.  Process-wide; call before the first launch.  The attention entry points use mode 1 only when
 * they are given a workspace (the packed operands live there); the GEMMs fall back to the fp32 kernels for shapes the
 * bf16 form does not take (K % 32 != 0, fewer than 32 output columns, operands not 16-byte aligned). */
/* test utility: fills the chip's LDS with `pattern` (no result of this library may depend on stale LDS contents) */
int r3d_debug_poison_lds(unsigned pattern, unsigned* sink /* 1 device word */, void* stream);
/* test / A-B utility: the CG's SpMV runs on its LDS-resident form when the launch holds at least min_blocks 128-row
 * workgroups (default 256; 0 = always, INT_MAX = never); both forms give the same bits.  Returns the previous value. */
int r3d_debug_set_cg_spmv_lds_min_blocks(int min_blocks);
int r3d_set_matrix_arith(int mode);
int r3d_get_matrix_arith(void);
/* A/B knob under mode 1: bit 0 the point-wise GEMM takes the bf16 form, bit 1 the weight-gradient GEMM does (default 3). */
int r3d_debug_set_gemm_bx3(int mask);

/* ---- layout conversion at the forward() boundary (models/mpti.py:433-437) -------- */
int r3d_cm_to_pm(const float* in /*(B,C,N)*/, int B, int C, int N, float* out /*(B*N,ld)*/, long ld, void* stream);
int r3d_pm_to_cm(const float* in /*(B*N,ld)*/, long ld, int B, int C, int N, float* out /*(B,C,N)*/, void* stream);
int r3d_pm_to_cm_pitched(const float* in, long ld, int B, int C, int N, float* out /*(B,C,pitch)*/, long pitch, void* stream);
long r3d_cm_pitch(int N); /* row pitch (floats) of internal channel-major copies: avoids power-of-two channel strides */
int r3d_copy_cols(const float* src, long ld_src, float* dst, long ld_dst, long M, int C, void* stream);

/* ---- k nearest neighbours ----------------------------------------------------------
 * mode 0: models/dgcnn.py:17-23 knn(x,k): score = -xx[j] + 2<xi,xj> - xx[i], k largest.
 * mode 1: models/mpti.py:733-735 faiss.IndexFlatL2.search: score = -max(0,|xi|^2+|xj|^2-2<xi,xj>).
 * Inner products are channel-ascending fp32 fma chains (bit-exact vs oracle/r3d_oracle.c);
 * ties resolve to the lower index; columns are sorted best first.
 * x (B*N, ldx); norm_ws (B*N) scratch; idx_out (B,N,k) int32; score_out optional (B,N,k);
 * n_valid_dev optional device int: only rows < *n_valid_dev take part. 1 <= k <= min(N,256).
 * x_cm: optional (B,C,N) channel-major copy (the reference's own tensor layout); the streamed
 * kernel (k <= 32, C <= 64) reads its operands from it and makes the copy into cm_ws (B*C*N
 * floats) when x_cm is NULL.
 * status: optional device int32.  For k > 32 a non-NULL status selects the two-pass
 * append-and-rank kernel; bit 0 set afterwards means its survivor buffer overflowed and the
 * call must be repeated with status == NULL (insertion kernel, always exact). */
int r3d_sqnorm(const float* x, long ldx, long rows, int C, float* out, void* stream);
long r3d_knn_norm_ws_words(int B, int N);  /* floats of norm_ws: B*N norms + per-tile overflow flags */
int r3d_knn_topk(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                 const int32_t* n_valid_dev, float* norm_ws, float* cm_ws, int32_t* idx_out, float* score_out,
                 int32_t* status, void* stream);
/* the same with a scratch of r3d_knn_split_ws_words(B, N, k) floats: for k > 32 with status != NULL and so few query
 * tiles that even twice as many workgroups fit the chip in one round (2 B ceil(N/32) <= 256), the candidate axis is
 * dealt to two workgroups per tile and their sorted lists are merged -- same result, bit for bit */
long r3d_knn_split_ws_words(int B, int N, int k);
int r3d_knn_topk_split(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                       const int32_t* n_valid_dev, float* norm_ws, float* cm_ws, int32_t* idx_out, float* score_out,
                       int32_t* status, float* split_ws, long split_ws_words, void* stream);

/* B point sets with their own valid counts: set b has n_valid_dev[b * n_valid_stride] rows (the graph nodes of B episodes'
 * label-propagation systems, each at its capacity N).  status: ONE word for the batch. */
int r3d_knn_topk_batched(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                         const int32_t* n_valid_dev, int n_valid_stride, float* norm_ws, float* cm_ws, int32_t* idx_out,
                         float* score_out, int32_t* status, float* split_ws, long split_ws_words, float* bf_ws,
                         long bf_ws_words, void* stream);
/* bf_ws: optional scratch of r3d_knn_bf_ws_words(B, N, C) floats (needs x and C % 64 == 0): the streamed kernels then run
 * their THRESHOLD pass -- which only needs a lower bound of every score -- on the bf16 matrix core, and (x 16-byte aligned,
 * ldx % 4 == 0) their second pass as a bf16 FILTER: a candidate whose score's upper bound reaches the threshold is kept,
 * and only the kept ones (~k + 10 per query) get their exact score -- the fp32 fmaf chain in channel order, bit for bit
 * the accumulation of the all-pairs fp32 pass it replaces.  Indices and scores are bit-identical with and without bf_ws. */
long r3d_knn_bf_ws_words(int B, int N, int C);
/* test / A-B utility: 0 keeps the threshold pass on the fp32 core even when bf_ws is given (same results).  Returns the
 * previous setting. */
int r3d_debug_set_knn_bf16_threshold(int on);
/* the same for the second pass: 0 keeps it the all-pairs fp32 pass (same results) */
int r3d_debug_set_knn_bf16_filter(int on);
/* 1: launches captured into a hipGraph may use their stream's packed-weight scratch of the bf16 x 3 point-wise GEMM (it
 * must exist already: run the sequence eagerly on that stream first).  The caller promises that the captured graph is the
 * only user of that stream's scratch while it replays.  Default 0: captured launches take the kernel that cuts W itself
 * (same bits).  Returns the previous setting. */
int r3d_set_wpack_in_capture(int on);

/* ---- 1x1 convolution + folded BatchNorm/bias + activation ---------------------------
 * models/dgcnn.py:64-80 conv1d, models/mpti.py:18-40 BaseLearner, models/attention.py:39-41.
 * Out[m][j] = act(scale[j] * sum_k X[m][k] W[j][k] + shift[j]); act 0 none, 1 ReLU, 2 LeakyReLU(0.2).
 * scale/shift may be NULL (1 / 0). */
int r3d_pointwise_conv(const float* X, long ldx, const float* W /*(Co,K)*/, long M, int K, int Co,
                       const float* scale, const float* shift, int act, float* Out, long ldo, void* stream);

/* ---- fused EdgeConv: gather + 2-layer edge MLP + max over K --------------------------
 * models/dgcnn.py:26-61,117-118.  PQ (B*N,128) = [s1*Wa x | s1*(Wb-Wa) x + t1] per point
 * (from r3d_pointwise_conv), idx (B,N,K) local neighbour ids, W2 (64,64), s2/t2 (64) folded BN2.
 * out (B*N, ldo) 64 columns; argmax_out optional (B*N,64) winning neighbour slot.
 * Shapes: N a multiple of 4, K in {4, 8, ..., 32}; neighbour ids outside [0, N) are clamped, never dereferenced. */
int r3d_edgeconv_fwd(const float* PQ, const int32_t* idx, const float* W2, const float* s2, const float* t2,
                     float* out, long ldo, int B, int N, int K, int32_t* argmax_out, void* stream);

/* ---- point self-attention, d = 64 (models/attention.py:43-46) ------------------------
 * qkv (B*N, ld): q*(1/8) | k | v at columns 0 | 64 | 128.  out (B*N, ldo) 64 columns.
 * lse_out optional (B*N). */
/* ws (optional, r3d_attention_ws_words(B, N) floats): enables the streamed-axis split -- small grids (B*N/128
 * workgroups) are cut along the key axis so that ~512 workgroups exist, partials merged in a fixed order */
long r3d_attention_ws_words(int B, int N);
int r3d_attention_fwd(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out, float* ws,
                      void* stream);

/* ---- multi-prototype extraction (models/mpti.py:597-715) -----------------------------
 * FPS (start index 0, ties lowest index) -> sorted unique seeds -> nearest-seed assignment ->
 * cluster means, for background + each way, then query rows appended: fills node rows
 * [0, n_proto) and [n_proto, n_proto + n_query_pts) of `nodes` and the label matrix Y.
 * desc: device int32[r3d_head_desc_words()] = {seg_count[8], seg_m[8], seg_poff[8], n_proto, n_nodes,..}.
 * Seeds per segment of n > k points: what torch_cluster.fps(feat, None, ratio = k / n) draws at the call site
 * models/mpti.py:612-613, ceil(float32(n) * float32(k / n)) = k or k + 1 (101 for 5.8 % of the n <= 20480 at k = 100),
 * so a segment can hold k + 1 prototypes: nodes / node_labels need (n_way + 1) * (k + 1) + n_query_pts rows, k < r3d_head_max_k(). */
int r3d_head_desc_words(void);
int r3d_head_max_k(void);
/* out[n], n in [0, n_max): the seeds a segment of n points gets at k (n <= k: n), as the device evaluates the count above
 * (device int32 out; for tests that hold it to the host arithmetic over every n). */
int r3d_fps_sample_count_table(int k, int n_max, int32_t* out, void* stream);
long r3d_head_proto_ws_words(int n_way, int k_shot, int N);
int r3d_head_proto_ws_offsets(int n_way, int k_shot, int N, long* out6 /* comp,mind,assign,cand,sel,seeds */);
/* flags: R3D_HEAD_FPS_ONE_LAUNCH = all FPS rounds in one persistent launch (points stay in registers; needs the
 * grid co-resident: episodes in flight x support points / 256 <= ~384 workgroups; desc word 26 (HD_FPS_TIMEOUT) reports
 * a wait time-out); 0 = one launch per round. */
#define R3D_HEAD_FPS_ONE_LAUNCH 1
int r3d_head_prototypes(const int32_t* support_y /*(n_way*k_shot,N)*/, const int32_t* shot_keep /*opt (n_way*k_shot)*/,
                        const float* feat /*(S*N,ldf)*/, long ldf, const float* featT /*(S,D,N)*/,
                        const float* qfeat /*(n_q*N,ldq)*/, long ldq, int n_way, int k_shot, int N, int D,
                        int n_query_pts, int k, float* nodes, long ldn, float* node_labels /*(n_cap,4)*/,
                        int32_t* desc, int32_t* assign_out /*opt (2*S*N)*/, int32_t* cluster_count /*opt (n_cap)*/,
                        int32_t* ws, long ws_words, int flags, void* stream);

/* n_ep episodes in one launch sequence (every pointer addresses episode 0).  Strides between consecutive episodes:
 * support_y / shot_keep / desc / assign / cluster_count / ws in int32 words (ws_stride even, >= the scratch size), feat /
 * qfeat / nodes in ROWS.  fps_group: episodes whose farthest-point samplings share one persistent launch (their workgroups
 * must be co-resident: ~500 workgroup slots at D <= 192, 250 above; an episode needs ceil(S N / 256) + n_way + 1). */
int r3d_head_prototypes_batched(int n_ep, int fps_group, const int32_t* support_y, long sy_stride, const int32_t* shot_keep,
                                long keep_stride, const float* feat, long ldf, long feat_ep_rows, const float* qfeat, long ldq,
                                long qfeat_ep_rows, int n_way, int k_shot, int N, int D, int n_query_pts, int k, float* nodes,
                                long ldn, long nodes_ep_rows, float* node_labels, int32_t* desc, long desc_stride,
                                int32_t* assign_out, long assign_stride, int32_t* cluster_count, long ccount_stride,
                                int32_t* ws, long ws_words, long ws_stride, int flags, void* stream);

/* ---- affinity + label propagation (models/mpti.py:717-776) ---------------------------
 * nbr (n_cap, kp1) from r3d_knn_topk mode 1 (column 0 is dropped as in mpti.py:736).
 * Z = (I - alpha D^-1/2 A D^-1/2)^-1 Y (all columns at once), A the symmetrised gaussian kNN affinity with zero
 * diagonal, solved by two-level conjugate gradients: coarse space = D^1/2 x indicators of 64 aggregates (nodes
 * grouped around an even subsample of the first *n_proto_dev rows, the prototypes), A-DEF2 preconditioner.
 * nodes rows are read as float4 (ldn % 4 == 0, 16-byte aligned); Y, Z (n_cap, 4) fp32, 16-byte aligned; n_cap <= 32768.
 * ws: r3d_lp_ws_words(n_cap, kp1) int32 words, 16-byte aligned; it keeps the graph, the coarse space and the
 * directed weights for r3d_label_propagate_bwd.  stats_out optional device int32[2] = {converged, iterations}.
 * r3d_lp_ws_offsets: int32-word offsets inside ws of {row_ptr, col (uint16 entries), val, dinv, aggregate ids,
 * solver state} for tests and tools that read the system back. */
long r3d_lp_ws_words(int n_cap, int kp1);
int r3d_lp_ws_offsets(int n_cap, int kp1, long* out6);
int r3d_label_propagate(const float* nodes, long ldn, int D, const int32_t* nbr, int kp1, const float* Y,
                        const int32_t* n_dev, const int32_t* n_proto_dev, int n_cap, float sigma, float alpha,
                        int max_iter, float tol, float* Z, int32_t* ws, long ws_words, int32_t* stats_out, void* stream);

/* n_ep systems at once: system e = rows [e n_cap, (e + 1) n_cap) of nodes / nbr / Y / Z, counts at n_dev[e desc_stride] /
 * n_proto_dev[e desc_stride], scratch ws + e ws_stride (a multiple of 4 words, >= r3d_lp_ws_words), {converged, iterations}
 * at stats_out + e stats_stride.  Every CG launch serves all systems (two launches per iteration for the whole batch). */
int r3d_label_propagate_batched(int n_ep, const float* nodes, long ldn, int D, const int32_t* nbr, int kp1, const float* Y,
                                const int32_t* n_dev, const int32_t* n_proto_dev, long desc_stride, int n_cap, float sigma,
                                float alpha, int max_iter, float tol, float* Z, int32_t* ws, long ws_words, long ws_stride,
                                int32_t* stats_out, long stats_stride, void* stream);
/* Runtime guard for schedules with several streams in flight: recomputes the directed gaussian weights of
 * the graphs a preceding r3d_label_propagate(_batched) left in ws -- to be called with nothing else on the chip -- and adds
 * to *mismatch_out the number of entries whose bits differ from the ones the solve used.  scratch:
 * n_ep * r3d_graph_weights_verify_words(n_cap, kp1) floats. */
long r3d_graph_weights_verify_words(int n_cap, int kp1);
int r3d_graph_weights_verify(int n_ep, const float* nodes, long ldn, int D, const int32_t* n_dev, long desc_stride, int n_cap,
                             int kp1, float sigma, int32_t* ws, long ws_words, long ws_stride, float* scratch,
                             int32_t* mismatch_out, void* stream);

/* More than 3 ways (5..8 classes; models/mpti.py:49,58 take any n_way): label columns travel as float4 per node, so Y, Z
 * (and G, lambda of the backward) are TWO planes of 4 columns, (2, n_ep * n_cap, 4), plane 1 = classes 4..7, written /
 * read that way by r3d_head_prototypes_batched, r3d_query_logits_ce_batched, r3d_ce_grad_batched and
 * r3d_train_metrics_batched.  The label propagation is column-wise independent: r3d_label_propagate_batched solves plane 0
 * and leaves graph, weights and preconditioner in ws; this entry point solves further right-hand sides (plane 1) on them.
 * The backward is called once per plane (its outputs add). */
int r3d_label_propagate_solve_batched(int n_ep, const float* Y, const int32_t* n_dev, long desc_stride, int n_cap, int kp1,
                                      float alpha, int max_iter, float tol, float* Z, int32_t* ws, long ws_words,
                                      long ws_stride, int32_t* stats_out, long stats_stride, void* stream);

/* Captured episodes: enable the CG kernel nodes (three per iteration) of iterations < budget in an instantiated hipGraph holding
 * r3d_label_propagate / r3d_label_propagate_bwd launches, disable the rest (no dispatch for them).  graph: the
 * hipGraph_t the hipGraphExec_t graph_exec was instantiated from.  n_cg (optional, host): CG nodes found.
 * No reference counterpart (the reference inverts the dense matrix, models/mpti.py:758-776). */
int r3d_graph_set_lp_budget(void* graph, void* graph_exec, int budget, int* n_cg);

/* ---- query logits + cross entropy (models/mpti.py:558-559, 778-781) ------------------ */
int r3d_query_logits_ce(const float* Z, const int32_t* n_proto_dev, int n_q, int N, int n_classes,
                        const int64_t* labels /*opt (n_q,N)*/, float* logits /*(n_q,n_classes,N)*/,
                        float* loss_out /*opt*/, int32_t* pred_out /*opt (n_q*N)*/, void* stream);

/* per system of a batch: Z rows [e z_ep_rows, ...) -> logits / loss / pred number e of the batch arrays */
int r3d_query_logits_ce_batched(int n_ep, const float* Z, long z_ep_rows, const int32_t* n_proto_dev, long desc_stride, int n_q,
                                int N, int n_classes, const int64_t* labels, float* logits, float* loss_out, int32_t* pred_out,
                                void* stream);

/* ==== training mode (BatchNorm with batch statistics, backward) =======================
 * A conv+BN+act layer: z = r3d_pointwise_conv (no affine) -> r3d_colstats mode 0 -> r3d_bn_fold ->
 * r3d_affine_act.  Backward: r3d_colstats mode 1 (sum du, sum du*zhat = dbeta, dgamma) ->
 * r3d_bn_bwd_apply (dz) -> r3d_pointwise_conv(_acc)(dz, W^T) for dX, r3d_gemm_tn(dz, X) for dW.
 * Reference: nn.BatchNorm{1,2}d in train mode inside models/dgcnn.py:45-80, models/mpti.py:31-39. */
int r3d_pointwise_conv_acc(const float* X, long ldx, const float* W, long M, int K, int Co, const float* scale,
                           const float* shift, int act, float* Out, long ldo, void* stream);
/* z = X W^T together with its column sums (sum z, sum z^2) from the GEMM epilogue: the batch statistics of the
 * layer without a second pass over z.  ws: r3d_pointwise_conv_stats_ws_words(M, Co) floats. */
long r3d_pointwise_conv_stats_ws_words(long M, int Co);
int r3d_pointwise_conv_stats(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out, long ldo,
                             float* sums_out /*[2][Co]*/, float* ws, void* stream);
/* the same with separate statistics for rows [0, M_first) and [M_first, M) (support and query clouds of an episode in
 * one launch; mpti.py:434,436 normalise them separately).  M_first: a positive multiple of 64. */
int r3d_pointwise_conv_stats2(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out, long ldo,
                              long M_first, float* sums_a /*[2][Co]*/, float* sums_b /*[2][Co]*/, float* ws, void* stream);
/* ... and over the alternating row segments of a batch of episodes (rows_a, rows_b multiples of 64): ONE GEMM launch,
 * sums_out [seg][2][Co] */
int r3d_pointwise_conv_stats_seg(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out, long ldo,
                                 long rows_a, long rows_b, float* sums_out, float* ws, void* stream);
int r3d_colreduce(const float* part /*[chunks][2][C]*/, int chunks, int C, float* sums_out /*[2][C]*/, void* stream);
int r3d_colreduce_seg(const float* part, int count_a, int count_b, int n_seg, int C, float* sums_out /*[seg][2][C]*/, void* stream);
long r3d_colstats_ws_words(long M, int C);
int r3d_colstats(const float* X, long ldx, const float* DY, long lddy, long M, int C, int mode, const float* scale,
                 const float* shift, const float* mean, const float* invstd, int act, float* sums_out /*[2][C]*/,
                 float* ws, void* stream);
/* rec (optional): instead of updating the running statistics, record (batch mean, unbiased batch variance) as 2 C floats at
 * rec + *rec_index_dev * rec_stride; r3d_bn_running_update then applies n_records such records in order, bit for bit
 * what the updates would have given one after the other (captured episodes of several streams record, the owner
 * applies them in episode order after the step). */
int r3d_bn_fold(const float* sums, double count, int C, const float* gamma, const float* beta, float eps,
                float momentum, float* running_mean /*opt*/, float* running_var /*opt*/, float* mean, float* invstd,
                float* scale, float* shift, float* rec /*opt*/, const int32_t* rec_index_dev /*opt*/, long rec_stride,
                void* stream);
/* the segmented forms (sums / outputs [seg][...], BatchNorm vectors of segment s at pointer + s * bn_stride, running
 * statistics or records updated in segment order = the reference's order of getFeatures calls) */
long r3d_colstats_seg_ws_words(long M, int C, long rows_a, long rows_b);
int r3d_colstats_seg(const float* X, long ldx, const float* DY, long lddy, long M, int C, long rows_a, long rows_b, int mode,
                     const float* scale, const float* shift, const float* mean, const float* invstd, long bn_stride, int act,
                     float* sums_out /*[seg][2][C]*/, float* ws, void* stream);
int r3d_bn_fold_seg(const float* sums /*[seg][2][C]*/, int n_seg, double count_a, double count_b, int C, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean /*opt*/, float* running_var /*opt*/,
                    float* mean, float* invstd, float* scale, float* shift, long bn_stride, float* rec /*opt*/,
                    const int32_t* rec_index_dev /*opt*/, long rec_stride, void* stream);
int r3d_affine_act_seg(const float* Z, long ldz, long M, int C, long rows_a, long rows_b, const float* scale, const float* shift,
                       long bn_stride, int act, float* Y, long ldy, void* stream);
int r3d_bn_bwd_apply_seg(const float* Z, long ldz, const float* DY, long lddy, long M, int C, long rows_a, long rows_b,
                         const float* scale, const float* shift, const float* mean, const float* invstd, long bn_stride, int act,
                         const float* sums /*[seg][2][C]*/, double count_a, double count_b, float* DZ, long lddz, void* stream);
int r3d_bn_running_update(const float* rec, int n_records, long rec_stride, int C, float momentum,
                          const float* bias /*opt: conv bias in front of the BatchNorm*/, float* running_mean,
                          float* running_var, void* stream);
int r3d_affine_act(const float* Z, long ldz, long M, int C, const float* scale, const float* shift, int act, float* Y,
                   long ldy, void* stream);
int r3d_bn_bwd_apply(const float* Z, long ldz, const float* DY, long lddy, long M, int C, const float* scale,
                     const float* shift, const float* mean, const float* invstd, int act, const float* sums,
                     double count, float* DZ, long lddz, void* stream);
long r3d_gemm_tn_ws_words(long M, int Ca, int Cb);
int r3d_gemm_tn(const float* A, long lda, const float* B, long ldb, long M, int Ca, int Cb, float alpha, float* out,
                int accumulate, float* ws, void* stream);
int r3d_add_cols(const float* src, long ld_src, float* dst, long ld_dst, long M, int C, void* stream);

/* EdgeConv with batch statistics over the edges of every segment of clouds (models/dgcnn.py:53-57 in train mode; segments
 * of clouds_a / clouds_b clouds alternating, clouds_b == 0: equal segments -- the S support and Q query clouds of every
 * episode of a batch) and its backward; PQ is the RAW point-wise GEMM [Wa x | (Wb-Wa) x].  One launch per pass over ALL
 * clouds; statistics are per segment ([seg][2][64]), BatchNorm vectors of segment s at pointer + s * bn_stride.
 * ws: r3d_edgeconv_train_ws_words(B, N) floats. */
long r3d_edgeconv_train_ws_words(int B, int N);
int r3d_edge_stats1(const float* PQ, const int32_t* idx, int B, int N, int K, int clouds_a, int clouds_b,
                    float* sums_out /*[seg][2][64]*/, float* esum /*opt (B*N,64): sum_t e1 of every point*/, float* ws,
                    void* stream);
/* one-pass training forward: z2 statistics + per point/channel max and min of z2 over the K edges; BatchNorm2 +
 * LeakyReLU is monotone per channel, so r3d_edge_select finishes the layer once the statistics are folded */
int r3d_edgeconv_train_fwd_minmax(const float* PQ, const int32_t* idx, const float* s1, const float* t1, long bn_stride,
                                  const float* W2, int B, int N, int K, int clouds_a, int clouds_b, float* zmax, float* zmin,
                                  int32_t* argmax, int32_t* argmin, float* sums_out /*[seg][2][64]*/, float* ws, void* stream);
int r3d_edge_select(float* zmax /*in: max, out: selected z*/, const float* zmin, int32_t* argmax /*in/out*/,
                    const int32_t* argmin, const float* s2, const float* t2, long bn_stride, long M, long rows_a, long rows_b,
                    float* out, long ldo, void* stream);
/* reverse neighbour list (for every point the edges that name it, ascending): rev_ws = r3d_edge_reverse_ws_words int32
 * words.  The backward gathers along it instead of scattering with float atomics: deterministic gradients. */
long r3d_edge_reverse_ws_words(int B, int N, int K);
int r3d_edge_reverse(const int32_t* idx, int B, int N, int K, int32_t* rev_ws, long ws_words, void* stream);
/* bn2_sums [seg][2][64] in; dW2 (64,64) summed over the WHOLE batch, bn1_sums [seg][2][64], dPQ (B*N,128) out.
 * zwin (B*N,64): z2 of every max-pool winner (zmax after r3d_edge_select); esum: from r3d_edge_stats1.  Both given, N % 8 == 0
 * and r3d_set_matrix_arith(1): the three edge GEMMs of the backward run on the bf16 matrix core in three-piece arithmetic
 * (none decides an index: the winner and its side of the LeakyReLU kink come from argmax / zwin); else on the fp32 core. */
int r3d_edgeconv_bwd(const float* PQ, const int32_t* idx, const float* s1, const float* t1, const float* mean1,
                     const float* invstd1, const float* W2, const float* s2, const float* t2, const float* mean2,
                     const float* invstd2, long bn_stride, const float* bn2_sums, const float* dout, long lddo,
                     const int32_t* argmax, const float* zwin /*opt*/, const float* esum /*opt*/, int B, int N, int K,
                     int clouds_a, int clouds_b, float* DY1 /*(B*N*K,64) scratch*/, float* BE /*(B*N,128) scratch*/,
                     const int32_t* rev_ws /* r3d_edge_reverse of the same idx */, float* dW2, float* bn1_sums, float* dPQ,
                     float* ws, void* stream);

/* attention with dropout on the weights (attention.py:45) and flash-style backward */
/* effective dropout seed = seed + *seed_dev (seed_dev may be NULL); a captured hipGraph bumps *seed_dev per replay */
int r3d_attention_fwd_train(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out, float p_drop,
                            unsigned seed, const unsigned* seed_dev, float* ws /*opt, as above*/, void* stream);
int r3d_attention_bwd(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO, long lddo,
                      const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev, float q_scale, float* dqkv,
                      long ldd, float* ws /* r3d_attention_ws_words(B, N) floats */, void* stream);
/* the same; ws_holds_packed_qkv != 0: ws is the workspace r3d_attention_fwd_train ran with on this qkv, untouched since
 * (the bf16 x 3 kernels reuse the packed q | k | v pieces it holds instead of cutting them again) */
int r3d_attention_bwd_ws(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO, long lddo,
                         const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev, float q_scale,
                         float* dqkv, long ldd, float* ws, int ws_holds_packed_qkv, void* stream);

/* batches of episodes: clouds [e seed_group, (e + 1) seed_group) are episode e, whose dropout mask is the one a call on
 * those clouds alone draws with seed + 2 e (the one-episode schedule advances its seed by 2 per episode); outputs are
 * bit for bit those of that call (p_drop = 0: the inference forward of a batch) */
long r3d_attention_ws_words_ep(int B, int N, int seed_group); /* workspace of the _ep calls: the key-axis split is the one
                                                                * of ONE episode (seed_group clouds), whatever the batch */
int r3d_attention_fwd_train_ep(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out, float p_drop,
                               unsigned seed, const unsigned* seed_dev, int seed_group, float* ws, void* stream);
int r3d_attention_bwd_ep(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO, long lddo,
                         const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev, int seed_group, float q_scale,
                         float* dqkv, long ldd, float* ws, int ws_holds_packed_qkv, void* stream);

/* head backward (reference: autograd through models/mpti.py:488-512,571).  r3d_ce_grad -> G = dL/dZ (scaled by the
 * device scalar *gscale); r3d_label_propagate_bwd: adjoint CG solve on the graph r3d_label_propagate left in ws,
 * then gradients w.r.t. the node features; r3d_head_prototypes_bwd: cluster-mean / query-row backward. */
int r3d_ce_grad(const float* Z, const int32_t* n_proto_dev, int n_cap, int n_query_pts, int n_classes,
                const int64_t* labels, const float* gscale_dev, float* G, void* stream);
int r3d_label_propagate_bwd(const float* nodes, long ldn, int D, int kp1, const float* Z, const float* G,
                            const int32_t* n_dev, int n_cap, float sigma, float alpha, int max_iter, float tol, float* lam,
                            float* dnodes, long ldd, int32_t* ws, long ws_words, int32_t* stats_out, void* stream);
int r3d_head_prototypes_bwd(const float* dnodes, long ldd, int n_way, int k_shot, int N, int D, int n_query_pts,
                            const int32_t* desc, const int32_t* assign, const int32_t* cluster_count, const int32_t* ws,
                            float* dsfeat, long lds_, float* dqfeat, long ldq, void* stream);

/* the same for n_ep episodes (layouts as the forward `_batched` calls; G, lam, dnodes: n_cap rows per system; *gscale_dev
 * scales every episode alike: the step's loss is the SUM of the episodes' losses) */
int r3d_ce_grad_batched(int n_ep, const float* Z, const int32_t* n_proto_dev, long desc_stride, int n_cap, int n_query_pts,
                        int n_classes, const int64_t* labels, const float* gscale_dev, float* G, void* stream);
int r3d_label_propagate_bwd_batched(int n_ep, const float* nodes, long ldn, int D, int kp1, const float* Z, const float* G,
                                    const int32_t* n_dev, long desc_stride, int n_cap, float sigma, float alpha, int max_iter,
                                    float tol, float* lam, float* dnodes, long ldd, int32_t* ws, long ws_words, long ws_stride,
                                    int32_t* stats_out, long stats_stride, void* stream);
int r3d_head_prototypes_bwd_batched(int n_ep, const float* dnodes, long ldd, long nodes_ep_rows, int n_way, int k_shot, int N,
                                    int D, int n_query_pts, const int32_t* desc, long desc_stride, const int32_t* assign,
                                    long assign_stride, const int32_t* cluster_count, long ccount_stride, const int32_t* ws,
                                    long ws_stride, float* dsfeat, long lds_, long dsfeat_ep_rows, float* dqfeat, long ldq,
                                    long dqfeat_ep_rows, void* stream);

/* per-way supervised contrastive loss, train only (models/mpti.py:226-313): per shot FPS(4) prototypes of the
 * foreground points -> proj Linear(D,128) -> L2 normalise -> SupCon(temp); mean over ways.  ws keeps what
 * r3d_contrast_bwd needs (prototype gradients, assignments, per-way parameter gradients). */
long r3d_contrast_ws_words(int n_way, int k_shot, int N);
int r3d_contrast_fwd(const float* feat, long ldf, int D, const int32_t* support_y, const int32_t* support_flag, int n_way,
                     int k_shot, int N, const float* W, const float* bias, float temp, float* loss_out, float* ws,
                     long ws_words, void* stream);
int r3d_contrast_bwd(int D, int n_way, int k_shot, int N, const float* gscale_dev, float* dfeat, long ldd, float* dW,
                     float* db, float* ws, void* stream);
/* n_ep episodes: features of episode e at feat + e feat_ep_rows ldf, masks / flags entry e of (n_ep, S, N) / (n_ep, S),
 * scratch ws + e ws_stride, loss_out[e]; the backward sums dW / db over the batch */
int r3d_contrast_fwd_batched(int n_ep, const float* feat, long ldf, long feat_ep_rows, int D, const int32_t* support_y,
                             const int32_t* support_flag, int n_way, int k_shot, int N, const float* W, const float* bias,
                             float temp, float* loss_out, float* ws, long ws_words, long ws_stride, void* stream);
int r3d_contrast_bwd_batched(int n_ep, int D, int n_way, int k_shot, int N, const float* gscale_dev, float* dfeat, long ldd,
                             long dfeat_ep_rows, float* dW, float* db, float* ws, long ws_stride, void* stream);
int r3d_train_metrics_batched(int n_ep, const int32_t* pred, const int64_t* query_y, const int64_t* gt_query_y, int n_query_pts,
                              const float* Z, long z_ep_rows, const int32_t* desc, long desc_stride, const int32_t* proto_ws,
                              long pws_stride, const int32_t* assign, long assign_stride, const int32_t* gt_support_y, int n_way,
                              int k_shot, int N, float* out4 /*(n_ep,4)*/, void* stream);
/* training-only debug metrics (mpti.py:515-568): out4 = query_acc_LP, query_acc_original, clean_ratio_LP_avg,
 * clean_ratio_original_avg */
int r3d_train_metrics(const int32_t* pred, const int64_t* query_y, const int64_t* gt_query_y, int n_query_pts, const float* Z,
                      const int32_t* desc, const int32_t* proto_ws, const int32_t* assign, const int32_t* gt_support_y,
                      int n_way, int k_shot, int N, float* out4, void* stream);

/* ---- clean-shot detection, eval only (models/mpti.py:87-223, 316-371) -----------------
 * Per shot: box means of foreground features at scales (1,1,1) and (2,2,1) -> cosine map ->
 * majority vote -> shot_keep (n_way*k_shot) int32 (0 = drop the shot's foreground). */
long r3d_clean_ws_words(int n_way, int k_shot);
int r3d_clean_shot_detect(const float* feat /*(S*N,ldf)*/, long ldf, int D, const float* support_x /*(S,Cin,N)*/,
                          int Cin, const int32_t* support_y, int n_way, int k_shot, int N, int32_t* shot_keep,
                          float* dbg_cos_sum /*opt (n_way,2,4*k_shot)*/, int32_t* ws, void* stream);

int r3d_clean_shot_detect_batched(int n_ep, const float* feat, long ldf, long feat_ep_rows, int D, const float* support_x,
                                  int Cin, const int32_t* support_y, int n_way, int k_shot, int N, int32_t* shot_keep,
                                  float* dbg_cos_sum, int32_t* ws, long ws_stride, void* stream);

/* ---- ProtoNet head (models/protonet.py:295-349): masked average pooling + similarity ----
 * method 0 cosine * scaler, 1 -euclidean^2; anything else returns non-zero like the reference's
 * NotImplementedError.  Z (n_query_pts, 4) similarity rows; ws S*2*256 floats. */
int r3d_protonet_head(const float* sfeat, long ldf, const float* qfeat, long ldq, int D, const int32_t* support_y,
                      int n_way, int k_shot, int N, int n_query_pts, int method, float scaler, float* Z, float* ws,
                      void* stream);

/* ---- mIoU accumulator (eval_noise.py:23-72): hist (3, n_classes) uint64 = GT | predicted | TP */
int r3d_miou_accumulate(const int32_t* pred, const int64_t* gt, long n, const int32_t* lut, int n_lut, int n_classes,
                        uint64_t* hist, void* stream);

#ifdef __cplusplus
}
#endif
#endif
