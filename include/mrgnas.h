/*
 * mrgnas.h -- C ABI of libmrgnas_hip.so: the MI355X (gfx950) implementation of
 * MR-GNAS's relational message-passing hot path.
 *
 * The reference (Amanda-Zheng/MR-GNAS) is pure Python on PyTorch + DGL and has
 * no FFI of its own; each entry point below replaces the ATen/DGL kernel
 * sequence that one reference operator launches.  The citation on each
 * function is the reference interface it stands in for (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *  - float tensors are float32, row-major, contiguous (leading dimension == D);
 *  - graph index arrays are int32;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *    call only enqueues work on that stream -- no allocation, no
 *    synchronisation, safe to capture into a hipGraph;
 *  - scratch memory is caller-owned: ask mrg_*_workspace_bytes(), pass `ws`;
 *  - return value: 0 = success, < 0 = argument error (MRG_E_*), > 0 = the
 *    hipError_t of the failed launch.  mrg_error_string() explains either.
 *  - row layout of every [M, D] edge tensor, M = E + N: rows [0, b0) original
 *    direction ("in") edges, [b0, b1) inverse ("out") edges, [b1, M) one
 *    self-loop row per node (reference models/model_lp.py:126-129,
 *    models/model_search_lp.py:135-139).  The reference always has b0 = E/2,
 *    b1 = E; relation-block shards pass their local boundaries.
 */
#ifndef MRGNAS_H
#define MRGNAS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRG_ABI_VERSION 14   /* 14: mrg_clip_sgd_step, mrg_optim_chunk (clip_grad_norm_ + SGD with momentum over every parameter tensor in three launches), mrg_gated_branch.act (tanh behind the BatchNorm: CompGraphConv's tail on the epilogue kernels), mrg_gemm_set_small (few-row products on two-tile column blocks), mrg_segmax_bwd_input (a_max's input gradient without a dense product); 13: mrg_set_dynamic_rows (device-side row counts: the sampled search step as one replayable HIP graph), mrg_seg_reduce_bwd_ordered (aggregator backward walked in destination order), mrg_gemm_set_q (the 16 x 16 x 32 row GEMM at three workgroups per CU for 129..224 output columns); 12: mrg_act_grad_transpose (the [B, N] scorer's output gradient, activation folded in, as [N, B] rows); 11: mrg_gemm_set_wide8 (eight-tile column block for D = 256); 10: mrg_gemm_set_epilogue(2) (transposed accumulators: a tested comparison point); 9: mrg_gated_branch (the MixedOp epilogue recomputes f_dense_comp's output from its gate and f_sparse_comp's from its row factor), mrg_gate_row_fwd / _bwd, mrg_sum_rows_gather, mrg_wgrad_set_variant, mrg_dense_filter_fwd3 out == NULL; 8: mrg_zero_* (cell-zero MixedOp recomputed from the tables), mrg_linear_bwd_input3_pair, mrg_sample_edge_neighborhood; 7: mrg_gemm_set_epilogue (row-order stores of the split-core row GEMM), mrg_set_stream_blocks, mrg_gemm_set_mode(2); 6: fused a_mean (run-sum epilogue, heads reducer, bit-mask backward), mrg_mix_stats_coef; 5: three-segment dense filter entry points; 4: mrg_linear_relu_segmax_fwd (fused a_max); 3: device graph / plan builders, samplers, [B, N] scorers, ranking; 2: GEMM workspaces, span_gcs ext_scal */

#define MRG_OK            0
#define MRG_E_NULLPTR    -1   /* a required pointer is NULL */
#define MRG_E_SHAPE      -2   /* negative size, or a size the kernels do not cover (D > 1024, or D > 256 with D % 4 != 0) */
#define MRG_E_ENUM       -3   /* unknown op / mode / act code */
#define MRG_E_WORKSPACE  -4   /* workspace pointer missing */

/* compose codes (a1) */
#define MRG_COMPOSE_MULT 0
#define MRG_COMPOSE_SUB  1
#define MRG_COMPOSE_ADD  2
/* reducer codes (a4-a6) */
#define MRG_REDUCE_SUM   0
#define MRG_REDUCE_MEAN  1
#define MRG_REDUCE_MAX   2
/* compose codes of mrg_fused_gcs: x = X[xi[e]], y = Y[yi[e]], s = scal[e] (1 if scal is NULL) */
#define MRG_GCS_SUB      0   /* x - y*s                          CompGCN 'sub'  (u_sub_e)            */
#define MRG_GCS_MUL      1   /* x * (y*s)                        CompGCN 'mul'  (u_mul_e)            */
#define MRG_GCS_COPY     2   /* x*s                (Y unused)    gather backward, d/dh of 'sub'       */
#define MRG_GCS_NEGS     3   /* -(x*s)             (Y unused)    d/dr of 'sub'                        */
#define MRG_GCS_CCORR    4   /* sum_i x[i] * y[(i+k)%D] * s      CompGCN 'ccorr' and its d/dh         */
#define MRG_GCS_CCONV    5   /* s * sum_i x[i] * y[(k-i)%D]      d/dr of 'ccorr'                      */
/* activation codes for mrg_linear_fwd */
#define MRG_ACT_NONE     0
#define MRG_ACT_RELU     1
#define MRG_ACT_SIGMOID  2   /* the [B, N] score functions: sigmoid((sub * rel) all_ent^T) */

int mrg_abi_version(void);
/* Upper bound of the grid of the FLAT HBM-streaming kernels (compose, K-way sums, the MixedOp combine / statistics /
 * gradient-reduction passes; 256-thread blocks that walk the tensors with a grid stride).  Default 512: the fastest streaming
 * grid on MI355X (tools/stream_lab.hip: 5.9 TB/s, 2048 blocks 5.3).  64 <= blocks <= 4096 (the MixedOp reductions use at most
 * 1024).  The row-per-wave kernels (gates, gathers, reducers) are not affected.  Process-wide; a tuning knob, results do not
 * depend on it except for the summation order of the column reductions. */
int mrg_set_stream_blocks(int blocks);
const char *mrg_error_string(int code);
/* Name of the code object's target, e.g. "gfx950". */
const char *mrg_target_arch(void);

/* ---- a1: compose ops ---------------------------------------------------
 * pre_mult_op / pre_sub_op / pre_add_op .forward(g, src_emb, hr)
 * reference models/operations_lp.py:71-98.   out = s (*|-|+) hr, [rows, D]. */
int mrg_compose_fwd(int op, const float *s, const float *hr, float *out,
                    int64_t rows, int D, void *stream);
/* autograd of the above: gs/ghr may be NULL when not needed. */
int mrg_compose_bwd(int op, const float *gout, const float *s, const float *hr,
                    float *gs, float *ghr, int64_t rows, int D, void *stream);

/* ---- G + a1: gather feeding the compose ----------------------------------
 * all_ent_emb[src_id_final] (op) rel_embed[edge_type_final]
 * reference models/model_lp.py:126-131, models/model_search_lp.py:135-145.
 * out[i,:] = ent[ent_idx[i],:] (op) rel[rel_idx[i],:];  op = -1 copies ent rows
 * only (plain gather, rel ignored). Bit-exact. */
int mrg_gather_compose_fwd(int op, const float *ent, const float *rel,
                           const int32_t *ent_idx, const int32_t *rel_idx,
                           float *out, int64_t rows, int D, void *stream);

/* floats per (u, v, c) slot of a gate: u at [0, D), v at [D, 2D), c at index
 * in_dim (2D with s_in, D without); padded so every slot stays 16-byte aligned. */
#define MRG_GATE_LD(D) (2 * (D) + 4)

/* ---- a2 / a3: collapsed scalar gates --------------------------------------
 * f_sparse_op_comp.forward  reference models/operations_lp.py:317-343
 * f_sparse_op_last.forward  reference models/operations_lp.py:412-416
 *
 * a_x(W_x [s ; s_in] + b_x) has no non-linearity inside, so it equals
 * u_x . s + v_x . s_in + c_x with  [u_x ; v_x] = W_x^T a_x,  c_x = a_x . b_x.
 *
 * mrg_gate_collapse:   uvc[0 .. in_dim) = W^T a,  uvc[in_dim] = a . b
 *   W [D, in_dim] (nn.Linear weight), b [D] or NULL, a [D] (nn.Linear(D,1).weight). */
int mrg_gate_collapse(const float *W, const float *b, const float *a, float *uvc,
                      int D, int in_dim, void *stream);
/* Batched forms over the three direction segments (in, out, self) of one operator -- one launch instead of three: *_host are
 * HOST arrays of 3 device pointers, NULL for an absent segment; uvc / d_uvc are [3][MRG_GATE_LD(D)].
 * fold != 0 (both operands of the operator are the same rows, reference models/cell_lp.py:95-104: op(g, h_in, h_in)): the
 * parameters are nn.Linear(2D, D) but the gate is u.s + v.s = (u + v).s -- uvc holds u + v at [0, D) and c at index D
 * (the layout of a gate without a second operand), and the parameter gradient spreads d(u + v) over both weight halves. */
int mrg_gate_collapse3(const float *const *W_host, const float *const *b_host, const float *const *a_host, float *uvc,
                       int D, int in_dim, int fold, void *stream);
int mrg_gate_param_grad3(const float *const *W_host, const float *const *b_host, const float *const *a_host,
                         const float *d_uvc, float *const *gW_host, float *const *gb_host, float *const *ga_host,
                         int D, int in_dim, int fold, void *stream);
/* The same fold for the dense filters: Wt [3][D][D], Wt_i = W_i[:, :D] + W_i[:, D:] (W_i is [D][2D]); and its adjoint
 * gW_i = [gWt_i | gWt_i] (gWt_host: 3 device pointers to [D][D]; a NULL source gives a zero gradient). */
int mrg_fold_halves3(const float *const *W_host, float *Wt, int D, void *stream);
int mrg_unfold_halves3(const float *const *gWt_host, float *const *gW_host, int D, void *stream);

/* out[i,:] = sigmoid(u_x.s_i + v_x.s_in_i + c_x) * s_i * scale * (i < b1 ? norm[i] : 1)
 *   x = segment of row i (0: i < b0, 1: b0 <= i < b1, 2: i >= b1)
 *   uvc  [3][MRG_GATE_LD(D)]  (u, v, c per segment); s_in NULL => v ignored (f_sparse_last)
 *   norm [b1] or NULL (=> 1).  f_sparse_comp: scale = 1/3; f_sparse_last: b0 = b1 = 0, scale = 1. */
int mrg_gate_fwd(const float *s, const float *s_in, const float *norm, const float *uvc,
                 float *out, int64_t b0, int64_t b1, int64_t M, int D, float scale, void *stream);
int64_t mrg_gate_bwd_workspace_bytes(int64_t M, int D);
/* gs, gs_in [M, D] (gs_in NULL allowed iff s_in NULL); d_uvc [3][MRG_GATE_LD(D)] receives
 * d(loss)/d(u, v, c) per segment (deterministic two-stage reduction, no atomics). */
int mrg_gate_bwd(const float *gout, const float *s, const float *s_in, const float *norm,
                 const float *uvc, float *gs, float *gs_in, float *d_uvc, void *ws,
                 int64_t b0, int64_t b1, int64_t M, int D, float scale, void *stream);
/* f_sparse_op_comp as a row factor: the candidate y = s * fvec[r] is recomputed by the MixedOp epilogue (mrg_gated_branch.row_k)
 * instead of being stored.  fvec[r] = sigmoid(u.s + v.s_in + c0) * t_r is mrg_gate_fwd's factor (same expression and dot order:
 * s * fvec[r] equals its output bit for bit); hvec[r] = t_r * gate * (1 - gate).  fvec, hvec: [M]. */
int mrg_gate_row_fwd(const float *s, const float *s_in, const float *norm, const float *uvc, float *fvec, float *hvec,
                     int64_t b0, int64_t b1, int64_t M, int D, float scale, void *stream);
/* q [M] (mrg_mix_bwd_apply's row_dq: the gradient w.r.t. fvec), dz_r = q_r * hvec[r] -> gs_in [M, D] = dz_r * v (s_in != NULL)
 * and d_uvc [3][MRG_GATE_LD(D)] for mrg_gate_param_grad3; the gradient w.r.t. s was added by the epilogue.
 * ws: mrg_gate_bwd_workspace_bytes(M, D). */
int mrg_gate_row_bwd(const float *q, const float *hvec, const float *s, const float *s_in, const float *uvc, float *gs_in, float *d_uvc,
                     void *ws, int64_t b0, int64_t b1, int64_t M, int D, void *stream);
/* chain rule back to the nn.Linear parameters:  d_uvc [in_dim+1] ->
 * gW [D, in_dim] = a (x) d_uv,  gb [D] = a * d_c (NULL ok),  ga [D] = W d_uv + b d_c. */
int mrg_gate_param_grad(const float *W, const float *b, const float *a, const float *d_uvc,
                        float *gW, float *gb, float *ga, int D, int in_dim, void *stream);

/* ---- a4 / a5 / a6: destination-segmented reducers --------------------------
 * block.update_all(fn.copy_edge('msg_e','m'), fn.max|sum|mean('m','h')) + residual
 * reference models/operations_lp.py:232-234, 247-249, 261-263  (DGL 0.5.3 gspmm).
 *
 * The in-edges of node v are eid[rowptr[v] .. rowptr[v+1]) (edge ids in the
 * caller's order, ascending).  Long lists are cut into chunks:
 *   chunk c covers CSR positions [chunk_start[c], chunk_end[c]) of node chunk_node[c];
 *   every node has >= 1 chunk (an empty one if it has no in-edge);
 *   chunk_slot[c] = -1 if the chunk is its node's whole list (result written
 *   straight to out), else the index of its partial result in the workspace;
 *   nodes with > 1 chunk are listed in hub_node[], their partial slots are
 *   hub_first[j] .. hub_first[j] + hub_count[j] - 1 (consecutive, list order).
 * out[v,:] = reduce_{e in N(v)} msg[e,:]  (+ self_rows[v,:] if not NULL); rows
 * without in-edges reduce to 0.  arg [N, D] (max only) = the lowest edge id
 * attaining the max, -1 where there is no in-edge. */
int64_t mrg_seg_reduce_workspace_bytes(int64_t n_slots, int D);
int mrg_seg_reduce_fwd(int mode, const float *msg, const float *self_rows,
                       const int32_t *eid,
                       const int32_t *chunk_node, const int32_t *chunk_start, const int32_t *chunk_end,
                       const int32_t *chunk_slot, int64_t n_chunks,
                       const int32_t *hub_node, const int32_t *hub_first, const int32_t *hub_count, int64_t n_hubs,
                       int64_t n_slots, const int32_t *in_degree,
                       float *out, int32_t *arg, void *ws,
                       int64_t N, int D, void *stream);
/* gmsg [E, D]: sum  gmsg[e] = gout[dst[e]];  mean  gout[dst[e]] / max(deg,1);
 * max  gmsg[e,c] = (arg[dst[e],c] == e) ? gout[dst[e],c] : 0.
 * gself [N, D] = gout (skipped when NULL).  relu_src [E, D] (NULL ok): the messages were ReLU
 * outputs (a_max / a_mean), gmsg is additionally zeroed where relu_src <= 0. */
int mrg_seg_reduce_bwd(int mode, const float *gout, const int32_t *dst, const int32_t *in_degree,
                       const int32_t *arg, float *gmsg, float *gself, const float *relu_src,
                       int64_t E, int64_t N, int D, void *stream);
/* The same (relu_src [E, D] or relu_bits [E, ceil(D / 32)] or neither) with the edges WALKED in destination order: order [E] = the edge
 * ids sorted by destination (the eid list of mrg_chunk_plan_build; NULL = edge-id order).  Same values; the gathered rows of the [N, D]
 * tables gout / arg are then re-used from cache by consecutive edges instead of being fetched per edge (C5: 5.05 -> see DESIGN.md). */
int mrg_seg_reduce_bwd_ordered(int mode, const float *gout, const int32_t *dst, const int32_t *in_degree, const int32_t *arg,
                               float *gmsg, float *gself, const float *relu_src, const unsigned *relu_bits, const int32_t *order,
                               int64_t E, int64_t N, int D, void *stream);
/* ABI 14.  a_max's backward without the dense input-gradient product (reference models/operations_lp.py:230-235): the gradient w.r.t.
 * the messages has one non-zero per (node, column) -- the arg-max edge, where the maximum is positive -- so
 *   gmsg[e][c] = (arg[dst[e]][c] == e && (mx == NULL || mx[dst[e]][c] > 0)) ? gout[dst[e]][c] : 0        [E, D]  (NULL: not written)
 *   gx[e][k]   = sum_c gmsg[e][c] * W[c][k]                                                             [E, Kin]
 * in ONE pass per edge row: the rows W[c, :] of the columns an edge won are added from a copy of W in LDS.  W [D, Kin] row-major must
 * fit 160 KB (mrg_segmax_bwd_input_ok); D, Kin multiples of 4, at most 256.  order: NULL or the edge ids sorted by destination.
 * Deterministic; exact f32 arithmetic. */
int mrg_segmax_bwd_input_ok(int D, int Kin);
int mrg_segmax_bwd_input(const float *gout, const float *mx, const int32_t *dst, const int32_t *arg, const float *W, float *gmsg,
                         float *gx, const int32_t *order, int64_t E, int64_t N, int D, int Kin, void *stream);

/* ---- a9: fused gather -> compose -> segmented sum ------------------------------
 * CompGraphConv.forward steps 1-3, reference models/compgcn.py:58-87:
 *   g.edata['h'] = r_feats[etype] * norm;  apply_edges(u_sub_e | u_mul_e | ccorr);
 *   mask + scatter (in/out edges);  update_all(copy_e, sum)
 * (the per-direction linears W_O / W_I commute with the sum and run on the [N, D] result).
 *   out[seg,:] = sum_{e in list(seg)} combine(mode, X[xi[e],:], Y[yi[e],:], scal[e])
 * Segment lists use the chunk plan of mrg_seg_reduce_fwd (eid = element ids grouped by
 * segment; seg_len[nseg] = list lengths).  The same entry point with other index arrays is
 * the backward (segments keyed by src / by etype) and the backward of row gathers. */
int mrg_fused_gcs(int mode, const float *X, const int32_t *xi, const float *Y, const int32_t *yi,
                  const float *scal, const int32_t *eid,
                  const int32_t *chunk_node, const int32_t *chunk_start, const int32_t *chunk_end,
                  const int32_t *chunk_slot, int64_t n_chunks,
                  const int32_t *hub_node, const int32_t *hub_first, const int32_t *hub_count, int64_t n_hubs,
                  int64_t n_slots, const int32_t *seg_len,
                  float *out, void *ws, int64_t nseg, int D, void *stream);

/* Span form of the elementwise modes (SUB, MUL, COPY, NEGS) of mrg_fused_gcs: the elements are
 * pre-sorted by segment and packed as int32x4 {seg, xi, yi, float-bits of scal} in `meta` [E];
 * every lane group reduces `span` consecutive sorted elements (perfect load balance, 8 gathered
 * rows in flight).  span_slot [n_spans][2] = workspace slot of the first / last run of a span
 * when that run does not cover its whole segment (-1 otherwise); hub_* list the segments made
 * of partial runs (consecutive slots, list order) AND, with hub_count 0, the segments that have
 * no element at all, so that every row of `out` [nseg, D] is written.
 * ext_scal (NULL ok): when given, meta.w is an element index and the scale is ext_scal[meta.w]
 * (per-call scales, e.g. upstream gradients, without rebuilding the packed metadata). */
int mrg_span_gcs(int mode, const float *X, const float *Y, const void *meta, const float *ext_scal, int64_t E, int span,
                 const int32_t *span_slot, const int32_t *span_start, int64_t n_spans,
                 const int32_t *hub_seg, const int32_t *hub_first, const int32_t *hub_count, int64_t n_hubs,
                 int64_t n_slots, const int32_t *seg_len,
                 float *out, void *ws, int64_t nseg, int D, void *stream);

/* ---- gradient fan-in -------------------------------------------------------------------
 * A state read by several candidate operators (MixedOp, reference models/cell_lp.py:25-33; h_in read by
 * every MixedOp of a cell, models/cell_lp.py:150-186) receives one gradient per reader:
 *   out[i] = (accumulate ? out[i] : 0) + sum_k xs_host[k][i],  i < n,  1 <= K <= 8
 * xs_host: HOST array of K device pointers.  Order of summation k = 0..K-1. */
int mrg_sum_buffers(const float *const *xs_host, int K, float *out, int64_t n, int accumulate, void *stream);
/* out[r] = sum_k xs[k][r] + (r < E ? gather_edge[dst[r]] : gather_self[r - E]),  rows of D floats, 0 <= K <= 8.
 * The fan-in of a state read by a_sum_op (reference models/operations_lp.py:252-264) among others: a_sum's gradient w.r.t. the
 * [M, D] state is a gather of the [N, D] node gradient (edge rows: gather_edge = g * dropout mask, self rows: gather_self = g,
 * NULL = zero), so it is read from that tensor instead of from a materialised [M, D] copy. */
int mrg_sum_rows_gather(const float *const *xs_host, int K, const float *gather_edge, const float *gather_self,
                        const int32_t *dst, int64_t E, int64_t rows, int D, float *out, void *stream);

/* ---- DistMult scores (the step after the path) ---------------------------------------
 * Network.calc_score, reference models/model_search_lp.py:169-176:
 *   score[t] = sum_c ent[s_t, c] * rel[r_t, c] * ent[o_t, c]
 * without materialising the three [T, D] gathers; backward = three mrg_span_gcs (mode MUL, ext_scal = dscore). */
int mrg_distmult_score(const float *ent, const float *rel, const int32_t *s_idx, const int32_t *r_idx,
                       const int32_t *o_idx, float *score, int64_t T, int D, void *stream);

/* ---- X: MixedOp epilogue  out = sum_k w_k * ReLU(BatchNorm_k(y_k)) -----------------
 * MixedOp.forward / op_forward, reference models/cell_lp.py:25-33, with nn.BatchNorm1d in
 * training mode (:21).  K <= 8 operator outputs y_k [rows, D]; a NULL y_k is an all-zero
 * output (f_zero).  *_host arguments are HOST arrays of K DEVICE pointers.
 *   mrg_mix_colstats      sums[k][0|1][c] = sum_r y_k, sum_r y_k^2        (float64 [K][2][D])
 *   mrg_mix_finalize_fwd  coef[k][0..3] = scale (gamma*invstd), shift (beta - mean*scale),
 *                         invstd, mean*invstd; updates running_mean / running_var like torch
 *                         (momentum, unbiased variance) when their pointers are given.
 *                         `total_rows` = rows of the whole (possibly sharded) tensor.
 *   mrg_mix_fwd           out = sum_k w[k] * relu(y_k * scale_k + shift_k)
 *   mrg_mix_bwd_reduce    red[k][0] = sum_r gr, [1] = sum_r gr*xhat, [2] = sum_r g*relu(z),
 *                         gr = w[k] * g * [z > 0]                         (float32 [K][3][D])
 *   mrg_mix_finalize_bwd  coef2[k][0|1] = red[k][0|1] / total_rows; dw[k] = sum_c red[k][2][c];
 *                         optional dgamma_k = red[k][1], dbeta_k = red[k][0]
 *   mrg_mix_bwd_apply     gy_k = (gr - coef2[k][0] - xhat * coef2[k][1]) * scale_k  (NULL gy_k skipped)
 * Statistics may be all-reduced between colstats/finalize (and reduce/finalize) when rows
 * are sharded over GPUs.  Needs K*6*D*4 <= 64 KiB of LDS. */
int64_t mrg_mix_workspace_bytes(int K, int D);
/* "Static step graphs" (round 5).  The reference's search loop draws a new step graph every step (search/mr_lp_search.py:187-214,
 * utils/utils_rgcn.py:79-118) whose node count depends on the draw.  To replay the whole step -- sampler included -- from ONE captured
 * HIP graph, the step graph is padded to a host-known node capacity and the true counts stay in DEVICE memory: after
 *   mrg_set_dynamic_rows(cap_m, count_m, cap_n, count_n)
 * every mrg_mix_* / mrg_zero_* launch whose `rows` equals cap_m (the [M, D] edge + node rows) or cap_n (the [N, D] node rows) treats
 * the rows at and beyond *count_m / *count_n (int32, device) as PADDING: left out of the BatchNorm statistics and of every
 * gradient reduction, not counted in the statistics' row total, and written as zeros by mrg_mix_fwd / mrg_zero_fwd and by the
 * gradient stores.  Zero rows stay zero through every operator of the search space (a zero state row yields a zero candidate),
 * so no other entry point needs the counts.  NULL counts switch a slot off.  Process-wide; the pointers are read by the kernels at
 * run time (a captured launch re-reads them on every replay). */
int mrg_set_dynamic_rows(int64_t cap_m, const int32_t *count_m, int64_t cap_n, const int32_t *count_n);

/* ---- the step's tail: gradient clipping + SGD -------------------------------------------
 * torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.SGD(momentum, weight_decay, dampening 0, no Nesterov).step()
 * (reference search/mr_lp_search.py:118-119,243-245) over ALL parameter tensors in three launches: chunked sum of squares (double),
 * one-workgroup ordered total -> norm_coef[0] = total 2-norm, norm_coef[1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0),
 * then per element g = coef * grad + weight_decay * p; buf = momentum * buf + g; p -= lr * buf.
 * params / grads / bufs: DEVICE arrays of n_tensors device pointers (float32, contiguous); a null gradient pointer leaves that
 * tensor (and its momentum buffer) untouched, like torch's skip of parameters without a gradient.  Momentum buffers start at zero
 * (torch initialises buf = g on the first step: the same value).  The tensors are cut into chunks of mrg_optim_chunk() elements by
 * the caller: chunk b covers elements [chunk_off[b], chunk_off[b] + chunk_len[b]) of tensor chunk_tensor[b] (device arrays; shapes
 * never change, so they are built once).  partial: n_chunks doubles of workspace.  Deterministic. */
int mrg_optim_chunk(void);
int mrg_clip_sgd_step(void *const *params, const void *const *grads, void *const *bufs, const int32_t *chunk_tensor,
                      const int64_t *chunk_off, const int32_t *chunk_len, int64_t n_chunks, double *partial, float *norm_coef,
                      float max_norm, float lr, float momentum, float weight_decay, void *stream);
/* `gated` (HOST pointer, NULL or k < 0 = none) of the five entry points that read the candidates: candidate k is the gated
 * filter f_dense_op_comp (reference models/operations_lp.py:356-390) and is NOT stored -- y_host[k] holds its gate
 * sigmoid(W [s ; s_in] + b) (mrg_dense_filter_fwd3 with out == NULL) and its value is recomputed wherever it is read as
 *   y_k[r][c] = gate[r][c] * s[r][c] * rowscale[r]
 * -- the expression and the order of the row GEMM's gate epilogue, so coefficients, output and gradients are those of the
 * stored form bit for bit.  rowscale covers ALL rows: the caller expands scale_edge * norm[r] on edge rows and scale_self on
 * self rows once per graph (in float32, as the epilogue multiplies them).  One [rows, D] write per MixedOp and two reads of its
 * backward less; s is the MixedOp's input state.
 * In mrg_mix_bwd_apply, when rs_on[k] == 2 for this k, fold_s[k] must be `s` and fold_gate[k] must be y_host[k]. */
typedef struct mrg_gated_branch {
  int32_t k;               /* the gated candidate, < 0: none */
  const float *s;          /* [rows, D] device */
  const float *rowscale;   /* [rows] device */
  /* The ROW-SCALED candidate (f_sparse_op_comp, reference models/operations_lp.py:304-343: a scalar gate per row), never stored
   * either: y_row_k[r][c] = s[r][c] * row_f[r] with row_f from mrg_gate_row_fwd (bit for bit mrg_gate_fwd's output);
   * y_host[row_k] must be `s` itself.  row_k < 0: none. */
  int32_t row_k;
  const float *row_f;      /* [rows] device */
  /* mrg_mix_bwd_apply only (the other entry points ignore them).  The candidate gets NO gradient tensor (gy_host[row_k] must be
   * NULL): with gy its gradient w.r.t. y, the kernel writes row_dq[r] = sum_c gy * s -- the gradient w.r.t. row_f[r], which
   * mrg_gate_row_bwd turns into the parameter and s_in gradients -- and, with dz_r = row_dq[r] * row_h[r], ADDS the gradient
   * w.r.t. s, gy * row_f[r] + dz_r * u_seg[c], into fold_gs[k] of the gated candidate k (required: rs_on[k] == 2; both are
   * gradients w.r.t. the rows s).
   * row_uvc: mrg_gate_collapse3's [3][row_ld] vectors; rows [0, b0) use segment 0, [b0, b1) 1, [b1, rows) 2.  D <= 256. */
  const float *row_h;      /* [rows] device */
  const float *row_uvc;
  int32_t row_ld;
  int64_t b0, b1;
  float *row_dq;           /* [rows] device, out */
  /* ABI 14: the activation behind the BatchNorm in mrg_mix_fwd / _bwd_reduce / _bwd_apply: 0 = ReLU (the MixedOp, reference
   * models/cell_lp.py:25-33), 1 = tanh (CompGraphConv's BatchNorm -> tanh tail, reference models/compgcn.py:100-111).  A descriptor
   * with k < 0 and row_k < 0 carries only this field (s may then be NULL); tanh needs exactly that (no recomputed candidate)
   * and K <= 5, else MRG_E_SHAPE. */
  int32_t act;
} mrg_gated_branch;
int mrg_mix_colstats(const float *const *y_host, int K, int64_t rows, int D, double *sums, void *ws,
                     const mrg_gated_branch *gated, void *stream);
int mrg_mix_finalize_fwd(const double *sums, const float *const *gamma_host, const float *const *beta_host,
                         float *const *rmean_host, float *const *rvar_host, int K, double total_rows, int D,
                         float eps, float momentum, float *coef, void *stream);
/* mrg_mix_colstats + mrg_mix_finalize_fwd in two launches instead of three (the ordered reduction of the per-block
 * statistics and the finalize step are one kernel): for callers without a collective between the two -- the single-GPU
 * step.  Bit-identical coefficients and running statistics.  ws: mrg_mix_workspace_bytes(K, D). */
int mrg_mix_stats_coef(const float *const *y, const float *const *gamma, const float *const *beta,
                       float *const *running_mean, float *const *running_var, int K, int64_t rows, double total_rows,
                       int D, float eps, float momentum, float *coef, void *ws, const mrg_gated_branch *gated, void *stream);
/* out = (addend ? addend : 0) + sum_k w[k] ReLU(y_k scale_k + shift_k).  addend (may be NULL, may NOT alias out): the output
 * of the MixedOp this one is summed with -- a state fed by several MixedOps (reference models/cell_lp.py:104-113) is
 * accumulated here instead of by separate full-size add kernels. */
int mrg_mix_fwd(const float *const *y_host, int K, const float *coef, const float *w, const float *addend, float *out,
                int64_t rows, int D, const mrg_gated_branch *gated, void *stream);
int mrg_mix_bwd_reduce(const float *g, const float *const *y_host, int K, const float *coef, const float *w,
                       float *red, void *ws, int64_t rows, int D, const mrg_gated_branch *gated, void *stream);
int mrg_mix_finalize_bwd(const float *red, int K, double total_rows, int D, float *coef2,
                         float *const *dgamma_host, float *const *dbeta_host, float *dw, void *stream);
/* rs_on (HOST int[K], may be NULL = none): candidate k's output gradient is written already multiplied by its consumer's row
 * scale, gy_k[r] *= r < rs_edge_rows[k] ? rs_scale[k] * (rs[k] ? rs[k][r] : 1) : rs_self[k] -- exactly the `dz = g * c` pass of
 * f_comp_op's backward (mrg_dense_filter_dz, kind 1), which the caller then skips (bit-identical values). */
/* rs_on[k] == 2: the gated form, mrg_dense_filter_dz kind 0 of f_dense_op_comp's backward: with gc = gy * c, gy_k receives
 * dz = gc * s * gate * (1 - gate) and fold_gs[k] the direct term gc * gate (fold_s / fold_gate / fold_gs: HOST arrays of K
 * device pointers, used for those k only). */
int mrg_mix_bwd_apply(const float *g, const float *const *y_host, float *const *gy_host, int K,
                      const float *coef, const float *coef2, const float *w, const float *const *rs, const float *rs_scale,
                      const float *rs_self, const int64_t *rs_edge_rows, const int *rs_on, const float *const *rs_full,
                      const float *const *fold_s,
                      const float *const *fold_gate, float *const *fold_gs, const int *fold_add_from, int64_t rows, int D,
                      const mrg_gated_branch *gated, void *stream);
/* rs_full (HOST array of K device pointers, NULL or NULL entries = none): candidate k's multiplier expanded over ALL rows
 * (rs_scale[k] * rs[k][r] on the edge rows, rs_self[k] on the others), which the kernel then loads with the candidates' rows
 * instead of deriving it from rs / rs_scale / rs_self / rs_edge_rows between the arithmetic and the stores. */
/* fold_add_from (HOST array of K ints, NULL = none): for a gated candidate k (rs_on[k] == 2), the index q of the candidate
 * whose OUTPUT is k's operand s (f_identity of the same MixedOp: y_host[q] == fold_s[k]); its gradient gy_q is added to
 * fold_gs[k] and gy_host[q] may be NULL -- both are gradients w.r.t. the same rows. */

/* ---- Cell zero: the MixedOp over the compose candidates, recomputed from the tables ---------------------------------------------
 * reference models/cell_lp.py:53-68 (Cell_Zero: ONE MixedOp over PRE_OPS), :25-33 (MixedOp.forward / op_forward),
 * models/operations_lp.py:71-98 (pre_mult / pre_sub / pre_add) and the gather feeding them, models/model_search_lp.py:135-145:
 *   out[r] = sum_k w[k] * ReLU(BN_k(ent[ent_idx[r]] (op_k) rel[rel_idx[r]]))          ops: MRG_COMPOSE_*, 1 <= K <= 3
 * The candidates are elementwise functions of two cache-resident table rows, so no candidate output is ever stored: the
 * statistics, combine and gradient passes recompute them, and the backward writes the two combined per-row gradients
 *   g_ent_rows[r] = sum_k gy_k * d y_k / d ent-row,   g_rel_rows[r] = sum_k gy_k * d y_k / d rel-row
 * which two mrg_span_gcs(COPY) launches turn into the table gradients.  The entry points mirror mrg_mix_colstats /
 * mrg_mix_stats_coef / mrg_mix_fwd / mrg_mix_bwd_reduce / mrg_mix_bwd_apply (same workspaces, same coef / red / coef2 layouts,
 * mrg_mix_finalize_fwd / _bwd in between, statistics may be all-reduced when the rows are sharded; ws: mrg_zero_workspace_bytes(D)).
 * Same values as the stored form; the statistics are summed over more, smaller blocks (their order differs in the last bits).  ops: HOST array of K codes. */
int64_t mrg_zero_workspace_bytes(int D);      /* ws of mrg_zero_colstats / _stats_coef / _bwd_reduce (up to 2048 blocks of partials) */
int mrg_zero_colstats(const float *ent, const float *rel, const int32_t *ent_idx, const int32_t *rel_idx, const int *ops, int K,
                      int64_t rows, int D, double *sums, void *ws, void *stream);
int mrg_zero_stats_coef(const float *ent, const float *rel, const int32_t *ent_idx, const int32_t *rel_idx, const int *ops, int K,
                        const float *const *gamma, const float *const *beta, float *const *running_mean,
                        float *const *running_var, int64_t rows, double total_rows, int D, float eps, float momentum,
                        float *coef, void *ws, void *stream);
int mrg_zero_fwd(const float *ent, const float *rel, const int32_t *ent_idx, const int32_t *rel_idx, const int *ops, int K,
                 const float *coef, const float *w, float *out, int64_t rows, int D, void *stream);
int mrg_zero_bwd_reduce(const float *g, const float *ent, const float *rel, const int32_t *ent_idx, const int32_t *rel_idx,
                        const int *ops, int K, const float *coef, const float *w, float *red, void *ws, int64_t rows, int D,
                        void *stream);
int mrg_zero_bwd_apply(const float *g, const float *ent, const float *rel, const int32_t *ent_idx, const int32_t *rel_idx,
                       const int *ops, int K, const float *coef, const float *coef2, const float *w, float *g_ent_rows,
                       float *g_rel_rows, int64_t rows, int D, void *stream);


/* ---- dense linear on edge / node rows (fp32 MFMA) ---------------------------
 * nn.Linear inside a_max_op / a_mean_op (reference models/operations_lp.py:228,231,246)
 * and the post-aggregation linears of CompGraphConv (reference models/compgcn.py:77-78,100,103).
 * Y[rows, Nout] = act(X[rows, K] W[Nout, K]^T + bias)
 *
 * Two matrix cores serve every tall-skinny product of the library (this one, mrg_linear_bwd_input,
 * mrg_dense_filter_fwd):
 *   - split core: each f32 operand is written as the sum of three bf16 numbers (error <= 2^-26 relative) and the
 *     product is accumulated in f32 from the six leading cross terms on the bf16 matrix pipe -- same error class
 *     as an f32 FMA chain (tests pin it against float64 next to the exact core), 2-3x the throughput.  Needs a
 *     workspace of mrg_gemm_workspace_bytes(K, Nout) bytes for the pre-split weight, K % 4 == 0, K > 48 and
 *     16-byte aligned rows;
 *   - exact core: v_mfma_f32_32x32x2_f32; taken when ws == NULL, the operands do not qualify, or after
 *     mrg_gemm_set_mode(1).
 * ws: NULL, or mrg_gemm_workspace_bytes(K, Nout) bytes of device memory private to this call. */
int64_t mrg_gemm_workspace_bytes(int K, int Nout);
/* 0 (default): split core where possible, on the kernel that shares the pre-split weight slabs of a 128-row workgroup through
 * LDS (two workgroups per CU); 1: exact-f32 core only; 2: split core on the wave-autonomous one-wave-per-SIMD kernel (the
 * default of rounds 1-2; also what mode 0 falls back to for operands the LDS-weight kernel does not take): a tested comparison
 * point with bit-identical results.  Modes 3 / 4 of rounds 2-3 (persistent and two-waves-per-SIMD kernels)
 * left the library in round 4 (tools/lab/).  Process-wide. */
int mrg_gemm_set_mode(int mode);
/* Store order of the split-core row GEMM's elementwise epilogues (bias / activation, gate, scale, accumulate):
 * 0 (default): accumulator-order 4-byte stores (two 128-byte row pieces per store instruction);
 * 2 (round 4): the LDS-weight kernel exchanges the MFMA operands, so that a lane holds one output row's 4-column chunks and
 * every epilogue load and store is 16 bytes per lane with no LDS round trip (needs N % 4 == 0 and 16-byte aligned rows of every
 * [rows, N] operand and of the bias; otherwise, and for the fused aggregators' epilogues, mode 0) -- bit-identical, measured
 * 8-30 % slower per launch, kept as a tested comparison point;
 * 1: (one-wave kernel, mrg_gemm_set_mode(2)) a 32-row strip of results goes through wave-private LDS and leaves as
 * row-order 16-byte stores (1 KB of consecutive addresses per store instruction; the gate multiplicand / accumulate input are
 * read the same way) whenever N % 4 == 0 and the rows of every [rows, N] operand are 16-byte aligned.  Bit-identical results;
 * measured slower for the plain epilogue and in the whole step, kept as a tested comparison point.
 * Process-wide. */
int mrg_gemm_set_epilogue(int mode);
/* 1 (default): a split-core row GEMM whose output is eight 32-column tiles wide (225 .. 256 columns: D = 256) and that has no
 * direction groups, a single source and no gate epilogue runs as ONE column block (128 accumulator registers per lane, a weight ring of
 * two slabs): the activation operand is read once.  0: two four-tile column blocks (rounds 1-3).  Bit-identical.  Process-wide. */
int mrg_gemm_set_wide8(int on);
/* 1 (default, round 5): a split-core row GEMM of 129 .. 224 output columns over a reduction dimension K > 224 (D = 200: the dense
 * filters of models/operations_lp.py:266-288,356-390 forward on two operands, K = 2 D, and the paired input gradient of a MixedOp's
 * two dense candidates) whose epilogue is not a fused aggregator runs on rowgemm_x3q_k (csrc/gemm_x3q.hpp): 16 x 16 x 32 MFMA tiles,
 * 16 rows per wave, 64-row workgroups at THREE per CU.  2 (lab, tests): every K > 48.  0: rowgemm_x3s_k for all of them (rounds 3-4:
 * 32 x 32 x 16 tiles, two workgroups per CU).  Same six cross terms in the same order per accumulator; sums over k are formed 32 at a
 * time instead of 16, so the two agree to rounding (both pinned <= 1.5 x the exact-f32 core's error against float64).  Measured:
 * -5 .. -12 % per launch at K = 400, equal at K = 200 (profiles/r5_rowgemm_q.txt).  Process-wide. */
int mrg_gemm_set_q(int on);
/* ABI 14.  1 (default): split-core products of at most 16 384 rows (a sampled step graph, a rank's node chunk) run on the wave-
 * autonomous kernel with two-tile column blocks -- 3.5 x more waves with a 3.5 x shorter instruction chain each; same k-order per
 * output element, bit-identical results.  0: one kernel for every row count.  on > 1 (lab): the row bound itself. */
int mrg_gemm_set_small(int on);
/* ABI 14 (lab).  mrg_linear_bwd_weight3 sizes the row blocks of its (up to three) ranges so that the ranges TOGETHER get about one
 * workgroup per CU.  A caller that launches the ranges one by one with mrg_linear_bwd_weight and wants the SAME partial sums (bit
 * for bit) announces the number of ranges (1..3) around those calls; 1 (default) = a launch is alone. */
int mrg_wgrad_set_share(int n);
/* The split-core weight gradient (mrg_linear_bwd_weight / _weight3): 1 (default) = every 32-column x 16-row operand fragment is
 * split into its bf16 planes ONCE per workgroup and shared through LDS (wgrad_x3v_k), 0 = by every wave that multiplies it
 * (wgrad_x3_k, rounds 1-2).  Same operands and products in the same order: bit-identical gradients for any shape. */
int mrg_wgrad_set_variant(int variant);
int mrg_linear_fwd(const float *X, const float *W, const float *bias, float *Y, void *ws,
                   int64_t rows, int K, int Nout, int act, void *stream);
/* a_max_op.forward as ONE GEMM (reference models/operations_lp.py:230-235; SURVEY section 2b K_lin_relu_segmax):
 *   out[v] = max over in-edges e of v of ReLU(X[e] W^T + bias)  + self_rows[v]        (0 + self row without in-edge)
 * The split-core GEMM walks the edge rows in destination order (eid = the by-destination edge list of the chunk plan,
 * dst[e] = destination of edge e) and its epilogue reduces ReLU(acc + bias) over the runs of equal destination in
 * registers, publishing 64-bit atomic maxima of (value bits, 0xFFFFFFFF - list position): exact, order-independent,
 * lowest edge id among equal values (DGL's argmax).  The [E, Nout] message tensor is never written.
 * arg [N, Nout] int32 (winning edge id, -1 without in-edge) and mx [N, Nout] (the maximum itself: the backward's ReLU
 * mask is mx > 0) are optional.  ws: mrg_linear_relu_segmax_workspace_bytes(N, K, Nout) bytes; that function returns 0
 * when the split core cannot take the shape (K <= 48 or K % 4 != 0): use mrg_linear_fwd + mrg_seg_reduce_fwd then. */
int64_t mrg_linear_relu_segmax_workspace_bytes(int64_t N, int K, int Nout);
int mrg_linear_relu_segmax_fwd(const float *X, const float *W, const float *bias, const int32_t *eid, const int32_t *dst,
                               const float *self_rows, float *out, int32_t *arg, float *mx, void *ws,
                               int64_t E, int64_t N, int K, int Nout, void *stream);
/* a_mean_op.forward (reference models/operations_lp.py:245-250; SURVEY section 2b K_lin_relu_segmean) without the [E, Nout]
 * message tensor, in two launches:
 *   1. mrg_linear_relu_segsum_fwd: the split-core GEMM over the edges in destination order; its epilogue adds
 *      ReLU(X[e] W^T + bias) over the runs of equal destination inside each lane's 16 rows of a 32-row strip (a fixed
 *      order) and stores a run's sum at the position of its first row: part [E, Nout], only those "head" rows are written.
 *      relu_bits [E, ceil(Nout / 32)] keeps y > 0 per element of ORIGINAL edge row e (bit = column % 32) for the backward.
 *   2. mrg_seg_reduce_heads_fwd: out[v] = (sum of v's head rows, ascending) / max(deg, 1) + self_rows[v] (MRG_REDUCE_MEAN;
 *      MRG_REDUCE_SUM without the division) over the chunk plan; rowptr [N + 1] = list starts.  Which rows are heads
 *      follows from the list positions alone.  Deterministic: no float atomics.
 * Backward of the reducer with the bit mask: mrg_seg_reduce_bwd_bits.  Shapes as mrg_linear_relu_segmax_fwd. */
int mrg_linear_relu_segsum_fwd(const float *X, const float *W, const float *bias, const int32_t *eid, const int32_t *dst,
                               float *part, unsigned *relu_bits, void *ws, int64_t E, int K, int Nout, void *stream);
int mrg_seg_reduce_heads_fwd(int mode, const float *part, const float *self_rows, const int32_t *rowptr,
                             const int32_t *chunk_node, const int32_t *chunk_start, const int32_t *chunk_end,
                             const int32_t *chunk_slot, int64_t n_chunks, const int32_t *hub_node, const int32_t *hub_first,
                             const int32_t *hub_count, int64_t n_hubs, int64_t n_slots, const int32_t *in_degree, float *out,
                             void *ws, int64_t N, int D, void *stream);
int mrg_seg_reduce_bwd_bits(int mode, const float *gout, const int32_t *dst, const int32_t *in_degree, float *gmsg, float *gself,
                            const unsigned *relu_bits, int64_t E, int64_t N, int D, void *stream);
/* gX[rows, K] (+)= gY[rows, Nout] W[:, 0:K]   (gY already masked by the activation).  W is
 * [Nout][ldw] row-major, ldw >= K (a column block of a wider weight, e.g. one half of an
 * nn.Linear(2D, D)); accumulate != 0 adds into gX.  ws (mandatory) holds the split / transposed block. */
int64_t mrg_linear_bwd_input_workspace_bytes(int K, int Nout);
int mrg_linear_bwd_input(const float *gY, const float *W, float *gX, void *ws,
                         int64_t rows, int K, int Nout, int ldw, int accumulate, void *stream);
/* gW[Nout, K1+K2] = gY^T [X1 | X2]  (X2 NULL / K2 = 0: single source; the reference's
 * torch.cat([s, s_in], 1) is never materialised),  gbias[Nout] = column sums of gY (NULL ok). */
int64_t mrg_linear_bwd_weight_workspace_bytes(int64_t rows, int K, int Nout);
int mrg_linear_bwd_weight(const float *gY, const float *X1, const float *X2, float *gW, float *gbias, void *ws,
                          int64_t rows, int K1, int K2, int Nout, void *stream);

/* ---- dense (per-feature) filters, one direction segment per call ------------------
 * f_dense_op_comp / f_comp_op / f_dense_op_last / f_dense_op .forward,
 * reference models/operations_lp.py:356-390, 266-288, 392-401, 345-354.
 *   z = [s | s_in] W^T + bias        (s_in NULL: z = s W^T + bias; W is [D, 2D] or [D, D])
 *   kind 0:  out = sigmoid(z) * s * c      (gate [rows, D] receives sigmoid(z) for the backward)
 *   kind 1:  out = z * c
 *   c = scale * (rowscale ? rowscale[row] : 1)     (the reference's 1/3 and edge norm)
 * torch.cat([s, s_in], 1) is never materialised: the GEMM reads both sources; gate, scale and
 * norm are applied in its epilogue.  ws: NULL or mrg_gemm_workspace_bytes(K, D) bytes, K = D or 2D (see mrg_linear_fwd). */
int mrg_dense_filter_fwd(int kind, const float *s, const float *s_in, const float *W, const float *bias,
                         const float *rowscale, float scale, float *out, float *gate, void *ws,
                         int64_t rows, int D, void *stream);
/* The three direction segments of one dense filter in ONE launch each (weight split + grouped row GEMM) instead of three:
 * rows [0, b0) use W[0] / bias[0], [b0, b1) W[1] / bias[1] (edge rows: c = scale_edge * norm[row]), [b1, M) W[2] / bias[2]
 * (self rows: c = scale_self).  W / bias: HOST arrays of three device pointers (bias entries may be NULL).  Split core
 * only: mrg_dense_filter3_workspace_bytes(D, K) returns 0 when the shape does not qualify (then use the per-segment
 * entry point).  Same arithmetic per row as mrg_dense_filter_fwd: results are bit-identical.
 * kind 0 with out == NULL: only the gate is stored; the MixedOp epilogue recomputes the output (mrg_gated_branch). */
int64_t mrg_dense_filter3_workspace_bytes(int D, int K);
int mrg_dense_filter_fwd3(int kind, const float *s, const float *s_in, const float *const *W, const float *const *bias,
                          const float *norm, float scale_edge, float scale_self, float *out, float *gate, void *ws,
                          int64_t b0, int64_t b1, int64_t M, int D, void *stream);
/* dz (and the direct term of gs) for all M rows of the three segments in one launch: rows [0, b1) c = scale_edge * norm[row],
 * rows [b1, M) c = scale_self (see mrg_dense_filter_dz). */
int mrg_dense_filter_dz3(int kind, const float *g, const float *s, const float *gate, const float *norm, float scale_edge,
                         float scale_self, float *dz, float *gs, int64_t b1, int64_t M, int D, void *stream);
/* mrg_linear_bwd_input / mrg_linear_bwd_weight for the three row ranges [0, b0) [b0, b1) [b1, M) with their own weights
 * W[0..2] (HOST array of device pointers, each [Nout][ldw]) in one launch each.  The *_workspace_bytes queries return 0
 * when the split core cannot take the shape or mrg_gemm_set_mode(1) is active: use the per-range entry points then. */
int64_t mrg_linear_bwd_input3_workspace_bytes(int K, int Nout);
int mrg_linear_bwd_input3(const float *gY, const float *const *W, float *gX, void *ws, int64_t b0, int64_t b1, int64_t M,
                          int K, int Nout, int ldw, int accumulate, void *stream);
/* The input gradient of TWO candidates that read the same rows as one product over the concatenated reduction dimension:
 * gX rows of range s (+)= [gY1 | gY2] [W1[s][:, 0:K] ; W2[s][:, 0:K]] -- f_dense_comp and f_comp of one MixedOp (reference
 * models/cell_lp.py:95-113; models/operations_lp.py:266-288, 356-390) leave ONE gradient w.r.t. their shared operand instead
 * of two that a fan-in pass would add.  gY1, gY2: [M, Nout]; W1, W2: HOST arrays of three device pointers, each [Nout][ldw].
 * Split core only (the workspace query returns 0 otherwise). */
int64_t mrg_linear_bwd_input3_pair_workspace_bytes(int K, int Nout);
int mrg_linear_bwd_input3_pair(const float *gY1, const float *gY2, const float *const *W1, const float *const *W2, float *gX,
                               void *ws, int64_t b0, int64_t b1, int64_t M, int K, int Nout, int ldw, int accumulate,
                               void *stream);
int64_t mrg_linear_bwd_weight3_workspace_bytes(int64_t b0, int64_t b1, int64_t M, int K1, int K2, int Nout);
int mrg_linear_bwd_weight3(const float *gY, const float *X1, const float *X2, float *const *gW, float *const *gbias, void *ws,
                           int64_t b0, int64_t b1, int64_t M, int K1, int K2, int Nout, void *stream);
/* backward, step 1:  kind 0: dz = g*s*c*gate*(1-gate), gs = g*c*gate (direct term);  kind 1: dz = g*c.
 * Steps 2-3 are mrg_linear_bwd_input (gs += dz W[:, :D]; gs_in = dz W[:, D:]) and
 * mrg_linear_bwd_weight (gW = dz^T [s | s_in], gbias = column sums of dz). */
int mrg_dense_filter_dz(int kind, const float *g, const float *s, const float *gate, const float *rowscale,
                        float scale, float *dz, float *gs, int64_t rows, int D, void *stream);

/* ---- f2: graph construction, edge ordering and index plans on the device ------------------
 * build_graph_from_triplets + comp_deg_norm  reference utils/utils_rgcn.py:120-158
 * node_norm_to_edge_norm                     reference search/mr_lp_search.py:30-36
 * build_graph                                reference train/mr_lp_train.py:77-89
 * Integer work: bit-exact with the reference's numpy formulation.
 *
 * mrg_build_graph: triples [T][3] int64 (s, r, o) -> E = 2T directed edges (e < T: s -> o with relation r; e >= T:
 * o -> s with r + R).  sorted != 0: edges ordered by (relation, dst, src) like `sorted(zip(rel, dst, src))`
 * (utils_rgcn.py:151); sorted == 0: the train driver's un-sorted halves.  in_degree [N] counts edges per destination.
 * norm[e] = deg_norm_table[in_degree[dst]] * deg_norm_table[in_degree[src]] where the HOST supplies
 * deg_norm_table[d] = float32(d) ** float32(-0.5), 0 for d = 0 (numpy's own values: a device pow / rsqrt may differ
 * in the last bit); the caller must make table_len exceed the largest in-degree (max_degree [1] reports it).
 * src32 / dst32 / etype32 (NULL ok): int32 copies for the kernels.  norm NULL skips the norm (and the int32 copies). */
int64_t mrg_build_graph_workspace_bytes(int64_t T);
int mrg_build_graph(const int64_t *triples, int64_t T, int64_t N, int R, int sorted,
                    const float *deg_norm_table, int64_t table_len,
                    int64_t *src, int64_t *dst, int64_t *etype, float *norm, int32_t *in_degree,
                    int32_t *src32, int32_t *dst32, int32_t *etype32, int32_t *max_degree,
                    void *ws, int64_t ws_bytes, void *stream);

/* Span plan of mrg_span_gcs for segment ids seg [E] in [0, nseg): perm [E] = stable argsort, seg_sorted [E],
 * seg_len [nseg], span_slot [2 * n_spans] (n_spans = ceil(E / span)), hub_seg / hub_first / hub_count with capacity
 * 2 * n_spans + nseg, counts [2] = {n_hubs, n_slots} (device memory; the host reads it once).  Same results as the
 * tensor formulation it replaces (mr-gnas_amd/graph.py:span_plan), which stays as the test's cross-check. */
int64_t mrg_plan_workspace_bytes(int64_t E, int64_t nseg, int span);
int mrg_span_plan_build(const int32_t *seg, int64_t E, int64_t nseg, int span, int snap,
                        int32_t *perm, int32_t *seg_sorted, int32_t *seg_len, int32_t *span_slot, int32_t *span_start,
                        int32_t *hub_seg, int32_t *hub_first, int32_t *hub_count, int32_t *counts,
                        void *ws, int64_t ws_bytes, void *stream);
/* span_start [n_spans + 1] (ABI 8): span i covers sorted elements [span_start[i], span_start[i + 1]).  A cut is nominally
 * i * span; where that position falls inside a segment it moves to the nearer end of that segment if that is at most `snap`
 * (0 .. span / 4; 0 = fixed spans) elements away, so segments up to 2 * snap long are never split over spans (no partial runs,
 * workspace slots or hub-pass work for them) and every span stays within +- 2 * snap of the nominal length.  The last span may
 * be empty. */
/* meta[j] = {seg_sorted[j], xi[perm[j]] (perm[j] if xi NULL), yi[perm[j]] (0 if NULL), w} with w = float bits of
 * scal[perm[j]] (1.0f if scal NULL), or perm[j] itself when w_is_index != 0 (external per-call scales). */
int mrg_span_meta_pack(const int32_t *perm, const int32_t *seg_sorted, const int32_t *xi, const int32_t *yi,
                       const float *scal, int w_is_index, void *meta, int64_t E, void *stream);

/* Chunk plan of mrg_seg_reduce_fwd / mrg_fused_gcs (CSR by destination, lists cut into chunks of `chunk`):
 * eid [E], rowptr [N + 1], in_degree [N], chunk_* with capacity N + E / chunk + 1, hub_* with capacity
 * E / chunk + 1, counts [3] = {n_chunks, n_hubs, n_slots}. */
int64_t mrg_chunk_plan_workspace_bytes(int64_t E, int64_t N);
int mrg_chunk_plan_build(const int32_t *dst, int64_t E, int64_t N, int chunk,
                         int32_t *eid, int32_t *rowptr, int32_t *in_degree,
                         int32_t *chunk_node, int32_t *chunk_start, int32_t *chunk_end, int32_t *chunk_slot,
                         int32_t *hub_node, int32_t *hub_first, int32_t *hub_count, int32_t *counts,
                         void *ws, int64_t ws_bytes, void *stream);

/* ---- f4: sampling-side data preparation on the device ----------------------------------------
 * Random draws are inputs (the host draws them on the device; tests replay numpy's), so each call is deterministic and
 * bit-exact with the reference for the same draws.
 *
 * negative_sampling, reference utils/utils_rgcn.py:191-204: samples [(rate+1) B][3]: rows [0, B) = pos (label 1), row
 * B + j = pos[j % B] with the subject (choices[j] > 0.5) or the object replaced by values[j] (label 0). */
int mrg_negative_sampling(const int64_t *pos, int64_t B, int rate, const int64_t *values, const double *choices,
                          int64_t *samples, float *labels, void *stream);
/* `uniq_v, edges = np.unique((src, dst), return_inverse=True)`, reference utils/utils_rgcn.py:97-101: uniq [<= min(2n,
 * num_nodes)] = the distinct node ids in ascending order, new_src / new_dst their ranks, count [1] (device) = len(uniq). */
/* sample_edge_neighborhood (reference utils/utils_rgcn.py:30-71; `--edge_sampler neighbor`, search/mr_lp_search.py:323-324):
 * sample_size dependent picks by neighbourhood expansion in ONE launch of one persistent workgroup.  Adjacency in the
 * reference's append order (get_adj_and_degrees, :18-28) as a CSR: rowptr [N + 1], adj_edge / adj_other [2 T] (triple id and the
 * vertex at its other end), degrees [N].  Draws are inputs: u_vertex [sample_size] float64 uniforms for the vertex picks
 * (inverse cdf over sample_counts * seen), and EITHER tries [n_tries] -- the reference's own sequence of tried adjacency slots,
 * rejected ones included (replay: bit-exact) -- OR u_edge [sample_size] float64 uniforms (one draw per pick selects uniformly
 * among the vertex's unpicked entries: the reference's distribution without its rejection loop).  edges [sample_size] int32.
 * status [3] int64: picks made, tries consumed, failure code (0 ok; 1 graph exhausted; 2 / 3 bad or missing tries).
 * ws: mrg_sample_neighborhood_workspace_bytes(N, T). */
int64_t mrg_sample_neighborhood_workspace_bytes(int64_t N, int64_t T);
int mrg_sample_edge_neighborhood(const int32_t *rowptr, const int32_t *adj_edge, const int32_t *adj_other, const int32_t *degrees,
                                 int64_t N, int64_t T, int64_t sample_size, const double *u_vertex, const int64_t *tries,
                                 int64_t n_tries, const double *u_edge, int32_t *edges, int64_t *status, void *ws,
                                 int64_t ws_bytes, void *stream);
int64_t mrg_relabel_workspace_bytes(int64_t num_nodes);
int mrg_relabel_nodes(const int64_t *src, const int64_t *dst, int64_t n, int64_t num_nodes, int64_t *uniq,
                      int64_t *new_src, int64_t *new_dst, int32_t *count, void *ws, int64_t ws_bytes, void *stream);
/* Dense targets of a (subject, relation) batch: process() + TrainDataset / TestDataset.get_label (+ label smoothing),
 * reference utils/process_data.py:4-31, utils/data_set.py:15-33.  keys [U] ascending = subject * (2R) + relation of the
 * known pairs, rowptr [U + 1] / objs = their object lists; out [B][num_ent] = v_zero everywhere, v_one at the objects of
 * query[b] (a query without a list gives an all-v_zero row). */
int mrg_multi_hot_labels(const int64_t *keys, const int32_t *rowptr, const int32_t *objs, const int64_t *query,
                         int64_t B, int64_t U, int64_t num_ent, float v_zero, float v_one, float *out, void *stream);

/* ---- f3: filtered ranking and the [B, N] score functions ---------------------------------------------
 * predict(), reference train/mr_lp_train.py:290-299: entries with a non-zero label are pushed to -1e7, the target keeps
 * its score; ranks[b] = 1 + #(greater) + #(equal at a lower index)  (the position in a stable descending sort). */
int mrg_rank_filtered(const float *pred, const float *labels, const int64_t *obj, int64_t B, int64_t N,
                      int64_t *ranks, void *stream);
/* sf_TransE_op.forward, reference models/operations_lp.py:101-112:
 *   score[b, n] = sigmoid(gamma - sum_c |sub[b, c] + rel[b, c] - ent[n, c]|)
 * backward: gobj [B][D] (the gradient of both sub and rel) and gent [N][D]; either may be NULL.
 * (sf_DisMult_op, :115-127, is mrg_compose_fwd(MULT) + mrg_linear_fwd(act = MRG_ACT_SIGMOID).) */
int mrg_transe_score_fwd(const float *ent, const float *sub, const float *rel, float gamma, float *score,
                         int64_t B, int64_t N, int D, void *stream);
int mrg_transe_score_bwd(const float *ent, const float *sub, const float *rel, const float *gscore, const float *score,
                         float *gent, float *gobj, int64_t B, int64_t N, int D, void *stream);
/* gT[n][b] = g[b][n] * act'(y[b][n]): the output gradient of a wide, short Linear (the [B, N] scorers: N = all entities) with the
 * activation's derivative folded in, transposed to [N][B] rows -- the layout both of its gradient products stream
 * (functional._Linear.backward; replaces torch's mul / rsub / mul / strided copy).  act: MRG_ACT_NONE (y may be NULL) / _RELU /
 * _SIGMOID, y = the forward's output.  B <= 64 * 65535. */
int mrg_act_grad_transpose(const float *g, const float *y, float *gT, int64_t B, int64_t N, int act, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MRGNAS_H */
