/* gcnn_hip.h -- C ABI of libgcnn_hip.so: the MI355X (gfx950) implementation of the bipartite GCNN hot path of
 * stefanvanberkum/gcnn-cut-selector.
 *
 * The reference has no native layer and no FFI: its hot path is Python calling TensorFlow ops
 * (/root/reference/model.py).  Each entry point below therefore cites the reference *Python* interface whose
 * arithmetic it replaces.  All pointers are DEVICE pointers unless marked "host"; all matrices are row-major fp32,
 * indices int32; `stream` is a hipStream_t passed as void*.  Every function returns 0 on success, a negative
 * GCNN_E_* code on bad arguments and a positive hipError_t on a HIP failure; no function allocates device memory
 * (workspaces are passed in), synchronises the device or throws.
 */
#ifndef GCNN_HIP_H
#define GCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCNN_EMB 64
#define GCNN_N_PARAMS 62          /* arrays in a checkpoint, model.py:53-56 */
#define GCNN_E_BADARG (-1)
#define GCNN_E_WORKSPACE (-2)
#define GCNN_E_UNSUPPORTED (-4)   /* sizes outside what a specialised entry point handles: use the general path */
#define GCNN_E_HIP (-3)           /* a HIP call failed inside an entry point whose return value is a count */

/* ---- parameter layout --------------------------------------------------------------------------------------
 * All 62 model variables live in ONE flat fp32 buffer, in the reference's checkpoint order
 * (model.py:53-56, 215; shapes model.py:174-208, 486-508), each tensor starting at a multiple of 4 floats.
 * Gradients and Adam moments use the same layout. */
int gcnn_abi_version(void);
int gcnn_param_count(void);                                   /* 62 */
int gcnn_param_total_floats(void);                            /* size of the flat buffer */
int gcnn_param_info(int index, int* offset, int* rows, int* cols, int* trainable);

/* ---- per-launch timing (a measuring aid; bench.py's roofline_step) ------------------------------------------------
 * Between gcnn_profile_begin() and gcnn_profile_end() every kernel launch the library makes is bracketed by two HIP events
 * on its stream.  gcnn_profile_end waits for those events (the ONE entry point that synchronises), stores up to `capacity`
 * kernel names (static strings) and durations in milliseconds, launch by launch, and returns the number of launches seen
 * (possibly > capacity; at most 512 are recorded), or GCNN_E_HIP.  Single-threaded use only, one session at a time, all launches of
 * a session on streams of the device that was current at gcnn_profile_begin (events are per device). */
int gcnn_profile_begin(void);
int gcnn_profile_end(int32_t capacity, const char** names /* host, optional */, float* ms /* host, optional */);

/* ---- sizes ------------------------------------------------------------------------------------------------ */
typedef struct gcnn_dims {
    int32_t n_cons, n_vars, n_cuts;        /* TOTAL node counts of the stacked batch (model.py:273-275) */
    int32_t n_cons_edges, n_cut_edges;     /* E1, E2 */
} gcnn_dims;

/* One edge set in both receiver orders (built by gcnn_graph_build): by-left CSR and by-variable CSR. */
typedef struct gcnn_graph {
    const int32_t* l_ptr;   /* [n_left+1]  segment offsets, edges grouped by left (constraint/cut) node */
    const int32_t* l_oth;   /* [E]         variable index of each edge, by-left order */
    const float*   l_coef;  /* [E]         raw edge feature, by-left order */
    const int32_t* v_ptr;   /* [n_var+1]   segment offsets, edges grouped by variable node */
    const int32_t* v_oth;   /* [E]         left index of each edge, by-variable order */
    const float*   v_coef;  /* [E]         raw edge feature, by-variable order */
    int32_t l_max_deg;      /* longest by-left segment, 0 = unknown.  Edge passes give segments that are long for the list's */
    int32_t v_max_deg;      /* longest by-variable segment, 0 = unknown.  mean degree a wave of their own in an extra launch,
                               which is skipped when the list is known to hold none. */
} gcnn_graph;

/* ---- graph plan: COO -> receiver-sorted CSR in both orders -------------------------------------------------
 * Replaces nothing arithmetic in the reference; it is the index structure that lets tf.scatter_nd
 * (model.py:568-569) and the gradients of tf.gather (model.py:564-565) run as atomic-free segmented sums.
 * edge_inds is the reference's [2,E] int32 tensor (row 0 = left id, row 1 = variable id, utils.py:110,234);
 * any edge order is accepted (stable sort => deterministic summation order). */
size_t gcnn_graph_temp_bytes(int32_t n_edges);
/* one pass over the list: flags[0] != 0 <=> some index is out of range; flags[1] != 0 <=> NOT sorted by left id */
int gcnn_graph_check(const int32_t* edge_inds, int32_t n_edges, int32_t n_left, int32_t n_var, int32_t* flags,
                     void* stream);
/* left_sorted != 0 (as established by gcnn_graph_check; the reference's get_state emits (row, col)-sorted lists,
 * utils.py:102-104) skips the by-left sort: the by-left order is then the input order. */
int gcnn_graph_build(const int32_t* edge_inds, const float* edge_feats, int32_t n_edges, int32_t n_left,
                     int32_t n_var, int32_t left_sorted, int32_t* l_ptr, int32_t* l_oth, float* l_coef, int32_t* v_ptr, int32_t* v_oth,
                     float* v_coef, int32_t* l_perm /* optional [E]: by-left position -> input edge id */, void* temp,
                     size_t temp_bytes, void* stream);

/* ---- device-side batch collation: utils.load_batch's stacking (utils.py:389-426) on a device-resident sample store
 * The store keeps every sample's arrays (features, targets and both CSR orders of both edge sets) concatenated in
 * HBM with SAMPLE-LOCAL index values.  Because a mini-batch is a disjoint union, its arrays are the chosen samples'
 * segments laid end to end with indices shifted by the sample's position in the batch (utils.py:401-407) -- CSR
 * included, so no sort runs per batch.  One launch performs all copies.  A job copies `width` 4-byte words per unit
 * (node row / edge / cut); units of sample-slot s go from src unit src_off[unit_kind][s].. to dst unit
 * dst_off[unit_kind][s]..; int32 arrays add dst_off[add_kind][s] (add_kind < 0: plain copy); is_ptr arrays
 * (segment offsets, width 1) get one trailing entry = dst_off[add_kind][batch].
 * jobs: HOST array (<= 24); src_off [n_kinds][batch] and dst_off [n_kinds][batch+1]: DEVICE int64;
 * max_words: the largest job's output size in words (sizes the grid). */
typedef struct gcnn_collate_job {
    const void* src;
    void* dst;
    int32_t unit_kind, width, add_kind, is_ptr;
} gcnn_collate_job;
int gcnn_collate(const gcnn_collate_job* jobs, int32_t n_jobs, const int64_t* src_off, const int64_t* dst_off,
                 int32_t batch, int64_t max_words, void* stream);

/* ---- standalone scatter-sum pass (K9): tf.scatter_nd(updates=[E,64], indices, shape=[R,64]), model.py:568-569
 * seg_ptr[R+1] are receiver-sorted segment offsets; perm (optional) maps sorted position -> row of `msg`
 * (NULL when msg is already receiver-sorted).  out[r] = sum of the segment's rows, 0 for empty segments. */
int gcnn_seg_sum_f32(const float* msg, const int32_t* seg_ptr, const int32_t* perm, int32_t n_recv, float* out,
                     void* stream);
/* its transpose (the gradient of the pass): d_msg[perm[i]] = d_out[recv(i)] */
int gcnn_seg_bcast_f32(const float* d_out, const int32_t* seg_ptr, const int32_t* perm, int32_t n_recv,
                       float* d_msg, void* stream);

/* ---- node GEMMs on the fp32 MFMA: Keras Dense(64) layers, model.py:174-208, 486-508 ---------------------------
 * forward:  y = act( (sa*xa) @ wa [+ xb @ wb] [+ bias] [+ deg (x) bd] ),  deg_r = seg_ptr[r+1]-seg_ptr[r]
 *           (the deg term is the hoisted bias of feature_module_final, model.py:499-500 summed by :568).
 *           xa, xb, y: [n,64]; wa, wb: [64,64] Keras (in,out) layout; sa: optional device scalar.
 * backward: dy <- dy * (ymask > 0) in place when ymask != NULL, then
 *           dx (=|+=) so * (dy @ wa^T)  and optionally  dx2 (=|+=) dy @ wb^T   (beta 0 = overwrite, 1 = accumulate) */
int gcnn_linear_fwd(const float* xa, const float* sa, const float* wa, const float* xb, const float* wb,
                    const float* bias, const float* bd, const int32_t* seg_ptr, int32_t relu, float* y, int32_t n,
                    void* stream);
int gcnn_linear_bwd(float* dy, const float* ymask, const float* wa, const float* so, float* dx, int32_t beta,
                    const float* wb, float* dx2, int32_t beta2, int32_t n, void* stream);

/* ---- fused edge pass of PartialGraphConvolution.call, model.py:563-569, with Dense(feature_module_final) hoisted --
 * forward:  s_out[r] = sum_{e in seg(r)} relu(s1 * (PL[l_e] + c_e*w_edge + PR[v_e])),  c_e = (coef_e+e_shift)*e_scale
 *           p_recv = projected table of the receiving side [n_recv,64] (constraint/cut side when from_v=True,
 *           model.py:553-556), p_oth = the other side's table, gathered by oth[e].
 *           Optional output for the backward pass: n_rows [n_recv,64] = number of active edges ([s1*J_e > 0]) per
 *           receiver and channel.  Nothing is stored per edge.  Tables hold at most 2^24 rows (gathers use 32-bit byte offsets).
 * bwd_recv: element-wise, because d_s[r] is constant over a segment: d_p_recv = s1*d_s*n_rows.
 * bwd_send: segments grouped by the SENDING node u; the ReLU pattern is recomputed from the two projected tables with the
 *           forward's own expression (bit-identical): with r = oth[e], J_e = (c_e*w_edge + p_send[u]) + p_recv[r] and
 *           t_e = [s1*J_e > 0] * d_s[r]:
 *           d_p_send[u] = s1*sum_{e in seg(u)} t_e ;  the gradient of feature_module_edge's kernel (model.py:490-492),
 *           s1*sum_e c_e*t_e, is left as *n_parts partial rows dw_partial[i][64] (one per thread block, fixed summation
 *           order; the caller adds the rows up).  dw_partial must hold GCNN_EDGE_DW_PARTS rows. */
#define GCNN_EDGE_DW_PARTS 16384
int gcnn_conv_edge_fwd(const int32_t* seg_ptr, const int32_t* oth, const float* coef, int32_t n_recv, int32_t n_edges,
                       const float* p_recv, const float* p_oth, const float* w_edge, const float* e_shift,
                       const float* e_scale, const float* s1, float* s_out, float* n_rows /* optional */,
                       int32_t max_degree /* longest segment, 0 = unknown */, void* stream);
int gcnn_conv_edge_bwd_recv(const float* d_s, const float* n_rows, const float* s1, int32_t n_recv, float* d_p_recv,
                            void* stream);
int gcnn_conv_edge_bwd_send(const int32_t* seg_ptr, const int32_t* oth, const float* coef, int32_t n_send, int32_t n_edges,
                            const float* p_send, const float* p_recv, const float* w_edge, const float* e_shift,
                            const float* e_scale, const float* s1, const float* d_s, float* d_p_send, float* dw_partial,
                            int32_t* n_parts /* host */, int32_t max_degree /* longest segment, 0 = unknown */, void* stream);

/* ---- whole-model forward: GCNN.call, model.py:257-300 ------------------------------------------------------
 * params: flat buffer (layout above).  cons/var/cut feats: [C,4], [V,14], [K,6] raw features (PreNorm applied
 * inside, model.py:365-382).  workspace: gcnn_workspace_floats(dims) floats.  save_for_backward = 1 leaves the
 * activations and edge statistics (the N rows) gcnn_backward needs in the workspace; 0 (inference) skips those stores.
 * The per-edge Dense (feature_module_final, model.py:499-500) and the upper half of output_module's first layer (model.py:505)
 * have nothing between them but the PreNorm scale (model.py:503), so the pass multiplies by their product
 * M = s2*Wf*W1a (made once per call); save_for_backward = 2 runs the two-layer form instead and also stores the tensor between
 * them (the scatter-sum output A = post_conv_module's input) -- what gcnn_prenorm_stats reads for layers 6, 8 and 10.
 * scores: [n_cuts] (model.py:300). */
size_t gcnn_workspace_floats(const gcnn_dims* dims);
int gcnn_forward(const gcnn_dims* dims, const float* params, const float* cons_feats, const float* var_feats,
                 const float* cut_feats, const gcnn_graph* cons_graph, const gcnn_graph* cut_graph,
                 float* workspace, size_t workspace_floats, float* scores, int32_t save_for_backward, void* stream);

/* ---- MSE head: MeanSquaredError on 1-D input, model_trainer.py:132,271 ----------------------------------------
 * loss_out[0] = scale * sum_k (scores_k - targets_k)^2 ; d_scores_k = 2*scale*(scores_k - targets_k).
 * scale = 1/n gives Keras' mean; data-parallel callers pass 1/global_cut_count.  loss_out / d_scores may be NULL. */
int gcnn_mse_loss(const float* scores, const float* targets, int32_t n, float scale, float* loss_out,
                  float* d_scores, void* stream);

/* ---- single-state inference: what the SCIP cut selector does per separation round, model_evaluator.py:82-111 --------
 * get_state -> ten tf.convert_to_tensor -> get_improvements(state, False).numpy() -> sorted(range(n), key=quality, reverse=True)
 * as ONE call on ONE stream: one host->device copy of the packed inputs, a three-launch graph plan specialised to a single
 * (row, col)-sorted state (utils.py:102-104), the inference forward pass, optionally the descending stable ranking of the
 * scores, one device->host copy.  Nothing is synchronised: after the call returns, wait on `stream`, then read host_out.
 *
 * host_in  (pinned): gcnn_infer_layout.in_bytes bytes; in_off[0] .. in_off[1] = a block the CALLER keeps zero (counters, flags
 *          and offset arrays of the plan ride in the upload instead of a memset), in_off[1..7] = cons_feats [C,4] f32, cons_edge_inds
 *          [2,E1] i32, cons_edge_feats [E1] f32, var_feats [V,14] f32, cut_feats [K,6] f32, cut_edge_inds [2,E2] i32,
 *          cut_edge_feats [E2] f32 (the reference's input tuple, model.py:263-275).
 * host_out (pinned): out_bytes bytes; out_off[0] scores [K] f32 (model.py:300), out_off[1] order [K] i32 (only when
 *          want_order: order[0] = index of the best cut, equal scores in index order), out_off[2] four int32 flags:
 *          [0] an edge index out of range, [1] / [2] constraint / cut edges not sorted by row, [3] a variable with more than
 *          2,048 edges.  Any flag set => the scores are NOT valid: raise on [0], otherwise use gcnn_graph_build + gcnn_forward.
 * arena    (device, 256-byte aligned, arena_bytes): inputs, plan, outputs and the forward workspace; caller-owned, reusable.
 * Returns GCNN_E_UNSUPPORTED for more than 32,768 variables (or want_order with more than 4,096 cuts). */
typedef struct gcnn_infer_layout {
    size_t in_bytes, in_off[8];
    size_t out_bytes, out_off[3];
    size_t arena_bytes, dev_off[8];   /* dev_off: internal carving of the arena behind the uploaded block */
} gcnn_infer_layout;
int gcnn_infer_layout_for(const gcnn_dims* dims, gcnn_infer_layout* layout /* host */);
int gcnn_infer(const gcnn_dims* dims, const float* params, const void* host_in, void* host_out, void* arena,
               size_t arena_bytes, int32_t want_order, void* stream);

/* Host helper of gcnn_infer (no device work): an edge list that is NOT sorted by row -- get_state emits sorted lists
 * (utils.py:102-104), other producers may not -- is brought into row order while it is packed into the staging buffer: a stable
 * counting sort (entries of a row keep their input order), O(E + n_left), a few tens of microseconds for a few 10^4 entries.
 * rows / cols / vals: the list (host); out_inds [2,E] and out_vals [E] (host, e.g. inside host_in); scratch: n_left + 1 ints (host).
 * Returns 0, or GCNN_E_BADARG when a row id lies outside [0, n_left) (nothing is written then: the device check reports it). */
int gcnn_host_sort_edges_by_row(const int32_t* rows, const int32_t* cols, const float* vals, int32_t n_edges, int32_t n_left,
                                int32_t* out_inds, float* out_vals, int32_t* scratch);
/* The packing step itself, same arguments: copies the list into the staging buffer and checks its order on the way (one pass, no
 * temporaries); a list that is not sorted by row goes through gcnn_host_sort_edges_by_row.  Returns 0 (was sorted: copied), 1
 * (sorted here), 2 (not sorted and a row id outside [0, n_left): copied as it is, the device check of gcnn_infer reports it), or
 * GCNN_E_BADARG for null pointers / negative sizes. */
int gcnn_host_pack_edges(const int32_t* rows, const int32_t* cols, const float* vals, int32_t n_edges, int32_t n_left,
                         int32_t* out_inds, float* out_vals, int32_t* scratch);

/* Keras-form Adam step (see gcnn_adam_step) to run right behind a backward pass. */
typedef struct gcnn_adam_args {
    float* params; float* m; float* v;   /* flat buffers, gcnn_param_total_floats() each; params is updated in place */
    float lr_t, beta1, beta2, eps;
} gcnn_adam_args;

/* ---- forward + loss head in one pass (the training step's forward, model_trainer.py:269-271) -------------------
 * As gcnn_forward(save_for_backward = 1); the last launch also evaluates the MSE head on its own scores,
 *   loss = loss_scale * sum_k (score_k - targets_k)^2    (loss_scale = 1/n_cuts: Keras' mean),
 * and the gradient of the readout's Dense(64->1) w.r.t. that loss, and -- the same rows, nothing in between -- the receiver-side
 * gradients of the cut rows (through the readout's hidden layer and conv v->k's update), leaving all of them in the workspace:
 * follow with gcnn_backward(d_scores = NULL, ..., loss_out), which starts at conv v->k's sender pass.  Saves the separate
 * gcnn_mse_loss launch and the first two backward launches.  (A caller that only wants loss and scores may stop here.) */
int gcnn_forward_loss(const gcnn_dims* dims, const float* params, const float* cons_feats, const float* var_feats,
                      const float* cut_feats, const gcnn_graph* cons_graph, const gcnn_graph* cut_graph,
                      float* workspace, size_t workspace_floats, float* scores, const float* targets,
                      float loss_scale, void* stream);

/* ---- backward: the vector-Jacobian product tf.GradientTape computes for GCNN.call, model_trainer.py:269-272 ----
 * d_scores: [n_cuts] gradient of the loss w.r.t. the scores; must follow gcnn_forward(save_for_backward = 1) on the same
 * workspace, inputs and parameters.  d_scores = NULL: continue from gcnn_forward_loss instead (the loss head already ran);
 * loss_out (optional) then receives that loss.  Gradients w.r.t. the 46 trainable tensors are written to `grads` (flat
 * layout; non-trainable and padding slots are left untouched -- keep them zero). */
int gcnn_backward(const gcnn_dims* dims, const float* params, const float* cons_feats, const float* var_feats,
                  const float* cut_feats, const gcnn_graph* cons_graph, const gcnn_graph* cut_graph,
                  float* workspace, size_t workspace_floats, const float* d_scores, float* grads,
                  float* cut_count_out /* optional: receives (float)n_cuts, the slot data-parallel callers all-reduce
                                          together with the gradients */,
                  float* loss_out /* optional, see above */,
                  const gcnn_adam_args* adam /* optional: apply gcnn_adam_step(params, grads, ...) right behind, fused into the
                                                last launch whenever the gradients allow it */, void* stream);

/* ---- PreNorm fitting statistics: PreNormLayer.update_params, model.py:394-423 -----------------------------------
 * For ONE batch and ONE of the 11 PreNorm layers (call order: 0 cons, 1 cons-edge, 2 var, 3 cut, 4 cut-edge, then
 * 5+2k / 6+2k = feature_module_final / post_conv_module of convolution k) writes the population mean [units] followed by
 * the mean squared deviation [units] of that layer's input to out_mean_var (device doubles; units = 4,1,14,6,1 for the
 * input layers, 1 otherwise).  Layers >= 5 read activations of a preceding gcnn_forward(save_for_backward=2) on the same
 * workspace, inputs and parameters.  The streaming merge over batches (Chan et al.) is the caller's, as in the reference. */
int gcnn_prenorm_stats(const gcnn_dims* dims, const float* params, const float* cons_feats, const float* var_feats,
                       const float* cut_feats, const gcnn_graph* cons_graph, const gcnn_graph* cut_graph,
                       float* workspace, size_t workspace_floats, int32_t layer, double* out_mean_var, void* stream);

/* ---- Keras-form Adam over the flat buffer: model_trainer.py:131,273 ------------------------------------------
 * theta -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller (host double).
 * grad_scale (optional device scalar, may be NULL) multiplies every gradient first -- or divides it when
 * scale_is_divisor != 0 (data parallel: the all-reduced global cut count).  A divisor that is not > 0 (a global batch
 * without a single cut: model_trainer.py:271 would average over nothing) makes the call a no-op: parameters and moments
 * keep their values instead of turning into NaN. */
int gcnn_adam_step(float* params, const float* grads, float* m, float* v, int32_t n, float lr_t, float beta1,
                   float beta2, float eps, const float* grad_scale, int32_t scale_is_divisor, void* stream);

/* The same update with hyper-parameters and step counter on the device, so that a captured hipGraph of a whole training
 * step can be replayed: opt_state = {lr, beta1, beta2, eps, t, lr_t} (6 floats, device).  Each call advances t by one and
 * recomputes lr_t; the caller changes lr (the plateau schedule of model_trainer.py:177-179) by writing opt_state[0].
 * With a divisor that is not > 0 neither t nor any parameter changes (see gcnn_adam_step). */
int gcnn_adam_step_dev(float* params, const float* grads, float* m, float* v, int32_t n, float* opt_state,
                       const float* grad_scale, int32_t scale_is_divisor, void* stream);

/* ---- ranking-prefix accuracy on the device: model_trainer.py:280-302 / model_tester.py:205-224 ----------------------
 * Per sample s (cuts offsets[s] .. offsets[s+1]-1 of the stacked vectors): rank by pred and by truth, descending, ties in
 * index order (Python's stable sorted(reverse=True)); frac = first differing position / #cuts (1 if none).
 * acc[f] += [frac >= fractions[f]] (accumulates across calls); frac_out[s] = frac (optional).  max_cuts = largest sample
 * (host value, <= 4096, else GCNN_E_WORKSPACE).  Optionally also accumulates the cut-weighted loss of
 * model_trainer.py:304: loss_acc[0] += loss_in[0] * loss_weight. */
int gcnn_ranking_metric(const float* pred, const float* truth, const int32_t* offsets, int32_t n_samples,
                        int32_t max_cuts, const float* fractions, int32_t n_fractions, float* acc, float* frac_out,
                        const float* loss_in, float loss_weight, float* loss_acc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCNN_HIP_H */
