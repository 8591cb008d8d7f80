/*
 * ampconv.h -- C ABI of libampconv.so: the MI355X (gfx950) implementation of the
 * AMPConv hot path of HarryL-Git/ampnet.
 *
 * The reference has no FFI layer of its own: its boundary for this path is the
 * Python class `AMPConv` (reference src/ampnet/conv/amp_conv.py:9-51) on top of
 * torch.nn.MultiheadAttention and torch_geometric.nn.MessagePassing.  Each entry
 * point below names the reference lines whose arithmetic it replaces.  The host
 * side that binds these with ctypes is ampnet_amd/conv/amp_conv.py; the stub a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library
 *     allocates nothing and keeps no state between calls (re-entrant);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     no call synchronises the device;
 *   - every function returns 0 on success, a negative AMPCONV_E_* code for
 *     argument errors detected on the host, or a positive hipError_t;
 *     nothing throws or aborts;
 *   - N = nodes, E = edges, L = tokens per node, D = embed_dim, H = heads,
 *     dh = D / H.  Messages flow src = edge_index[0] -> dst = edge_index[1]
 *     (amp_conv.py:40-41).
 */
#ifndef AMPCONV_H_
#define AMPCONV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPCONV_VERSION 107

enum {
  AMPCONV_OK = 0,
  AMPCONV_E_BADARG = -1,   /* null pointer, negative size, D % H != 0, ... */
  AMPCONV_E_DTYPE = -2,    /* dtype not supported by this build */
  AMPCONV_E_WORKSPACE = -3 /* workspace too small */
};

/* dtype codes.
 * AMPCONV_F32:  every view is fp32 in HBM; the per-edge products of the one-wave-per-unit shapes (L <= 20, dh = 32 or 16)
 *               run on v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics), those of the workgroup-per-unit shapes
 *               (other even dh <= 64, L <= 64) on the 16-bit matrix cores with every fp32 tile split exactly into three
 *               bf16 planes on its way into LDS (fp32-grade error; csrc/edge_block_x3.hip), the per-node projections
 *               (ampconv_proj_*) the same way.
 * AMPCONV_BF16: every view is bf16 in HBM (Q/K/V/O and gradients), fp32 softmax and accumulation.
 *               L <= 20 with dh = 32 (BASELINE config 5) or dh = 16 (half-filled tiles): products on the bf16 MFMA,
 *               no softmax statistics.  Other even dh <= 64 with L <= 64 (e.g. the reference's class default L = 40,
 *               dh = 50): the workgroup-per-unit kernels on the bf16 rows as they lie (one bf16 MFMA per product,
 *               softmax weights and dS rounded to bf16 for their second product); their source pass needs the
 *               statistics buffer of ampconv_softmax_stats_bytes like the fp32 call.  Anything else: AMPCONV_E_DTYPE.
 * (Versions <= 102 had three more codes for split-operand edge kernels; they were slower than the native fp32 MFMA
 * kernels on MI355X and were removed in 103.) */
enum { AMPCONV_F32 = 0, AMPCONV_BF16 = 1 };

/*
 * Strided view of a per-node token matrix: element (node n, token l, head h,
 * channel c) lives at  ptr + n*node_stride + l*row_stride + h*head_stride + c
 * (strides in ELEMENTS).  A row-major [N, L, D] tensor is
 * {ptr, L*D, D, dh}; the K third of a packed [N*L, 3D] projection is
 * {ptr + D, L*3*D, 3*D, dh}.
 */
typedef struct {
  void *ptr;
  int64_t node_stride;
  int64_t row_stride;
  int64_t head_stride;
} ampconv_view_t;

int ampconv_version(void);
const char *ampconv_error_string(int code);

/* ---- graph preparation ------------------------------------------------------
 * Replaces what PyG's propagate() does implicitly with index_select/scatter on
 * the unsorted edge list (amp_conv.py:25).  Builds, with stable sorts so that
 * every later floating-point sum has a fixed order:
 *   dst-sorted CSR: rowptr[N+1], col[E] (source of each sorted edge),
 *                   eperm[E] (original edge id at each sorted position)
 *   src-sorted CSC: cscptr[N+1], crow[E] (destination), cperm[E],
 *                   cinv[E] = 1 / in-degree(crow[p]) (the weight of edge p in its
 *                   destination's mean, read sequentially by the source pass)
 * `oob` (device int32) is set non-zero if any index is outside [0, N); such
 * indices are clamped so that no kernel faults.  */
size_t ampconv_csr_workspace_bytes(int64_t N, int64_t E);
int ampconv_csr_build(const int64_t *edge_index, int64_t E, int64_t N,
                      int32_t *rowptr, int32_t *col, int32_t *eperm,
                      int32_t *cscptr, int32_t *crow, int32_t *cperm, float *cinv,
                      int32_t *oob, void *workspace, size_t workspace_bytes,
                      void *stream);

/* csr_build and both long-segment plans (below; chunk <= 0 or a NULL plan: none) in one call.  Graphs of at most
 * 12 288 edges and 16 384 nodes -- GraphSAINT batches, Cora: the reference's own regime -- are prepared by ONE launch
 * (two workgroups, the stable sorts in LDS); larger ones by the calls above.  Same outputs either way.
 * `status` (device int32[4]) = {bounds flag as `oob` above, chunks of plan_dst, chunks of plan_src, 0}: everything the
 * host needs back, in one 16-byte read.  `by_edge` (E int32, may be NULL): CSC position of every ORIGINAL edge id,
 * from which ampconv_csc_positions_from derives `spos` (below) in one launch.  */
int ampconv_graph_build(const int64_t *edge_index, int64_t E, int64_t N,
                        int32_t *rowptr, int32_t *col, int32_t *eperm,
                        int32_t *cscptr, int32_t *crow, int32_t *cperm, float *cinv,
                        int32_t *status, int chunk, void *plan_dst, void *plan_src,
                        int32_t *by_edge, void *workspace, size_t workspace_bytes, void *stream);
int ampconv_csc_positions_from(const int32_t *eperm, const int32_t *by_edge, int64_t E,
                               int32_t *spos, void *stream);

/* spos[p] = position in the src-sorted CSC of the edge at position p of the dst-sorted CSR
 * (eperm, cperm of ampconv_csr_build; `scratch` = E int32).  Needed only to hand softmax
 * statistics from the destination pass to the source pass (below).  */
int ampconv_csc_positions(const int32_t *eperm, const int32_t *cperm, int64_t E,
                          int32_t *scratch, int32_t *spos, void *stream);

/* Node lists for the projections' `nodes` argument (below): the ascending ids of the nodes with at least one in-edge
 * (which = 1), one out-edge (2) or either (3).  list: N + 8 int32 (the entries behind the count are padding the
 * projection kernels may read); ptr: N + 1 int32, ptr[n + 1] - ptr[n] = 1 iff node n is listed, ptr[N] = the count
 * -- a CSR-shaped array, so ampconv_mask_rows(Y, ptr, ...) zeroes exactly the rows of the nodes NOT listed;
 * count: one device int32.  */
size_t ampconv_active_nodes_workspace_bytes(int64_t N);
int ampconv_active_nodes(const int32_t *rowptr, const int32_t *cscptr, int64_t N, int which,
                         int32_t *list, int32_t *ptr, int32_t *count, void *workspace,
                         size_t workspace_bytes, void *stream);

/* ---- long segments ("hubs": power-law graphs, BASELINE config 5) ---------------------------
 * One wavefront per (row, head) runs as long as its longest segment.  A plan cuts every CSR
 * (or CSC) segment longer than `chunk` edges into chunks; the edge kernels then reduce each
 * chunk in its own wavefront into a partial tile in `hub_ws` and an ordered pass adds a row's
 * partial tiles (fixed order: bitwise reproducible).  plan = int32 header {n_chunks, chunk, 0, 0}
 * followed by 16-byte descriptors; the caller reads header[0] back once (n_chunks), sizes
 * hub_ws with ampconv_hub_workspace_bytes and passes both to the edge calls (plan = NULL or
 * n_chunks = 0: no splitting).  n_tiles = 1 (forward, dst pass) or 2 (src pass: dK and dV).  */
size_t ampconv_hub_plan_bytes(int64_t E, int chunk);
int ampconv_hub_plan(const int32_t *ptr, int64_t N, int64_t E, int chunk, void *plan, void *stream);
size_t ampconv_hub_workspace_bytes(int64_t n_chunks, int L, int D, int n_tiles);

/* ---- edge phase, forward ----------------------------------------------------
 * For every row r < n_rows (destination d = qidx ? qidx[r] : r) and head h:
 *   O[r,:,h] = (1/deg_r) * sum_{p in [rowptr[r], rowptr[r+1])}
 *                 softmax_rows(Q[d,:,h] K[col[p],:,h]^T / sqrt(dh)) V[col[p],:,h]
 * and O[r] = 0 when deg_r = 0.  Replaces, per edge: torch functional.py:6578
 * (scale), :6589 (QK^T), :6590 (softmax), :6594 (PV), and PyG's mean
 * aggregation (amp_conv.py:11).  Q/K/V are the per-NODE projections.  */
int ampconv_fwd_edge(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                     const int32_t *rowptr, const int32_t *col,
                     const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                     ampconv_view_t O, const void *hub_plan, int64_t hub_chunks,
                     void *hub_ws, int dtype, void *stream);

/* ---- edge phase, backward (autograd of the above; amp_conv.py has no custom
 * backward, cora_benchmark_graphsaint.py:110 calls loss.backward()) ------------
 * dObar is the gradient w.r.t. the MEAN (the kernels apply 1/deg).
 * _dst: one pass over the dst-sorted CSR, writes dQ[r] for every row.
 * _src: one pass over the src-sorted CSC, writes dK[s], dV[s] for every source;
 *       `cinv[p]` = 1/in-degree of the destination of CSC edge p (ampconv_csr_build).
 * No atomics: every output row is owned by one wavefront.
 * Softmax statistics (optional): both passes need, per edge, head and destination token i, the
 * softmax normaliser and delta_i = sum_j P_ij dP_ij.  The destination pass has them as a by-product
 * (its softmax runs inside a lane); the source pass otherwise re-reduces them across lanes.  With
 * `stats` (ampconv_softmax_stats_bytes(E, ...) bytes, 16-byte aligned; 0 = this dtype/shape keeps
 * none and `stats` must be NULL) the destination pass stores them at the edge's CSC position
 * (`spos`, ampconv_csc_positions) as 20 log2-sum-exp + 20 delta floats per (edge, head), and the
 * source pass -- which must then run AFTER the destination pass -- reads them back sequentially.
 * `out_absmax` (may be NULL; AMPCONV_F32 only): a device float that receives, by atomic max, the largest finite
 * magnitude of what the pass writes to dQ (dK and dV) -- the scale source of the projections that consume the
 * gradient (SCALED MODE below); the caller zeroes it, both passes may share one.  The destination pass's tile kernels
 * record it as they store (one compare per wave); the source pass and the other kernel families get it from one
 * ampconv_absmax pass over the output, run by the entry point -- the output must then be a plain row-major matrix (rows
 * of D channels, a node's L rows consecutive).  */
size_t ampconv_softmax_stats_bytes(int64_t E, int L, int D, int H, int dtype);
int ampconv_bwd_edge_dst(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                         ampconv_view_t dObar, const int32_t *rowptr,
                         const int32_t *col, int64_t n_rows, int L, int D, int H,
                         ampconv_view_t dQ, const void *hub_plan, int64_t hub_chunks,
                         void *hub_ws, const int32_t *spos, float *stats, float *out_absmax,
                         int dtype, void *stream);
int ampconv_bwd_edge_src(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                         ampconv_view_t dObar, const int32_t *cscptr,
                         const int32_t *crow, const float *cinv,
                         int64_t n_src, int L, int D, int H, ampconv_view_t dK,
                         ampconv_view_t dV, const void *hub_plan, int64_t hub_chunks,
                         void *hub_ws, const float *stats, float *out_absmax, int dtype,
                         void *stream);

/* ---- edge phase on fp16 PLANES: fp32-grade results off the FP32 pipe (csrc/edge_mfma_f16x2.hip, ABI 106) --------
 * The same three passes (same reference arithmetic: torch functional.py:6578-6594 per edge, amp_conv.py:11, SURVEY.md
 * A.2) for L <= 20, dh = 32 or 16, with Q, K, V and dObar in the PLANE FORMAT: the 4 dh-byte slot of the dh fp32 channels
 * of one (token row, head) holds dh fp16 `hi` then dh fp16 `lo` with hi + lo = x * 2^e (to 2^-22 |x|, absolute 2^-25 in
 * scaled units below that), ONE exponent per tensor: e = 14 - floor(log2 bound) for a device-side upper bound of the
 * tensor's magnitudes -- bounds[0] for Q | K | V (written by ONE ampconv_proj_rows_planes call), bounds[1] for dObar.
 * `bounds` (device, 4 floats) also carries what the backward passes scale dS = P (dP - delta) by before they split it:
 * bounds[2] = the largest |V| and bounds[3] = the largest |dObar| (true fp32 magnitudes, as recorded by the out_absmax
 * of the two ampconv_proj_rows_planes calls; any upper bound serves).  The forward pass reads bounds[0] only.
 * Views keep the strides, in 4-byte elements, of the fp32 tensor the planes replace; head_stride must be dh.  Every
 * product is the fp32 sum of three v_mfma_f32_16x16x32_f16 partial products (dropped: lo x lo <= 2^-22 of the product),
 * softmax and all sums are fp32, the outputs (Obar, dQ, dK, dV) plain fp32 views.
 * dObar must arrive DIVIDED by the in-degree of its node (ampconv_proj_rows_planes, row_scale = 1): the passes carry no
 * per-edge weight.  Softmax statistics as above (`stats`: E * H * 40 floats, 16-byte aligned, or NULL; `spos` with
 * them; delta is handed over in the units of the scaled dObar V^T product, which both passes share).  out_absmax as
 * above (both backward passes record it themselves).
 * planes_supported: 1 if (L, D, H) is served.  Same accuracy class as the fp32 kernels on tensors whose rows lie
 * within ~2^12 of the tensor's maximum (ampconv_absmax_stats measures exactly that; callers fall back to the fp32
 * entry points otherwise).  */
int ampconv_planes_supported(int L, int D, int H);
int ampconv_fwd_edge_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                            const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D,
                            int H, ampconv_view_t O, const void *hub_plan, int64_t hub_chunks,
                            void *hub_ws, const float *bounds, void *stream);
int ampconv_bwd_edge_dst_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                ampconv_view_t dObar, const int32_t *rowptr, const int32_t *col,
                                int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                                const void *hub_plan, int64_t hub_chunks, void *hub_ws,
                                const float *bounds, const int32_t *spos, float *stats,
                                float *out_absmax, void *stream);
int ampconv_bwd_edge_src_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                ampconv_view_t dObar, const int32_t *cscptr, const int32_t *crow,
                                int64_t n_src, int L, int D, int H, ampconv_view_t dK,
                                ampconv_view_t dV, const void *hub_plan, int64_t hub_chunks,
                                void *hub_ws, const float *bounds, const float *stats,
                                float *out_absmax, void *stream);

/* ---- edge phase on fp32 VIEWS with operand bounds: the workgroup-per-unit shapes off the FP32 pipe (csrc/edge_block_x3.hip,
 * ABI 107) ----------------------------------------------------------------------------------------------------------
 * The same three passes for L <= 64 and even dh <= 64 outside the one-wave-per-unit kernels' shapes (L <= 20 with dh = 32 or
 * 16) -- the reference's AMPGCN class defaults L = 40, D = 100, H = 2 (src/ampnet/module/amp_gcn.py:21-35), or 40 tokens at
 * its 128 / 4 -- on plain fp32 views -- what ampconv_fwd_edge / _bwd_edge_dst / _bwd_edge_src take --
 * plus the `bounds` of the plane-format entry points above: bounds[0] >= max |Q|K|V|, bounds[1] >= max |dObar|,
 * bounds[2] >= max |V|, bounds[3] >= max |dObar| (device, 4 floats; the forward pass reads bounds[0] only).  With them
 * the kernels split every fp32 tile, on its way into LDS, into TWO fp16 planes of x * 2^(14 - floor(log2 bound)) and
 * run each product as three v_mfma_f32_16x16x32_f16 partial products (half the matrix-pipe cycles and two thirds of
 * the vector instructions of the bound-free kernels behind the fp32 entry points, which split into three bf16 planes).
 * dObar is the gradient of the mean as in the fp32 entry points (the passes apply 1 / in-degree; `cinv` as in
 * ampconv_bwd_edge_src).  Softmax statistics: ampconv_softmax_stats_bytes(E, L, D, H, AMPCONV_F32) bytes, REQUIRED by
 * the source pass, delta in the units of the scaled dObar V^T product.  out_absmax as above.  Same accuracy class as the
 * fp32 kernels on tensors whose rows lie within ~2^12 of the tensor's maximum (see ampconv_planes_supported).  */
int ampconv_scaled_supported(int L, int D, int H);
int ampconv_fwd_edge_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                            const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D,
                            int H, ampconv_view_t O, const void *hub_plan, int64_t hub_chunks,
                            void *hub_ws, const float *bounds, void *stream);
int ampconv_bwd_edge_dst_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                ampconv_view_t dObar, const int32_t *rowptr, const int32_t *col,
                                int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                                const void *hub_plan, int64_t hub_chunks, void *hub_ws,
                                const float *bounds, const int32_t *spos, float *stats,
                                float *out_absmax, void *stream);
int ampconv_bwd_edge_src_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                ampconv_view_t dObar, const int32_t *cscptr, const int32_t *crow,
                                const float *cinv, int64_t n_src, int L, int D, int H,
                                ampconv_view_t dK, ampconv_view_t dV, const void *hub_plan,
                                int64_t hub_chunks, void *hub_ws, const float *bounds,
                                const float *stats, float *out_absmax, void *stream);

/* ---- per-edge side outputs, ORIGINAL edge order ------------------------------
 * attn_weights: W[e] = mean_h softmax_rows(Q[dst e,:,h] K[src e,:,h]^T/sqrt(dh)),
 * [E, L, L] fp32 -- `self.attn_output_weights` (amp_conv.py:39,43-47; torch
 * functional.py:6604-6606).  */
int ampconv_attn_weights(ampconv_view_t Q, ampconv_view_t K,
                         const int64_t *edge_index, int64_t E, int L, int D,
                         int H, float *W, int dtype, void *stream);

/* attn_scores: the same without the softmax, W[e] = mean_h Q K^T / sqrt(dh) -- the per-edge weights of
 * the reference's softmax-free attention (custom_multihead_attn_forward.py:4173-4184, :4441-4442).  */
int ampconv_attn_scores(ampconv_view_t Q, ampconv_view_t K,
                        const int64_t *edge_index, int64_t E, int L, int D,
                        int H, float *W, int dtype, void *stream);

/* ---- node-side helpers --------------------------------------------------------
 * segment_mean: PyG aggr='mean' on an [E, F] message matrix (amp_conv.py:11,
 * testing_message_passing_pyg.py:37-40): out[n] = mean of msg[eperm[p]] over the
 * CSR segment of n, 0 for empty segments.
 * mask_rows: zero, in place, the rows of Y[N, F] whose CSR segment is empty
 * (out-projection bias must not leak into nodes nobody sends to).
 * masked_colsum: out[f] = sum over rows with a non-empty segment of dY[n, f],
 * folded over the L tokens: out has D fp32 entries (out_proj.bias gradient).
 * `dtype` of Y / dY: AMPCONV_F32 or AMPCONV_BF16.  */
int ampconv_segment_mean(const float *msg, const int32_t *rowptr,
                         const int32_t *eperm, int64_t N, int64_t F, float *out,
                         void *stream);
int ampconv_mask_rows(void *Y, const int32_t *rowptr, int64_t N, int64_t F,
                      int dtype, void *stream);
/* gather_segment_sum: out[r, :] = scale_r * sum_{p in [ptr[r], ptr[r+1])} (w ? w[p] : 1) * rows[idx[p], :]
 * with F (a multiple of 4) fp32 per row, scale_r = 1/segment length if `mean` else 1, 0 for empty
 * segments.  The whole edge phase of the softmax-free variant ("next" row 3 of SURVEY.md 8f): without
 * softmax, sum_e Q_d K_s^T V_s = Q_d sum_e (K_s^T V_s), so the per-edge work collapses to this
 * segment reduction of the per-source dh x dh matrices K_s^T V_s (forward: CSR, mean; backward:
 * CSC, weights cinv).  */
int ampconv_gather_segment_sum(const float *rows, const int32_t *ptr, const int32_t *idx,
                               const float *w, int mean, int64_t N, int64_t F, float *out,
                               void *stream);
/* The per-node products around it (custom_multihead_attn_forward.py:4173-4184 re-associated), one
 * [L, dh] tile per (node, head) on either side, M = [N, H, dh, dh] fp32 row-major:
 *   linear_outer: M[n,h] = scale * A[n,:,h]^T B[n,:,h]      (K^T V; Q^T dObar for the backward)
 *   linear_apply: Out[n,:,h] = scale * A[n,:,h] M[n,h]      (transpose != 0: ... M[n,h]^T)
 *                 (Q Mbar; dObar Mbar^T, V dM^T, K dM for the backward)  */
int ampconv_linear_outer(ampconv_view_t A, ampconv_view_t B, int64_t N, int L, int D, int H,
                         float scale, float *M, void *stream);
int ampconv_linear_apply(ampconv_view_t A, const float *M, int transpose, int64_t N, int L,
                         int D, int H, float scale, ampconv_view_t Out, void *stream);
int ampconv_masked_colsum(const void *dY, const int32_t *rowptr, int64_t N,
                          int L, int D, float *out, int dtype, void *stream);

/* ---- node phase: the per-node projections ---------------------------------------------------
 * Replaces the packed in-projection (torch functional.py:5785-5862 `_in_projection_packed`; the
 * reference's copy src/ampnet/conv/custom_multihead_attn_forward.py:4070-4077) and the
 * out-projection (torch functional.py:6600), once per NODE instead of once per edge, and their
 * autograd backward (dObar = dY Wo, dX = dQKV Win, dW = dOut^T In, db = colsum).
 * `dtype` names the storage of EVERY tensor of a call (rows, weights, bias, outputs, gradients):
 *   AMPCONV_F32   fp32 in, fp32 out, fp32 accumulate; every operand element is split EXACTLY into three bf16 terms
 *                 and a product is the sum of the six partial products of order >= 2^-16 on
 *                 v_mfma_f32_32x32x16_bf16 (error of the dropped terms <= 3 * 2^-26 per product: fp32 grade).
 *                 Non-finite inputs: NaN propagates; +-Inf (and |x| above the largest bf16, 3.39e38) comes out as
 *                 NaN where an fp32 GEMM would return +-Inf (the residual planes are inf - inf).
 *   AMPCONV_BF16  bf16 in HBM (BASELINE config 5), ONE product per fragment pair on the same instruction, fp32
 *                 accumulate, results rounded to bf16 once (weight / bias gradients: after the ordered fp32 sum over
 *                 the row slices).  The weight-gradient PRODUCT is not masked in this mode (the in-degree mask acts
 *                 on the column sums only): the rows of nodes without in-edges contribute nothing because the other
 *                 operand (the forward pass's Obar) is exactly 0 there -- as in a plain dY^T Obar.
 *   proj_supported     : 1 if (N, K) is served, else 0 -- the caller then uses a library GEMM.  fp32: both multiples
 *                        of 4, bf16: of 8 (rows are read and written in 16-byte pieces); tiles are padded
 *                        internally, so the reference's default embed_dim = 100 is served in fp32
 *   proj_weight_image  : B[n][k] = W[n * stride_n + k * stride_k] (N x K) -> `image`
 *                        (proj_weight_image_bytes(N, K, dtype) bytes, 16-byte aligned): the weight as ready MFMA
 *                        fragments (fp32: its three bf16 planes), zero-padded.  (stride_n, stride_k) = (K, 1) uses a
 *                        row-major [N, K] weight as it stands (forward), (1, N) its transpose (backward).
 *   proj_weight_images : up to 8 of them in ONE launch (forward and transposed images of both weights of a layer)
 *   proj_rows          : out[m, :N] = (A[m, :K] B^T + bias) * (rowptr ? [node m / L has an in-edge] : 1)
 *                        A row-major with leading dimension lda (elements), out with ldc
 *   proj_wgrad         : dW[Na, Nb] = sum_m (mask_m A[m, :Na])^T B[m, :Nb] and colsum[Na] = sum_m mask_m A[m, :Na]
 *                        (mask as above, rowptr may be NULL; bf16: see above); deterministic: fixed row slices,
 *                        ordered sum.  `workspace`: proj_wgrad_workspace_bytes(M, Na, Nb, dtype) bytes.
 *   NODE LISTS (`nodes` != NULL, AMPCONV_BF16 only, 16 <= L <= 128; NULL: every row): only the L rows of each of the
 *   n_nodes listed nodes (ascending node ids < M / L; the array must be readable for 8 entries past n_nodes:
 *   ampconv_active_nodes makes such lists) are read, multiplied and -- proj_rows -- written; all other rows of
 *   `out` are left untouched.  proj_wgrad's workspace is then sized for the listed rows:
 *   proj_wgrad_workspace_bytes(n_nodes * L, ...).  The per-node formulation computes a projection for EVERY node, the reference one per EDGE:
 *   a node without in-edges needs no Q row and no output row (it is 0), one without out-edges no K / V rows -- on the
 *   R-MAT graph of BASELINE config 5 that is 48 % of the nodes on either side.  proj_rows with a list takes no mask
 *   (rowptr must be NULL: list the nodes that pass it); proj_wgrad sums over the listed rows only.
 *   SCALED MODE (AMPCONV_F32 only; `a_absmax` (+ `b_absmax` for proj_wgrad) != NULL: device floats holding the largest
 *   finite magnitude of the operand, or any upper bound of it within a few binades): the operands are scaled by a power
 *   of two into fp16's range and split into TWO fp16 planes, a product is three fp16 matrix products instead of six bf16
 *   ones -- as close to the fp64 result as the six-product form on operands whose elements lie within 2^17 of the
 *   maximum, absolute error <= 2^-39 of the maximum per element below that (csrc/proj_gemm.hip); the
 *   weight's scale is part of its image.  NULL: the six-product form, exact split at any range.  ampconv_absmax
 *   computes such a maximum (one pass over X[M, K], row stride ld; merged into *out by atomic max, `reset` zeroes it
 *   first; NaN and infinities are skipped); proj_rows in this mode records the maximum of what it WRITES into `out_absmax` (may be
 *   NULL; atomic max, the caller zeroes it) -- the next product's operand then needs no pass of its own.
 * Developer switches read from the environment at the first call (A/B measurements; the defaults are the
 * shipped configuration, nothing else keeps state): AMPCONV_PROJ_ROWS=1 (256 x 256 row tiles),
 * AMPCONV_PROJ_WGRAD_TI=128, AMPCONV_PROJ_WGRAD_BF16_T=128 (smaller weight-gradient tiles), and for the edge phase AMPCONV_FORCE_GENERIC=1,
 * AMPCONV_SMALL=0 (L <= 4 on the tile kernels instead of the short-sequence family), AMPCONV_{FWD,DST,SRC}_T4=0,
 * AMPCONV_{FWD,DST,SRC}_NT4=0 (older tilings of the same kernels, kept as cross-checks for the tests),
 * AMPCONV_CSR_SMALL=0 (graph preparation by the multi-launch path).  */
int ampconv_proj_supported(int N, int K, int dtype);
size_t ampconv_proj_weight_image_bytes(int N, int K, int dtype);
int ampconv_proj_weight_image(const void *W, int64_t stride_n, int64_t stride_k, int N, int K,
                              void *image, int dtype, void *stream);
typedef struct {
  const void *W;
  int64_t stride_n, stride_k;
  int N, K;
  void *image;
} ampconv_weight_image_t;
int ampconv_proj_weight_images(int count, const ampconv_weight_image_t *jobs, int dtype, void *stream);
int ampconv_proj_rows(const void *A, int64_t lda, int64_t M, int K, const void *wimage, int N,
                      const void *bias, const int32_t *rowptr, int L, void *out, int64_t ldc,
                      const int32_t *nodes, int64_t n_nodes, const float *a_absmax,
                      float *out_absmax, int dtype, void *stream);
size_t ampconv_proj_wgrad_workspace_bytes(int64_t M, int Na, int Nb, int dtype);
int ampconv_proj_wgrad(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int Na,
                       int Nb, const int32_t *rowptr, int L, void *dW, void *colsum,
                       void *workspace, size_t workspace_bytes, const int32_t *nodes, int64_t n_nodes,
                       const float *a_absmax, const float *b_absmax, int dtype, void *stream);
int ampconv_absmax(const void *X, int64_t ld, int64_t M, int K, int dtype, float *out, int reset,
                   void *stream);
/* PLANE OUTPUT (fp32 storage, scaled mode; N % 128 == 0, K % 32 == 0, ldc % 32 == 0): proj_rows whose result leaves in
 * the plane format of the edge kernels above instead of fp32 -- same bytes, same layout of 128-byte slots.
 *   proj_out_bound  : out[0] = *a_absmax * max_n sum_k |B[n][k]| + max_n |bias[n]| (B as in proj_weight_image): an upper
 *                     bound of |A B^T + bias| that is known BEFORE the product runs; the scale of its planes
 *   proj_rows_planes: as proj_rows; `out_bound` = that device float; plane_dh = 32 or 16: the slot width (head dimension); row_scale = 1 (with rowptr): rows are also DIVIDED
 *                     by their node's segment length (dObar / in-degree); out_absmax (may be NULL) records the largest
 *                     finite magnitude of the fp32 values behind the planes over the columns >= absmax_col0 only (the V
 *                     third of a packed in-projection bounds Obar, a mean of convex combinations of V rows)
 *   planes_to_f32   : the reverse: out[m, k] = (hi + lo) * 2^-e of X[M, K] in plane format (side outputs, fall-backs)
 *   absmax_stats    : out[0] = largest finite magnitude of X[M, K] (fp32), out[1] = the smallest NON-ZERO maximum of any
 *                     group of 8 consecutive 16-byte pieces (32 channels: a head slot); out[1] * 2^12 < out[0] says that
 *                     whole rows / heads lie far below the tensor's maximum: one scale per tensor then costs them their
 *                     low plane and the caller should use the exact kernels (six-product projections, fp32 edge passes)  */
int ampconv_proj_out_bound(const void *W, int64_t stride_n, int64_t stride_k, int N, int K,
                           const void *bias, const float *a_absmax, float *out, void *stream);
int ampconv_proj_rows_planes(const void *A, int64_t lda, int64_t M, int K, const void *wimage, int N,
                             const void *bias, const int32_t *rowptr, int L, int row_scale, void *out,
                             int64_t ldc, const float *a_absmax, const float *out_bound,
                             float *out_absmax, int absmax_col0, int plane_dh, void *stream);
int ampconv_planes_to_f32(const void *X, int64_t ld, int64_t M, int K, int plane_dh, const float *bound,
                          void *out, int64_t ldo, void *stream);
int ampconv_absmax_stats(const void *X, int64_t ld, int64_t M, int K, float *out, void *stream);

/* ---- GraphSAINT random-walk sampler ("next" row: the step before the hot path) -------------
 * In-tree spec: the reference's vendored PyG sampler, visualization/visualize_graphsaint_subgraphs.py
 * :195-199 (walks), :107-110 (unique nodes + induced sub-graph), :137-173 (norms).  The graph is
 * given as the src-sorted CSC of ampconv_csr_build (cscptr, crow = destinations, cperm = original
 * edge ids).  The caller reads n_sub / e_sub (device int32) back to size the next outputs.
 *   random_walk : walks[b, 0] = start[b]; each step moves to a uniform random out-neighbour
 *                 (stays if none); counter-based generator keyed by (seed, walk, step)
 *   nodes       : walked nodes (with repeats) -> mark[N], relabel[N], node_idx (sorted unique)
 *   count/fill  : induced sub-graph, edges grouped by source in CSC order: relabelled
 *                 edge_index [2, e_sub] and the original edge ids (count_edges_bounded: the same before
 *                 the host knows n_sub -- an upper bound sizes cnt / off, n_sub is read on the device --
 *                 so that n_sub and e_sub come back in ONE read)
 *   add_counts  : count[idx[i]] += 1 (occurrence statistics of nodes / edges)
 *   norms       : edge_norm = clamp(node_count[src] / edge_count, 0, 1e4) (NaN -> 0.1),
 *                 node_norm = num_samples / max(node_count, 0.1 if 0) / N               */
int ampconv_saint_random_walk(const int32_t *cscptr, const int32_t *crow, const int64_t *start,
                              int64_t B, int walk_length, uint64_t seed, int64_t *walks, void *stream);
size_t ampconv_saint_workspace_bytes(int64_t N);
int ampconv_saint_nodes(const int64_t *nodes, int64_t n, int64_t N, int32_t *mark, int32_t *relabel,
                        int64_t *node_idx, int32_t *n_sub, void *workspace, size_t workspace_bytes,
                        void *stream);
int ampconv_saint_count_edges(const int64_t *node_idx, int64_t n_sub, const int32_t *cscptr,
                              const int32_t *crow, const int32_t *mark, int32_t *cnt, int32_t *off,
                              int32_t *e_sub, void *workspace, size_t workspace_bytes, void *stream);
int ampconv_saint_count_edges_bounded(const int64_t *node_idx, int64_t n_bound, const int32_t *n_sub_dev,
                                      const int32_t *cscptr, const int32_t *crow, const int32_t *mark,
                                      int32_t *cnt, int32_t *off, int32_t *e_sub, void *workspace,
                                      size_t workspace_bytes, void *stream);
int ampconv_saint_fill_edges(const int64_t *node_idx, int64_t n_sub, const int32_t *cscptr,
                             const int32_t *crow, const int32_t *cperm, const int32_t *mark,
                             const int32_t *relabel, const int32_t *off, int64_t E_sub,
                             int64_t *edge_index, int64_t *edge_id, void *stream);
int ampconv_saint_add_counts(const int64_t *idx, int64_t n, float *count, void *stream);
int ampconv_saint_norms(const float *node_count, const float *edge_count, const int64_t *edge_src,
                        int64_t N, int64_t E, float num_samples, float *node_norm, float *edge_norm,
                        void *stream);
/* gather_rows: the batch's rows of a resident per-node tensor, dst[i, :] = src[idx[i], :] (the collate step of the
 * vendored sampler, visualize_graphsaint_subgraphs.py:112-135: `item[node_idx]`).  Rows of row_bytes bytes (a multiple of
 * 16, both tensors 16-byte aligned, src rows src_stride_bytes apart), idx in [0, n_src).  */
int ampconv_saint_gather_rows(const void *src, int64_t src_stride_bytes, int64_t row_bytes, const int64_t *idx,
                              int64_t n, void *dst, void *stream);

/* ---- AMPGCN featuriser ("next" row: the step right before the first AMPConv layer) ----------
 * Reference src/ampnet/module/amp_gcn.py:120-183: z-score of the node features (:122-125), L present
 * (non-zero) features sampled per node with replacement (:132-135), token = concat(embedding row,
 * z-scored value) (:146-147).  x is [N, F] fp32, idx [N, L] int32 (-1 = node without any present
 * feature; `empty_flag` is raised), table [F, De], out [N, L, De + 1].  table_grad zeroes dtable and
 * accumulates the token gradients with float atomics.  */
int ampconv_feat_zscore_stats(const float *x, int64_t N, int64_t F, float *mean, float *inv_std,
                              void *stream);
int ampconv_feat_sample_present(const float *x, int64_t N, int F, int L, uint64_t seed, int32_t *idx,
                                int32_t *empty_flag, void *stream);
int ampconv_feat_build(const float *x, const float *mean, const float *inv_std, const int32_t *idx,
                       const float *table, int64_t N, int F, int L, int De, float *out, void *stream);
int ampconv_feat_table_grad(const float *dout, const int32_t *idx, int64_t N, int L, int De, int F,
                            float *dtable, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AMPCONV_H_ */
