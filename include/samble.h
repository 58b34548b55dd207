/* samble.h - C ABI of libsamble_hip.so: the MI355X (gfx950) kernels behind SAMBLE's
 * attention-score downsampling path.
 *
 * The reference (stevenczwu/SAMBLE) is pure Python on PyTorch: it has no native boundary, so
 * each entry point below replaces a chain of ATen calls made by one reference function (cited
 * per entry as file:line).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add and how models/downsample.py / utils/ops.py call sites map onto it.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated otherwise;
 *   - nothing is allocated, freed or synchronised: inputs, outputs and workspace belong to the
 *     caller, every kernel (and memset) is enqueued on `stream` (a hipStream_t, may be NULL);
 *   - return 0 on success, a negative code otherwise (SAMBLE_E_*); the message of the last
 *     failure on the calling thread is available from samble_last_error();
 *   - re-entrant and thread-safe; the current HIP device is the caller's business;
 *   - strides are in ELEMENTS; "bs" = batch stride, "rs" = row stride;
 *   - fp32 everywhere, indices int32 (neighbour lists) or int64 (sampled indices, as the
 *     reference's (B,1,M) int64 tensor).  Attention kernels require D == 128 in this round.
 */
#ifndef SAMBLE_H
#define SAMBLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMBLE_OK 0
#define SAMBLE_E_INVALID (-22)     /* bad argument / unsupported shape */
#define SAMBLE_E_WORKSPACE (-12)   /* workspace too small */
#define SAMBLE_E_HIP_BASE (-1000)  /* -1000 - hipError_t */
#define SAMBLE_E_TIMEOUT (-110)    /* a grid barrier of the fused select chain gave up (reported through the chain's status
                                      word, samble_select_chain_status_async: no entry point synchronises to return it) */

/* score modes (reference models/downsample.py:315-340) */
#define SAMBLE_SCORE_SPARSE_COL_SUM 0
#define SAMBLE_SCORE_SPARSE_COL_AVG 1
#define SAMBLE_SCORE_SPARSE_COL_SQR 2
#define SAMBLE_SCORE_SPARSE_ROW_SUM 3
#define SAMBLE_SCORE_SPARSE_ROW_STD 4
/* sample modes (reference utils/ops.py:476, 507) */
#define SAMBLE_SAMPLE_TOPK 0
#define SAMBLE_SAMPLE_UNIFORM 1
#define SAMBLE_SAMPLE_RANDOM 2
/* plain top-k of the score without the +1e-8 (DownSampleGlobal, models/downsample.py:1403) and
 * bottom-k (topk(.., largest=False), models/downsample.py:1299-1301) */
#define SAMBLE_SAMPLE_TOP_RAW 3
#define SAMBLE_SAMPLE_BOTTOM_RAW 4
/* Boltzmann temperature: fixed inverse temperature, or members_in_bin / divisor
 * (reference utils/ops.py:524-548: mode_1 -> divisor 100, mode_3 -> divisor 200) */
#define SAMBLE_TEMP_FIXED 0
#define SAMBLE_TEMP_COUNT 1

const char* samble_version(void);
const char* samble_last_error(void);
/* Bumped whenever an exported prototype changes its argument list (entry points are otherwise only added): a consumer
 * compares samble_abi_version() with the SAMBLE_ABI_VERSION it was built against BEFORE the first call -- a stale library
 * or binding then fails at load time instead of passing shifted pointers (samble_amd/_lib.py does exactly that). */
#define SAMBLE_ABI_VERSION 6
int samble_abi_version(void);

/* ---- utils/ops.py:17-44  knn(a, b, k) -------------------------------------------------------
 * xq (B,C,Nq), xk (B,C,Nk) channel-major (the layout the reference's callers hold before
 * their permute(0,2,1)).  idx_out (B,Nq,K) int32, nearest first, self included when xq == xk.
 * dist_out (B,Nq,K) or NULL: POSITIVE distance of the reference-normalised points (centred on
 * xq's mean, divided by the mean unbiased per-channel std), i.e. -1 * the reference's first
 * return value.  K in {1,3,8,16,20,32,40,64}.
 * variant (stateless kernel choice, same results contract): 0 = the default for the shape (K in {16,32},
 * Nk >= 2 K, C = 128 or C <= 64: fused Gram + top-K on the fp16 matrix cores, operands = two fp16 planes
 * of the points centred on xq's mean (as the reference centres them) under a per-cloud power-of-two
 * scale, C < 64 zero-padded to 64 channels; other K with C <= 8: exact sum (a-b)^2 on the vector ALU);
 * SAMBLE_KNN_FP32_MFMA = the fused fp32-MFMA kernel for C in {64,128};
 * SAMBLE_KNN_TWO_KERNEL = key matrix through HBM + row select (what every other shape falls back to).
 * dist_out of the matrix-core kernels is formed as |a|^2 + |b|^2 - 2 a.b: absolute error ~2^-21 |a||b| in
 * d^2 (the self match comes out as sqrt of that, not as 0); the vector-ALU path is exact to fp32 rounding. */
#define SAMBLE_KNN_FP32_MFMA 1
#define SAMBLE_KNN_TWO_KERNEL 2
size_t samble_knn_workspace_bytes(int B, int C, int Nq, int Nk, int K, int variant);
int samble_knn_f32(const float* xq, int64_t q_bs, int Nq, const float* xk, int64_t k_bs, int Nk, int B, int C, int K,
                   int variant, int32_t* idx_out, float* dist_out, void* ws, size_t ws_bytes, void* stream);

/* ---- models/downsample.py:116-137  q_conv / k_conv / v_conv (bias-free 1x1 Conv1d) ------------
 * x (B,C,N) channel-major, tokens (C,nt) = bin_tokens[0], W (3C,C) row-major = [Wq; Wk; Wv]
 * (each conv weight (C,C,1) squeezed).  qkv (B,N+nt,3C) point-major rows [Q|K|V] with the given
 * strides: rows 0..N-1 project the points, rows N.. the bin tokens (Q of a token row is unused).
 * C = 128 in this round.  The backward writes dx (B,C,N) (NULL to skip), dW (3C,C) and
 * dtokens (C,nt) (dW NULL skips both), all deterministic. */
size_t samble_proj_workspace_bytes(int B, int N);
int samble_proj_fwd_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt, const float* W,
                        float* qkv, int64_t o_bs, int64_t o_rs, void* ws, size_t ws_bytes, void* stream);
int samble_proj_bwd_f32(const float* dqkv, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C,
                        int N, const float* tokens, int nt, const float* W, float* dx, int64_t dx_bs, float* dW,
                        float* dtokens, void* ws, size_t ws_bytes, void* stream);
/* The same two entries with the forward and dx on the bf16 matrix cores (fp32 operands split into three bf16
 * planes, six products each, fp32 accumulation: see the "split fp32 operands" block below); dW stays fp32 MFMA.
 * Same arguments; the workspaces also hold an operand image of W.
 * Wk / Wv (samble_proj_bwd_tri_f32 here, samble_proj_fwd_split_tri_f32 below): NULL, NULL -- W is the (3C, C) block
 * [Wq; Wk; Wv]; both non-NULL -- W is Wq and the three (C, C) weights are tensors of their own, as the reference's
 * q_conv / k_conv / v_conv hold them (models/downsample.py:54-56): no concatenation on the caller's side.  The
 * backward takes the three only together with w_tr_image (dx reads the image; W itself is read for the token rows). */
size_t samble_proj_fwd_tri_workspace_bytes(void);
int samble_proj_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt,
                            const float* W, float* qkv, int64_t o_bs, int64_t o_rs, void* ws, size_t ws_bytes,
                            void* stream);
size_t samble_proj_bwd_tri_workspace_bytes(int B, int N);
int samble_proj_bwd_tri_f32(const float* dqkv, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C,
                            int N, const float* tokens, int nt, const float* W, const float* Wk /* or NULL */,
                            const float* Wv /* or NULL */, const void* w_tr_image /* or NULL */, float* dx, int64_t dx_bs,
                            float* dW, float* dtokens, const float* dx_residual, void* ws, size_t ws_bytes, void* stream);
/* residual (samble_n2p_attn_fwd_f32, samble_linear_dx_tri_f32) / dx_residual (samble_proj_bwd_tri_f32): optional tensor of
 * the OUTPUT's layout that the kernel adds on its way out -- the `x + f(x)` of the layers around the sampler
 * (models/attention.py:187-192) and, backward, the gradient that reaches the same tensor along the residual branch --
 * instead of an elementwise pass of its own.  May alias the output (in-place accumulation); NULL: plain result. */

/* ---- models/downsample.py:139-153 + 242-252  energy / softmax / (all rows of) A @ V^T --------
 * Q (B,N,D), K and V (B,N+nt,D) point-major with explicit strides (the nt bin-token rows follow
 * the N point rows).  O (B,N,D) contiguous: row i = softmax(Q_i K^T / sqrt(D)) V, the row the
 * reference gathers if i is sampled; lse (B,N) log-sum-exp of the scaled logits over all N+nt
 * columns; tok (B,N,nt) = attention_bins_beforesoftmax; row_std (B,N) or NULL = unbiased std of each
 * row of the point-to-point block A[:, :N] (idx_mode "row_std", models/downsample.py:319-320). */
int samble_attn_fwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                        const float* V, int64_t v_bs, int64_t v_rs, int B, int N, int nt, int D, float* O, float* lse,
                        float* tok, float* row_std, void* stream);

/* idx_mode "col_sum" (models/downsample.py:315-318): colsum (B,N) = column sums of A[:, :N], from Q, the
 * point rows of K and the forward's lse. */
int samble_attn_colsum_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                           const float* lse, int B, int N, int D, float* colsum, void* stream);

/* score = stat with NaN -> 0 (models/downsample.py:342) and its per-cloud z-score, for the dense modes */
int samble_stat_score_f32(const float* stat, int B, int N, float* score, float* z, void* stream);

/* ---- models/downsample.py:300-344  calculate_attention_score (sparse_* modes) ----------------
 * nn (B,N,KN) int32 = neighbour lists from samble_knn_f32 on the layer input.  score (B,N),
 * z (B,N) = per-cloud z-score of the score (utils/ops.py:450-452), indeg_out (B,N) int32 or NULL
 * = kNN in-degree (sparse_num without its 1e-8).  Any N: up to 12 800 points a workgroup keeps the cloud's column
 * accumulators in LDS, longer clouds add the same integer terms to global memory directly (same bits). */
size_t samble_score_workspace_bytes(int B, int N);
int samble_sparse_score_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                            const float* lse, const int32_t* nn, int B, int N, int KN, int D, int mode, float* score,
                            float* z, int32_t* indeg_out, void* ws, size_t ws_bytes, void* stream);

/* z-score alone (utils/ops.py:450-452 / 517-520) */
int samble_zscore_f32(const float* score, int B, int N, float* z, void* stream);

/* ---- utils/ops.py:180-189  the nb-1 batch quantiles of all B*N z-scores ----------------------
 * out (nb-1) floats, descending.  The caller then averages over ranks (ops.py:191-199) and
 * blends with momentum (ops.py:201-233): five floats, host-side. */
size_t samble_quantiles_workspace_bytes(void);
int samble_batch_quantiles_f32(const float* z, int64_t n, int nb, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- utils/ops.py:201-233  boundary state from the (rank-averaged) quantiles ------------------
 * upper / lower: the two (nb) boundary vectors of the module state (upper[0] = +inf, lower[nb-1] = -inf).
 * first != 0 initialises them from the quantiles; otherwise blended in place,
 * old * momentum + one_minus_momentum * q, as two fp32 products and a sum (no FMA), the reference's order.
 * one_minus_momentum is passed separately because the reference forms it in double (1 - 0.99) before the
 * fp32 multiply. */
int samble_blend_boundaries_f32(const float* quantiles, float* upper, float* lower, int nb, float momentum,
                                float one_minus_momentum, int first, void* stream);

/* ---- utils/ops.py:454-463 + models/downsample.py:264-284  bin membership and bin weights ------
 * upper/lower (nb) = the two (1,1,1,nb) boundary tensors.  tok (B,N,nt), nt == nb or 1.
 * member (B,N) uint8 bit t = point in bin t; cap (B,nb) int32; w_pre (B,nb) = weights before
 * relu; w (B,nb).  relu_first = 1 for relu_mean_order == "relu_mean". */
int samble_bin_assign_f32(const float* z, const float* tok, int nt, const float* upper, const float* lower, int B,
                          int N, int nb, int relu_first, uint8_t* member, int32_t* cap, float* w_pre, float* w,
                          void* stream);

/* ---- utils/ops.py:385-432  calculate_num_points_to_choose ------------------------------------ */
int samble_alloc_counts_f32(const float* w, const int32_t* cap, int B, int nb, int M, int32_t* counts, void* stream);

/* ---- utils/ops.py:467-619  generating_downsampled_index --------------------------------------
 * noise (B*nb, N) = the Exp(1) draw torch.multinomial makes internally (row b*nb+t); NULL for
 * topk.  temp: inverse temperature (TEMP_FIXED) or divisor (TEMP_COUNT).  idx_out (B,M) int64.
 * N <= 16384 (a cloud's selection keys live in one workgroup's LDS): the longest cloud the sampler layer takes --
 * twice BASELINE's largest configuration (configs[4], N = 8192). */
int samble_bin_select_f32(const float* score, const float* z, const uint8_t* member, const int32_t* counts,
                          const float* noise, int B, int N, int nb, int M, int sample_mode, int temp_mode, float temp,
                          int64_t* idx_out, void* stream);
/* The same with the Exp(1) draw made inside the kernel (no (B*nb, N) tensor, no launch to fill it): element (row,
 * n) of the draw is Philox4x32-10 under (seed, offset) in the counter layout of torch's device generator -- offset
 * in 32-bit outputs, a multiple of 4; a caller that owns such a generator passes its state and advances it by 4 --
 * first output word, 23 bits into (0, 1), -log.  samble_exp1_noise_f32 writes that (rows = B*nb, N) tensor out:
 * samble_bin_select_f32 on it returns the same indices, bit for bit (tests/test_gpu_stages.py). */
int samble_bin_select_seeded_f32(const float* score, const float* z, const uint8_t* member, const int32_t* counts,
                                 uint64_t seed, uint64_t offset, int B, int N, int nb, int M, int sample_mode,
                                 int temp_mode, float temp, int64_t* idx_out, void* stream);
int samble_exp1_noise_f32(uint64_t seed, uint64_t offset, int rows, int N, float* noise, void* stream);

/* ---- models/downsample.py:242-252  gather of the sampled rows -> x_ds (B,D,M) ---------------- */
int samble_gather_rows_f32(const float* O, int64_t o_bs, int64_t o_rs, const int64_t* idx, int B, int M, int D,
                           float* x_ds, void* stream);

/* ---- utils/ops.py:136-145  gather_by_idx(pcd (B,C,N), idx (B,1,M)) -> (B,C,M) ---------------- */
int samble_gather_points_f32(const float* pcd, int B, int C, int N, const int64_t* idx, int M, float* out,
                             void* stream);

/* ---- models/embedding.py:7-39  EdgeConv body: conv1 + BN + LReLU + conv2 + BN + LReLU + max over K ------
 * A 1x1 conv over [x_i ; x_j - x_i] is a_i + b_j with two per-point projections; the caller folds
 * BatchNorm-1 into them (ap, bp: (B*N, 64) point-major).  nn (B,N,32) neighbour lists.  64 channels, K = 32.
 *   samble_edge_gather_sums_f32   S[p] = sum_k bp[j(p,k)], Q[p] = sum_k bp[j(p,k)]^2  (BN1 batch statistics and
 *                                 the backward's closed forms are built from these per-point sums)
 *   samble_edge_mlp_fwd_f32       y = W2 LReLU(ap_i + bp_j) per edge on the matrix cores (fp32 operands as three bf16
 *                                 planes, six products: fp32-equivalent sums; round 2-3: v_mfma_f32_32x32x2_f32); returns per (point,
 *                                 channel) max_k y and min_k y (LReLU o BN2 is monotone, so max_k commutes with
 *                                 it), the edge index that attains each (kmax, kmin; first one on ties) and
 *                                 samble_edge_partial_count() x (2,64) double partial sums of y, y^2
 *   samble_edge_mlp_bwd_f32       recomputes the edge tensors; dy = c0 + c1 y + [edge == kext] sdv;
 *                                 writes du (B*N, 32, 64) = gradient of the pre-activation ap_i + bp_j per
 *                                 edge, dusum (B*N, 64) = its sum over a point's edges (NULL: not wanted) and
 *                                 samble_edge_partial_count() x (64,64) partials of dW2 */
int samble_edge_partial_count(void);
int samble_edge_gather_sums_f32(const float* bp, const int32_t* nn, int B, int N, int K, int C, float* S, float* Q,
                                void* stream);
int samble_edge_mlp_fwd_f32(const float* ap, const float* bp, const int32_t* nn, const float* W2, int B, int N, int K,
                            int C, float* ymax, float* ymin, uint8_t* kmax, uint8_t* kmin, double* partials,
                            void* stream);
int samble_edge_mlp_bwd_f32(const float* ap, const float* bp, const int32_t* nn, const float* W2, const uint8_t* kext,
                            const float* sdv, const float* c0c1, int B, int N, int K, int C, float* du, float* dusum,
                            float* dw2_partials, void* stream);

/* ---- utils/ops.py:47-65, 83-112  select_neighbors / group: gather of the neighbour tensor -----
 * x (B,C,N), nn (B,N,K) -> out (B, C or 2C, N, K).  mode: 0 neighbor, 1 diff, 2 center_neighbor, 3 center_diff. */
int samble_group_gather_f32(const float* x, const int32_t* nn, int B, int C, int N, int K, int mode, float* out,
                            void* stream);

/* ---- utils/ops.py:622-643  farthest_point_sample -------------------------------------------
 * xyz (B,3,N) channel-major (the layout the models hold; the reference permutes to (B,N,3) first),
 * start (B) = the first centroid of each cloud (the reference draws it with torch.randint), out
 * (B,npoint) int64 in selection order.  N <= 32768 (points and distances in registers up to 16384;
 * beyond, distances in LDS and the points re-read from the L2 every round). */
int samble_fps_f32(const float* xyz, const int64_t* start, int B, int N, int npoint, int64_t* out, void* stream);

/* ---- models/attention.py:165-250  Neighbor2PointAttention, attention part (scalar_dot, asm dot) --
 * qkv (B,N,3C) point-major rows [Q|K|V] = samble_proj_fwd_f32 of the layer input with the three
 * Conv2d 1x1 weights (nt = 0); nn (B,N,KN) neighbour lists of the layer input.  diff != 0:
 * group_type "diff" (keys/values are neighbour minus centre: by linearity (Wx)_j - (Wx)_i), else
 * "neighbor".  out (B,C,N) = sum_j softmax_j(q_i . k_ij / sqrt(C/heads)) v_ij, heads laid out
 * head-major along C as the reference's split_heads.  C = 128; heads = 4 (N2P) or 1 (the local
 * attention of DownSampleLocal, models/downsample.py:885-963).  att: optional (B,N,KN) softmax
 * probabilities of the single head (the reference's attention_map), heads == 1 and KN <= 64 only. */
int samble_n2p_attn_fwd_f32(const float* qkv, int64_t bs, int64_t rs, const int32_t* nn, int B, int N, int KN, int C,
                            int heads, int diff, float* out, float* att, const float* residual, void* stream);

/* Backward of samble_n2p_attn_fwd_f32: g (B,C,N) = gradient of its output -> dqkv (B,N,3C) point-major
 * rows [dQ|dK|dV] (feed it to samble_proj_bwd_f32).  Deterministic (no atomics).  K <= 32. */
size_t samble_n2p_attn_bwd_workspace_bytes(int B, int N, int KN);
int samble_n2p_attn_bwd_f32(const float* qkv, int64_t bs, int64_t rs, const int32_t* nn, const float* g, int B, int N,
                            int KN, int C, int heads, int diff, float* dqkv, int64_t dbs, int64_t drs,
                            const int32_t* inv_order, const int32_t* inv_offsets, void* ws, size_t ws_bytes,
                            void* stream);

/* Inverse neighbour lists (optional input above, required below): inv_order (B*N*K) = the edge ids
 * e = (b*N + i)*K + k sorted by target b*N + nn[e], ties in ascending e (a STABLE sort of the neighbour
 * table; the host builds it once per kNN), inv_offsets (B*N + 1) = the group boundaries.  With them the
 * scatter-add over neighbours becomes a gather in a fixed order: deterministic, no atomics, no table scan.
 * samble_segment_sum_rows_f32: out[t][0:C] = sum over the incoming edges e of target t of
 * src[per_edge ? e : e / K][0:C]  (C = 64): EdgeConv's backward (sum of per-edge gradients / of the
 * sources' per-point rows over a point's reverse neighbours). */
int samble_segment_sum_rows_f32(const float* src, int64_t src_row_stride, const int32_t* inv_order, const int32_t* inv_offsets,
                                int K, int C, int per_edge, int64_t n_targets, float* out, void* stream);
/* both forms in one pass over the lists (EdgeConv's backward needs D = the per-edge sum and R = the per-point sum over the
 * same reverse neighbours): out_edge / out_point are bit for bit what two calls of the entry above give */
int samble_segment_sum_rows_pair_f32(const float* src_edge, int64_t edge_row_stride, const float* src_point,
                                     int64_t point_row_stride, const int32_t* inv_order, const int32_t* inv_offsets, int K, int C,
                                     int64_t n_targets, float* out_edge, float* out_point, void* stream);
/* The lists themselves, on the device and without a sort (a query lists a target at most once -- the rows of nn hold
 * distinct indices, as samble_knn_f32 writes them -- so "ascending edge id" inside a group is "ascending query": a
 * bit matrix targets x queries, prefix popcounts, one placement pass).  nn (B,N,KN) with entries in [0, N);
 * inv_order (B*N*KN), inv_offsets (B*N + 1), indegree (B*N) or NULL.  N < 65536. */
size_t samble_inverse_neighbors_workspace_bytes(int B, int N);
int samble_inverse_neighbors(const int32_t* nn, int B, int N, int KN, int32_t* inv_order, int32_t* inv_offsets,
                             int32_t* indegree, void* ws, size_t ws_bytes, void* stream);

/* ---- autograd of downsample.py:139-147 + 242-252 ----------------------------------------------
 * g (B,D,M) = gradient w.r.t. x_ds.  Writes dQ rows idx (other rows are zeroed), dK and dV rows
 * 0..N+nt-1, each with its own strides. */
size_t samble_attn_bwd_workspace_bytes(int B, int N, int M, int D);
/* variant (stateless kernel choice, same results): 0 = fused kernel (5 matrix products per tile),
 * 1 = two kernels, dQ query-stationary + dK/dV key-stationary (7 products) */
int samble_attn_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                        const float* V, int64_t v_bs, int64_t v_rs, const float* O, const float* lse,
                        const int64_t* idx, const float* g, int B, int N, int nt, int M, int D, float* dQ,
                        int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs, int64_t dk_rs, float* dV,
                        int64_t dv_bs, int64_t dv_rs, int variant, void* ws, size_t ws_bytes, void* stream);

/* ---- models/attention.py:253-355  Point2PointAttention: global self-attention over H heads of depth D ----------
 * Q, K, V (B, N, H*D) point-major rows with explicit strides (head h = columns h*D .. h*D + D - 1; the three may be
 * column blocks of one [Q|K|V] row); per head
 *     O = softmax((qk_mul * Q K^T + key_bias_j) / sqrt(D)) V          (no (B, H, N, N) tensor)
 * asm "dot" (attention.py:341): qk_mul = 1, key_bias NULL; "l2" (:343, energy -|q - k|^2): qk_mul = 2, key_bias_j =
 * -|k_j|^2; "l2+" (:345): qk_mul = -2, key_bias_j = +|k_j|^2 -- the row term |q_i|^2 is constant along a softmax row.
 * key_bias (B, H, N).  O (B, N, H*D), lse (B, H, N).  D in 4..128, a multiple of 4; all bases 16-byte aligned, strides
 * multiples of 4.  True fp32 matrix products (v_mfma_f32_32x32x2_f32).
 * Backward: dO (B, N, H*D) -> dQ, dK, dV (same layout as their operands; dK holds qk_mul * dS^T Q only) and
 * bias_grad (B, H, N) = d loss / d key_bias (the column sums of dS; required when key_bias is given: the caller adds
 * bias_grad_j * d key_bias_j / d k_j to dK).  Workspace: samble_attn_heads_bwd_workspace_bytes. */
int samble_attn_heads_fwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                              const float* V, int64_t v_bs, int64_t v_rs, const float* key_bias /* or NULL */, float qk_mul,
                              int B, int N, int H, int D, float* O, int64_t o_bs, int64_t o_rs, float* lse, void* stream);
size_t samble_attn_heads_bwd_workspace_bytes(int B, int N, int H);
int samble_attn_heads_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                              const float* V, int64_t v_bs, int64_t v_rs, const float* key_bias /* or NULL */, float qk_mul,
                              int B, int N, int H, int D, const float* O, int64_t o_bs, int64_t o_rs, const float* lse,
                              const float* dO, int64_t g_bs, int64_t g_rs, float* dQ, int64_t dq_bs, int64_t dq_rs,
                              float* dK, int64_t dk_bs, int64_t dk_rs, float* dV, int64_t dv_bs, int64_t dv_rs,
                              float* bias_grad /* or NULL */, void* ws, size_t ws_bytes, void* stream);

/* ---- two-pass forward with the logit map kept in HBM (same reference lines as samble_attn_fwd_f32) ---
 * In exact fp32 on MI355X reloading a logit (4 bytes) is ~3x cheaper than recomputing it (2*D flop),
 * so S = Q K^T / sqrt(D) is computed once and kept: smap (B, N, ld) row-major,
 * ld >= samble_attn_map_row_stride(N, nt) = 32 * ceil((N+nt)/32), columns N..N+nt-1 = token logits,
 * columns >= N+nt = -inf.  Caller-owned like every other buffer (538 MB at B=32, N=2048).
 *   samble_attn_stats_f32       pass 1, all N rows: smap, lse (B,N), tok (B,N,nt).  asm "dot": q_sqnorm =
 *                               k_sqnorm = NULL.  asm "l2" (downsample.py:154-175, S = -|q-k|^2/sqrt(D)):
 *                               q_sqnorm (B,N) = |q_i|^2 and k_sqnorm (B,ld), zero padded, = |k_j|^2
 *                               replaces q@k, /sqrt(D), the softmax normaliser and the token split
 *                               (models/downsample.py:139-153)
 *   samble_sparse_score_map_f32 = samble_sparse_score_f32 reading A_ij = exp(S_ij - lse_i) from the map
 *                               (ld == 0: `smap` is the compact (B, N, KN) neighbour-logit array of
 *                               samble_attn_stats_nl_tri_f32 and nn its ascending lists, see below)
 *                               (models/downsample.py:300-344)
 *   samble_attn_rows_fwd_f32    pass 2, the M sampled rows: x_ds (B,D,M) = softmax(S[idx]) V
 *                               replaces gather + @v + permute (models/downsample.py:242-252)
 *   samble_attn_rows_bwd_f32    = samble_attn_bwd_f32 reading S from the map (4 matrix products per
 *                               tile instead of 5) and O from x_ds (B,D,M).  ds_colsum: NULL for asm "dot".
 *                               For "l2" pass a (B, N+nt) buffer: it receives c_j = sum_i dS_ij, the token
 *                               logits are taken as -|q-k|^2, and dQ / dK come out as for "dot"; the caller
 *                               finishes with dQ *= 2, dK_j = 2 dK_j - 2 c_j k_j (dS rows sum to zero, so
 *                               the |q|^2 term drops out) */
int samble_attn_map_row_stride(int N, int nt);
int samble_attn_stats_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs, int B,
                          int N, int nt, int D, float* smap, int ld, float* lse, float* tok, const float* q_sqnorm,
                          const float* k_sqnorm, void* stream);
int samble_sparse_score_map_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int B, int N, int KN,
                                int mode, float* score, float* z, int32_t* indeg_out, void* ws, size_t ws_bytes,
                                void* stream);
int samble_attn_rows_fwd_f32(const float* smap, int ld, const float* lse, const float* V, int64_t v_bs, int64_t v_rs,
                             const int64_t* idx, int B, int N, int nt, int M, int D, float* x_ds, void* stream);
int samble_attn_rows_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                             const float* V, int64_t v_bs, int64_t v_rs, const float* smap, int ld, const float* lse,
                             const float* x_ds, const int64_t* idx, const float* g, int B, int N, int nt, int M, int D,
                             float* dQ, int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs, int64_t dk_rs, float* dV,
                             int64_t dv_bs, int64_t dv_rs, float* ds_colsum, void* ws, size_t ws_bytes, void* stream);

/* ---- the integer tail as two launches, or one (same reference lines as the stand-alone entries above) ----------------
 * For shapes samble_select_chain_supported(B, N, nb) accepts (one 1024-thread workgroup per cloud, all resident:
 * B <= min(128, CUs / 2), B * nb <= 1024, N <= 12800), over ONE caller-owned workspace of
 * samble_select_chain_workspace_bytes(B, N).
 * The grid barrier's poll is BOUNDED and its give-up is CLEAN: if the workgroups turn out not to be co-resident (a
 * co-tenant, a CU mask), the poll stops after ~1 s, raises the workspace's status word and every workgroup leaves the
 * kernel with valid placeholder integers for its cloud (all points in bin 0, all M picks from it) -- nothing traps, the
 * HIP context lives on and the kernels downstream run on memory they own.  The step's results are then meaningless:
 * the caller reads the word with samble_select_chain_status_async (an async copy into its pinned memory: 1 = SAMBLE_E_TIMEOUT)
 * whenever it next synchronises, and switches to the stand-alone stage entries above:
 *   samble_sparse_score_map_quantiles_f32  = samble_sparse_score_map_f32 (models/downsample.py:300-344, score + z,
 *       in-degree) + samble_batch_quantiles_f32 (utils/ops.py:180-189) in two launches (accumulation; finalize +
 *       radix select with grid barriers).  quantiles_out: NB floats -- the nb-1 quantiles and a validity count of 1.0 --
 *       or NULL (static boundaries: no quantiles).  A give-up (below) leaves all nb floats at ZERO.
 *   [the caller all-reduces (SUM) the nb floats over the ranks here, utils/ops.py:191-197: element nb-1 then holds the
 *    number of ranks whose quantiles are valid -- the world size, unless a rank's chain gave up, whose zeros then
 *    neither enter the mean nor count towards it]
 *   samble_bin_plan_f32  = samble_blend_boundaries_f32 (utils/ops.py:201-233; quantiles NULL: boundaries are used as
 *       they are) + samble_bin_assign_f32 + samble_alloc_counts_f32 (utils/ops.py:385-464) in one launch.  Must follow
 *       the call above on the same stream and workspace (it holds the barrier counters that call zeroed).
 *       quantile_divisor (device pointer, may be NULL = 1): every quantile is DIVIDED by *quantile_divisor on its way
 *       in -- utils/ops.py:199 `bin_boundaries / world_size` as the reference's true division, without a launch of its
 *       own: pass quantiles + (nb - 1) after the all-reduce above.
 *   Boundary state whose interior is NaN counts as "no state" in every entry that blends (this one, the one-launch chain,
 *   samble_blend_boundaries_f32): the quantiles initialise it as on a first call.  A caller that allocates a first call's
 *   state fills it with NaN, so that a chain which gave up before writing it leaves something the next call repairs.
 *   samble_select_chain_f32  = the two entries above in ONE launch, for a caller with nothing to exchange between them (a
 *       single rank): arguments as theirs (smap / lse / nn NULL when samble_attn_stats_nl_tri_f32 filled the workspace;
 *       want_quantiles 0: static boundaries).
 *   samble_select_chain_status_async  copies the workspace's status word to host_flag (pinned host memory) on the stream
 *       (for callers that pass no host_status).
 * host_status (all three launching entries, may be NULL): an int32 in pinned host memory (hipHostMalloc) that a barrier
 * which gives up sets to 1 with a system-scope store -- the status without a copy behind every launch; the caller clears it.
 * spin_budget (all three launching entries): poll rounds a grid barrier waits before it gives up; 0 = the default
 * (2^20, about a second).  A caller that knows its kernels share the device may shorten it; 0xFFFFFFFF injects the fault:
 * every barrier gives up without polling (tests of the give-up path).
 * Same integers as the stand-alone entries: both run the same device functions (csrc/select_dev.h). */
int samble_select_chain_supported(int B, int N, int nb);
size_t samble_select_chain_workspace_bytes(int B, int N);
int samble_select_chain_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int KN, int mode, const float* tok,
                            int nt, int want_quantiles, float* quantiles_out, float* upper, float* lower, int first,
                            float momentum, float one_minus_momentum, int B, int N, int nb, int relu_first, int M,
                            float* score, float* z, int32_t* indeg_out, uint8_t* member, int32_t* cap, float* w_pre, float* w,
                            int32_t* counts, void* ws, size_t ws_bytes, unsigned int spin_budget, int32_t* host_status,
                            void* stream);
int samble_select_chain_status_async(const void* ws, int B, int N, int32_t* host_flag, void* stream);
int samble_sparse_score_map_quantiles_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int B, int N,
                                          int KN, int mode, int nb, float* score, float* z, int32_t* indeg_out,
                                          float* quantiles_out, void* ws, size_t ws_bytes, unsigned int spin_budget,
                                          int32_t* host_status, void* stream);
int samble_bin_plan_f32(const float* z, const float* tok, int nt, const float* quantiles, const float* quantile_divisor,
                        float* upper, float* lower, int first, float momentum, float one_minus_momentum, int B, int N, int nb, int relu_first, int M,
                        uint8_t* member, int32_t* cap, float* w_pre, float* w, int32_t* counts, void* ws, size_t ws_bytes,
                        unsigned int spin_budget, int32_t* host_status, void* stream);

/* ---- the same passes on the bf16 matrix cores with split fp32 operands ------------------------------
 * An fp32 operand is carried as three bf16 planes h + m + l (all 24 significand bits); a product keeps
 * the six partial products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm), each one bf16 MFMA with fp32
 * accumulation: error vs fp64 equal to the fp32 MFMA's (tools/micro/split_mfma_bench.hip) at 2.6x its
 * rate.  Operands are handed over as IMAGES (layout: samble_amd/csrc/tri_dev.h), caller-owned:
 *   samble_tri_image_bytes(B, rows, transposed)   size of an image of a (B, rows, 128) matrix
 *   samble_tri_split_f32     fp32 rows -> row image (contraction over channels: Q, K) and / or transposed
 *                            image (contraction over rows: V in P V); either pointer may be NULL
 *   samble_tri_split_qkv_f32 one launch for the projection output (B, N+nt, 3*128) = [Q|K|V]: row image of Q (N
 *                            rows), row image of K and transposed image of V (N+nt rows); optionally (non-NULL)
 *                            the two images the backward wants: transposed K, row V
 *                            -- the two ROW images of K and V leave in their LOGIT FORM (below)
 *   samble_tri_k_logit_form  a row image (as samble_tri_split_f32 writes it) -> its logit form, in place, once: each
 *                            32-row tile is rewritten as two fp16 planes of its values x 2^e (e per tile; 2^-e kept in
 *                            the tile) -- for the products that are formed tile by tile with nothing accumulated across
 *                            tiles: S = Q K^T (row image of K; the query row is converted in registers under its own
 *                            scale) and the backward's dP = dO V^T (row image of V; the dO rows come from the
 *                            backward's own preparation in the same form).  Those kernels run three fp16 MFMA products
 *                            per k-step instead of six bf16 ones and take the scales out of the finished accumulator,
 *                            exactly: 22 significant bits per operand, power-of-two scaling, fp32 accumulation
 *                            (samble_amd/csrc/tri_dev.h; the backward's dV / dK accumulation does the same on images it
 *                            prepares itself; P V and dQ stay on three bf16 planes).  samble_tri_split_qkv_f32 and
 *                            samble_proj_fwd_split_tri_f32 apply it to k_image and v_rm_image themselves; the K row
 *                            image arguments of the three logit entries and the v_rm_image argument of
 *                            samble_attn_rows_bwd_tri_f32 expect it.  IDEMPOTENT since ABI 0.2: a converted tile carries a
 *                            tag (a bit pattern no three-plane tile can hold) beside its 2^-e and is left alone by a
 *                            second call; a RAW samble_tri_split_f32 image handed to those entries is still the caller's
 *                            error (they cannot check it without reading the image: wrong logits, no fault)
 *   samble_attn_stats_tri_f32  = samble_attn_stats_f32 on a Q image (N rows) and a K image (N+nt rows, logit form)
 *   samble_attn_rows_fwd_tri_f32 = samble_attn_rows_fwd_f32 on the transposed image of V (N+nt rows)
 *   samble_attn_rows_bwd_tri_f32 = samble_attn_rows_bwd_f32 (same outputs, same ds_colsum contract).  variant 0:
 *                            query-stationary dQ kernel that also writes a dS map, then dV and dK accumulated
 *                            key-stationary from the two maps (4 products per tile, no dQ slabs); variant 1: fused
 *                            dP / dV / dK kernel (5 products, no dS map).  Workspace:
 *                            samble_attn_rows_bwd_tri_workspace_bytes */
size_t samble_tri_image_bytes(int B, int rows, int transposed);
int samble_tri_split_f32(const float* src, int64_t bs, int64_t rs, int B, int rows, int D, void* rm_image,
                         void* tr_image, void* stream);
int samble_tri_k_logit_form(void* row_image, int B, int rows, void* stream);
int samble_attn_stats_tri_f32(const void* q_image, const void* k_image, int B, int N, int nt, int D, float* smap,
                              int ld, float* lse, float* tok, const float* q_sqnorm, const float* k_sqnorm,
                              void* stream);

int samble_tri_split_qkv_f32(const float* qkv, int64_t bs, int64_t rs, int B, int N, int nt, int D, void* q_image,
                             void* k_image, void* v_tr_image, void* k_tr_image, void* v_rm_image, void* stream);
/* samble_proj_fwd_tri_f32 and samble_tri_split_qkv_f32 in one: the projection kernel writes the operand images of
 * every full 32-point tile from its accumulators (the fp32 rows are not read back), a split launch over the remaining
 * tiles (token rows, ragged end) completes them.  Same bytes in qkv and in all images as the two separate calls.
 * k_tr_image / v_rm_image: both or neither.  Workspace: samble_proj_fwd_tri_workspace_bytes().
 * rows = SAMBLE_PROJ_ROWS_Q_ONLY: the K and V columns of the fp32 point rows are left UNWRITTEN wherever the images
 * carry the tile (every full 32-point tile; a ragged last tile and the token rows are always written in full) -- for
 * callers that go on with the images only, as the map-free forward and its backward do (they read the Q rows, the
 * token rows and the images): one sixth less written by the kernel. */
#define SAMBLE_PROJ_ROWS_ALL 0
#define SAMBLE_PROJ_ROWS_Q_ONLY 1
/* w_tr_image (optional, samble_proj_w_image_bytes() bytes): receives the transposed operand image of W, which
 * samble_proj_bwd_tri_f32 takes back as its w_tr_image (no split launch in the backward; W must be unchanged). */
size_t samble_proj_w_image_bytes(void);
int samble_proj_fwd_split_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt,
                                  const float* W, const float* Wk /* or NULL */, const float* Wv /* or NULL */, float* qkv,
                                  int64_t o_bs, int64_t o_rs, void* q_image, void* k_image, void* v_tr_image,
                                  void* k_tr_image, void* v_rm_image, int rows, void* w_tr_image, void* ws, size_t ws_bytes,
                                  void* stream);
size_t samble_attn_rows_bwd_tri_workspace_bytes(int B, int N, int M, int D);
int samble_attn_rows_bwd_tri_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs, int64_t k_rs,
                                 const float* V, int64_t v_bs, int64_t v_rs, const void* k_tr_image,
                                 const void* v_rm_image, const float* smap, int ld, const float* lse, const float* x_ds,
                                 const int64_t* idx, const float* g, int B, int N, int nt, int M, int D, float* dQ,
                                 int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs, int64_t dk_rs, float* dV,
                                 int64_t dv_bs, int64_t dv_rs, float* ds_colsum, int variant, void* ws, size_t ws_bytes,
                                 void* stream);
int samble_attn_rows_fwd_tri_f32(const float* smap, int ld, const float* lse, const void* v_tr_image,
                                 const int64_t* idx, int B, int N, int nt, int M, int D, float* x_ds, void* stream);

/* ---- the map-free forward (split-bf16 images, asm "dot", the sparse_* score modes) ------------------------------
 * The sparse score modes (models/downsample.py:300-344) read only the K kNN entries of each attention row, and the
 * gathered product (downsample.py:242-252) only the M sampled rows: the N x (N+nt) logit map never has to exist.
 *   samble_nn_prepare            neighbour lists (B, N, KN) of samble_knn_f32 -> the same lists in ascending index
 *                                order (nn_sorted) and one 32-bit word per (cloud, tile of 32 keys, query):
 *                                masks (B, ceil(N/32), N), bit k of word (b, t, i) set <=> 32 t + k in nn[b][i].
 *                                KN in {16, 32}.  samble_nn_masks_bytes(B, N) = size of `masks`.
 *                                clear / clear_bytes (optional): the kernel also zeroes that range on its way -- the
 *                                score workspace of the statistics pass that follows (then pass score_ws_cleared = 1
 *                                there: no memset launch between the two).
 *   samble_attn_stats_nl_tri_f32 = samble_attn_stats_tri_f32 without the map: lse (B,N), tok (B,N,nt) and
 *                                nl (B, N, KN), nl[b][i][k] = S[b][i][nn_sorted[b][i][k]] (bit-identical to the map's
 *                                entries).  1 <= KN <= 32.  nl may be NULL when score_ws is given:
 *                                with score_ws (samble_score_workspace_bytes or, for the fused chain,
 *                                samble_select_chain_workspace_bytes; ALL score_ws_bytes are zeroed first unless
 *                                score_ws_cleared says the caller did: samble_nn_prepare's `clear`) the
 *                                pass also accumulates the sparse_* statistics of score_mode (A_ij = exp(S_ij -
 *                                lse_i), i over all rows, j in nn_sorted[i]) into it as samble_sparse_score_map_f32
 *                                would (N <= 8192): follow with samble_sparse_score_map_f32 or its _quantiles variant passing
 *                                smap = NULL.
 *   samble_sparse_score_map_f32 / samble_sparse_score_map_quantiles_f32 take (smap = nl, ld = 0, nn = nn_sorted),
 *                                or smap = NULL: the statistics are in `ws` already (see above).
 *   samble_attn_rows_fwd_recompute_tri_f32 = samble_attn_rows_fwd_tri_f32 recomputing the logits of the M sampled
 *                                rows from the Q / K images; pmap (optional, (B, M, ld), ld >=
 *                                samble_attn_map_row_stride(N, nt)) receives P = softmax rows of the sampled rows,
 *                                which samble_attn_rows_bwd_tri_f32 takes as `smap` with variant
 *                                SAMBLE_ROWS_BWD_PMAP (no exponentials, no row indirection in the backward). */
#define SAMBLE_ROWS_BWD_FUSED_DKDV 1 /* fused dP / dV / dK kernel instead of the dS map (logit map only) */
#define SAMBLE_ROWS_BWD_PMAP 2       /* `smap` is the (B, M, ld) P map of samble_attn_rows_fwd_recompute_tri_f32 */
/* or-ed into `variant` of any backward entry: dQ has nt more rows behind its N point rows (same strides) -- the Q
 * columns of the projection's [Q|K|V] gradient block, whose token rows are keys only -- and they receive zeros
 * (no fill launch on the caller's side) */
#define SAMBLE_BWD_DQ_TOKEN_ROWS 8
size_t samble_nn_masks_bytes(int B, int N);
/* nn and nn_sorted 16-byte aligned (the rows move in 16-byte pieces); KN 16 or 32 */
int samble_nn_prepare(const int32_t* nn, int B, int N, int KN, int32_t* nn_sorted, uint32_t* masks, void* clear /* or NULL */,
                      size_t clear_bytes, void* stream);
int samble_attn_stats_nl_tri_f32(const void* q_image, const void* k_image, int B, int N, int nt, int D,
                                 const uint32_t* masks, int KN, float* nl, float* lse, float* tok,
                                 const int32_t* nn_sorted, int score_mode, void* score_ws, size_t score_ws_bytes,
                                 int score_ws_cleared, void* stream);
int samble_attn_rows_fwd_recompute_tri_f32(const void* q_image, const void* k_image, const void* v_tr_image,
                                           const float* lse, const int64_t* idx, int B, int N, int nt, int M, int D,
                                           float* x_ds, float* pmap, int ld, void* stream);

/* ---- models/upsample.py:181-213  UpSampleInterpolation.interpolate: the inverse-distance blend (csrc/interp.hip) ----
 * After the cross-set search (samble_knn_f32 with distances: idx (B,N,K) int32 into the M coarse points, dist (B,N,K)):
 *   samble_interp_blend_fwd_f32  out[b][c][n] = sum_k w_k feat[b][c][idx_k], w_k = (1/(d_k + 1e-8)) / sum (…) as the
 *                                reference forms it; feat (B,C,M) and out (B,C,N) channel-major; the weights (B,N,K) are
 *                                kept for the backward.  The (B,C,N,K) neighbour tensor is never built.
 *   samble_interp_blend_bwd_f32  dfeat[b][c][j] = sum over the edges (n,k) -> j of w g[b][c][n], in the order of the inverse
 *                                lists that samble_inverse_neighbors builds from idx (targets b N + j, M <= N):
 *                                deterministic, no atomics; works on point-major copies of g and dfeat in the workspace
 *                                (samble_interp_blend_bwd_workspace_bytes).  1 <= K <= 8. */
int samble_interp_blend_fwd_f32(const float* feat, int B, int C, int M, const int32_t* idx, const float* dist, int N, int K,
                                float* weights, float* out, void* stream);
size_t samble_interp_blend_bwd_workspace_bytes(int B, int C, int N, int M);
int samble_interp_blend_bwd_f32(const float* g, int B, int C, int N, const float* weights, const int32_t* inv_order,
                                const int32_t* inv_offsets, int K, int M, float* dfeat, void* ws, size_t ws_bytes,
                                void* stream);

/* ---- the closed forms around the fused EdgeConv's two MLP sweeps (csrc/edge_glue.hip) ------------------------------
 * EdgeConv (models/embedding.py:7-39) here = per-point projections a, b (B, N, 64) of conv1, samble_edge_mlp_fwd_f32 /
 * samble_edge_mlp_bwd_f32 over the B N K edges (above), and between them the two BatchNorm2d layers in TRAINING mode,
 * the LeakyReLU and the max over the K neighbours in closed form on (points, 64) rows.  These entries are that closed
 * form: per-channel sums in double, workgroup partials added in index order (deterministic, no atomics, no host round
 * trip).  One caller-owned block each of samble_edge_glue_constants_bytes() (floats the sweeps read: BN scales / shifts,
 * the backward's correction terms), samble_edge_glue_statistics_bytes() (doubles: mu / sigma of both layers; keep both
 * from forward to backward) and samble_edge_glue_partials_bytes() (scratch).
 *   samble_edge_bn1_f32       S_i = sum_k b_j, Q_i = sum_k b_j^2; BN1 batch statistics over the edges; the folded
 *                             projections ap = a sc1 + sh1, bp = b sc1 the sweeps read; running statistics updated in
 *                             place when given (momentum as nn.BatchNorm2d: unbiased running variance)
 *   samble_edge_bn2_out_f32   BN2 statistics from samble_edge_mlp_fwd_f32's partials; ext / kext = the max (gamma2 >= 0)
 *                             or min of conv2's output over a point's edges and the edge that attains it; out (B, 64, N)
 *                             = LeakyReLU(BN2(ext)), channel-major
 *   samble_edge_bwd_pre_f32   g (B, 64, N) -> sdv = sc2 g LeakyReLU'(v) rows for samble_edge_mlp_bwd_f32, whose `c0c1`
 *                             argument is constants + 256 after this call; d gamma2, d beta2
 *   samble_edge_bwd_post_f32  from dusum (samble_edge_mlp_bwd_f32), D_i = sum of du over the INCOMING edges of i and R_i = sum of a over them
 *                             (samble_segment_sum_rows_f32 over samble_inverse_neighbors' lists; indeg = their counts):
 *                             d gamma1, d beta1, the per-point gradients da, db and dW2 (64, 64) = the ordered sum of
 *                             samble_edge_mlp_bwd_f32's per-wave partials
 * nn.SyncBatchNorm (the reference trainer converts every BatchNorm, train_modelnet.py:245-246): each of the four entries
 * runs in two halves around the CALLER's all-reduce (SUM) of `pooled` -- samble_edge_glue_pooled_bytes() of float64:
 * [total 0 (64) | total 1 (64) | edge count] -- over its process group:
 *   phase SAMBLE_EDGE_ALL    one rank: everything (pooled unused, may be NULL)
 *   phase SAMBLE_EDGE_SUMS   the statistics kernel and this rank's totals into pooled (backward entries: d gamma / d beta
 *                            as well -- they stay per rank, DistributedDataParallel averages parameter gradients itself)
 *   phase SAMBLE_EDGE_APPLY  from the constants on, with the all-reduced pooled in place of the partials */
#define SAMBLE_EDGE_ALL 0
#define SAMBLE_EDGE_SUMS 1
#define SAMBLE_EDGE_APPLY 2
size_t samble_edge_glue_pooled_bytes(void);
size_t samble_edge_glue_partials_bytes(void);
size_t samble_edge_glue_constants_bytes(void);
size_t samble_edge_glue_statistics_bytes(void);
/* a, b: rows of C floats at ab_row_stride floats (= C for two separate tensors; 2 C for the two halves [a | b] of one
 * (B, N, 2 C) projection output: the layer's two per-point projections come from ONE 1x1 convolution and are read where they
 * are).  da, db of samble_edge_bwd_post_f32 likewise at dab_row_stride: written as the halves of one (B, N, 2 C) gradient. */
int samble_edge_bn1_f32(const float* a, const float* b, int64_t ab_row_stride, const int32_t* nn, int B, int N, int K, int C, const float* gamma1,
                        const float* beta1, float eps, float* running_mean, float* running_var, float momentum,
                        int64_t* num_batches_tracked, float* S, float* Q, float* ap, float* bp, float* constants,
                        double* statistics, double* partials, int phase, double* pooled, void* stream);
int samble_edge_bn2_out_f32(const float* ymax, const float* ymin, const uint8_t* kmax, const uint8_t* kmin,
                            const double* mlp_partials, int n_partials, int B, int N, int C, const float* gamma2,
                            const float* beta2, float eps, float* running_mean, float* running_var, float momentum,
                            int64_t* num_batches_tracked, float* constants, double* statistics, float* ext, uint8_t* kext,
                            float* out, int phase, double* pooled, void* stream);
/* (num_batches_tracked, both entries: nn.BatchNorm2d's int64 counter on the device, incremented by the kernel; may be NULL) */
int samble_edge_bwd_pre_f32(const float* g, const float* ext, int B, int N, int C, const float* gamma2, float* constants,
                            const double* statistics, float* sdv, float* dgamma2, float* dbeta2, double* partials,
                            int phase, double* pooled, void* stream);
int samble_edge_bwd_post_f32(const float* a, const float* b, int64_t ab_row_stride, const float* S, const float* R,
                             const float* dusum, const float* D, const int32_t* indeg, int B, int N, int K, int C,
                             float* constants, const double* statistics, const float* dw2_partials, int n_partials, float* da,
                             float* db, int64_t dab_row_stride, float* dgamma1, float* dbeta1, float* dW2, double* partials,
                             int phase, double* pooled, void* stream);

/* ---- 1x1 convolutions over C = 128 input channels next to the neighbour / sampler kernels (csrc/linear.hip) --------
 * Replace, in the layers that sandwich the sampler:
 *   models/attention.py:187-192 (Neighbor2PointAttention / Point2PointAttention): ff = Conv1d(128 -> 512, bias=False),
 *       LeakyReLU(0.2), Conv1d(512 -> 128, bias=False) and its autograd;
 *   models/cls_model.py:136 (FeatureLearningBlock): conv_list[i](x).max(dim=-1)[0] -- Conv1d(128 -> 1024, bias=False)
 *       followed by the maximum over the points -- and its autograd.
 *   models/embedding.py:20-28 (EdgeConv): conv1 over [x_i ; x_j - x_i] = two per-point projections of x (C = 3 / 64
 *       input channels -> 64 + 64 outputs) and their autograd.
 * fp32 in, fp32 out; products on the bf16 matrix cores with each fp32 operand split into three bf16 planes (six partial
 * products, fp32 accumulation: fp32-equivalent, csrc/tri_dev.h).  Layouts: x and dx channel-major (B, C, N) as the
 * modules hold them (x_bs = elements between clouds), 1 <= C <= 128 (the kernels contract over 128 channels: the missing
 * ones are zeros, and so are W's columns C .. 127); the wide side point-major rows (B, N, O) (o_bs / o_rs = elements
 * between clouds / rows; rows 16-byte aligned); W (O, 128) row-major, O a multiple of 32, handed over as OPERAND IMAGES
 * of samble_linear_image_bytes(O) bytes each, written by samble_linear_weight_images_f32 (either pointer may be NULL):
 *   rm_image  contraction over the 128 channels   (samble_linear_fwd_tri_f32, samble_linear_amax_fwd_tri_f32)
 *   tr_image  contraction over the O outputs      (samble_linear_dx_tri_f32)
 *   samble_linear_fwd_tri_f32       out[b][n][o] = epilogue(sum_c W[o][c] x[b][c][n]); epilogue SAMBLE_LIN_PLAIN,
 *                                   SAMBLE_LIN_LEAKY (LeakyReLU 0.2), SAMBLE_LIN_LEAKY_MASK (times 1 where ref > 0, else
 *                                   0.2; ref: a (B, N, O) tensor laid out like out -- the backward of the activation);
 *                                   SAMBLE_LIN_LEAKY_BITS / SAMBLE_LIN_LEAKY_MASK_BITS: the same pair with the activation's
 *                                   sign as ONE BIT per value -- `ref` then points to samble_linear_sign_bytes(B, N, O)
 *                                   bytes of sign words, WRITTEN by LEAKY_BITS (beside out) and read by LEAKY_MASK_BITS
 *                                   in place of the activation itself (4 MB instead of 134 MB at B = 32, N = 2048, O = 512);
 *                                   results bit for bit those of LEAKY / LEAKY_MASK
 *   samble_linear_amax_fwd_tri_f32  y[b][o] = max_n sum_c W[o][c] x[b][c][n], arg[b][o] = the first point that reaches
 *                                   it; the (B, O, N) tensor is never written
 *   samble_linear_dx_tri_f32        dx[b][c][n] = sum_o W[o][c] g[b][n][o]
 *   samble_linear_dw_tri_f32        dW[o][c] = sum_{b,n} g[b][n][o] x[b][c][n] as (O, 128); O a multiple of 128; per-workgroup
 *                                   partials in ws, summed in a fixed order (deterministic, no float atomics)
 *   samble_amax_bwd_f32             backward of samble_linear_amax_fwd_tri_f32 for upstream gy (B, O): the arg-max columns
 *                                   are ADDED to dx_inout (zeros, or another gradient of the same tensor: one wave owns a
 *                                   column) and dW (O, 128) is written; outputs grouped by point with a
 *                                   counting sort, sums in ascending output / cloud order (deterministic).  An exact tie of
 *                                   the maximum goes to the lowest point index (torch.amax's backward splits it evenly). */
#define SAMBLE_LIN_PLAIN 0
#define SAMBLE_LIN_LEAKY 1
#define SAMBLE_LIN_LEAKY_MASK 2
#define SAMBLE_LIN_LEAKY_BITS 4
#define SAMBLE_LIN_LEAKY_MASK_BITS 5
size_t samble_linear_sign_bytes(int B, int N, int O);
size_t samble_linear_image_bytes(int O);
int samble_linear_weight_images_f32(const float* W, int O, int C, void* rm_image, void* tr_image, void* stream);
/* ... of the TRANSPOSE of Wt (128, O) row-major: the images samble_linear_weight_images_f32 would write for Wt^T (O, 128),
 * without the transposing copy (models/attention.py:187-192: the second FFN convolution's weight (128, 512, 1) feeds
 * samble_linear_dx_tri_f32 as W^T).  Two-plane build of csrc/linear.hip only (the default). */
int samble_linear_weight_images_t_f32(const float* Wt, int O, int C, void* rm_image, void* tr_image, void* stream);
/* ... and both weights of a feed-forward layer (models/attention.py:187-192: Conv1d 128->H, Conv1d H->128) in one launch:
   W1 (O1, 128) as it is, W2t (128, O2) transposed, each image pointer may be null */
int samble_linear_weight_images_pair_f32(const float* W1, int O1, void* rm1_image, void* tr1_image, const float* W2t, int O2,
                                         void* rm2_image, void* tr2_image, void* stream);
int samble_linear_two_plane_build(void); /* 1 in the default build; 0 in a -DSAMBLE_LIN_DUO=0 (three bf16 planes) A/B build */
int samble_linear_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O, int epilogue,
                              const float* ref, float* out, int64_t o_bs, int64_t o_rs, void* stream);
/* A feed-forward layer's two convolutions in ONE sweep over the points (models/attention.py:187-192 `ff` and its input
 * gradient): mid (B, N, H) = epilogue(Wa x) as samble_linear_fwd_tri_f32 writes it (may be NULL: not kept), then
 * out (B, 128, N) = Wb^T-contraction of mid over H as samble_linear_dx_tri_f32 forms it (+ residual, which may alias out) --
 * without reading mid back (it is what the first product's accumulators hold).  wa_rm_image: row image of Wa (H, 128);
 * wb_tr_image: transposed image of Wb (H, 128).  epilogue SAMBLE_LIN_LEAKY_BITS (forward: Wa = W1, Wb = W2^T; writes the
 * sign words) or SAMBLE_LIN_LEAKY_MASK_BITS (backward: Wa = W2^T, Wb = W1; reads them).  x 128 channels.  Two-plane build. */
int samble_linear_chain_f32(const float* x, int64_t x_bs, int B, int N, const void* wa_rm_image, const void* wb_tr_image, int H,
                            int epilogue, float* mid, int64_t mid_bs, int64_t mid_rs, void* sign_words, float* out,
                            int64_t out_bs, const float* residual, void* stream);
size_t samble_linear_amax_workspace_bytes(int B, int N, int O);
int samble_linear_amax_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O, float* y,
                                   int32_t* arg, void* ws, size_t ws_bytes, void* stream);
int samble_linear_dx_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const void* w_tr_image, int O, int B, int C, int N,
                             float* dx, int64_t dx_bs, const float* residual, void* stream);
size_t samble_linear_dw_workspace_bytes(int B, int N, int O);
int samble_linear_dw_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C, int N,
                             int O, float* dW, void* ws, size_t ws_bytes, void* stream);
/* ... written transposed: dWt (128, O) row-major -- that convolution's weight gradient in its parameter's own layout */
int samble_linear_dw_t_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C, int N,
                               int O, float* dWt, void* ws, size_t ws_bytes, void* stream);
/* Channel-major in, channel-major out (a 1x1 Conv1d as the reference's modules hold their tensors: the interpolation
   layers' `conv` and `res_conv`, models/upsample.py:142-150 -- F.conv1d there):
     samble_linear_fwd_cm_f32   out (B, O, N) [cloud stride o_bs] = W x, or out += W x when accumulate != 0 (the second
                                128 channels of a 256-channel input, without a concatenated copy of the two inputs)
     samble_linear_dw_cm_f32    dW (O, 128) = sum over clouds and points of g x^T, g (B, O, N) [cloud stride g_bs]
                                channel-major; workspace: samble_linear_dw_workspace_bytes(B, N, O); O a multiple of 128
   (the input gradient is the forward entry again with the image of W^T: samble_linear_weight_images_t_f32) */
int samble_linear_fwd_cm_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O, int accumulate,
                             float* out, int64_t o_bs, void* stream);
int samble_linear_dw_cm_f32(const float* g, int64_t g_bs, const float* x, int64_t x_bs, int B, int C, int N, int O, float* dW,
                            void* ws, size_t ws_bytes, void* stream);
/* nn.BatchNorm1d in training mode on x (B, C, N) channel-major (the attention layers' bn1 / bn2,
   models/attention.py:187-192): out = (x - mean) / sqrt(var + eps) * gamma + beta with the batch statistics over (B, N);
   save_mean / save_invstd (C) go to the backward; running_mean / running_var (may be null) get torch's momentum update
   (unbiased variance).  Statistics in float64, summed in a fixed order.  workspace (every entry that takes one):
   samble_bn_train_workspace_bytes(B, C).
     samble_bn_train_fwd_f32        one rank: statistics + normalisation (two launches)
     samble_bn_train_bwd_f32        one rank: dx = gamma invstd (dy - mean(dy) - xhat mean(dy xhat)), dgamma = sum dy xhat,
                                    dbeta = sum dy (may be null); dx may be dy
   nn.SyncBatchNorm (the reference trainer converts every BatchNorm, train_modelnet.py:245-246): the same kernels in two
   halves, between which the CALLER all-reduces (SUM) `pooled` over its process group --
     samble_bn_train_stats_f32      pooled (2 C + 1 float64) = [sum x | sum x^2 | B N] of this rank
     samble_bn_train_apply_f32      the normalisation from the all-reduced block (count = pooled[2 C])
     samble_bn_train_bwd_sums_f32   pooled (2 C float64) = [sum dy | sum dy xhat] of this rank; dgamma / dbeta (may be null)
                                    = the same sums as float32: they stay per rank (DistributedDataParallel averages
                                    parameter gradients itself, as torch's SyncBatchNorm leaves them)
     samble_bn_train_bwd_apply_f32  dx from the all-reduced sums; count = device pointer to the pooled element count
                                    (the forward's pooled + 2 C)
   act_slope: the LeakyReLU behind the normalisation (models/upsample.py:142-150: Conv1d, BatchNorm1d, LeakyReLU(0.2)) in the
   same passes -- forward out = v > 0 ? v : act_slope v in the normalisation's epilogue; backward: dy is the gradient of the
   ACTIVATED output and is masked by the sign of the normalised value, re-formed from x, save_mean, save_invstd, gamma and
   beta by the same float expression (the activation's output is never read).  1.0f = no activation (beta unused). */
size_t samble_bn_train_workspace_bytes(int B, int C);
int samble_bn_train_fwd_f32(const float* x, int B, int C, int N, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, float* out, float* save_mean, float* save_invstd,
                            float act_slope, void* ws, size_t ws_bytes, void* stream);
int samble_bn_train_bwd_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean, const float* save_invstd,
                            const float* gamma, float* dx, float* dgamma, float* dbeta, const float* beta, float act_slope,
                            void* ws, size_t ws_bytes, void* stream);
int samble_bn_train_stats_f32(const float* x, int B, int C, int N, double* pooled, void* ws, size_t ws_bytes, void* stream);
int samble_bn_train_apply_f32(const float* x, int B, int C, int N, const double* pooled, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* out, float* save_mean,
                              float* save_invstd, float act_slope, void* stream);
int samble_bn_train_bwd_sums_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean,
                                 const float* save_invstd, double* pooled, float* dgamma, float* dbeta, const float* gamma,
                                 const float* beta, float act_slope, void* ws, size_t ws_bytes, void* stream);
int samble_bn_train_bwd_apply_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean,
                                  const float* save_invstd, const float* gamma, const double* pooled, const double* count,
                                  float* dx, const float* beta, float act_slope, void* stream);
size_t samble_amax_bwd_workspace_bytes(int B, int N, int O);
int samble_amax_bwd_f32(const float* x, int64_t x_bs, int B, int C, int N, const int32_t* arg, const float* gy, const float* W,
                        int O, float* dx_inout, int64_t dx_bs, float* dW, void* ws, size_t ws_bytes, void* stream);

/* ---- measurement hook (bench.py; the only entry points that are not part of the path) ------------------
 * The library records HIP events around its own launches of the selected kernels, on the stream each is
 * launched on.  samble_timing_select(mask): bit SAMBLE_T_x set = time kernel x (0 = off; resets the samples);
 * samble_timing_read(id, ...) waits for that kernel's recorded launches (the last 32 at most) and returns
 * their mean / median duration in ms and how many launches were seen.  No effect on results; nothing on the data path
 * reads it.  THE ONE EXCEPTION to "re-entrant, no mutable global state" above: the selection mask and the event tables
 * are process-wide (csrc/abi.hip g_time_*), so select / read must not run while another thread is inside the library
 * (a benchmark selects, runs its steps, synchronises, reads).  With the mask at 0 -- the default, and what every test and
 * the product path run under -- the hook is two predictable branches per launch and touches nothing shared. */
#define SAMBLE_T_ATTN_STATS 1   /* attn_stats(_tri / _nl_tri): QK^T + softmax statistics over all rows */
#define SAMBLE_T_ATTN_ROWS 2    /* attn_rows(_tri / _rc_tri): P V of the sampled rows */
#define SAMBLE_T_BWD_DV 3       /* bwd_kacc_tri (dV) */
#define SAMBLE_T_KNN 4          /* knn_duo / knn_stream: fused Gram + top-K */
#define SAMBLE_T_ATTN_FWD 5     /* attn_fwd: single-pass flash forward */
#define SAMBLE_T_BWD_DQ 6       /* bwd_dq_tri (dP, dS map, dQ) */
#define SAMBLE_T_BWD_DK 7       /* bwd_kacc_tri<1> (dK) */
#define SAMBLE_T_PROJ_FWD 8
#define SAMBLE_T_PROJ_DX 9
#define SAMBLE_T_PROJ_DW 10
#define SAMBLE_T_TRI_SPLIT 11   /* operand images of Q, K, V */
#define SAMBLE_T_KNN_PREP 12    /* cloud means + centred operand image + norms of the points */
#define SAMBLE_T_SPARSE_SCORE 13
#define SAMBLE_T_QUANTILES 14
#define SAMBLE_T_BIN_ASSIGN 15
#define SAMBLE_T_ALLOC_COUNTS 16
#define SAMBLE_T_BIN_SELECT 17
#define SAMBLE_T_BWD_PREP 18
#define SAMBLE_T_GATHER 19
#define SAMBLE_T_BWD_ROWS_F32 22 /* bwd_rows (fp32-MFMA backward over the key blocks) */
#define SAMBLE_T_NN_PREPARE 23   /* nn_prepare: ascending neighbour lists + per-tile membership words */
#define SAMBLE_T_EDGE_FWD 24     /* edge_mlp_fwd (EdgeConv body) */
#define SAMBLE_T_EDGE_BWD 25     /* edge_mlp_bwd */
#define SAMBLE_T_N2P_FWD 26      /* n2p_attn_fwd */
#define SAMBLE_T_N2P_BWD 27      /* n2p backward: transpose + n2p_bwd_point + n2p_bwd_gather / scatter */
#define SAMBLE_T_INV_NN 28       /* samble_inverse_neighbors: mark + count + place */
#define SAMBLE_T_SEG_SUM 29      /* seg_sum_rows64 */
#define SAMBLE_T_EDGE_SUMS 30    /* edge_gather_sums */
#define SAMBLE_T_KNN_SMALL 31    /* knn_smallc_fused (xyz) */
#define SAMBLE_T_LIN_FWD 32      /* lin_fwd_tri (1x1 convolution, point-major output, leaky / mask epilogues) */
#define SAMBLE_T_LIN_DX 33       /* lin_dx_tri */
#define SAMBLE_T_LIN_DW 34       /* lin_dw_tri + its partial sum */
#define SAMBLE_T_LIN_AMAX 35     /* lin_fwd_tri<amax> + reduce: 1x1 convolution and max over the points */
#define SAMBLE_T_LIN_AMAX_BWD 36 /* amax_bwd + the sum over the clouds */
#define SAMBLE_T_BN_FWD 37       /* bn_stats + bn_apply: BatchNorm1d training forward */
#define SAMBLE_T_LIN_CHAIN 38    /* lin_chain: a feed-forward layer's two convolutions in one sweep */
#define SAMBLE_T_BN_BWD 39       /* bn_bwd_reduce + bn_bwd_apply: BatchNorm1d training backward */
int samble_timing_select(uint64_t kernel_mask);
int samble_timing_read(int kernel_id, float* mean_ms, float* median_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* SAMBLE_H */
