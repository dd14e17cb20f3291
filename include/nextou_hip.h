/*
 * nextou_hip.h — C-ABI of libnextou_hip.so, the MI355X (gfx950) implementation of the
 * NexToU graph hot path.
 *
 * The reference (PengchengShi1220/NexToU) is pure Python/PyTorch and has no FFI of its own;
 * every entry point below replaces a *sequence of ATen ops* in the reference and cites the
 * reference file:line it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add to call them.
 *
 * Conventions
 *  - all pointers are DEVICE pointers (HBM) unless the name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing here
 *    synchronises the device, allocates device memory or blocks the host, so every call is
 *    hipGraph-capturable;
 *  - tensors are dense row-major in the reference's own layouts:
 *        features  (B, C, N)        "channel-major", N contiguous   [reference (B,C,N,1)]
 *        nn_idx    (B, N, K)  int32 neighbour ids in [0, M)
 *        relpos    (N, M)     float, shared by every batch element [reference (1,N,M)]
 *  - return value: 0 on success, a positive hipError_t on a HIP failure, a negative
 *    NEXTOU_E* code on an argument error; nextou_last_error() returns a message.
 */
#ifndef NEXTOU_HIP_H
#define NEXTOU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEXTOU_ABI_VERSION 14

#define NEXTOU_EINVAL   (-1)  /* bad argument (null pointer, non-positive size, k > M ...) */
#define NEXTOU_ENOSPACE (-2)  /* workspace too small */
#define NEXTOU_ENOTSUP  (-3)  /* shape not supported by the requested algorithm */

/* algorithm selector of nextou_knn_graph */
#define NEXTOU_KNN_AUTO   0   /* fused MFMA kernel when k <= 32, else the naive pair */
#define NEXTOU_KNN_FUSED  1   /* fused normalised-distance (f32 MFMA) + streaming top-k */
#define NEXTOU_KNN_NAIVE  2   /* materialise (B,N,M) distances, then one wave per row */

typedef void* nextou_stream_t;

int nextou_abi_version(void);
const char* nextou_last_error(void);

/* Optional launch profiler (measurement only, used by bench.py's `roofline` object): while enabled,
 * every kernel launch of this library is bracketed by HIP events on its own stream.
 * nextou_profile_enable(n): n > 0 (re)starts recording with room for n launches, n == 0 stops.
 * nextou_profile_report(): after a device synchronise, writes a JSON array aggregated per launch
 *   label — {"kernel", "bound": "hbm"|"mfma", "launches", "ms", "work"} where work is the summed
 *   ALGORITHMIC bytes (hbm) or flops (mfma) — and returns its length (0 = buffer too small).
 * nextou_profile_dropped(): launches since the last enable that found the record pool full and went unrecorded (a caller
 *   that divides the report by a step count checks this is 0).
 * Do not enable while capturing a hipGraph. */
int nextou_profile_enable(int max_records);
size_t nextou_profile_report(char* buf, size_t cap);
int nextou_profile_dropped(void);

/* ------------------------------------------------------------------------------------------
 * K1  dense kNN graph.
 * Replaces DenseDilatedKnnGraph.forward + {dense,xy_dense}_knn_matrix + *_pairwise_distance
 *   (reference network_architecture/torch_edge.py:151-163, 58-110, 12-55): F.normalize over
 *   channels, dist = (|x|^2 + (-2 x.y)) + |y|^2 (+ relative_pos), topk(-dist, K).
 * Canonical arithmetic (bit-exact contract with oracle/nextou_oracle.c):
 *   den  = max(sqrtf(fma-chain_c x^2), 1e-12);  xn = x / den          (IEEE division)
 *   xs   = fma-chain_c xn^2 ;  inner = fma-chain_c xn*yn  (c ascending, one rounding per step)
 *   dist = ((xs + (-2*inner)) + ys) [+ relpos]
 *   neighbours = K smallest by (dist, index) lexicographic, emitted in that order.
 *   Non-finite input: a distance that comes out NaN or +inf is ordered as FLT_MAX (after every finite distance, by
 *   index among themselves), so every output id lies in [0, M) whatever the features hold — the reference's topk
 *   puts NaN first instead; neither order means anything, but an id outside the candidates would be an out-of-bounds
 *   gather in the aggregation (tests/test_gpu_knn_small.py::test_non_finite_features_still_give_valid_neighbour_ids).
 * y == NULL selects the self graph (M must equal N).  K here is k*dilation of the reference.
 * normalize = 1: the F.normalize step is part of the call (DenseDilatedKnnGraph.forward);
 * normalize = 0: inputs are taken as they are (dense_knn_matrix / xy_dense_knn_matrix called
 *   directly, torch_edge.py:58-110).
 * workspace: nextou_knn_workspace_bytes() bytes of device scratch (normalised copies, norms,
 *   and for the NAIVE algorithm the (B,N,M) distance matrix).
 * ---------------------------------------------------------------------------------------- */
size_t nextou_knn_workspace_bytes(int B, int C, int N, int M, int K, int has_y, int algo);

int nextou_knn_graph(const float* x, const float* y, const float* relpos,
                     int32_t* nn_idx,
                     void* workspace, size_t workspace_bytes,
                     int B, int C, int N, int M, int K, int algo, int normalize,
                     nextou_stream_t stream);

/* Materialised squared-distance matrix WITHOUT normalisation or bias — the reference's public
 * helpers pairwise_distance / part_pairwise_distance / xy_pairwise_distance
 * (torch_edge.py:12-23, 26-39, 42-55):  dist[b, n-row_start, m] = (|x_n|^2 + (-2 x_n.y_m)) + |y_m|^2
 * for rows n in [row_start, row_end).  y == NULL: y = x.  Same canonical fma-chain arithmetic. */
size_t nextou_pairwise_workspace_bytes(int B, int N, int M, int has_y);
int nextou_pairwise_distance(const float* x, const float* y, float* dist,
                             void* workspace, size_t workspace_bytes,
                             int B, int C, int N, int M, int row_start, int row_end,
                             nextou_stream_t stream);

/* Expands nn_idx (B,N,K_total) int32 into the reference's edge_index (2,B,N,K_total/dilation)
 * int64: [0] = nn_idx[..., ::dilation], [1] = centre ids (arange(N) broadcast).
 * Replaces torch_edge.py:89-90,109-110 (arange/repeat/transpose/stack) and :126-136 (::d). */
int nextou_edge_index_i64(const int32_t* nn_idx, int64_t* edge_index,
                          int B, int N, int K_total, int dilation,
                          nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  max-relative aggregation (the gather / sub / max / interleave part of MRConv.forward,
 *   reference network_architecture/NexToU_Encoder_Decoder.py:401-409 and the two
 *   batched_index_select calls, torch_nn.py:94-115).
 *     out[b, 2c,   n] = x[b, c, n]
 *     out[b, 2c+1, n] = max_j ( src[b, c, nn_idx[b,n,j*idx_step]] - x[b, c, ctr] ),  j < K
 *   src = y if y != NULL (M points) else x;  ctr = center_idx[b,n,j*idx_step] if
 *   center_idx != NULL else n.  idx rows have idx_stride int32 entries.
 * Backward (autograd of the above): gout (B,2C,N) -> dx (B,C,N), dy (B,C,M) (dy NULL iff y
 *   NULL).  nextou_mr_aggregate_bwd recomputes the arg-max (ties go to the first j, neighbour
 *   order) and works for every shape; dx and dy are fully overwritten.
 * ---------------------------------------------------------------------------------------- */
int nextou_mr_aggregate_fwd(const float* x, const float* y,
                            const int32_t* nn_idx, const int32_t* center_idx,
                            float* out, uint16_t* arg_out,
                            int B, int C, int N, int M, int K, int idx_stride, int idx_step,
                            nextou_stream_t stream);

/* Training fast path: when arg_out (B,C,N) uint16 is given (identity centres, K <= 32, M <= 65536 and
 * source rows that fit the LDS staging — query with nextou_mr_aggregate_has_arg), the forward also
 * records which source id won each max (first max over the rounded differences — the element
 * autograd's max backward routes to).  nextou_mr_aggregate_bwd_arg is then a pure scatter-add:
 *   dx[b,c,n] = g[b,2c,n] - g[b,2c+1,n] (+ scattered terms when dy == NULL, i.e. the self graph)
 *   d{y|x}[b,c,arg[b,c,n]] += g[b,2c+1,n]
 * with no gathers, no ids and no global atomics; x / y / nn_idx need not be kept for backward. */
int nextou_mr_aggregate_has_arg(int B, int C, int N, int M, int K);
int nextou_mr_aggregate_bwd_arg(const float* gout, const uint16_t* arg, float* dx, float* dy,
                                int B, int C, int N, int M, nextou_stream_t stream);

/* Self graphs of N = M <= 512 points (the Swin windows), given the neighbour ids as well: the same backward as a GATHER over
 * reverse neighbour lists built per window in LDS — no floating-point atomics, fixed summation order (bit-reproducible
 * gradients).  Measured slower than the scatter on MI355X (200 vs 133 us at cfg-2 stage 2), so
 * nextou_mr_aggregate_bwd_wants_idx() — the policy the autograd glue asks — says 1 only under NEXTOU_MR_BWD=rev. */
int nextou_mr_aggregate_bwd_wants_idx(int B, int C, int N, int K);
int nextou_mr_aggregate_bwd_arg_idx(const float* gout, const uint16_t* arg, const int32_t* nn_idx, float* dx,
                                    int B, int C, int N, int K, int idx_stride, int idx_step, nextou_stream_t stream);

int nextou_mr_aggregate_bwd(const float* gout, const float* x, const float* y,
                            const int32_t* nn_idx, const int32_t* center_idx,
                            float* dx, float* dy,
                            int B, int C, int N, int M, int K, int idx_stride, int idx_step,
                            nextou_stream_t stream);

/* K2 + K7 fused: the max-relative aggregation of Swin windows feeding MRConv's grouped 1x1 convolution in one
 * launch — reference NexToU_Encoder_Decoder.py:401-418 (MRConv.forward: aggregate, then BasicConv), torch_nn.py:66-92 (BasicConv's
 * grouped conv), :766-818 (window partition / reverse around them).  Replaces nextou_mr_aggregate_fwd -> nextou_window_scatter ->
 * nextou_pw_rows_fused(groups) for a self graph inside windows:
 *   windows   (B * nWin, C, Nw) channel-major rows of the shifted windows (nextou_window_gather's output), Nw = wd * wh * ww
 *   nn_idx    (B * nWin, Nw, idx_stride) neighbour ids inside the window; neighbours j * idx_step, j < K <= 32
 *   weight    (2C, 2C / groups) the grouped convolution's weights, no bias (folded into the norm behind it)
 *   a_rows    NULL or channels-last (B, D, H, W, 2C): the aggregate [x_0, mr_0, x_1, mr_1, ...] at the rows the window map assigns —
 *             bit-identical to nextou_window_scatter(nextou_mr_aggregate_fwd(...)); the weight gradient's operand, not needed in eval
 *   arg_out   NULL or (B * nWin, C, Nw) uint16: the arg-max tape of nextou_mr_aggregate_fwd (bit-identical)
 *   h_rows    channels-last (B, D, H, W, 2C): conv1x1(a, weight, groups) — bit-identical to nextou_pw_rows_fused on a_rows
 *   stats_partial  NULL or [2C][stats_tiles] (sum, sum of squares) float64 pairs of h per window, stats_tiles == B * nWin
 *                  (nextou_norm_finalize's `partial`)
 * C % groups == 0, C / groups even and <= 32 (or 44 / 54: the 264- / 324-channel stages), Nw <= 256: nextou_mr_grouped_rows_supported() says whether a shape is taken
 * (NEXTOU_ENOTSUP otherwise; NEXTOU_MR_GROUPED=0 switches the kernel off).  HBM-bound: 12 B'C Nw + 4 B' Nw K bytes in eval. */
int nextou_mr_grouped_rows_supported(int n_windows, int C, int groups, int Nw, int K);
/* nextou_mr_grouped_rows_bwd: the window tensor's gradient in one launch — autograd of the three ops above (grouped data-gradient
 * GEMM, window gather of the gradient rows, arg-tape scatter of nextou_mr_aggregate_bwd_arg):
 *   dh_rows  channels-last (B, D, H, W, 2C): gradient of h_rows;  weight, arg: as in the forward
 *   dx       (B * nWin, C, Nw) float32, overwritten: dx[c][n] = ga[n][2c] - ga[n][2c+1] + sum_{n': arg[c][n'] == n} ga[n'][2c+1],
 *            ga = dh W_g per group; sums in 64-bit fixed point (bit-reproducible), NaN-poisoned per (window, group) tile when a
 *            gradient in it is not finite.  (The weight gradient is nextou_pw_wgrad on (dh_rows, a_rows).) */
int nextou_mr_grouped_rows(const float* windows, const int32_t* nn_idx, int idx_stride, int idx_step, int K, const float* weight,
                           float* a_rows, uint16_t* arg_out, float* h_rows, double* stats_partial, int stats_tiles,
                           int B, int C, int D, int H, int W, int wd, int wh, int ww, int sd, int sh, int sw, int groups,
                           nextou_stream_t stream);
int nextou_mr_grouped_rows_bwd(const float* dh_rows, const float* weight, const uint16_t* arg, float* dx, int B, int C,
                               int D, int H, int W, int wd, int wh, int ww, int sd, int sh, int sw, int groups, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * batched_index_select (reference torch_nn.py:94-115):
 *     out[b, c, n, j] = src[b, c, idx[b, n, j]]        src (B,C,M), idx (B,N,K) -> (B,C,N,K)
 * backward: dsrc[b, c, m] = sum_{(n,j): idx[b,n,j]==m} gout[b,c,n,j]   (dsrc overwritten)
 * ---------------------------------------------------------------------------------------- */
int nextou_gather_fwd(const float* src, const int32_t* idx, float* out,
                      int B, int C, int M, int N, int K, nextou_stream_t stream);
int nextou_gather_bwd(const float* gout, const int32_t* idx, float* dsrc,
                      int B, int C, int M, int N, int K, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5  BTI critical-voxel map (reference loss/bti_loss.py:76-117 and :132-134).
 *   nextou_argmax_labels: labels[b,v] = argmax_l softmax(logits[b,:,v]) as the reference computes it (:132-134), with torch.argmax's
 *     first-index rule on EQUAL float32 softmax values.  For all but ~1e-6 of the voxels that is the first arg-max of the logits; when
 *     an earlier class lies within 2^-21 of the maximum the canonical softmax is evaluated — e_l = float32(exp(x_l - max)) (exp to
 *     1e-16 by a fixed fma sequence, identical in oracle/nextou_oracle.c), s = e_0 + e_1 + ... in class order (float32), q_l = e_l / s
 *     (IEEE division) — and the first l with q_l == q_max wins.  Gaps <= 2^-25 (every float32 exp gives exactly 1: device-independent
 *     in the reference too) are reproduced exactly; in the band up to ~2.4e-7 ATen's outcome depends on the last bit of its vectorised
 *     exp and this restatement matches ATen-CPU on 99.8 % of the band's voxels (tests: g7d_near_ties).
 *   nextou_bti_critical_map: labels (B,D,H,W) uint8 (D = 1 for 2-D) ->
 *     critical (B,D,H,W) uint8 in {0,1}.  lut_a[l] / lut_c[l] hold one bit per interaction:
 *     bit i of lut_a[l] = (l in A_i); bit i of lut_c[l] = (l in C_i) for an exclusion pair,
 *     (l not in A_i and not in C_i) for an inclusion pair (:90-98).  A voxel is critical iff
 *     (OR_nbhd(c) & a) | (OR_nbhd(a) & c) != 0 with zero padding (:101-115); the
 *     neighbourhood is the (2*min_thick+1)^dim box for connectivity 26 / 8 and the 6 / 4
 *     cross otherwise (:52-73).  Up to 32 interactions per call.
 * ---------------------------------------------------------------------------------------- */
int nextou_argmax_labels(const float* logits, uint8_t* labels,
                         int B, int L, int64_t V, int64_t stride_l, int64_t stride_v, nextou_stream_t stream);

/* nextou_labels_u8: a target label map (n values; src_dtype 0 = float32 as nnU-Net hands it, 1 = int64, 2 = uint8) -> the uint8 volume
 *   the K5 kernels read, truncating as Tensor.long() does; flag[0] |= 1 (device uint32, never cleared here) when a value lies outside
 *   [0, n_classes) — the case the reference's CrossEntropyLoss raises for (bti_loss.py:141) — so the check needs no host read. */
int nextou_labels_u8(const void* src, int src_dtype, uint8_t* out, int64_t n, int n_classes, unsigned* flag, nextou_stream_t stream);

int nextou_bti_critical_map(const uint8_t* labels,
                            const uint32_t* lut_a, const uint32_t* lut_c, int n_labels,
                            uint8_t* critical,
                            int B, int D, int H, int W,
                            int connectivity, int min_thick,
                            nextou_stream_t stream);

/* K5b  critical-voxel cross-entropy, float64 arithmetic (reference loss/bti_loss.py:141-143:
 *   CrossEntropyLoss(reduction='none')(x.double(), y) * critical, summed over voxels).
 *   fwd: partial[b, i], i < nextou_bti_ce_partials(), are block-wise sums of
 *        critical[b,v] * (logsumexp_l x[b,l,v] - x[b,target[b,v],v]); the caller adds them (fixed order).
 *   bwd: grad_logits[b,l,v] = scale_dev[b] * critical[b,v] * (softmax_l(x[b,:,v]) - [l == target[b,v]]),
 *        fp32, every voxel written; scale_dev (B) are DEVICE doubles (the upstream gradient of every
 *        sample's sum) so that no host synchronisation is needed.  target / critical: uint8 (B,V); targets >= L
 *        contribute nothing.
 *   Layout (ABI v9, also nextou_argmax_labels): logits element (b, l, v) at b * L * V + l * stride_l + v * stride_v with
 *        (stride_l, stride_v) = (V, 1) — NCDHW planes — or (1, L) — channels-last rows, what the network's heads emit: no (B, L, V)
 *        copy of the logits is made on the way in, and grad_logits comes back in the same layout. */
int nextou_bti_ce_partials(void);
int nextou_bti_ce_fwd(const float* logits, const uint8_t* target, const uint8_t* critical,
                      double* partial, int B, int L, int64_t V, int64_t stride_l, int64_t stride_v, nextou_stream_t stream);
int nextou_bti_ce_bwd(const float* logits, const uint8_t* target, const uint8_t* critical,
                      const double* scale_dev, float* grad_logits, int B, int L, int64_t V,
                      int64_t stride_l, int64_t stride_v, nextou_stream_t stream);

/* K5d  soft-Dice statistics of the segmentation logits (ABI v9) — the three volume reductions nnU-Net's (MemoryEfficient)SoftDiceLoss needs
 * (reference loss/compound_bti_loss.py:29-30, :53-55 -> nnunetv2/training/loss/dice.py, softmax_helper_dim1 as apply_nonlin), which ATen runs as
 * softmax, a one-hot scatter, three products and three reductions over every logit.  p = softmax over the L classes (fp32), w = mask or 1:
 *   fwd: partial[((b * T + t) * L + l) * 3 + {0, 1, 2}], T = nextou_dice_stats_partials(): block-wise doubles of
 *        (sum_v w p[l] [target == l], sum_v w p[l], sum_v w [target == l]); the caller adds the T blocks (fixed order);
 *   bwd: g_intersect / g_sum_pred: DEVICE doubles (B, L) = d loss / d intersect, d loss / d sum_pred;
 *        grad_logits[k] = p_k (G_k - sum_l G_l p_l), G_l = w (g_sum_pred[b, l] + g_intersect[b, l] [target == l]), in the logits' layout.
 *   logits layout as nextou_ce_mean_*: (stride_l, stride_v) = (1, L) channels-last rows (L even) or (V, 1) planes; 2 <= L <= 32.
 *   target: uint8 (B, V) class indices (>= L: counted nowhere); mask: uint8 (B, V) or NULL. */
int nextou_dice_stats_partials(void);
int nextou_dice_stats_fwd(const float* logits, const uint8_t* target, const uint8_t* mask, double* partial, int B, int L, int64_t V,
                          int64_t stride_l, int64_t stride_v, nextou_stream_t stream);
int nextou_dice_stats_bwd(const float* logits, const uint8_t* target, const uint8_t* mask, const double* g_intersect,
                          const double* g_sum_pred, float* grad_logits, int B, int L, int64_t V, int64_t stride_l, int64_t stride_v,
                          nextou_stream_t stream);

/* K5c  mean cross-entropy of the segmentation logits, fp32 (ABI v7) — the deep-supervision CE inside every NexToU trainer's loss
 * (nnUNetTrainer_NexToU.py / *_BTI_*.py -> nnU-Net's RobustCrossEntropyLoss = torch.nn.CrossEntropyLoss(reduction='mean')), which ATen runs as
 * log_softmax -> nll_loss over NCDHW tensors (with channels-last logits: a layout copy in, two passes each way, a copy of the gradient back).
 *   logits: element (b, l, v) at b * L * V + l * stride_l + v * stride_v; (stride_l, stride_v) = (1, L) — channels-last rows, what the network
 *           emits — or (V, 1) — NCDHW planes.  L <= 32.  target: int64 (B * V); voxels whose target is ignore_index do not count; a target outside [0, L) that is
 *           not ignore_index makes the loss NaN (torch raises a device assert there; no gradient flows to that voxel).
 *   fwd: partial[2 i + {0, 1}], i < nextou_ce_mean_partials(): (sum of -log softmax(x)[target], number of counted voxels) of block i as doubles;
 *        the caller adds them (fixed order) and divides.
 *   bwd: grad_logits (same layout as logits) = scale * (softmax(x) - onehot(target)) for counted voxels, 0 for the others; scale_dev: ONE device
 *        float = upstream gradient / count (no host synchronisation). */
int nextou_ce_mean_partials(void);
int nextou_ce_mean_fwd(const float* logits, const int64_t* target, double* partial, int B, int L, int64_t V,
                       int64_t stride_l, int64_t stride_v, int64_t ignore_index, nextou_stream_t stream);
int nextou_ce_mean_bwd(const float* logits, const int64_t* target, const float* scale_dev, float* grad_logits, int B, int L, int64_t V,
                       int64_t stride_l, int64_t stride_v, int64_t ignore_index, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K6  normalisation fused with LeakyReLU (batch norm and, with B = 1 and C = B*C, instance norm) —
 * the (norm -> nonlin) tail of every conv block on the path: reference torch_nn.py:84-90 (BasicConv),
 * NexToU_Encoder_Decoder.py:384-390 (FFN), :710-720 / :833-842 (fc1 / fc2) and the
 * ConvDropoutNormReLU blocks of the conv stages (:125-136, :281-298).  Replaces
 * batch_norm / instance_norm -> leaky_relu and their autograd.
 *   x, y, gy, gx : (B, C, S) contiguous (channels_last = 0), or (B, S, C) contiguous — PyTorch's channels_last /
 *                  channels_last_3d memory format of the same logical tensor (channels_last = 1; any C,
 *                  param_period = 0); dtype NEXTOU_DTYPE_F32, NEXTOU_DTYPE_BF16 or NEXTOU_DTYPE_F16
 *   weight, bias : float (param_period ? param_period : C) or NULL (= 1 / 0); channel c uses entry
 *                  c % param_period when param_period > 0 (instance norm: period = real channel count)
 *   training != 0: batch statistics (biased variance for the normalisation); when running_mean /
 *                  running_var are given they are updated in place,
 *                  r = (1 - momentum) * r + momentum * stat (unbiased variance, as torch does).
 *   training == 0: normalises with running_mean / running_var (required).
 *   y = leaky_relu(x_hat * weight + bias, slope); slope = 1 is the plain normalisation.
 *   pre_bias     : NULL, or the bias (indexed like weight) of the convolution that produced x when the
 *                  caller ran that convolution WITHOUT its bias: norm(x + b) == norm(x) under batch
 *                  statistics, so b only enters the running mean (training) and the shift (inference;
 *                  save_mean then holds running_mean - b).  Its gradient is identically 0 in training
 *                  and weight * invstd * gbias in inference — computed by the caller.
 *   save_mean / save_invstd (C floats, may be NULL for inference) feed the backward.
 *   fwd traffic: 2 reads + 1 write of the tensor (training), 1 + 1 (inference).
 *   bwd: gx, gweight[c] = sum dz * x_hat, gbias[c] = sum dz, with dz = gy * (z > 0 ? 1 : slope);
 *        4 reads + 1 write.  gweight / gbias have C entries (the caller folds instance-norm
 *        entries over the batch); either may be NULL.
 *   workspace: nextou_norm_act_workspace_bytes() bytes of device scratch, contents irrelevant.
 *   Sums are float64 in a fixed order: bit-reproducible.
 * ---------------------------------------------------------------------------------------- */
#define NEXTOU_DTYPE_F32  0
#define NEXTOU_DTYPE_BF16 1
#define NEXTOU_DTYPE_F16  2   /* IEEE half: torch.autocast("cuda")'s default dtype, what nnU-Net v2 trains under */

size_t nextou_norm_act_workspace_bytes(int B, int C, int64_t S, int dtype);

int nextou_norm_act_fwd(const void* x, const float* weight, const float* bias, const float* pre_bias,
                        float* running_mean, float* running_var,
                        void* y, float* save_mean, float* save_invstd,
                        void* workspace, size_t workspace_bytes,
                        int B, int C, int64_t S, int param_period, int dtype, int channels_last,
                        int training, float momentum, float eps, float slope, nextou_stream_t stream);

int nextou_norm_act_bwd(const void* x, const void* gy, const float* weight, const float* bias,
                        const float* save_mean, const float* save_invstd,
                        void* gx, float* gweight, float* gbias,
                        void* workspace, size_t workspace_bytes,
                        int B, int C, int64_t S, int param_period, int dtype, int channels_last,
                        int training, float slope, nextou_stream_t stream);

/* nextou_norm_act_bwd with TWO incoming gradients summed on load (ABI v14): the output of a plain encoder stage feeds the next stage AND,
 * as the skip connection, the decoder's concatenation (reference NexToU_Encoder_Decoder.py:143-150, :311-337); autograd would add the
 * two gradients in a pass of its own.  Channels-last fp32, C <= 128 and a multiple of 4; gy2: rows of C floats at row stride ld2 >= C
 * (a channel range of the concatenation's gradient where it lies).  Everything else as nextou_norm_act_bwd(channels_last = 1). */
int nextou_norm_act_bwd_two(const float* x, const float* gy, const float* gy2, int64_t ld2, const float* weight, const float* bias,
                            const float* save_mean, const float* save_invstd, float* gx, float* gweight, float* gbias,
                            void* workspace, size_t workspace_bytes, int B, int C, int64_t S, int training, float slope,
                            nextou_stream_t stream);

/* Per-channel sum over batch and space: out[c] = sum_{b,s} x[b,c,s] (float64 accumulation, fixed order) — the bias
 * gradient of a convolution that is not followed by a norm (segmentation heads, transposed convolutions), which
 * PyTorch-ROCm computes with a generic reduction that collapses on channels-last tensors (5.5 ms for 727 MB).
 * Same layouts / dtypes / workspace as nextou_norm_act_fwd. */
int nextou_channel_sum(const void* x, float* out, void* workspace, size_t workspace_bytes,
                       int B, int C, int64_t S, int dtype, int channels_last, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3 / K4  data movement around the graph kernels, fused with the layout change between the dense stages'
 * channels-last activations x_cl (B, D, H, W, C) — PyTorch's channels_last_3d memory of a (B, C, D, H, W) tensor; 2-D
 * models pass D = 1 — and the graph kernels' channel-major rows (B', C, N).  float32.  HBM-bound: one read + one
 * write of the tensor.
 *
 * nextou_window_gather    replaces torch.roll(x, -shift) -> window_partition (reference
 *     NexToU_Encoder_Decoder.py:781-790, :634-660):  out_cm[b * nWin + win, c, p] = x_cl[b, pos(win, p) + shift mod size, c],
 *     windows enumerated row-major over (D/wd, H/wh, W/ww), points row-major inside a window; 0 <= shift < size.
 * nextou_window_scatter   the inverse, replacing window_reverse -> torch.roll(x, +shift) [-> + shortcut]
 *     (:807-817, :662-693): out_cl[b, pos, c] = src_cm[...] (+ residual_cl[b, pos, c] when residual_cl != NULL).
 *     Each is the other's backward.
 * nextou_pool_rows        replaces MaxPool{2,3}d(pool, stride = pool, return_indices = True) (:524-530):
 *     values_cm[b, c, n] = max over the (pd, ph, pw) cell of pooled point n (row-major over (D/pd, H/ph, W/pw)),
 *     cell[b, n, c] (uint8, points-major) = the winning position inside the cell, (kd * ph + kh) * pw + kw; first
 *     maximum in scan order wins, NaN wins (ATen's rule).
 * nextou_cell_scatter     replaces MaxUnpool{2,3}d(out, cat(indices, indices)) (:536-549): out_cl (B, D, H, W, C2), C2 = C
 *     or 2C, is written completely — src_cm[b, c2, n] at the cell position cell[b, n, c2 mod C], zeros elsewhere (no
 *     memset, no index concatenation).  Also the backward of nextou_pool_rows (C2 = C).
 * nextou_cell_gather      out_cm[b, c2, n] = x_cl[b, cellpos(n, cell[b, n, c2 mod C]), c2]: the backward of
 *     nextou_cell_scatter, and the replay of recorded arg-max cells.
 * ---------------------------------------------------------------------------------------- */
int nextou_window_gather(const float* x_cl, float* out_cm, int B, int C, int D, int H, int W,
                         int wd, int wh, int ww, int sd, int sh, int sw, nextou_stream_t stream);
int nextou_window_scatter(const float* src_cm, const float* residual_cl, float* out_cl, int B, int C, int D, int H, int W,
                          int wd, int wh, int ww, int sd, int sh, int sw, nextou_stream_t stream);
int nextou_pool_rows(const float* x_cl, float* values_cm, uint8_t* cell, int B, int C, int D, int H, int W,
                     int pd, int ph, int pw, nextou_stream_t stream);
int nextou_cell_gather(const float* x_cl, const uint8_t* cell, float* out_cm, int B, int C2, int C, int D, int H, int W,
                       int pd, int ph, int pw, nextou_stream_t stream);
int nextou_cell_scatter(const float* src_cm, const uint8_t* cell, float* out_cl, int B, int C2, int C, int D, int H, int W,
                        int pd, int ph, int pw, nextou_stream_t stream);
/* nextou_depth_unroll     out_cl (B, D, H, W, 3C): out[b, d, h, w, kd*C + c] = x_cl[b, d + kd - 1, h, w, c], zeros outside the volume.
 *     The three depth taps of a [3,k,k] stride-1 convolution (the plain stages' ConvDropoutNormReLU, reference
 *     NexToU_Encoder_Decoder.py:125-136, :281-298) as input channels: its weight gradient becomes the 2-D problem
 *     (B*D, 3C, H, W) x (B*D, Cout, H, W) that MIOpen's 2-D kernels run 15-20 % faster than the 3-D one.  C % 4 == 0. */
int nextou_depth_unroll(const float* x_cl, float* out_cl, int B, int C, int D, int H, int W, nextou_stream_t stream);

/* nextou_cat_bias_rows (ABI v8)  out[p, :] = [a[p, :] + bias, b[p, :]] over channels-last rows (P, C1) and (P, C2), float32, C1 and C2 multiples
 *     of 4, C1 + C2 <= 1024, bias (C1 floats) may be NULL: the decoder's torch.cat((up-convolution output, skip), 1) (reference
 *     NexToU_Encoder_Decoder.py:311-337) with the up-convolution's bias folded in — the convolution runs bias-free and ATen's separate bias-add pass
 *     over its output never runs.  One read of a and b, one write. */
int nextou_cat_bias_rows(const float* a, const float* bias, const float* b, float* out, int64_t P, int C1, int C2, nextou_stream_t stream);

/* nextou_filter_flip_t (ABI v12)  out = the filter of the FORWARD convolution that computes the data gradient of a stride-1 convolution with
 *     filter w (Co, Ci, Kd, Kh, Kw): out (Ci, Co, Kd, Kh, Kw) stored channels-last — out[ci][kd][kh][kw][co] = w[co][ci][Kd-1-kd][Kh-1-kh][Kw-1-kw].
 *     w is read through its element strides (s_co, s_ci, s_kd, s_kh, s_kw): contiguous and channels-last filters alike; 2-D filters pass
 *     Kd = 1.  Replaces ATen's transpose -> flip -> contiguous(channels_last) (two copy kernels per convolution and step) in the backward
 *     of the plain stages' convolutions (reference NexToU_Encoder_Decoder.py:125-136, :281-298 ConvDropoutNormReLU; PyTorch-ROCm keeps
 *     the convolution itself). */
int nextou_filter_flip_t(const float* w, float* out, int Co, int Ci, int Kd, int Kh, int Kw, int64_t s_co, int64_t s_ci, int64_t s_kd,
                         int64_t s_kh, int64_t s_kw, nextou_stream_t stream);

/* nextou_narrow_copy_sum (ABI v13)  dst (P, C) dense = src[:, c_off : c_off + C] of channels-last rows with row stride ld floats, and
 *     sum_out[c] = sum_p dst[p, c] (float64 partial sums in nextou_channel_sum's order: bit-identical to nextou_channel_sum of the copy) —
 *     one read of the channel range instead of two.  The backward of the decoder's torch.cat((up-convolution output, skip), 1) (reference
 *     NexToU_Encoder_Decoder.py:311-337): the transposed convolution's backward needs the first C channels of the gradient as a dense
 *     tensor, its folded bias their sums.  float32; C, ld and c_off multiples of 4, C <= 128, 16-byte aligned pointers — NEXTOU_ENOTSUP
 *     otherwise (the caller keeps ATen's narrow().contiguous() + nextou_channel_sum).  ws: nextou_norm_act_workspace_bytes(1, C, P, f32). */
int nextou_narrow_copy_sum(const float* src, float* dst, float* sum_out, void* ws, size_t ws_bytes, int64_t P, int C, int64_t ld,
                           int c_off, nextou_stream_t stream);

/* nextou_upconv_cat_rows / nextou_upconv_cat_rows_bwd (ABI v13)  the decoder's torch.cat((transpconv(x) + bias, skip), 1) (reference
 *     NexToU_Encoder_Decoder.py:311-337; transpconvs are built with kernel == stride, :272-276) with the transposed convolution as a K7 GEMM: every
 *     input point p_in of the (B, D, H, W) volume produces the T = sd*sh*sw taps of its output block independently, so
 *     y2 (P_in, T*C1) = x (P_in, Cin) . w2^T with w2[(t*C1 + co), ci] = weight[ci, co, t] is one nextou_pw_rows call, and
 *     nextou_upconv_cat_rows      out[p, :] = [y2[p_in(p), t(p)*C1 : (t(p)+1)*C1] + bias, skip[p, :]] over the rows p of the
 *         (B, D*sd, H*sh, W*sw) output volume, channels-last, t = ((d2 % sd) sh + h2 % sh) sw + w2 % sw — the "pixel shuffle" rides
 *         on the concatenation pass (same traffic as nextou_cat_bias_rows);
 *     nextou_upconv_cat_rows_bwd  gy2 (P_in, T*C1) = the first C1 channels of the gradient rows g (P, C1 + C2), un-shuffled the same
 *         way, and gbias[c] = sum_p g[p, c] (nextou_narrow_copy_sum's pass and summation order); the data gradient is then
 *         nextou_pw_rows(gy2, weight viewed (Cin, T*C1)), the filter gradient nextou_pw_wgrad(gy2, x).
 *     float32, 2-D volumes pass D = sd = 1; C1, C2 multiples of 4, C1 + C2 <= 1024 (bwd: C1 <= 128, else NEXTOU_ENOTSUP); strides 1..4;
 *     ws of the backward: nextou_norm_act_workspace_bytes(1, C1, P, f32). */
int nextou_upconv_cat_rows(const float* y2, const float* bias, const float* skip, float* out, int B, int D, int H, int W, int sd, int sh,
                           int sw, int C1, int C2, nextou_stream_t stream);
int nextou_upconv_cat_rows_bwd(const float* g, float* gy2, float* gbias, void* ws, size_t ws_bytes, int B, int D, int H, int W, int sd,
                               int sh, int sw, int C1, int C2, nextou_stream_t stream);

/* nextou_pw_rows_up (ABI v14): the up-convolution's GEMM with the pixel shuffle in its store — the product row of input point p, column
 * t * cout + co, goes to channel co of the output point that tap t of p lands on, in rows of pitch ld_out (the concatenation buffer
 * (B, D sd, H sh, W sw, ld_out): its first cout channels), bias[co] added.  Strides 1, 2 or 4; cout, K, ldx, ld_out multiples of 4.
 * nextou_upconv_cat_rows with y2 == NULL then fills only the skip half: together they are nextou_pw_rows + nextou_upconv_cat_rows
 * without the (P_in, T cout) intermediate — one 881-MB write and read less at the full-resolution stage of cfg 2. */
int nextou_pw_rows_up(const float* x, const float* w, const float* bias, float* out, int N, int K, int64_t ldx, int64_t ld_out, int B, int D,
                      int H, int W, int sd, int sh, int sw, int cout, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Step glue (ABI v13): gradient clip + SGD update of a whole parameter list.  The step nnU-Net's trainer prescribes for the NexToU
 * plug-ins (nnUNetTrainer_NexToU inherits nnUNetTrainer.train_step: backward -> clip_grad_norm_(network.parameters(), 12) ->
 * SGD(momentum 0.99, nesterov, weight_decay 3e-5).step(); reference nnUNetTrainer_NexToU.py:17-91 overrides only build_network_architecture) runs both as
 * multi-tensor-apply launches of at most 36-110 tensors each — a few dozen launches (575 us replayed) for the 358 trainable tensors of cfg 2.  Here the
 * tensors are named by a table in DEVICE memory, so each stage is one launch whatever the tensor count:
 *   table   int64 [n_tensors][4]: parameter pointer, gradient pointer, momentum-buffer pointer (0: none), element count; the three
 *           tensors of a row are walked as flat float32 arrays and must share one dense layout
 *   chunks  int32 [n_chunks][2]: (row, chunk index within the row's tensor), chunk_elems elements per chunk (multiple of 4, >= 1024);
 *           every element of every tensor belongs to exactly one chunk
 * nextou_grad_norm_clip_coef   norm_coef[0] = || all gradients ||_2 (float64 partial sums per chunk into `partial` (n_chunks doubles),
 *           summed in a fixed order, rounded to float32), norm_coef[1] = min(1, max_norm / (norm + 1e-6)): clip_grad_norm_'s factor.
 * nextou_clip_sgd_update       per element: g <- g * norm_coef[1] and stored (norm_coef NULL: no clip, g is only read); d = g + wd * p;
 *           buf = momentum * buf + d; d = nesterov ? d + momentum * buf : buf; p = p - lr * d.  lr_dev (may be NULL) overrides lr with a
 *           float in device memory (schedulers under hipGraph replay).  A zero-filled buffer on the first step reproduces torch's
 *           "buf = clone(grad)".  The products with wd, momentum (Nesterov) and lr are fused multiply-adds, as in ATen's alpha-adds.
 * ---------------------------------------------------------------------------------------- */
/* nextou_device_write_i64   dst[0 .. n) <- host_values, carried as kernel arguments (448 values per launch): the way a table reaches
 *           device memory inside a hipGraph capture — the values live in the graph's nodes, no host buffer is read at replay. */
int nextou_device_write_i64(int64_t* dst, const int64_t* host_values, int64_t n, nextou_stream_t stream);
int nextou_grad_norm_clip_coef(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                               int64_t total_elems, double* partial, float max_norm, float* norm_coef, nextou_stream_t stream);
int nextou_clip_sgd_update(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems, int64_t total_elems,
                           const float* norm_coef, float lr, const float* lr_dev, float momentum, float weight_decay, int nesterov,
                           nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  point-wise (kernel 1, stride 1) convolutions on channels-last rows — the 1x1 convolutions of the Grapher / FFN
 * blocks (reference NexToU_Encoder_Decoder.py:368-390 FFN, :710-720 / :833-842 fc1 / fc2; torch_nn.py:66-92 the MRConv's
 * grouped BasicConv), which PyTorch-ROCm hands to MIOpen as convolutions.  x is the (P, groups*K) matrix a channels-last
 * activation already is in memory (P = B*D*H*W points, row stride ldx floats); weights are the reference's
 * (groups*N, K, 1[,1[,1]]) tensors.  float32 on v_mfma_f32_16x16x4_f32 (exact f32).  MFMA-bound for K, N >= 132,
 * HBM-bound below.
 *
 * nextou_pw_rows     y[p, g*N + n] = sum_k x[p, g*K + k] * w[g*N + n, k] (+ bias[g*N + n]; bias may be NULL).
 *                    The data gradient of the same convolution is this call on gy with the per-group transposed weight.
 *                    K and ldx must be multiples of 4, x and w 16-byte aligned.
 * nextou_pw_wgrad    dw[g*N + n, k] (+)= sum_p gy[p, g*N + n] * x[p, g*K + k]: split over points into `workspace`
 *                    (nextou_pw_wgrad_workspace bytes), summed in a fixed order — bit-reproducible.  accumulate != 0 adds
 *                    to dw.  N, K, ldg, ldx multiples of 4, gy and x 16-byte aligned.
 * ---------------------------------------------------------------------------------------- */
int nextou_pw_rows(const float* x, const float* w, const float* bias, float* y, int64_t P, int N, int K, int groups,
                   int64_t ldx, int64_t ldy, nextou_stream_t stream);
int nextou_pw_wgrad_workspace(int64_t P, int N, int K, int groups, size_t* bytes);
int nextou_pw_wgrad(const float* gy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int64_t P, int N,
                    int K, int groups, int64_t ldg, int64_t ldx, int accumulate, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7 + K6 fused (round 3): the point-wise pipeline of a Grapher / FFN block —
 *   conv 1x1 -> norm -> LeakyReLU -> conv 1x1 -> norm -> (+ shortcut)
 * (reference NexToU_Encoder_Decoder.py:368-390 FFN; :710-720, :833-842 fc1 / fc2; torch_nn.py:66-92 BasicConv) —
 * without the passes over the activation that the op-by-op form needs: the producing GEMM's epilogue delivers the batch
 * statistics of its output, the consuming GEMM's operand load applies norm + activation, so the activated tensor (363 MB for
 * the stage-2 FFN of cfg 2) is never written, and the statistics / apply passes of K6 over it never run.
 *
 * nextou_pw_rows_fused   y[p, g*N + n] = sum_k A[p, g*K + k] * w[g*N + n, k]  (no bias; N, K, ldx, ldy multiples of 4) with
 *     A = x, or — operand prologue, pro_scale != NULL — A[p, c] = leaky_relu(fmaf(x[p, c], pro_scale[c], pro_shift[c]), pro_slope)
 *         (c = g*K + k over all groups*K input channels; exactly K6's apply arithmetic);
 *     stats_partial != NULL, bwd_h == NULL: statistics epilogue — stats_partial[(c * T + t) * 2 + {0, 1}] = (sum, sum of squares)
 *         of y[:, c] over point tile t, T = nextou_pw_rows_tiles(P, N, K, groups, ldx, ldy, prologue, grad_stats) of the SAME launch
 *         arguments (stats_tiles hands it back: a mismatch is NEXTOU_EINVAL, never an overrun), c over groups*N output channels: the `partial`
 *         input of nextou_norm_finalize;
 *     bwd_h != NULL: gradient-statistics epilogue of a data-gradient GEMM (y = d loss / d activated): with h = bwd_h[p, c]
 *         (row stride ldh), z = fmaf(h, scale, shift), dz = y * (z > 0 ? 1 : bwd_slope), xhat = (h - bwd_mean[c]) * bwd_invstd[c]
 *         (scale = bwd_weight * bwd_invstd, shift = fmaf(-bwd_mean, scale, bwd_bias)):
 *         stats_partial[...] = (sum dz, sum dz * xhat) — the `partial` input of nextou_norm_bwd_finalize.  y is stored unchanged.
 *     Partial sums: fp32 over a wave's 32 points (fixed tree), float64 across waves and tiles, fixed order (bit-reproducible).
 * nextou_pw_wgrad_fused  nextou_pw_wgrad with the same operand prologue on x (the weight gradient of the second convolution).
 * nextou_norm_finalize   (sum, sum of squares) partials -> save_mean / save_invstd (training: batch statistics, running
 *     statistics updated, pre_bias as in nextou_norm_act_fwd; inference: from the running statistics) and, when scale / shift
 *     are given, the affine scale = weight * invstd, shift = fmaf(-mean, scale, bias) that an operand prologue consumes.
 * nextou_norm_apply_rows y = leaky_relu(fmaf(x, scale, shift), slope) [+ residual] over channels-last fp32 rows (rows, C):
 *     K6's apply pass alone, with the block's shortcut add folded in.
 * nextou_norm_bwd_finalize / nextou_norm_bwd_apply_rows   the two halves of K6's backward after the reduce: coeff (2C floats) +
 *     parameter gradients from (sum dz, sum dz*xhat) partials; gx = scale * ((dz - coeff[2c]) - xhat * coeff[2c+1]).
 * ---------------------------------------------------------------------------------------- */
int nextou_pw_rows_tiles(int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy, int prologue, int grad_stats);
int nextou_pw_rows_fused(const float* x, const float* w, float* y, int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy,
                         const float* pro_scale, const float* pro_shift, float pro_slope, double* stats_partial, int stats_tiles,
                         const float* bwd_h, int64_t ldh, const float* bwd_weight, const float* bwd_bias, const float* bwd_mean,
                         const float* bwd_invstd, float bwd_slope, nextou_stream_t stream);
int nextou_pw_wgrad_fused(const float* gy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int64_t P, int N,
                          int K, int groups, int64_t ldg, int64_t ldx, int accumulate, const float* pro_scale,
                          const float* pro_shift, float pro_slope, nextou_stream_t stream);
int nextou_norm_finalize(const double* partial, int tiles, double count, const float* pre_bias, float* running_mean,
                         float* running_var, float* save_mean, float* save_invstd, const float* weight, const float* bias,
                         float* scale, float* shift, int C, int training, float momentum, float eps, nextou_stream_t stream);
int nextou_norm_apply_rows(const float* x, const float* residual, float* y, const float* weight, const float* bias,
                           const float* save_mean, const float* save_invstd, int64_t rows, int C, float slope,
                           nextou_stream_t stream);
int nextou_norm_bwd_finalize(const double* partial, int tiles, double count, float* coeff, float* gweight, float* gbias, int C,
                             int training, nextou_stream_t stream);
int nextou_norm_bwd_apply_rows(const float* x, const float* gy, float* gx, const float* coeff, const float* weight,
                               const float* bias, const float* save_mean, const float* save_invstd, int64_t rows, int C,
                               float slope, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K8  segmentation heads (ABI v11): the decoder's deep-supervision `seg_layers` — nn.Conv{2,3}d(features, num_classes, 1, 1, 0,
 * bias=True) applied to every decoder stage's output (reference NexToU_Encoder_Decoder.py:253-258, :311-337) — and their autograd,
 * which PyTorch-ROCm runs as MIOpen convolutions.  Channels-last rows: x (P, C) with row stride ldx, logits y / gy (P, L) with row
 * stride ldy / ldg, weights w (L, C) dense, float32.  HBM-bound: 4 P (C + L) bytes each way.  Any C and L <= 64 (forward: any L);
 * 16-byte accesses when C % 4 == 0, ldx % 4 == 0 and the pointer is 16-byte aligned, scalar ones otherwise.
 *
 * nextou_head_rows_fwd            y[p, l] = bias[l] + sum_c x[p, c] w[l, c]   (bias may be NULL); c ascending in groups of four
 *                                 channels {16 j + 4 g + r : g = 0..3} — a fixed order, bit-reproducible.
 * nextou_head_rows_bwd            gx[p, c] = sum_l gy[p, l] w[l, c]           (gx NULL: skipped)
 *                                 gw[l, c] = sum_p gy[p, l] x[p, c], gb[l] = sum_p gy[p, l]   (both NULL: skipped; either may be NULL)
 *                                 split over points into `workspace` (nextou_head_rows_bwd_workspace bytes) and summed in a fixed
 *                                 order: bit-reproducible.
 * ---------------------------------------------------------------------------------------- */
int nextou_head_rows_fwd(const float* x, const float* w, const float* bias, float* y, int64_t P, int L, int C, int64_t ldx,
                         int64_t ldy, nextou_stream_t stream);
int nextou_head_rows_bwd_workspace(int64_t P, int L, int C, size_t* bytes);
int nextou_head_rows_bwd(const float* gy, const float* x, const float* w, float* gx, float* gw, float* gb, float* workspace,
                         size_t workspace_bytes, int64_t P, int L, int C, int64_t ldg, int64_t ldx, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2 + K7 for the pooled graphs (ABI v11): MRConv of a Pool-GNN block in one launch — max-relative aggregation
 * of a pooled (y != NULL, M candidates) or self (y == NULL, M == N) graph -> grouped 1x1 convolution -> InstanceNorm statistics
 * (reference NexToU_Encoder_Decoder.py:401-418 inside PoolDyGraphConv :516-551; torch_nn.py:66-92).  Channel-major tensors:
 * x (B, C, N), y (B, C, M), nn_idx (B, N, >= K) int32 with the row stride / element step of nextou_mr_aggregate_fwd, weight
 * (groups * Ng, 2C / groups) dense, h (B, groups * Ng, N) = the convolution of the interleaved aggregate [x_c, max_j(y_j - x)_c].
 * a_out (B, 2C, N): the aggregate itself (the weight gradient's operand), arg_out (B, C, N) uint16: the arg-max tape of
 * nextou_mr_aggregate_bwd_arg — both optional, both bit-identical to nextou_mr_aggregate_fwd's.  stats_partial: per (sample,
 * output channel) and 128-query tile one (sum, sum of squares) in float64, laid out [(b * groups * Ng + channel)][tile] — what
 * nextou_norm_act_fwd_partials consumes with B = 1, C = B * groups * Ng, param_period = groups * Ng (instance statistics);
 * stats_tiles must equal nextou_mr_grouped_cm_tiles(...), which returns 0 for a shape the kernel does not take (K > 32,
 * 2C / groups > 108, Ng > 112, a source too long for LDS; NEXTOU_MR_GROUPED_CM=0 switches it off).
 *
 * nextou_norm_act_fwd_partials   K6's forward (normalise + LeakyReLU, fp32, channel-major (B, C, S)) from ready-made statistics
 *     partials: n_partial per channel at partial[c * n_partial + t].  param_period as in nextou_norm_act_fwd.
 * ---------------------------------------------------------------------------------------- */
int nextou_mr_grouped_cm_tiles(int B, int C, int groups, int Ng, int N, int M, int K);
int nextou_mr_grouped_cm(const float* x, const float* y, const int32_t* nn_idx, int idx_stride, int idx_step, int K,
                         const float* weight, float* a_out, uint16_t* arg_out, float* h, double* stats_partial, int stats_tiles,
                         int B, int C, int N, int M, int groups, int Ng, nextou_stream_t stream);
int nextou_norm_act_fwd_partials(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                                 float* y, float* save_mean, float* save_invstd, const double* partial, int n_partial, int B, int C,
                                 int64_t S, int param_period, float momentum, float eps, float slope, nextou_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K9  the stem block (ABI v14): the network's first ConvDropoutNormReLU — conv_op(1, features[0], [1,]3x3, stride 1, padding 1, bias)
 * -> BatchNorm -> LeakyReLU on the ONE-channel image (reference NexToU_Encoder_Decoder.py:125-141, encoder.stages[0], first block)
 * and the autograd of the three, without ever storing the convolution's output.  Replaces, per step at cfg 2: the library convolution
 * (forward + weight gradient), K6's statistics / apply / backward-reduce / backward-apply passes over its 881-MB output.
 *   x        (B, D, H, W) fp32 image (D = 1 for a 2-D network), dense
 *   weight   (C, 9) = the filter (C, 1, [1,] 3, 3) as stored; pre_bias (C,) or NULL = the convolution's bias (folded into the
 *            statistics, as in nextou_norm_act_fwd); gamma / beta (C,) or NULL = the norm's affine
 *   y        (B, D, H, W, Cpad) channels-last rows, Cpad >= C a multiple of 4, <= 48; channels C .. Cpad-1 are written as zeros (the
 *            padding channels of channel_pad.py)
 *   act_mask NULL or (ceil(B D H / 8), W, Cpad / 4) uint32 out: bit 4 i + j of word (row block, column, q) = pre-activation of channel
 *            4q + j at image row 8 block + i > 0 — the LeakyReLU mask the backward applies (recomputing it there cost 40 fma and 36
 *            registers per thread; this is half a byte per voxel and quad, moved as dwords)
 *   save_mean / save_invstd (C,) out; moments (54 doubles) out: the tap sums X1[9] and the packed upper triangle of the 9 x 9 tap
 *            autocorrelation — what the backward needs besides (S1, S2); running_mean / running_var updated as F.batch_norm does
 *   training = 0: running statistics, no moments pass (forward only: the backward below is the batch-statistics one)
 * nextou_stem_bwd: gweight (C, 9), ggamma (C,), gbeta (C,) — any may be NULL — from gy (B, D, H, W, Cpad) and act_mask in one pass.  The image gets no
 * gradient (the caller routes an image that requires one through the library convolution); d/d pre_bias is exactly zero.
 * workspace: nextou_stem_workspace_bytes() bytes, contents irrelevant.  Float64 sums in a fixed order: bit-reproducible.
 * HBM-bound: V (4 + 4 Cpad + Cpad / 8) bytes each way for V = B D H W voxels. */
size_t nextou_stem_workspace_bytes(int B, int D, int H, int W, int Cpad);
int nextou_stem_fwd(const float* x, const float* weight, const float* pre_bias, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* y, uint32_t* act_mask, float* save_mean, float* save_invstd,
                    double* moments, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, int Cpad, int training,
                    float momentum, float eps, float slope, nextou_stream_t stream);
int nextou_stem_bwd(const float* x, const float* gy, const uint32_t* act_mask, const float* weight, const float* gamma,
                    const float* save_mean, const float* save_invstd, const double* moments, float* gweight, float* ggamma,
                    float* gbeta, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, int Cpad,
                    float slope, nextou_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEXTOU_HIP_H */
