/* mmtpsm.h -- C ABI of libmmtpsm.so, the MI355X (gfx950) implementation of the MMT-PSM hot path.
 *
 * Drop-in boundary: these are the entry points that the reference's Python binds through its
 * torch extension `maskrcnn_benchmark._C` (csrc/vision.cpp:7-13) and through ATen (conv / linear /
 * losses behind maskrcnn_benchmark.layers).  Plain pointers and sizes only, no torch types.
 * Every pointer is a DEVICE pointer unless marked [host].  Every call is asynchronous on `stream`
 * (a hipStream_t passed as void*; the reference launches on the current stream,
 * cuda/ROIAlign_cuda.cu:273).  Return value: 0 on success, otherwise a hipError_t (launch
 * failure) or a negative MMT_E* code (bad arguments); the Python side turns non-zero into
 * RuntimeError like the reference's AT_ASSERTM / AT_ERROR (csrc/ROIAlign.h:19-45).
 *
 * Layout: activations are NHWC fp32 (channels innermost); conv weights are [Cout][KH][KW][Cin]
 * (torch channels_last memory of an (O,I,H,W) tensor); ROI lists are [K][5] = (batch, x1, y1, x2, y2)
 * exactly as the reference (cuda/ROIAlign_cuda.cu:64-122).
 */
#ifndef MMTPSM_H
#define MMTPSM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMT_EINVAL (-22)

/* library / device info: writes "gfx950 ..." into buf [host]; returns ABI version */
int mmt_version(void);

/* ---------------------------------------------------------------- ROIAlign (NHWC, FPN-fused)
 * replaces _C.roi_align_forward / _C.roi_align_backward (csrc/ROIAlign.h:11-45,
 * cuda/ROIAlign_cuda.cu:64-122,177-254) and the per-level gather/scatter of Pooler.forward
 * (modeling/poolers.py:91-121): one launch pools all K ROIs from up to 4 pyramid levels.
 *   feats[l]   NHWC [N, H[l], W[l], C]         scales[l] spatial scale of level l
 *   rois       [K,5]                            levels int32 [K] (level of each ROI, 0..L-1)
 *   out        [K, PH, PW, C]
 * sampling_ratio > 0: fixed grid; <= 0: ceil(roi_size / pooled_size) like the reference. */
typedef struct {
  const float* feat[4];
  float* grad_feat[4]; /* backward only */
  int H[4];
  int W[4];
  float scale[4];
  int num_levels;
  int N;
  int C;
} mmt_pyramid;

int mmt_roi_align_forward(const mmt_pyramid* pyr /*[host]*/, const float* rois, const int32_t* levels,
                          int K, int PH, int PW, int sampling_ratio, float* out, void* stream);
/* the same with the pyramid levels stored as bf16 tensors (feat[l] point to bf16, C % 4 == 0): bf16 activation storage of
 * the bf16 arithmetic mode (mmt_conv_args.io_bf16); taps are widened exactly, arithmetic and output are fp32 */
int mmt_roi_align_forward_bf16(const mmt_pyramid* pyr /*[host]*/, const float* rois, const int32_t* levels,
                               int K, int PH, int PW, int sampling_ratio, float* out, void* stream);
/* grad_feat[l] must be zero-initialised by the caller; accumulates with fp32 atomics */
int mmt_roi_align_backward(const mmt_pyramid* pyr /*[host]*/, const float* rois, const int32_t* levels,
                           int K, int PH, int PW, int sampling_ratio, const float* grad_out, void* stream);
/* The same gradient without atomics and without the caller's clear (round 5): one block per 8 x 8 pixel tile of every level gathers
 * the bins that reach it in (k, ph, pw) order and WRITES every element of every grad_feat[l] once (zeros where no ROI reaches) --
 * a fixed summation order instead of the atomics' (cuda/ROIAlign_cuda.cu:177-254 adds in arrival order too).  Returns 1 and
 * touches nothing when it does not take the call (sampling_ratio != 2, C % 64 != 0, C > 256, K > 8192, or MMT_ROI_BWD_DENSE unset /
 * 0 -- it is OPT-IN: repeatable results, but crowded tiles serialise on one CU and the step is 0.3 ms slower than with the
 * atomics): the caller then clears and calls mmt_roi_align_backward. */
int mmt_roi_align_backward_dense(const mmt_pyramid* pyr /*[host]*/, const float* rois, const int32_t* levels,
                                 int K, int PH, int PW, int sampling_ratio, const float* grad_out, void* stream);

/* ---------------------------------------------------------------- batched NMS
 * replaces _C.nms (csrc/nms.h:10-28; CPU semantics cpu/nms_cpu.cpp:37-64: "+1" areas, suppress on
 * IoU >= thr) for B independent segments in one launch pair (RPN: one segment per (image, level),
 * rpn/inference.py:130-135; detections: one per (image, class), box_head/inference.py:124-126).
 *   boxes     [total,4] xyxy, each segment ALREADY SORTED by descending score
 *   seg_off   int32 [B+1] segment boundaries into boxes
 *   max_n     upper bound on segment length (multiple of 64 not required)
 *   mask_ws   workspace, B * max_n * ceil(max_n/64) uint64
 *   keep      int32 [B, max_n]: positions (within the sorted segment) of kept boxes, ascending
 *   keep_cnt  int32 [B]
 * The greedy sweep runs ON DEVICE (the reference copies the mask to the host, cuda/nms.cu:99-123). */
int mmt_nms_batched(const float* boxes, const int32_t* seg_off, int B, int max_n, float thr,
                    uint64_t* mask_ws, int32_t* keep, int32_t* keep_cnt, void* stream);

/* ---------------------------------------------------------------- input augmentation
 * replaces the PIL / torchvision calls behind data/transforms/transforms.py:28-205 (Resize, RandomHorizontalFlip,
 * AdjustBrightness, AdjustContrast, AdjustHue, RandomErasing, ToTensor, Normalize); images are uint8 [H][W][3] RGB on the
 * device.  Random numbers are drawn by the host mirror exactly as the reference draws them; the kernels are deterministic.
 *   mmt_resample_u8   one pass of Pillow's 8-bit bilinear (antialiased) resample along x (horizontal=1) or y: bounds
 *                     [out][2] (first tap, tap count), coeffs [out][ksize] 22-bit fixed point (host: precompute_coeffs)
 *   mmt_aug_views     V views of one base image: flip, brightness[v], contrast[v], hue_shift[v] (0..255; < 0: no colour
 *                     chain at all, the test-time pipeline), ToTensor,
 *                     BGR*255 - mean3; writes fp32 [H][W] pixels of out_C (>= 3) channels with row pitch out_W pixels into
 *                     out + v*view_stride (a zero-initialised, size-divisible-padded NHWC batch); sums_ws: V uint64
 *   mmt_aug_erase     R rectangles {view, top, left, h, w} filled from fills + fill_off[r] (h*w RGB bytes each) */
int mmt_resample_u8(const uint8_t* src, uint8_t* dst, int H, int W, int out_size, int horizontal, const int32_t* bounds,
                    const int32_t* coeffs, int ksize, void* stream);
int mmt_aug_views(const uint8_t* img, int H, int W, int flip, const float* brightness, const float* contrast,
                  const int32_t* hue_shift, unsigned long long* sums_ws, int V, const float* mean3 /*[host]*/, float* out,
                  long view_stride, int out_W, int out_C, void* stream);
int mmt_aug_erase(float* out, long view_stride, int out_W, int out_C, const int32_t* rects, const long* fill_off,
                  const uint8_t* fills, int R, const float* mean3 /*[host]*/, void* stream);

/* ---------------------------------------------------------------- ground-truth assignment (anchors / proposals)
 * replaces, for N images at once, boxlist_iou (structures/boxlist_ops.py:53-87) + Matcher (modeling/matcher.py:37-139) +
 * the label rules of rpn/loss.py:56-83 / box_head/loss.py:38-80 + BoxCoder.encode (modeling/box_coder.py:23-53).
 *   cand      [A_total,4] xyxy candidates, image n owns rows cand_off[n]..cand_off[n+1]-1 (cand_off, gt_off: device int32
 *             [N+1]); shared_cand=1: ONE [A,4] array (and visible[A]) shared by all images (the anchor grid)
 *   gt        [G_total,4], gt_labels [G_total] int64 (needed for labels_i); every image must own >= 1 gt
 *   matches   int32 [A_total]: index of the matched gt inside its image, -1 below `low`, -2 between `low` and `high`;
 *             allow_low_quality: candidates that are the (tied) best of some gt keep their argmax (top_ws: G_total uint32)
 *   labels_f  (RPN, optional) 1 / 0 / -1 (ignored or not visible);  labels_i (box head, optional) class / 0 / -1
 *   reg       (optional) [A_total,4] regression targets against the matched (or first) gt with weights (wx,wy,ww,wh)
 * Bit-identical to the tensor formulation (same expression order, no FMA contraction). */
/* ---------------------------------------------------------------- proposal selection
 * mmt_rpn_gather_decode: for every (image, level) and every index of that level's pre-NMS top-k (topk [N,k] int64, from
 *   the top-k over the [N, H*W*A] objectness logits): sigmoid(logit), the 4 deltas, the anchor, BoxCoder.decode with
 *   weights 1 and clip_to_image -> boxes / scores / idx / box_reg of shape [N, sumk(,4)], level l at columns
 *   [out_off, out_off + k).  head = the fused RPN head output of the level, NHWC with C = A + 4A channels
 *   (rpn/inference.py:86-113 without its top-k: ~12 launches per level -> 1 for all levels). */
typedef struct { const float* head; const float* anchors; const int64_t* topk; int HW; int k; int out_off; int pad; } mmt_rpn_level;
typedef struct {
  mmt_rpn_level lv[8];
  int L, N, A, C, sumk;
  float clipv;
  const float* lim;   /* [N][2] = (width - 1, height - 1) */
  float* boxes; float* scores; int64_t* idx; float* box_reg;
} mmt_rpn_select_args;
int mmt_rpn_gather_decode(const mmt_rpn_select_args* a /*[host]*/, void* stream);
/* mmt_rpn_topk: torch.topk(objectness, k, sorted=True)[1] of every (image, level) segment (rpn/inference.py:94-96) for all
 *   levels and images of a call in five launches (multi-block radix select + one sorting block per segment): topk [N][k]
 *   int64 indices px * A + a into the level's H*W*A logits (head = fused NHWC head output with C = 5A channels, logits
 *   first), descending by value, equal values: lower index first.  k <= 2048.  workspace: mmt_rpn_topk_workspace_bytes()
 *   bytes, 16-byte aligned. */
typedef struct { const float* head; int64_t* topk; int HW; int k; } mmt_rpn_topk_level;
long mmt_rpn_topk_workspace_bytes(int N, int L, long anchors_per_image);
int mmt_rpn_topk(const mmt_rpn_topk_level* levels /*[host][L]*/, int L, int N, int A, void* workspace, void* stream);
/* mmt_det_postprocess: PostProcessor.filter_results (box_head/inference.py:91-160) for the N images of a batch without a
 *   host round trip: per (image, foreground class j) the rows with prob[:, j] > score_thresh in stable descending score order,
 *   NMS (mmt_nms_batched semantics), survivors in ascending original row order, classes concatenated, and when more than
 *   detections_per_img (> 0) remain the cut `score >= kthvalue(scores, n - D + 1)` (ties kept).  prob [rows][nc], boxes
 *   [rows][nc * 4] (decoded, clipped), row_off [N + 1] on the device and on the host; images of at most 2048 rows, nc <= 64.
 *   Outputs: [N][capo]-shaped with capo = (largest image's rows) * (nc - 1), out_cnt [N] detections per image.
 *   workspace: mmt_det_workspace_bytes(rows, N, nc) bytes, 16-byte aligned. */
long mmt_det_workspace_bytes(int rows, int N, int nc);
int mmt_det_postprocess(const float* prob, const float* boxes, const int32_t* row_off, const int32_t* row_off_host /*[host]*/,
                        int N, int nc, float score_thresh, float nms_thresh, int detections_per_img, void* workspace,
                        float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cnt, void* stream);
/* mmt_rpn_post_select: after mmt_nms_batched over the N*L segments (keep [N*L,kmax], keep_cnt [N*L]): a kept candidate
 *   survives when its rank in its segment is < post_n and its position < own_pre[level] (rpn/inference.py:130-135); of the
 *   survivors the best fpn_post_n of the WHOLE BATCH (training: rpn/inference.py:223-234, written per image in (level,
 *   rank) order, then the image's gt boxes gt[gt_off[n]..gt_off[n+1]) appended with score 1, :55-76) or of each image in
 *   descending score order (inference, :235-242; fpn_post_n <= 2048).  Equal scores: the earlier candidate first.
 *   Outputs have a fixed capacity `cap` per image (>= fpn_post_n + the largest gt count); out_cnt [N] stays on the device. */
typedef struct {
  const float* boxes; const float* scores; const int64_t* idx; const float* box_reg;
  const int32_t* keep; const int32_t* keep_cnt;
  int seg_off[9], own_pre[8];
  int L, N, sumk, kmax, post_n, fpn_post_n, training, cap;
  int min_size_filter;   /* candidates removed by RPN.MIN_SIZE carry score -1 (and a far-away box): they hold no slot */
  int pad;
  const float* gt; const int32_t* gt_off;
  float* out_boxes; float* out_scores; int64_t* out_idx; float* out_reg; int32_t* out_level; int32_t* out_cnt;
  uint64_t* key_scratch;   /* workspace: N * L * min(post_n + 1, kmax) words */
} mmt_rpn_post_args;
int mmt_rpn_post_select(const mmt_rpn_post_args* a /*[host]*/, void* stream);
/* mmt_sample_fg_bg: BalancedPositiveNegativeSampler (balanced_positive_negative_sampler.py:20-72) for n_images label
 *   vectors labels[off[i]..off[i+1]) (float or int64: >= 1 positive, 0 negative, < 0 ignored) with caller-drawn uniform
 *   keys: pos_mask / neg_mask (bytes) mark the min(#pos, max_pos) positives and min(#neg, batch - num_pos) negatives with
 *   the smallest keys (equal keys: lower index first); counts [n_images][2].  One block per image. */
int mmt_sample_fg_bg(const void* labels, int labels_are_float, const float* keys, const int32_t* off, int n_images,
                     int batch_size_per_image, int max_pos, uint8_t* pos_mask, uint8_t* neg_mask, int32_t* counts, void* stream);
/* mmt_sample_fg_bg_wide: the same sampler, same results bit for bit, for long label vectors (the RPN's 262 k anchors per image):
 * the streaming passes run SW_CHUNK = 16384 labels per block instead of one block per image (count, collect the members below
 * tau of both classes into global lists, exact select per image, masks: four launches).  max_n >= every off[i+1] - off[i];
 * workspace: mmt_sample_fg_bg_workspace_bytes(n_images) bytes, 8-byte aligned; n_images <= 64. */
long mmt_sample_fg_bg_workspace_bytes(int n_images);
int mmt_sample_fg_bg_wide(const void* labels, int labels_are_float, const float* keys, const int32_t* off, int n_images, int max_n,
                          int batch_size_per_image, int max_pos, uint8_t* pos_mask, uint8_t* neg_mask, int32_t* counts,
                          void* workspace, void* stream);

/* BoxCoder.decode (modeling/box_coder.py:52-95) of codes [R, ncls*4] against boxes [R,4] with weights (wx,wy,ww,wh) and the
 * dw/dh clip, optionally followed by clip_to_image (structures/bounding_box.py:229-238): row r belongs to image i with
 * row_off[i] <= r < row_off[i+1] and is clamped to [0, lim[2i]] x [0, lim[2i+1]] (= width-1, height-1).  Replaces ~25
 * elementwise launches per call (rpn/inference.py:107-113, box_head/inference.py:60-75); equal to the reference's host arithmetic (true divisions by the weights) up to exp() of the math library. */
int mmt_box_decode(const float* codes, const float* boxes, int R, int ncls, float wx, float wy, float ww, float wh, float clip,
                   const int32_t* row_off /*[n_img+1] or NULL*/, const float* lim /*[n_img,2] or NULL*/, int n_img, float* out,
                   void* stream);
/* Pooler.convert_to_roi_format + LevelMapper (modeling/poolers.py:11-32, 91-104) for the boxes of all images of a call
 * (host arrays of n_img <= 32 device pointers to [count_i, 4] xyxy boxes, 16-byte aligned): rois [K, 5] = (image, box) and, when
 * `levels` is given, levels [K] = clamp(floor(lvl0 + log2(sqrt(area) / s0 + eps)), k_min, k_max) - k_min in the tensor
 * code's fp32 expression order (replaces ~40 elementwise / cat launches per pooler call) */
int mmt_roi_format_levels(const float* const* boxes /*[host]*/, const int32_t* counts /*[host]*/, int n_img, float s0, float lvl0,
                          float eps, int k_min, int k_max, float* rois, int32_t* levels /*or NULL*/, void* stream);
/* IR-Net relation NMS: the IoU-regression labels of the n ranked boxes per foreground class of one image against its ground
 * truth (reference modeling/relation/relation_module.py:323-391, numpy on the host): boxes [n][fg][4], score [n][fg], gt [G][4],
 * gt_labels [G] (class c + 1 belongs to class column c), thresholds [T <= 4] (host) -> out [n][fg][T]; numpy's first-index
 * tie rules; n <= 128, n * G <= 8192, G <= 256. */
int mmt_relation_reg_labels(const float* boxes, const float* score, const float* gt, const int64_t* gt_labels, int n, int fg, int G,
                            const float* thresholds /*[host]*/, int T, float* out, void* stream);
/* mmt_position_embedding: IR-Net's geometric position embedding of every ordered box pair of a class
 * (relation/relation_module.py:93-135 extract_multi_position_matrix): boxes [n][C][4] xyxy, freq [dim_g / 8] device =
 * wave_len^(-m / (dim_g / 8)), out [C][n][n][dim_g] = (sin(100 d_k f_m) for the four log-ratios d_k, then the cosines). */
int mmt_position_embedding(const float* boxes, int n, int C, int dim_g, const float* freq, float* out, void* stream);
/* mmt_relation_attention_{fwd,bwd}: the multi-head geometric relation attention of IR-Net's duplicate-removal network
 * (reference modeling/relation/relation_module.py:33-90, RelationModule.forward: two torch.bmm, log / clamp / add, topk,
 * softmax, scatter, permutes and the 16-group 1x1 conv1) in one launch forward, two backward.  C = classes (x images), N <= 128 ranked
 * boxes per class, G heads, DQ <= 128 query / key dims per head, DV <= 16 value dims per head.
 *   q, k [C*N][G*DQ] (rows in (class, box) order: the WQ / WK Linear outputs as they are), wg [C][N][N][G] (ReLU(WG(position
 *   embedding)), v [C*N][G*DV] = features x conv1.weight^T (the grouped 1x1 conv applied BEFORE the mix: the same bilinear
 *   form), bias [G*DV] (conv1.bias)
 *   S = scale q k^T + log(max(wg, 1e-6)); P = softmax over the top-k entries of every row of S (lower index wins a tie), zero
 *   elsewhere -> P [C][G][N][N] (kept for the backward); out [N][C][G*DV] = P v + bias.
 * bwd: dout [N][C][G*DV] -> dq, dk (like q), dwg (like wg; the clamp passes the gradient where wg >= 1e-6), dv (like v); a row
 * pass (softmax backward, dwg, dq; writes dS) and a column pass (dk, dv): every element is written by exactly one workgroup
 * (no atomics, nothing to zero). */
int mmt_relation_attention_fwd(const float* q, const float* k, const float* wg, const float* v, const float* bias, int C, int N,
                               int G, int DQ, int DV, int topk, float scale, float* P, float* out, void* stream);
int mmt_relation_attention_bwd(const float* q, const float* k, const float* wg, const float* v, const float* P, const float* dout,
                               int C, int N, int G, int DQ, int DV, float scale, float* dS /* workspace [C][G][N][N] */, float* dq,
                               float* dk, float* dwg, float* dv, void* stream);
/* mmt_ciam_{fwd,bwd}: IR-Net's cross-instance attention of the mask refinement (reference
 * modeling/relation/mask_relation_module.py:199-242, CIAM_Module.forward: bmm, max, mean, softmax, mm), all (image, class)
 * groups of the batch in one launch.  x [n][C <= 16][HW] fp32, group [n] int64 ids with equal ids CONTIGUOUS (the attention
 * stays inside a group), max_group >= the largest group (<= 512; n is always a valid bound), gamma [1] device.
 *   E[c][i][j] = <x[i,c,:], x[j,c,:]>; M[i][j] = mean_c (max_j' E[c][i][j'] - E[c][i][j]); A = softmax_j M over the group;
 *   out = gamma A x + x.   Kept for the backward: A [n][n] (zero outside the group), J [C][n] = the arg-max column of E[c][i][:]
 *   (first index on a tie).
 * bwd: dout -> dx [n][C][HW], dgamma [1] (zeroed here, then one atomic per instance); T [n][n], R [n] = workspace. */
int mmt_ciam_fwd(const float* x, const int64_t* group, int n, int C, int HW, int max_group, const float* gamma, float* A, int* J,
                 float* out, void* stream);
int mmt_ciam_bwd(const float* x, const int64_t* group, int n, int C, int HW, int max_group, const float* gamma, const float* A,
                 const int* J, const float* dout, float* T, float* R, float* dx, float* dgamma, void* stream);
/* mmt_rpn_loss: the RPN's two losses over all anchors of the batch (reference modeling/rpn/loss.py:183-194, with the sampler's
 * masks instead of index gathers): obj [R] logits, reg / regt [R][4] (16-byte aligned), labels [R] float (-1 / 0 / 1), pos / neg
 * [R] bool bytes.  n = max(#(pos | neg), 1); out[0] = sum_{pos | neg} BCEWithLogits(obj, max(label, 0)) / n; out[1] =
 * sum_{pos} smooth_l1(reg - regt, beta, sum) / n; dobj [R], dreg [R][4] = the gradients of out[0] w.r.t. obj and of out[1]
 * w.r.t. reg (zero outside the samples).  sums [3] = workspace (zeroed here).  Two launches. */
int mmt_rpn_loss(const float* obj, const float* reg, const float* labels, const float* regt, const uint8_t* pos, const uint8_t* neg,
                 long R, float beta, float* sums, float* out, float* dobj, float* dreg, void* stream);
/* mmt_box_loss: the box head's two losses (reference modeling/roi_heads/box_head/loss.py:118-162): logits [R][NC], breg
 * [R][4 NC], labels [R] int64 in [0, NC), regt [R][4].  out[0] = mean_i CE(logits_i, label_i); out[1] = sum_{label > 0}
 * smooth_l1(breg[i][4 label ..] - regt_i, beta = 1, sum) / R; dlogits, dbreg = their gradients.  out is zeroed here.  One launch. */
int mmt_box_loss(const float* logits, const float* breg, const int64_t* labels, const float* regt, int R, int NC, float* out,
                 float* dlogits, float* dbreg, void* stream);
/* the same on a FIXED-CAPACITY batch (round 6, SURVEY f-2: the sampled lists of box_head/loss.py:82-116 without a host read-back):
 * rows labelled -1 are padding behind an image's sampled set -- no loss, zero gradient --, and both means run over *n_rows (device:
 * the number of rows with label >= 0; NULL = R, i.e. mmt_box_loss) */
int mmt_box_loss_rows(const float* logits, const float* breg, const int64_t* labels, const float* regt, int R, int NC,
                      const int64_t* n_rows, float* out, float* dlogits, float* dbreg, void* stream);
int mmt_match_targets(const float* cand, const int32_t* cand_off, const float* gt, const int32_t* gt_off,
                      const int64_t* gt_labels, const uint8_t* visible, int N, int A_total, int G_total, int shared_cand,
                      float high, float low, int allow_low_quality, float wx, float wy, float ww, float wh, uint32_t* top_ws,
                      int32_t* matches, float* labels_f, int64_t* labels_i, float* reg, void* stream);

/* ---------------------------------------------------------------- implicit-GEMM convolution (fp32 MFMA)
 * replaces ATen/cuDNN conv2d + FrozenBatchNorm2d (layers/batch_norm.py:19-24) + ReLU + residual
 * add (backbone/resnet.py:254-274) + FPN lateral/top-down add (backbone/fpn.py:57-62), nn.Linear
 * (1x1 conv on a [R,1,1,C] tensor), and -- with transformed weights -- their data gradients.
 *   y[n,ho,wo,co] = epi( sum_{kh,kw,ci} x[n, ho*stride+kh-pad, wo*stride+kw-pad, ci] * w[co,kh,kw,ci] )
 *   epi(v) = v*scale[co] + shift[co]  (+ residual)  -> relu?  -> * (mask>0 ? mask_scale : 0)?  -> * mul?
 * residual modes: 0 none, 1 same shape, 2 nearest-x2 upsample of a [N,Ho/2,Wo/2,Cout] tensor (FPN
 * forward), 3 2x2 sum of a [N,2Ho,2Wo,Cout] tensor (FPN backward).
 * out_stride > 1 scatters row (n,ho,wo) to y[n, ho*out_stride, wo*out_stride] of a
 * [N, Ho*out_stride(+), Wo*out_stride(+), Cout] tensor the caller zeroed (data-grad of strided 1x1). */
typedef struct {
  const float* x;
  const float* w;
  const float* scale; /* [Cout] or NULL */
  const float* shift; /* [Cout] or NULL */
  const float* res;   /* residual or NULL */
  const float* mask;  /* [N,Ho,Wo,Cout] or NULL: output multiplied by (mask>0)*mask_scale */
  const float* mul;   /* [N,Ho,Wo,Cout] or NULL: output multiplied elementwise */
  float* y;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int relu, res_mode, out_stride, out_H, out_W; /* out_H/out_W: full dims of y when out_stride>1 */
  float mask_scale;
  /* optional (split-bf16 modes, see mmt_set_conv_precision): the weight tensor packed by mmt_pack_weight(s) --
   * plane q (q = 0..2) at w_planes + q * w_plane_stride bf16 elements.  16-byte aligned, stride % 8 == 0.
   * NULL: the kernel splits the fp32 weights itself (slower). */
  const void* w_planes;
  long w_plane_stride;
  /* optional: x pre-split into three bf16 planes (mmt_split_planes), each indexed exactly like x ([N][H][W][Cin]);
   * taken by the split-bf16 kernels on 128 x 128 tiles (mode 3), ignored otherwise.  Results are bit-identical to
   * the call without it: the same split, done once per tensor instead of once per use inside the kernel. */
  const void* x_planes;
  long x_plane_stride;
  /* optional: ALSO write the result as three bf16 planes (same split as mmt_split_planes, same indexing as y) from the
   * epilogue, for a following 3x3 convolution that wants x_planes; Cout % 4 == 0, out_stride == 1 */
  void* y_planes;
  long y_plane_stride;
  /* bf16 STORAGE of activations (mode 1, BASELINE configs[4] "bf16 MFMA path"): bit 0 (1) `x`, bit 1 (2) `y`, bit 2 (4)
   * `res`, bit 3 (8) `mask` point to bf16 tensors with the indexing of the fp32 tensor they stand for (the pointers keep
   * their declared type); bit 4 (16): `dy` of mmt_conv_wgrad is bf16.  A bf16 `x` is copied to LDS as it is -- it IS the
   * one-term plane -- and needs mode 1, w_planes, Cin % 16 == 0, Cout > 32; a bf16 `y` is the fp32 result rounded to
   * nearest even and replaces the fp32 store.  Values are those of the fp32-storage call on the rounded tensors. */
  int io_bf16;
  /* optional: device float, zeroed by the caller, into which max |y| of this launch's output is accumulated (the scale of a
   * consumer on the two-term fp16 split, mmt_conv3x3_strip_f16x2, without a reduction pass of its own) */
  void* y_amax;
  /* mmt_conv_wgrad only, optional, mode 3: device floats holding max |x| and max |dy| (e.g. recorded through y_amax by the
   * launches that produced them): the weight gradient then runs on the two-term fp16 split (3 products instead of 6) */
  const void* f16_x_amax;
  const void* f16_dy_amax;
  /* nonzero: y_amax points to a 33-float slot (zeroed by the caller); besides slot[0] = max |y| the launch accumulates,
   * over a sample of its tiles (every 64th block), sum |y| into slot[1 .. 16] and the number of sampled elements into
   * slot[17 .. 32].  max / mean = the crest factor of the tensor, by which the host decides
   * whether its consumers keep the two-term fp16 split or fall back to the three-term bf16 split (fp16 has 5 exponent
   * bits: a tensor dominated by a few huge elements pushes everything else below the range of the low term) */
  int y_amax_stats;
  /* fp16 split, range guard (round 4; optional, device pointers).  f16_guard_x: the 33-float statistics slot of `x` as a
   * producing launch recorded it (y_amax + y_amax_stats) or mmt_amax_stats computed it; f16_guard_dy: the same for `dy`
   * (mmt_conv_wgrad).  With it every block of an fp16-split launch (mmt_conv_forward_f16x2, mmt_conv3x3_strip_f16x2, the
   * fp16-split form of mmt_conv_wgrad) first derives the crest factor max / mean of ITS OWN operand on the device; above 2^17
   * -- one element 10^8 x the rest pushes everything else below the range of the low fp16 term -- the block computes its tile
   * with exact fp32 products from the fp32 operands instead (slow; until the host has moved the site to the 3-term bf16
   * split).  w_src / w_src_scale: for a call whose weight exists only as packed planes (`w` null: a data gradient), the
   * forward weight [Cin][KH][KW][Cout] and its per-row scale (or null) the planes were packed from (mmt_pack_weight_flipped):
   * the slow path reads the fp32 weights through the same flip / transpose.  Null guards switch the test off. */
  const void* f16_guard_x;
  const void* f16_guard_dy;
  const void* w_src;
  const void* w_src_scale;
  /* mmt_conv_wgrad on the fp16 split only, optional (round 4): a SECOND (x, dy) pair of the same shapes whose weight gradient
   * is accumulated by the same launch -- the labeled and the unlabeled student pass of a mean-teacher step share every weight
   * (engine/MTtrainer.py): dW = dY1^T im2col(X1) + dY2^T im2col(X2) as one launch over both pixel ranges instead of two
   * half-sized ones (half the launches, twice the pixels per block).  x2 / dy2 with their own maxima f16_x_amax2 / f16_dy_amax2
   * (and optional statistics slots f16_guard_x2 / f16_guard_dy2): every block works on one segment with that segment's
   * scales.  mmt_conv_wgrad_splits counts the slices of the two-segment launch when x2 is set. */
  const void* x2;
  const void* dy2;
  const void* f16_x_amax2;
  const void* f16_dy_amax2;
  const void* f16_guard_x2;
  const void* f16_guard_dy2;
  /* mmt_conv_forward_pg AND mmt_conv3x3_strip_f16x2 (the host side passes 1 to both by default; every other entry point refuses a
   * non-zero value): 0 = x_planes indexed like x ([N][H][W][Cin]); 1 = ROW-BLOCKED planes
   * [N * H][Cin / 16][W][16] (mmt_split_planes_f16_rb): the 16 channels of a 16-k step of consecutive pixels of an image row are
   * contiguous, so a copy instruction of the kernel reads runs of up to 1 KiB instead of 32-byte pieces of 32 cache lines */
  int x_planes_layout;
  /* Round 6: planes out of the PRODUCER's epilogue (the split pass in front of a plane-fed consumer disappears).  y_rb != NULL: the
   * launch also writes y as the two fp16 planes of y * s in the row-blocked order [N * Ho][Cout / 16][Wo][16] (plane q at y_rb +
   * q * y_rb_stride elements), s = *y_rb_scale -- a power of two the CALLER chose before the values exist (the host side keeps one
   * per producing site: the largest |y| any call of the site recorded during the previous step x 2 head-room, mmt_rb_scales_update).
   * Needs Cout % 16 == 0, out_stride == 1, fp32 y, N * Ho * Wo * Cout < 2^30; honoured by every epilogue form of
   * mmt_conv_forward_f16x2 / mmt_conv3x3_strip_f16x2 / mmt_conv_forward_pg (register-direct, LDS-staged, split-K finish,
   * row-resident), refused (MMT_EINVAL) elsewhere.  y_amax_next: a second device word that receives max |y| like y_amax[0]
   * (conditional atomic, once per block) -- the site's pending maximum, folded into its scale by mmt_rb_scales_update.
   * x_planes_lag = 1 tells a CONSUMER that the scale of its x_planes (*s_x) was chosen that way: its range guard then tests
   * max |x| * s_x against the fp16 range and the sampled mean against the low term's, with the actual scale, and takes the exact
   * fp32 path for the launch when either fails (a tensor that outgrew last step's head-room, or shrank below it).
   * mmt_conv_wgrad_planes: bit 0 = the planes of x, bit 1 = the planes of dy. */
  void* y_rb;
  long y_rb_stride;
  const float* y_rb_scale;
  void* y_amax_next;
  int x_planes_lag;
  /* y_rb for the leading y_rb_rows output pixels only (0 = all N * Ho * Wo): the planes are image-major, so "the first n images" is
   * n * Ho * Wo -- the teacher's K x flip batch, of which only view 0 feeds a plane-fed launch (the coarse inference's RPN head) */
  int y_rb_rows;
} mmt_conv_args;

int mmt_conv_forward(const mmt_conv_args* a /*[host]*/, void* stream);
/* which tile configuration mmt_conv_forward picks for these shapes: 0 = 128x32, 1 = 128x128 (the dominant
 * kernel of the step, conv_fwd_kernel<128,128,2,2>), 2 = 64x64.  Used by bench.py for the roofline line. */
int mmt_conv_variant(const mmt_conv_args* a /*[host]*/);
/* number of K ranges mmt_conv_forward would use for this call on the DMA-fed kernel (1 = un-split; > 1: partial tiles go
 * through the library's per-stream workspace and a finish launch).  Tuning / measurement aid like mmt_conv_variant. */
int mmt_conv_ksplit(const mmt_conv_args* a /*[host]*/);
/* 1 when mmt_conv_forward would run this call on the all-planes 3x3 kernel if x_planes were given (mode 3; 3x3, stride 1,
 * pad 1, Cin % 32 == 0, Cin >= 128, W % 64 == 0, enough 256-pixel tiles to fill the chip): the caller then splits x once
 * with mmt_split_planes (or has the producing convolution write the planes) and passes them. */
int mmt_conv_wants_planes(const mmt_conv_args* a /*[host]*/);
/* arithmetic of the convolution GEMMs -- forward, data gradient and weight gradient (process-wide; initial value from
 * the environment variable MMT_CONV_PRECISION, default 3):
 *   3  fp32 operands split on the fly into three bf16 terms x = x0 + x1 + x2 (round-to-nearest at each level, exact to
 *      2^-27 |x|); a*b is evaluated as the six products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 on the bf16 matrix pipe
 *      (v_mfma_f32_32x32x16_bf16, fp32 accumulate, small terms first).  Dropped terms <= 3*2^-27 |ab|, below the
 *      rounding of the fp32 product itself: measured error against fp64 is equal to or lower than mode 0's
 *      (profiles/r01_precision.txt) at 1.3-1.7x its speed -- the fp32-input MFMA runs at 1/16 of the bf16 rate.
 *   0  fp32-input MFMA (v_mfma_f32_32x32x2_f32): IEEE fp32 products and fp32 accumulate, the arithmetic of the
 *      reference's ATen fp32 convolution
 *   2  two-term split, 3 bf16 MFMAs (~2^-18 relative per product)
 *   1  plain bf16 inputs, fp32 accumulate (torch.autocast(bfloat16)-class arithmetic; BASELINE config 5)
 * Tensors stay fp32 in HBM in every mode.  Shapes the split kernels do not cover (Cin % 16 != 0 or Cout <= 32 forward,
 * Cout % 4 != 0 weight gradient) always run in mode 0.  Returns MMT_EINVAL for an unknown mode. */
int mmt_set_conv_precision(int mode);
/* EXPERIMENT (not used by the product path; mmt-psm_amd/tools/bench_f16x2.py): the tap-strip 3x3 kernel on
 * a TWO-term fp16 split -- x * s = h + l, 22 significant bits, 3 matrix products per multiply instead of 6.  The caller
 * scales each operand tensor by a power of two (largest magnitude near 2^14), passes the two fp16 planes of the input
 * (mmt_split_planes_f16) and of the packed weight (mmt_pack_weight_f16 / _flipped_f16) in x_planes / w_planes; the scales are
 * device scalars derived on the device from mmt_amax (no host round trip), the epilogue divides the sum by s_x s_w.
 * Strip shapes only (mmt_conv_wants_planes).  Opt-in from Python with MMT_F16X2=1. */
int mmt_amax(const float* x, long n, const float* rowscale, long inner, int rows, float* amax /*device, zeroed*/, void* stream);
/* the same reduction into a 33-float statistics slot (device, zeroed): slot[0] = max |x|, slot[1..16] += sums of |x| over a
 * sample of the tensor, slot[17..32] += the sample's element counts
 * (what mmt_conv_args.y_amax_stats makes a producing convolution record about its output) */
int mmt_amax_stats(const float* x, long n, float* slot /*device, zeroed*/, void* stream);
/* y = a + b (+ c) (+ d), elementwise in that order (n % 4 == 0, 16-byte aligned, c / d NULL when absent, y may be an input),
 * and the statistics of y into `slot` exactly as mmt_amax_stats records them.  The gradient of a tensor with several
 * consumers (a pyramid level read by the RPN head and two poolers; a ResNet stage output read by the next stage and the
 * FPN lateral; reference: autograd's own accumulation in engine/MTtrainer.py:101-104 `losses.backward()`) in one pass
 * instead of n - 1 additions plus a reduction pass. */
int mmt_sum_stats(const float* a, const float* b, const float* c, const float* d, float* y, long n,
                  float* slot /*device, zeroed*/, void* stream);
/* statistics slot of a tensor whose elements are convex combinations of elements of n source tensors (the pooled features of
 * mmt_roi_align_forward: bilinear taps of the pyramid levels, poolers.py:91-121): out[0] = the largest of the sources' maxima (an
 * upper bound: what a consumer's power-of-two scale needs), sums / counts added.  slots: HOST array of n <= 8 device pointers to
 * 33-float slots (they travel in the kernel arguments).  Replaces a reduction pass over the 51-205 MB pooled tensor in front of
 * fc6 / the mask head. */
int mmt_stats_combine(const float* const* slots /*[host]*/, int n, float* out /*device, 33 floats*/, void* stream);
int mmt_split_planes_f16(const float* x, void* planes, long plane_stride, long n, float scale, const float* amax /*device or NULL*/,
                         float* scale_out /*device or NULL*/, float* amax_next /*device or NULL: max |x| of THIS tensor is
                         accumulated here, for the scale of the next tensor in the same role (delayed scaling)*/,
                         float* zero_slot /*device or NULL: set to 0 (the accumulator of the call after this one)*/, void* stream);
int mmt_pack_weight_f16(const float* w, void* planes, long plane_stride, int Cout, int K, float scale, const float* amax,
                        float* scale_out, void* stream);
int mmt_pack_weight_flipped_f16(const float* w, const float* scale, void* planes, long plane_stride, int Cout, int KH, int KW,
                                int Cin, const float* amax, float* scale_out, void* stream);
int mmt_conv3x3_strip_f16x2(const mmt_conv_args* a /*[host]*/, const float* s_x /*device*/, const float* s_w /*device*/, void* stream);
/* the same arithmetic for every other shape of the DMA-fed kernel (1x1, strided, 3x3 on small maps): raw fp32 x, split in
 * registers after its scaling by the power of two of x_amax (device: max |x|, e.g. recorded through y_amax) */
int mmt_conv_forward_f16x2(const mmt_conv_args* a /*[host]*/, const float* x_amax /*device*/, const float* s_w /*device*/, void* stream);
int mmt_get_conv_precision(void);
/* Round 5: the ResNet stem as one launch (csrc/conv_stem.hip) -- conv 7x7 / stride 2 / pad 3 (3 -> 64) + FrozenBatchNorm + ReLU + max
 * pool 3x3 / stride 2 / pad 1: reference modeling/backbone/resnet.py:288-293 (StemWithFixedBatchNorm.forward) with
 * layers/batch_norm.py:19-24 folded into scale / shift.  x [N][3][H][W] fp32 NCHW (H, W multiples of 4), y [N][H/4][W/4][64] fp32
 * NHWC.  w_s2d: the filter as the 4x4 / stride-1 filter over the 2x2 space-to-depth image, [64][4][4][16] fp32 (channel =
 * (row parity, column parity, RGB + zero)); w_planes / s_w: its packed fp16 planes and their device scale (mmt_pack_weight_f16);
 * x_slot: 33-float statistics slot of x (mmt_amax_stats: [0] = max |x| gives the power-of-two scale of the fp16 split on the
 * device, the sums the range guard -- an image whose crest factor defeats fp16 takes exact fp32 products); y_slot (optional):
 * statistics of y are accumulated there like mmt_conv_args.y_amax with y_amax_stats.  Mode 3 (two-term fp16 split) only; results
 * are bit-identical to mmt_conv_forward_f16x2 on the space-to-depth image followed by mmt_maxpool3x3s2. */
int mmt_stem_fused(const float* x, int N, int H, int W, const float* w_s2d, const void* w_planes, long w_plane_stride,
                   const float* s_w /*device*/, const float* scale, const float* shift, const float* x_slot /*device*/, float* y,
                   float* y_slot /*device, zeroed, or NULL*/, void* stream);
/* Weight gradient of a stride-1, "same"-padded convolution from ROW-BLOCKED fp16 planes of both operands (round 5; replaces the
 * conv2d weight gradient behind layers/misc.py:30-43 where the planes exist anyway: the forward launch's input planes and the
 * data-gradient launch's gradient planes, mmt_split_planes_f16_rb): dw += rowscale[co] * dy^T im2col(x), dbias += column sums of dy.
 * a: shapes (+ x for the exact path, f16_guard_x / f16_guard_dy); s_x / s_dy: device scalars, the planes' scales.
 * mmt_conv_wgrad_planes_splits: 0 when the layer is not taken (needs W % 32 == 0, Cin % 128 == 0, Cout % 128 == 0, odd square
 * kernel with pad (k - 1) / 2, stride 1, mode 3; MMT_WGRAD_PLANES=0), else the number of pixel ranges across blocks: workspace
 * (device, splits * Cout * KH * KW * Cin floats) is needed when it is > 1.  mmt_conv_wgrad_planes returns 1 and touches nothing
 * when the layer is not taken.  Products (h l), (l h), (h h) per 16 pixels, fp32 accumulation: the arithmetic of mmt_conv_wgrad's
 * fp16-split form, another summation order. */
int mmt_conv_wgrad_planes_splits(const mmt_conv_args* a /*[host]*/);
int mmt_conv_wgrad_planes(const mmt_conv_args* a /*[host]*/, const float* dy, const void* x_planes, long x_plane_stride,
                          const void* dy_planes, long dy_plane_stride, const float* s_x, const float* s_dy, const float* rowscale,
                          float* dw, float* dbias, float* workspace, void* stream);
/* Host-side helper (no device work of its own): calls n recorded entry points of this library in order, each with its 16 recorded
 * integer / pointer arguments (entry points with fewer parameters ignore the rest; none with floating-point parameters may be
 * recorded), and stops at the first nonzero return code (-> that code, *failed_index = its position).  What the Python layer's
 * launch plans replay a no-grad pass through: one call from the interpreter instead of one per launch. */
typedef struct {
  void* fn;
  long a[16];
} mmt_call;
int mmt_replay(const mmt_call* calls /*[host]*/, int n, int* failed_index /*[host] or NULL*/);
/* x (NHWC fp32: `rows` = N * H image rows of W pixels, C channels, C % 16 == 0) -> the two fp16 planes of x * s in the row-blocked
 * order [rows][C / 16][W][16] that mmt_conv_forward_pg takes with mmt_conv_args.x_planes_layout = 1; s = the power of two derived on
 * the device from *amax (max |x|, e.g. a producer's statistics slot), also written to *scale_out.  Same values as
 * mmt_split_planes_f16, another order. */
int mmt_split_planes_f16_rb(const float* x, void* planes, long plane_stride, int rows, int W, int C, const float* amax,
                            float* scale_out, void* stream);
/* Round 6 (planes out of the producers, see mmt_conv_args.y_rb).
 * mmt_conv_writes_rb: 1 when the launch mmt_conv_forward_f16x2 / mmt_conv3x3_strip_f16x2 / mmt_conv_forward_pg would take for
 *   these shapes writes y_rb from its epilogue (asked by the host before it allocates the planes); 0: the consumer splits itself.
 * mmt_sum_stats_rb: mmt_sum_stats (y = a + b (+ c) (+ d), statistics into `slot`) over an NHWC tensor of `rows` = N * H image rows
 *   of W pixels x C channels that also writes y's row-blocked fp16 planes with *scale (planes NULL: the sum and its statistics
 *   alone -- a site's first step), and max |y| into *amax_next (or NULL) --
 *   replaces autograd's AccumulateGrad additions of a multi-consumer tensor (layers/fused.py::ForkFn) AND the split pass of the
 *   3x3 data gradient behind it (reference: autograd of backbone/fpn.py:57-69 / rpn/rpn.py:39-46).
 * mmt_rb_scales_update: once per training step over the host side's table of producing sites, state[2 i] = scale, state[2 i + 1] =
 *   pending maximum (what y_amax_next / amax_next accumulated): scale <- the power of two that puts 2 x pending into [2^13, 2^14),
 *   pending <- 0; sites without a pending maximum keep their scale. */
int mmt_conv_writes_rb(const mmt_conv_args* a /*[host]*/);
/* Round 6: the weight gradients of a BATCH of layers (what a backward pass hands over at a time) as grouped launches -- replaces the
 * per-layer conv2d weight gradients autograd issues behind layers/misc.py:30-43 for backbone/resnet.py:202-274, backbone/fpn.py:43-69,
 * rpn/rpn.py:39-46 and the heads.  Every job is what mmt_conv_wgrad takes (a, dy, rowscale, dw, dbias; a.f16_x_amax / f16_dy_amax and
 * the guards set for the fp16-split arithmetic) plus, optionally, both operands' row-blocked planes (mmt_conv_wgrad_planes' arguments).
 * Plane-fed jobs go out in groups of <= 12, fp16-split jobs without planes in groups of <= 6 per pixel-decode form, every slab of the
 * batch is summed by ONE reduce launch; a job no group takes (other arithmetic, a two-segment job, a group of one) is launched as
 * mmt_conv_wgrad(_planes) would launch it.  Inside a group the tiles of all layers fill the chip together: a layer is cut into a
 * quarter of the pixel ranges it would use alone (MMT_WGRAD_GROUP_DIV; 0 = as few as fill the chip as a group: measured slower in the
 * training step, whose latency-bound data-gradient chain shares the GPU with these launches).  workspace: *floats_out of mmt_conv_wgrad_group_workspace(jobs, n, &floats) floats (0: none needed), alive until the stream has
 * run the call.  n <= 96.  MMT_WGRAD_GROUP=0 (environment, read per call): every job as its single launch. */
typedef struct mmt_wgrad_job {
  mmt_conv_args a;
  const float* dy; const float* rowscale; float* dw; float* dbias;
  const void* x_planes; long x_plane_stride; const void* dy_planes; long dy_plane_stride; const float* s_x; const float* s_dy;
} mmt_wgrad_job;
int mmt_conv_wgrad_group_workspace(const mmt_wgrad_job* jobs /*[host]*/, int n, long* floats_out /*[host]*/);
int mmt_conv_wgrad_group(const mmt_wgrad_job* jobs /*[host]*/, int n, float* workspace, long workspace_floats, void* stream);
int mmt_sum_stats_rb(const float* a, const float* b, const float* c, const float* d, float* y, int rows, int W, int C, float* slot,
                     void* planes, long plane_stride, const float* scale, float* amax_next, void* stream);
int mmt_rb_scales_update(float* state, int n, void* stream);
/* Round 5: the plane-fed implicit GEMM (csrc/conv_pgemm.hip) -- the same arithmetic (two-term fp16 split, 3 products, mode 3) for
 * any (KH, KW, stride, pad) with Cin % 16 == 0, Cout > 32, res_mode <= 1, out_stride == 1, no `mul`, fp32 tensors: x_planes = the
 * two fp16 planes of x * s_x with x's NHWC indexing (mmt_split_planes_f16; x_planes_layout 0) or row-blocked (mmt_split_planes_f16_rb
 * or a producer's y_rb; x_planes_layout 1, with x_planes_lag as the planes' origin demands); x_planes must not be NULL, w_planes = the packed fp16 planes of w * s_w
 * (mmt_pack_weight_f16 / _flipped_f16), s_x / s_w device scalars, `x` (and `w`, or w_src) the fp32 tensors for the range guard's
 * exact path.  Replaces the ATen convolution / addmm behind layers/misc.py:30-43 `Conv2d`, backbone/resnet.py:254-274 (conv2 of
 * layer3 / layer4), backbone/fpn.py:57-66 and rpn/rpn.py:39-46 (3x3 on the small levels),
 * roi_heads/mask_head/roi_mask_feature_extractors.py:131-146, box_head/roi_box_feature_extractors.py:97-98 (fc6 / fc7), forward and
 * data gradient.  tile_rows: 64 | 128 | 256 (4 / 2 / 1 K groups inside a block) | 0 = the library's choice; ksplit: K ranges (> 1: partial tiles meet in the per-stream
 * workspace inside the SAME launch -- the last block of a tile to arrive adds them in the order 0 .. ksplit - 1 and runs the
 * epilogue), 0 = the library's choice.  Results are bit-identical to mmt_conv_forward_f16x2 on the tiled kernel for an equal number
 * of K ranges.  mmt_conv_pg_plan: the tile height and K ranges the library would pick (both 0: not a shape for this kernel). */
int mmt_conv_forward_pg(const mmt_conv_args* a /*[host]*/, const float* s_x /*device*/, const float* s_w /*device*/, int tile_rows,
                        int ksplit, void* stream);
int mmt_conv_pg_plan(const mmt_conv_args* a /*[host]*/, int* tile_rows, int* ksplit);
/* 1 when the library wants this call on mmt_conv_forward_pg with a plane-split pass of x in front (3x3 and larger kernels that
 * are not tap-strip shapes: mmt_conv_wants_planes is asked first); 0 otherwise.  MMT_PG=0 in the environment answers 0. */
int mmt_conv_pg_wanted(const mmt_conv_args* a /*[host]*/);
/* Packed bf16 planes of a weight matrix w[Cout][K] (K = KH*KW*Cin in the weight's own memory order, K % 16 == 0) for
 * the split-bf16 modes.  Plane q (q = 0..2, at planes + q*plane_stride bf16 elements) holds the q-th term of the
 * round-to-nearest bf16 expansion w = w0 + w1 + w2 (exact to 2^-27 |w|), tiled as
 * [K/16][ceil(Cout/32)][32 rows][2 halves][8] -- the LDS image of the kernels, so that their global->LDS DMA reads
 * 1 KiB of consecutive memory per instruction.  mmt_packed_weight_elems = elements per plane (-1 if unsupported).
 * mmt_pack_weights packs many matrices of one fp32 buffer in a single launch (engine/flat.py: once per SGD / EMA step):
 * descs[] (device) gives per matrix its offset in `base`, its offset inside a plane, Cout, K and its first unit index
 * (unit = one 1 KiB tile = 512 elements); unit_desc[u] (device) = index of the matrix unit u belongs to. */
typedef struct { long src_off, dst_off; int Cout, K, unit0, pad; } mmt_pack_desc;
long mmt_packed_weight_elems(int Cout, int K);
int mmt_pack_weight(const float* w, void* planes, long plane_stride, int Cout, int K, void* stream);
/* packed planes of the data-gradient weights of a convolution, straight from its forward weight w[Cout][KH][KW][Cin]:
 * the matrix [Cin][(KH-1-kh, KW-1-kw, co)] = w * scale[co] (what mmt_weight_flip_transpose produces in fp32) in the tiled
 * plane form; Cout % 16 == 0.  mmt_conv_forward accepts w == NULL when w_planes is given and the shape is one the
 * DMA-fed kernels take (Cin % 16 == 0, Cout > 32 of THAT call, mode != 0). */
int mmt_pack_weight_flipped(const float* w, const float* scale, void* planes, long plane_stride, int Cout, int KH, int KW,
                            int Cin, void* stream);
/* x[n] fp32 (n % 8 == 0, 16-byte aligned) -> three bf16 planes planes[q * plane_stride + i], x = p0 + p1 + p2 with
 * round-to-nearest at each level (|x - sum| <= 2^-27 |x|): the activation-side counterpart of mmt_pack_weight */
int mmt_split_planes(const float* x, void* planes, long plane_stride, long n, void* stream);
/* mmt_pack_weight_flipped for a whole table of weights in one launch (descs / unit_desc on the device; unit_desc[u] = index
 * of the descriptor that 512-element unit u belongs to, descs[d].unit0 = its first unit): once per optimiser step */
typedef struct { const float* w; const float* scale; void* dst; long plane_stride; int Cout, KH, KW, Cin, unit0, pad; } mmt_flip_desc;
int mmt_pack_weights_flipped(const mmt_flip_desc* descs /*[dev]*/, const int* unit_desc /*[dev]*/, int n_units, void* stream);
int mmt_pack_weights(const float* base, void* planes, long plane_stride, const mmt_pack_desc* descs /*[dev]*/,
                     const int* unit_desc /*[dev]*/, int n_units, void* stream);
/* the same two tables for the two-term fp16 split (the default arithmetic of mode 3): planes = two fp16 planes of
 * w_d * s_d, s_d the power of two that puts max |w_d| into [2^13, 2^14], derived on the device (a reduction launch and a
 * packing launch); stat[2 d] <- max |w_d| (x scale), stat[2 d + 1] <- s_d, the device scalar the convolutions take as s_w.
 * Replaces nothing in the reference: it keeps the packed form of engine/flat.py's parameter buffer current after
 * solver/build.py's SGD step and MTtrainer.py:277-281's EMA. */
int mmt_pack_weights_f16(const float* base, void* planes, long plane_stride, const mmt_pack_desc* descs /*[dev]*/,
                         const int* unit_desc /*[dev]*/, int n_units, int n_descs, float* stat /*[dev][n_descs][2]*/, void* stream);
int mmt_pack_weights_flipped_f16(const mmt_flip_desc* descs /*[dev]*/, const int* unit_desc /*[dev]*/, int n_units, int n_descs,
                                 float* stat /*[dev][n_descs][2]*/, void* stream);

/* weight gradient: dw[co,kh,kw,ci] += rowscale[co] * sum_{n,ho,wo} dy[n,ho,wo,co] * x[n,ho*s+kh-p,wo*s+kw-p,ci]
 * (dw = the caller's gradient buffer, same layout as the weight; Cin % 4 == 0); optional dbias[co] += sum dy[..,co].
 * The pixel dimension is split over `mmt_conv_wgrad_splits(a)` blocks; when that is > 1 the caller passes a
 * workspace of splits * Cout*KH*KW*Cin floats (partial tiles are stored there and summed by a second kernel --
 * fp32 atomics measured 1.6x slower on mid-size layers).  Uses N,H,W,Cin,Cout,KH,KW,stride,pad,Ho,Wo of a. */
int mmt_conv_wgrad_splits(const mmt_conv_args* a /*[host]*/);
int mmt_conv_wgrad(const mmt_conv_args* a /*[host]; x = input, mask/mul/res unused*/, const float* dy,
                   const float* rowscale /*[Cout] or NULL*/, float* dw, float* dbias /*or NULL*/,
                   float* workspace /*or NULL when splits == 1*/, void* stream);

/* bias gradient: out[c] += sum_m dy[m][c]  (dy [M,C] row-major, out zeroed or accumulated by the caller) */
int mmt_colsum(const float* dy, int M, int C, float* out, void* stream);

/* weight re-layout for data gradients: wd[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci] * scale[co] */
int mmt_weight_flip_transpose(const float* w, const float* scale /*or NULL*/, float* wd, int Cout, int KH, int KW,
                              int Cin, void* stream);

/* stem max-pool 3x3 s2 p1 (backbone/resnet.py:292), NHWC */
int mmt_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* the same on bf16 tensors (bf16 activation storage, see mmt_conv_args.io_bf16) */
int mmt_maxpool3x3s2_bf16(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------- losses (forward value + gradient in one launch)
 * mask-logit BCE (mask_head/loss.py:177-179): logits [P,28*28,NC] NHWC, labels int32 [P], targets [P,28*28] {0,1};
 * loss (1 float, accumulated; zero it first) = mean BCEWithLogits(logits[p,:,label_p], target); grad [P,28*28,NC]
 * fully written (zeros off the label channel), scaled by grad_scale. */
int mmt_mask_bce(const float* logits, const int32_t* labels, const float* targets, int P, int HW, int NC,
                 float grad_scale, float* loss, float* grad, void* stream);

/* MGD feature-hint loss (detector/generalized_rcnn.py:243-282), one pyramid LEVEL, all teacher pyramids:
 *   term_i = sum((s - t_i')^2 * m) / (sum(m)*C + 1e-7),  t_i' = horizontally flipped t_i if flip[i]
 * s, t_i NHWC [N,H,W,C]; m [N,H,W] {0,1}.
 * forward : acc[i] += sum((s-t_i')^2 m)  (i < nt),  acc[nt] += sum(m)      (acc zeroed by the caller)
 * backward: grad_s = 2 m * sum_i coef[i] (s - t_i')   (coef: DEVICE array [nt], = upstream/(n_terms*den_i)) */
typedef struct {
  const float* t[8];
  int flip[8];
  int nt;
} mmt_mgd_teachers;
int mmt_mgd_level_forward(const float* s, const mmt_mgd_teachers* T /*[host]*/, const float* m, int N, int H, int W,
                          int C, float* acc, void* stream);
int mmt_mgd_level_backward(const float* s, const mmt_mgd_teachers* T /*[host]*/, const float* m, int N, int H, int W,
                           int C, const float* coef, float* grad_s, void* stream);
/* binary mask pyramid level: m[n,h,w] = adaptive_avg_pool2d(seg[n], (H,W)) > 0.5, seg int32 [N,IH,IW] */
int mmt_mask_pool(const int32_t* seg, int N, int IH, int IW, int H, int W, float* m, void* stream);

/* PSM loss rows (box_head/loss.py:185-237,267-287).  For every ROI r:
 *   tbar = mean_k teacher[k,r,:];  t = softmax(tbar);  kind 0 ('ce'): t = sharpen(t, temp) if sharpen
 *   rowloss[r] = roww[r] * sum_c( -t_c * log_softmax(student[r])_c )          (kind 0)
 *              = roww[r] * sum_c( t_c * (log t_c - log_softmax(student[r])_c) ) (kind 1, 'kl')
 *   rowgrad[r,c] = roww[r] * (softmax(student[r])_c - t_c)
 * roww[r] in {0, 1, CLS_BALANCE_WEIGHT}: 0 = not selected, 1 = positive, w = kept hard negative.  The caller
 * normalises by 1/(S*3) ('ce': .mean(0).sum()/3) or 1/(S*NC) ('kl', 'mse': element mean), S = #selected.
 * kind: 0 soft-target CE ('bce'/'ce'), 1 KL, 2 MSE on the logits (rowloss = roww * sum_c (s_c - mean_k t_kc)^2). */
int mmt_psm_rows(const float* teacher, int Kaug, const float* student, int R, int NC, const float* roww,
                 float temp, int sharpen, int kind, float* rowloss, float* rowgrad, void* stream);
/* per-ROI perturbation sensitivity v[r] = sum_c std_k(q[k,r,c]) (unbiased, :164-173,191-194); q = softmax of the
 * teacher logits when use_softmax (MT.CLS_LOSS_TYPE == 'bce'), the raw logits otherwise */
int mmt_psm_variance(const float* teacher, int Kaug, int R, int NC, int use_softmax, float* v, void* stream);

/* ---------------------------------------------------------------- optimiser / EMA on flat parameter storage
 * EMA teacher update (engine/MTtrainer.py:277-281): t.mul_(alpha).add_(s, alpha=1-alpha); alpha is the Python
 * double, the kernel uses (float)alpha and (float)(1-alpha) like the reference's scalar casts */
int mmt_ema_update(float* teacher, const float* student, int64_t n, double alpha, void* stream);
/* SGD with momentum, torch.optim.SGD semantics (solver/build.py:5-23): g += wd*p; buf = first ? g : mom*buf + g;
 * p -= lr*buf */
int mmt_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float wd, float momentum,
                     int first, void* stream);

/* teacher pseudo-mask (mask_head/inference.py:29-65,169-246 + generalized_rcnn.py:129-132): for every detection
 * d of image img[d]: prob = sigmoid(logits[d,:,:,label[d]]) (logits NHWC [D,M,M,NC]), zero-pad by 1, expand the
 * box by (M+2)/M, bilinear-resize (align_corners=False) to the integer box, threshold, and add 1 to
 * seg[img[d]] (int32 [N,IH,IW], zeroed by the caller) at every pixel above thresh. */
int mmt_paste_masks(const float* logits, const int32_t* labels, const float* boxes /*[D,4]*/, const int32_t* img,
                    int D, int M, int NC, int IH, int IW, float thresh, int32_t* seg, void* stream);
/* the evaluator's form of the same paste (Masker.forward_single_image, mask_head/inference.py:221-246, as
 * data/datasets/evaluation/pap/pap_eval.py:107-109 calls it on predictions whose masks are still M x M): prob [D,M,M] = the
 * PROBABILITIES of the predicted class (MaskPostProcessor's `mask` field), boxes [D,4] in the target image's frame;
 * stack uint8 [D,IH,IW], zeroed by the caller, gets 1 wherever detection d's pasted probability exceeds thresh. */
int mmt_paste_mask_stack(const float* prob, const float* boxes /*[D,4]*/, int D, int M, int IH, int IW, float thresh,
                         uint8_t* stack, void* stream);

/* polygon -> MxM mask targets (mask_head/loss.py:37-75, structures/segmentation_mask.py:96-133,
 * pycoco/maskApi.c:166-206 rleFrPoly + :53-74 union) for P positive ROIs.
 *   poly_xy   float32 concatenated vertices (x,y interleaved) of all polygons
 *   poly_off  int32 [NP+1] vertex offsets of each polygon
 *   roi_poly  int32 [P,2]: polygons roi_poly[p][0]..roi_poly[p][1]-1 belong to ROI p (its matched instance)
 *   boxes     [P,4] proposals;  out [P,M,M] float {0,1};  overflow int32[1] set if a ROI exceeded the
 *   crossing-list capacity (never for 28x28 targets of sane polygons) */
int mmt_polygon_targets(const float* poly_xy, const int32_t* poly_off, const int32_t* roi_poly, const float* boxes,
                        int P, int M, float* out, int32_t* overflow, void* stream);

#ifdef __cplusplus
}
#endif
#endif
