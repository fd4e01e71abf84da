/*
 * retinahead.h - C ABI of libretinahead.so: the RetinaNet per-anchor dense-head
 * path (anchors, IoU assignment, focal + smooth-L1 loss forward/backward, box
 * decode/clip, score threshold + NMS) as hand-written gfx950 (MI355X) kernels.
 *
 * Conventions for every entry point
 *   - Pointers named in the comments as "device" are raw HIP device pointers;
 *     the library never allocates, frees or copies to the host.  Temporary
 *     storage is a caller-owned workspace whose size the matching
 *     *_workspace_bytes() query returns (align it to 256 bytes; its contents
 *     need no initialisation, no entry point reads what it has not written).
 *   - `stream` is a hipStream_t (NULL = the default stream).  Work is enqueued
 *     on it for the CALLER'S CURRENT DEVICE; nothing synchronises.
 *   - Return value: RH_OK or a negative RH_E* code; rh_last_error() gives the
 *     text for the calling thread.  Shapes are validated on the host before
 *     any launch, so a bad argument never reaches a kernel.
 *   - All tensors are dense, row-major float32 unless a comment says otherwise.
 *   - Re-entrant: no mutable global state; safe from several host threads.  (The one
 *     exception is rh_nms_static, which says so below.)
 *
 * The reference (mesnico/pytorch-retinanet) has no FFI: each entry point names
 * the Python lines it stands in for.
 */
#ifndef RETINAHEAD_H
#define RETINAHEAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_VERSION 100 /* 0.1.0 */

typedef void* rh_stream_t; /* hipStream_t */

enum rh_status {
  RH_OK = 0,
  RH_EINVAL = -1,     /* null pointer / bad scalar argument            */
  RH_ESHAPE = -2,     /* unsupported or inconsistent shape             */
  RH_ELAUNCH = -3,    /* HIP refused a launch                          */
  RH_EWORKSPACE = -4, /* workspace missing or too small                */
  RH_EALIGN = -5      /* pointer not aligned as the entry point needs  */
};

/* Per-anchor assignment states written by rh_assign (losses.py:70-72). */
enum rh_anchor_state { RH_NEGATIVE = 0, RH_IGNORE = 1, RH_POSITIVE = 2 };

/* Bits of the per-image device status words written by the loss entry points. */
enum rh_status_bits {
  RH_STATUS_LABEL_RANGE = 1,  /* a valid GT row has a label outside [0, C): the
                                 reference raises IndexError at losses.py:79; here
                                 the anchor is positive with an all-zero target row */
  RH_STATUS_NONFINITE_BOX = 2, /* a valid GT row has a NaN or infinite coordinate.  In the
                                 reference the NaN IoU wins torch.max (losses.py:61), so the
                                 anchor is neither below 0.4 nor at or above 0.5 and is
                                 ignored; here comparisons skip the non-finite pair and the
                                 anchor is assigned from its other boxes: results for such
                                  an image are not the reference's */
  RH_STATUS_REDUCTION_GAVE_UP = 4 /* the reduction inside the stream launch (RH_STAGE_STREAM |
                                 RH_STAGE_FINALIZE in one call) stopped waiting for a record of
                                 this image after about two seconds; the call's losses are NaN.
                                 Not expected to happen: it means a workgroup of the launch never
                                 ran or the workspace was written by someone else meanwhile */
};

int rh_version(void);
/* Copies the calling thread's last error text (NUL-terminated, truncated to cap);
 * returns the untruncated length. */
size_t rh_last_error(char* buf, size_t cap);

/* ------------------------------------------------------------------------- *
 * Anchors  -  replaces Anchors.forward, models/retinanet/anchors.py:22-41
 * (generate_anchors :48-79, shift :115-135).
 * ------------------------------------------------------------------------- */
#define RH_MAX_LEVELS 8
#define RH_MAX_RATIOS 8
#define RH_MAX_SCALES 8
typedef struct rh_anchor_cfg {
  int32_t num_levels;
  int32_t levels[RH_MAX_LEVELS];  /* pyramid level l: feature map = ceil(H/2^l) x ceil(W/2^l) */
  double strides[RH_MAX_LEVELS];  /* default 2^l       (anchors.py:14) */
  double sizes[RH_MAX_LEVELS];    /* default 2^(l+2)   (anchors.py:16) */
  int32_t num_ratios;
  double ratios[RH_MAX_RATIOS];   /* default 0.5, 1, 2 (anchors.py:18) */
  int32_t num_scales;
  double scales[RH_MAX_SCALES];   /* default 2^0, 2^(1/3), 2^(2/3) (anchors.py:20) */
} rh_anchor_cfg;

/* Fills *cfg with the reference defaults (levels 3..7). */
void rh_anchor_cfg_default(rh_anchor_cfg* cfg);
/* Number of anchors N for an H x W image, or a negative RH_E* code. */
int64_t rh_anchor_count(const rh_anchor_cfg* cfg, int32_t height, int32_t width);
/* out: device [N,4] (x1,y1,x2,y2), N = rh_anchor_count(...).  Values are formed
 * in float64 and narrowed once, exactly like the numpy reference. */
int rh_anchors_f32(const rh_anchor_cfg* cfg, int32_t height, int32_t width, float* out,
                   int64_t n, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * IoU matrix  -  replaces calc_iou, models/retinanet/losses.py:6-23.
 * a: device [n,4], b: device [m,4], out: device [n,m].
 * ------------------------------------------------------------------------- */
int rh_calc_iou_f32(const float* a, int64_t n, const float* b, int64_t m, float* out,
                    rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Anchor assignment  -  replaces losses.py:48-49 (GT filter), :59-61 (IoU,
 * max/argmax over GT, first index on ties), :70-79 (thresholds, one-hot label).
 *   anchors      device [N,4]
 *   annotations  device [B,M,5] (x1,y1,x2,y2,label), rows with label == -1 are padding
 *   state        device u8 [B,N]   rh_anchor_state; an image without valid GT is all RH_IGNORE
 *   argmax       device i32 [B,N]  index into the image's valid-GT list (order kept)
 *   label        device i32 [B,N]  class of positives, -1 otherwise
 *   iou_max      device f32 [B,N]
 *   num_pos      device i32 [B]
 * Any of state/argmax/label/iou_max may be NULL.  status: device i32 [B] (rh_status_bits per
 * image, written by every call) or NULL.
 * ------------------------------------------------------------------------- */
size_t rh_assign_workspace_bytes(int32_t batch, int64_t n, int32_t m);
int rh_assign(const float* anchors, const float* annotations, int32_t batch, int64_t n, int32_t m,
              int32_t num_classes, float neg_thresh, float pos_thresh, uint8_t* state,
              int32_t* argmax, int32_t* label, float* iou_max, int32_t* num_pos, int32_t* status,
              void* workspace, size_t workspace_bytes, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Fused dense-head loss, forward + backward  -  replaces FocalLoss.forward,
 * models/retinanet/losses.py:29-141, and the autograd graph it records.
 *   classifications device [B,N,C]  post-sigmoid probabilities (model.py:177,193)
 *   regressions     device [B,N,4]
 *   anchors         device [N,4]    (the reference's [1,N,4])
 *   annotations     device [B,M,5]
 *   d_classifications device [B,N,C], d_regressions device [B,N,4]:
 *       gradients of grad_cls*classification_loss + grad_reg*regression_loss
 *   losses          device f32 [5]  {classification_loss, regression_loss} (batch means, :139-141),
 *                                   then the call's positive anchors and its images as floats
 *                                   ([0..3] is the vector data-parallel ranks sum), then
 *                                   classification_loss + regression_loss in float32 (the total
 *                                   train.py:185 forms)
 *   per_image       device f32 [2,B] or NULL  (per-image losses, :94 and :134)
 *   num_pos         device i32 [B] or NULL
 *   status          device i32 [B] or NULL  (rh_status_bits per image, written by every call)
 * batch_divisor: number of images the batch mean runs over; 0 means B.  A
 * data-parallel shard passes the global batch size.
 *
 * normalizer = RH_NORM_GLOBAL replaces the reference's per-image normalisers (losses.py:94,
 * :134) and batch mean (:139-141) by one positive count P for the whole batch:
 *   classification_loss = sum over images of the focal sums / max(P, 1),
 *   regression_loss     = sum over images of the smooth-L1 sums / (4 P)   (0 when P == 0),
 * per_image[.., j] is image j's share of those sums, batch_divisor is not used.  The reference
 * has no such mode; with equal positive counts in every image both modes agree.  P is
 * *num_pos_total when that pointer is given, else the positives of this call's images, counted
 * on the device in the RH_STAGE_ASSIGN stage (which then also writes num_pos).  A data-parallel
 * shard runs RH_STAGE_ASSIGN, sums num_pos over all ranks, and passes the result as
 * num_pos_total to the remaining stages; its losses are then its share of the global ones.
 * ------------------------------------------------------------------------- */
enum rh_normalizer { RH_NORM_PER_IMAGE = 0, RH_NORM_GLOBAL = 1 };

typedef struct rh_focal_params {
  float alpha;          /* 0.25  losses.py:30 (gamma is fixed at 2, :31) */
  float neg_thresh;     /* 0.4   losses.py:70 */
  float pos_thresh;     /* 0.5   losses.py:72 */
  float grad_cls;       /* upstream gradient of classification_loss */
  float grad_reg;       /* upstream gradient of regression_loss     */
  int32_t batch_divisor;
  int32_t from_logits;  /* 0: classifications are probabilities, the reference's signature.
                           1: they are logits; the head's nn.Sigmoid (model.py:177, 193) is folded
                           into the stream, d_classifications is the gradient with respect to the
                           logits (dL/dp * p * (1 - p), as autograd through torch.sigmoid). */
  int32_t normalizer;   /* rh_normalizer; RH_NORM_PER_IMAGE is the reference */
  const float* num_pos_total; /* device f32 [1] or NULL; RH_NORM_GLOBAL only */
} rh_focal_params;

void rh_focal_params_default(rh_focal_params* p);

/* Stages of the fused call; RH_STAGE_ALL runs the whole path.  The stages of one step share its
 * workspace and may be issued by separate calls.  RH_STAGE_ASSIGN alone reads only anchors and
 * annotations (the prediction, gradient and loss pointers may be NULL; status and, under
 * RH_NORM_GLOBAL, num_pos are written), so a trainer that has the next batch's annotations early
 * can run it on a second stream, into a second workspace, while the current step's
 * RH_STAGE_STREAM is still running: the stream stage leaves the vector ALUs mostly idle.  The
 * caller orders the streams (the later stages of a step after its assignment stage; a
 * workspace's next assignment after the stages that still read it).
 * RH_STAGE_STREAM | RH_STAGE_FINALIZE in one call reduce inside the stream launch (no further
 * kernel); run apart they give the same bits.  Either way the workspace must have had its
 * RH_STAGE_ASSIGN for these annotations since it was allocated or last used for anything else:
 * that is the only requirement on its contents.  The workspace needs no initialisation: the
 * reduction inside the launch tells the records of the current call from older bytes by a 32-bit
 * tag that is never zero and changes with every call, and the assignment stage clears the tags of
 * every record the call will poll, so nothing the memory held before can pass for one. */
enum rh_focal_stage {
  RH_STAGE_ASSIGN = 1,   /* GT compaction + IoU assignment + num_pos            */
  RH_STAGE_STREAM = 2,   /* N x C focal stream + smooth-L1, losses and gradients */
  RH_STAGE_FINALIZE = 4, /* deterministic reduction of the per-tile partial sums */
  RH_STAGE_ALL = 7
};

size_t rh_focal_loss_workspace_bytes(int32_t batch, int64_t n, int32_t num_classes, int32_t m);
int rh_focal_loss_fwd_bwd(const float* classifications, const float* regressions,
                          const float* anchors, const float* annotations, int32_t batch, int64_t n,
                          int32_t num_classes, int32_t m, const rh_focal_params* params,
                          float* d_classifications, float* d_regressions, float* losses,
                          float* per_image, int32_t* num_pos, int32_t* status, void* workspace,
                          size_t workspace_bytes, int32_t stages, rh_stream_t stream);

/* The same call, which also runs the RH_STAGE_ASSIGN stage of the NEXT step (losses.py:57-79 for
 * the next batch's annotations: same anchors, B, N and class count; its own row count m) inside
 * this step's stream launch, where it costs about a fifth of a launch of its own.  The next step
 * is then called with stages = RH_STAGE_STREAM | RH_STAGE_FINALIZE, next->workspace as its
 * workspace and next->status / next->num_pos as the outputs the assignment wrote (num_pos only
 * under RH_NORM_GLOBAL; either may be NULL).  `stages` must include RH_STAGE_STREAM; next == NULL
 * is rh_focal_loss_fwd_bwd.  The reference has no counterpart: its assignment is part of the
 * loss's forward, losses.py:57-79, and runs when the predictions arrive. */
typedef struct rh_assign_ahead {
  const float* annotations;  /* device [B,m,5], the next step's */
  int32_t m;
  void* workspace;           /* the next step's workspace, not this step's */
  size_t workspace_bytes;    /* >= rh_focal_loss_workspace_bytes(batch, n, num_classes, m) */
  int32_t* status;           /* device i32 [B] or NULL */
  int32_t* num_pos;          /* device i32 [B] or NULL */
} rh_assign_ahead;

int rh_focal_loss_fwd_bwd_ahead(const float* classifications, const float* regressions,
                                const float* anchors, const float* annotations, int32_t batch,
                                int64_t n, int32_t num_classes, int32_t m,
                                const rh_focal_params* params, float* d_classifications,
                                float* d_regressions, float* losses, float* per_image,
                                int32_t* num_pos, int32_t* status, void* workspace,
                                size_t workspace_bytes, int32_t stages, const rh_assign_ahead* next,
                                rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * The same fused loss, reading the heads' outputs where they lie: per pyramid level l a
 * classification tensor [B, A*C, H_l, W_l] and a regression tensor [B, A*4, H_l, W_l] (the
 * convolutions' NCHW outputs), instead of the flat [B,N,C] / [B,N,4] tensors the reference
 * builds from them by permute + contiguous + view + cat
 * (models/retinanet/model.py:152-154, 196-202, 311-313).  Anchor n = off_l + (y*W_l + x)*A + a
 * is channel group a of cell (y, x) of level l; N = A * sum_l H_l*W_l must equal the number of
 * anchor rows.  Gradients are written in the layout of the inputs.
 *   cls_levels, reg_levels, d_cls_levels, d_reg_levels: HOST arrays of num_levels device pointers
 *   heights, widths: HOST arrays [num_levels]; num_levels <= 8
 * Everything else as rh_focal_loss_fwd_bwd (params->from_logits included).
 * ------------------------------------------------------------------------- */
size_t rh_focal_loss_levels_workspace_bytes(int32_t batch, int32_t num_levels, const int32_t* heights,
                                            const int32_t* widths, int32_t anchors_per_cell,
                                            int32_t m);
int rh_focal_loss_levels_fwd_bwd(const float* const* cls_levels, const float* const* reg_levels,
                                 const float* anchors, const float* annotations, int32_t batch,
                                 int32_t num_levels, const int32_t* heights, const int32_t* widths,
                                 int32_t anchors_per_cell, int32_t num_classes, int32_t m,
                                 const rh_focal_params* params, float* const* d_cls_levels,
                                 float* const* d_reg_levels, float* losses, float* per_image,
                                 int32_t* num_pos, int32_t* status, void* workspace,
                                 size_t workspace_bytes, int32_t stages, rh_stream_t stream);

/* Applies upstream gradients to the stored gradients without a host read:
 * x *= (*scale_x) / assumed and y *= (*scale_y) / assumed, in place, one launch.  An array whose
 * scale equals `assumed` is not touched (no memory traffic); a NULL scale pointer zeroes the
 * array (that loss took no part in the backward pass); a NULL array is skipped.  scale_x /
 * scale_y are device f32 [1].  Used by the autograd wrapper for train.py:185-190. */
int rh_apply_upstream_f32(float* x, int64_t nx, const float* scale_x, float* y, int64_t ny,
                          const float* scale_y, float assumed, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Generic elementwise focal BCE, forward + backward  -  replaces
 * FocalLoss.forward, models/focal_loss.py:14-25.
 *   inputs, targets device [count]; loss_elem device [count] or NULL;
 *   loss_mean device f32 [1] or NULL; grad device [count] (d/d inputs of the
 *   mean when reduce != 0, of the sum of elements otherwise).
 * ------------------------------------------------------------------------- */
size_t rh_generic_focal_workspace_bytes(int64_t count);
int rh_generic_focal_fwd_bwd(const float* inputs, const float* targets, int64_t count, float alpha,
                             float gamma, int32_t logits, int32_t reduce, float* loss_elem,
                             float* loss_mean, float* grad, void* workspace,
                             size_t workspace_bytes, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Detection-to-annotation matching for evaluation  -  the inner loop of evaluate,
 * csv_eval.py:188-208, with compute_overlap (csv_eval.py:11-35) in float64.
 *   det_boxes device f32 [B,D,4] and det_labels device i64 [B,D]: each image's detections in
 *   descending score order (the order rh_postprocess writes); det_count device i32 [B];
 *   annotations device f32 [B,M,5] (label -1 rows are padding);
 *   tp device u8 [B,D]: 1 = true positive, 0 = false positive (0 beyond det_count).
 * ------------------------------------------------------------------------- */
int rh_match_detections(const float* det_boxes, const int64_t* det_labels, const int32_t* det_count,
                        const float* annotations, int32_t batch, int32_t d, int32_t m,
                        double iou_threshold, uint8_t* tp, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Annotation collation on the device  -  the padding step of collater,
 * models/retinanet/model.py:43-55: out[b, m, :] = (box, label) of image b's m-th annotation
 * for m < count_b, five -1 otherwise.
 *   boxes device [T,4], labels device f32 [T] (all images back to back; either may be NULL when
 *   T == 0), offsets device i32 [B+1] (prefix sums of the counts, offsets[0] = 0),
 *   out device [B, m, 5] with m >= max count and m >= 1.
 * ------------------------------------------------------------------------- */
int rh_collate_annotations_f32(const float* boxes, const float* labels, const int32_t* offsets,
                               int32_t batch, int32_t m, float* out, rh_stream_t stream);

/* The same with the datasets' annotation filter folded in (csv_dataset.py:129-136,
 * coco_dataset.py:97-108): a row whose width or height is below min_size is dropped (the
 * reference uses 1; a NaN extent is kept, as there), rows given as [x, y, w, h] (xywh != 0, the
 * COCO form, width / height tested as stored) become corners by x + w, y + h; survivors keep their
 * order, the rest of out[b] is -1.  kept: device i32 [B] or NULL, rows kept per image (out needs
 * m >= the largest of them; m >= the largest raw count always suffices). */
int rh_sanitize_collate_annotations_f32(const float* boxes, const float* labels,
                                        const int32_t* offsets, int32_t batch, int32_t m,
                                        float min_size, int32_t xywh, float* out, int32_t* kept,
                                        rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Box decode and clip  -  replace BBoxTransform.forward (utils.py:96-120) and
 * ClipBoxes.forward (utils.py:133-142).
 *   boxes device [N,4] (broadcast over the batch), deltas device [B,N,4],
 *   out device [B,N,4]; mean/std: host float[4] (NULL = 0,0,0,0 / .1,.1,.2,.2).
 *   rh_clip_boxes_f32 works in place: x1,y1 >= 0, x2 <= width, y2 <= height.
 * ------------------------------------------------------------------------- */
int rh_bbox_transform_f32(const float* boxes, const float* deltas, int32_t batch, int64_t n,
                          const float* mean, const float* std, float* out, rh_stream_t stream);
int rh_clip_boxes_f32(float* boxes, int64_t rows, float width, float height, rh_stream_t stream);

/* out[i] = float32 sigmoid(x[i]), the classification head's output activation
 * (models/retinanet/model.py:177, 193), bit for bit the probability the logits-in modes of
 * rh_focal_loss_fwd_bwd and rh_postprocess derive from a logit.  x, out device [count]. */
int rh_sigmoid_f32(const float* x, float* out, int64_t count, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * NMS  -  replaces torchvision.ops.nms as called at model.py:336 (greedy, drop
 * when IoU > iou_threshold, ties in score keep the lower index first).
 *   boxes device [K,4], scores device [K], labels device i32 [K] or NULL
 *   (non-NULL: only equal labels suppress each other),
 *   keep device i64 [K] (first *keep_count entries valid, descending score),
 *   keep_count device i32 [1].
 * ------------------------------------------------------------------------- */
size_t rh_nms_workspace_bytes(int64_t k);
int rh_nms(const float* boxes, const float* scores, const int32_t* labels, int64_t k,
           float iou_threshold, int64_t* keep, int32_t* keep_count, void* workspace,
           size_t workspace_bytes, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * rh_nms without a workspace argument and without any allocation.  It has no counterpart
 * in the reference; it exists for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Arguments and
 * results are those of rh_nms (keep lists are identical), for k <= RH_NMS_STATIC_MAX_BOXES;
 * a larger k is refused with RH_ESHAPE.
 * The temporaries live in a fixed device array of 3 MiB inside the library, one copy per
 * GPU the library runs on, held until the process ends.  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on
 * different GPUs do not.  The first call on each device looks the array's address up
 * (hipGetSymbolAddress) and should therefore be made outside stream capture; later calls
 * only enqueue kernels.
 * ------------------------------------------------------------------------- */
#define RH_NMS_STATIC_MAX_BOXES 4096
int rh_nms_static(const float* boxes, const float* scores, const int32_t* labels, int64_t k,
                  float iou_threshold, int64_t* keep, int32_t* keep_count, rh_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Inference post-processing for a batch  -  replaces model.py:320-340 (which
 * handles image 0 only) for every image of the batch.
 *   classifications device [B,N,C], regressions device [B,N,4], anchors device [N,4]
 *   cand_cap  : capacity of the per-image candidate list (anchors with
 *               max-class score > score_thresh).  Candidates beyond it are
 *               dropped and counted in overflow[b]; cand_cap = N can never overflow.
 *   per_class : 0 = class-agnostic NMS (the reference), 1 = only equal labels suppress
 *   pre_nms_topk : > 0 keeps only the best-scoring candidates before NMS
 *   max_det   : > 0 truncates each image's kept list
 *   out_scores device f32 [B,out_cap], out_labels device i64 [B,out_cap],
 *   out_boxes device f32 [B,out_cap,4], out_count device i32 [B],
 *   cand_count device i32 [B] or NULL, overflow device i32 [B] or NULL.
 *   out_cap = max_det if max_det > 0, else min(cand_cap, pre_nms_topk or cand_cap).
 * ------------------------------------------------------------------------- */
typedef struct rh_post_params {
  float height, width;   /* image size used by the clip (utils.py:134) */
  float score_thresh;    /* 0.05 model.py:325 */
  float iou_thresh;      /* 0.5  model.py:336 */
  int32_t per_class;
  int32_t pre_nms_topk;
  int32_t max_det;
  int32_t cand_cap;
  int32_t from_logits;   /* 1: classifications are the head's logits; scores are their float32
                            sigmoid (model.py:177, 193) as rh_sigmoid_f32 computes it, and the
                            result equals the probability-mode result for those probabilities */
} rh_post_params;

size_t rh_postprocess_workspace_bytes(int32_t batch, int64_t n, int32_t num_classes,
                                      const rh_post_params* params);
int rh_postprocess(const float* classifications, const float* regressions, const float* anchors,
                   int32_t batch, int64_t n, int32_t num_classes, const rh_post_params* params,
                   float* out_scores, int64_t* out_labels, float* out_boxes, int32_t* out_count,
                   int32_t* cand_count, int32_t* overflow, void* workspace,
                   size_t workspace_bytes, int32_t stages, rh_stream_t stream);

/* The same post-processing on the heads' own outputs: per pyramid level l a classification tensor
 * [B, A*C, H_l, W_l] and a regression tensor [B, A*4, H_l, W_l], instead of the flat tensors the
 * reference builds from them (models/retinanet/model.py:152-154, 196-202, 311-313).  Results are
 * identical to rh_postprocess on the flattened tensors.  cls_levels / reg_levels, heights, widths:
 * HOST arrays [num_levels <= 8]; anchors device [N,4], N = A * sum_l H_l*W_l; everything else as
 * rh_postprocess. */
size_t rh_postprocess_levels_workspace_bytes(int32_t batch, int32_t num_levels,
                                             const int32_t* heights, const int32_t* widths,
                                             int32_t anchors_per_cell, int32_t num_classes,
                                             const rh_post_params* params);
int rh_postprocess_levels(const float* const* cls_levels, const float* const* reg_levels,
                          const float* anchors, int32_t batch, int32_t num_levels,
                          const int32_t* heights, const int32_t* widths, int32_t anchors_per_cell,
                          int32_t num_classes, const rh_post_params* params, float* out_scores,
                          int64_t* out_labels, float* out_boxes, int32_t* out_count,
                          int32_t* cand_count, int32_t* overflow, void* workspace,
                          size_t workspace_bytes, int32_t stages, rh_stream_t stream);

enum rh_post_stage {
  RH_POST_DECODE = 1, /* class max + threshold + decode + clip + compaction */
  RH_POST_SORT = 2,   /* per-image sort by (score desc, anchor index asc)   */
  RH_POST_NMS = 4,    /* overlapping pairs + greedy resolve + output gather */
  RH_POST_ALL = 7
};

#ifdef __cplusplus
}
#endif
#endif /* RETINAHEAD_H */
