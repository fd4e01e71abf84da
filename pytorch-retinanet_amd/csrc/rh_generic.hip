// Generic elementwise focal BCE on gfx950, forward + backward in one pass.
//
// Reference behaviour: models/focal_loss.py:14-25 (BCE on logits or probabilities,
// pt = exp(-BCE), F = alpha (1-pt)^gamma BCE, mean or per-element output).
#include "rh_common.h"

namespace rh {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void generic_focal_kernel(
    const float* __restrict__ x, const float* __restrict__ t, long long count, float alpha,
    float gamma, int logits, float grad_scale, float* __restrict__ loss_elem,
    float* __restrict__ grad, double* __restrict__ partials) {
  __shared__ double s_red[kThreads / 64];
  double acc = 0.0;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count; i += stride) {
    const float xv = x[i], tv = t[i];
    float bce, dbce;
    if (logits) {
      // binary_cross_entropy_with_logits: max(x,0) - x t + log1p(exp(-|x|))
      bce = (fmaxf(xv, 0.0f) - xv * tv) + log1pf(expf(-fabsf(xv)));
      dbce = 1.0f / (1.0f + expf(-xv)) - tv;
    } else {
      // binary_cross_entropy clamps both logs at -100; its backward divides by max((1-x)x, 1e-12)
      const float lp = fmaxf(logf(xv), -100.0f), lq = fmaxf(logf(1.0f - xv), -100.0f);
      bce = -(tv * lp + (1.0f - tv) * lq);
      dbce = (xv - tv) / fmaxf((1.0f - xv) * xv, 1e-12f);
    }
    const float pt = expf(-bce);
    const float om = 1.0f - pt;
    float omg, omg1;   // om^gamma, om^(gamma-1)
    if (gamma == 2.0f) {
      omg = om * om;
      omg1 = om;
    } else {
      omg = powf(om, gamma);
      omg1 = powf(om, gamma - 1.0f);
    }
    const float f = alpha * omg * bce;
    const float df = alpha * (gamma * omg1 * pt * bce + omg);
    if (loss_elem) loss_elem[i] = f;
    grad[i] = df * dbce * grad_scale;
    acc += (double)f;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(64) void generic_focal_mean_kernel(const double* __restrict__ partials,
                                                                int blocks, long long count,
                                                                float* __restrict__ loss_mean) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 64) acc += partials[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) *loss_mean = (float)(acc / (double)count);
}

// Pads per-image annotation lists to the [B, M, 5] layout the loss reads: row m of image b is
// (x1, y1, x2, y2, label) of its m-th annotation, or five -1 beyond the image's count
// (models/retinanet/model.py:43-55).  boxes [T,4] and labels [T] hold all images' annotations
// back to back; offsets [B+1] are the prefix sums of the counts.
__global__ __launch_bounds__(kThreads) void collate_annotations_kernel(
    const float* __restrict__ boxes, const float* __restrict__ labels,
    const int* __restrict__ offsets, float* __restrict__ out, int B, int M) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;   // one output row
  if (i >= (long long)B * M) return;
  const int b = (int)(i / M), m = (int)(i - (long long)b * M);
  const int lo = offsets[b], count = offsets[b + 1] - lo;
  float r[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
  if (m < count) {
    const size_t t = (size_t)lo + m;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = boxes[t * 4 + k];
    r[4] = labels[t];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) out[i * 5 + k] = r[k];
}

// Collation with the datasets' annotation filter folded in: rows narrower or lower than
// min_size are dropped (csv_dataset.py:129-136 on x2 - x1 / y2 - y1, coco_dataset.py:97-99 on the
// stored width / height; a NaN extent is kept, as `nan < 1` is false there), [x, y, w, h] rows
// become corners by x + w, y + h (coco_dataset.py:106-108), the survivors keep their order.
// One wave per image, 64 rows at a time, positions from a ballot prefix count.
__global__ __launch_bounds__(64) void sanitize_collate_kernel(
    const float* __restrict__ boxes, const float* __restrict__ labels,
    const int* __restrict__ offsets, float* __restrict__ out, int* __restrict__ kept, int M,
    float min_size, int xywh) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int lo = offsets[b], count = offsets[b + 1] - lo;
  float* ob = out + (size_t)b * M * 5;
  int cnt = 0;
  for (int m0 = 0; m0 < count; m0 += 64) {
    const int m = m0 + lane;
    const bool have = m < count;
    const size_t t = (size_t)lo + (have ? m : count - 1);
    float x1 = boxes[t * 4 + 0], y1 = boxes[t * 4 + 1], x2 = boxes[t * 4 + 2], y2 = boxes[t * 4 + 3];
    const float lab = labels[t];
    float w, h;
    if (xywh) {
      w = x2;
      h = y2;
      x2 = x1 + w;
      y2 = y1 + h;
    } else {
      w = x2 - x1;
      h = y2 - y1;
    }
    const bool keep = have && !(w < min_size || h < min_size);
    const unsigned long long km = __ballot(keep);
    const int at = cnt + __popcll(km & ((1ull << lane) - 1ull));
    if (keep && at < M) {
      float* o = ob + (size_t)at * 5;
      o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = lab;
    }
    cnt += __popcll(km);
  }
  for (int i = min(cnt, M) * 5 + lane; i < M * 5; i += 64) ob[i] = -1.0f;
  if (kept && lane == 0) kept[b] = cnt;
}

// Greedy matching of one image's detections to its annotations for detection evaluation
// (csv_eval.py:188-208).  Detections come in descending score order; a detection is a true
// positive when its best-overlapping annotation of the same class (first on ties) reaches the
// IoU threshold and no earlier detection has claimed it.  IoU in float64 with the operation
// order of csv_eval.compute_overlap (csv_eval.py:21-35).  One wave per image: lanes share the
// annotations, detections are walked in order.
__global__ __launch_bounds__(64) void match_detections_kernel(
    const float4* __restrict__ det_boxes, const long long* __restrict__ det_labels,
    const int* __restrict__ det_count, const float* __restrict__ ann, unsigned char* __restrict__ tp,
    int D, int M, double thresh) {
  extern __shared__ unsigned char s_taken[];
  const int j = blockIdx.x, lane = threadIdx.x;
  for (int m = lane; m < M; m += 64) s_taken[m] = 0;
  __syncthreads();
  const float* aj = ann + (size_t)j * M * 5;
  const int count = min(det_count[j], D);
  for (int d = 0; d < D; ++d) {
    unsigned char flag = 0;
    if (d < count) {
      const float4 bf = det_boxes[(size_t)j * D + d];
      const double x1 = bf.x, y1 = bf.y, x2 = bf.z, y2 = bf.w;
      const double lab = (double)det_labels[(size_t)j * D + d];
      const double area_d = (x2 - x1) * (y2 - y1);
      double best = -1.0;
      int best_m = M;
      for (int m = lane; m < M; m += 64) {
        const double al = aj[m * 5 + 4];
        if (al == -1.0 || al != lab) continue;
        const double ax1 = aj[m * 5 + 0], ay1 = aj[m * 5 + 1], ax2 = aj[m * 5 + 2], ay2 = aj[m * 5 + 3];
        const double area = (ax2 - ax1) * (ay2 - ay1);
        double iw = fmin(x2, ax2) - fmax(x1, ax1);
        double ih = fmin(y2, ay2) - fmax(y1, ay1);
        iw = fmax(iw, 0.0);
        ih = fmax(ih, 0.0);
        double ua = area_d + area - iw * ih;
        ua = fmax(ua, 2.220446049250313e-16);
        const double iou = iw * ih / ua;
        if (iou > best) {
          best = iou;
          best_m = m;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int om = __shfl_xor(best_m, o, 64);
        if (ob > best || (ob == best && om < best_m)) {
          best = ob;
          best_m = om;
        }
      }
      if (best_m < M && best >= thresh && !s_taken[best_m]) {
        flag = 1;
        if (lane == 0) s_taken[best_m] = 1;
      }
      __syncthreads();   // the claim is visible before the next detection looks
    }
    if (lane == 0) tp[(size_t)j * D + d] = flag;
  }
}

}  // namespace rh

using namespace rh;

extern "C" {

size_t rh_generic_focal_workspace_bytes(int64_t count) {
  (void)count;
  return 2048 * sizeof(double);
}

int rh_generic_focal_fwd_bwd(const float* inputs, const float* targets, int64_t count, float alpha,
                             float gamma, int32_t logits, int32_t reduce, float* loss_elem,
                             float* loss_mean, float* grad, void* workspace,
                             size_t workspace_bytes, rh_stream_t stream) {
  RH_REQUIRE(count >= 1, RH_ESHAPE, "count=%lld", (long long)count);
  RH_REQUIRE(inputs && targets && grad, RH_EINVAL, "null pointer");
  RH_REQUIRE(!reduce || loss_mean, RH_EINVAL, "reduce needs loss_mean");
  RH_REQUIRE(workspace && workspace_bytes >= rh_generic_focal_workspace_bytes(count), RH_EWORKSPACE,
             "workspace too small: need %zu bytes", rh_generic_focal_workspace_bytes(count));
  long long blocks = (count + kThreads - 1) / kThreads;
  if (blocks > 2048) blocks = 2048;
  double* partials = static_cast<double*>(workspace);
  const float gs = reduce ? (float)(1.0 / (double)count) : 1.0f;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(generic_focal_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, inputs,
                     targets, (long long)count, alpha, gamma, logits, gs, loss_elem, grad, partials);
  RH_CHECK_LAUNCH("generic_focal_kernel");
  if (loss_mean) {
    hipLaunchKernelGGL(generic_focal_mean_kernel, dim3(1), dim3(64), 0, s, partials, (int)blocks,
                       (long long)count, loss_mean);
    RH_CHECK_LAUNCH("generic_focal_mean_kernel");
  }
  return RH_OK;
}

int rh_collate_annotations_f32(const float* boxes, const float* labels, const int32_t* offsets,
                               int32_t batch, int32_t m, float* out, rh_stream_t stream) {
  RH_REQUIRE(batch >= 1 && m >= 1, RH_ESHAPE, "bad shape batch=%d m=%d", batch, m);
  RH_REQUIRE(offsets && out, RH_EINVAL, "null pointer");
  const long long rows = (long long)batch * m;
  hipLaunchKernelGGL(collate_annotations_kernel, dim3((unsigned)((rows + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, boxes, labels, offsets, out, batch, m);
  RH_CHECK_LAUNCH("collate_annotations_kernel");
  return RH_OK;
}

int rh_sanitize_collate_annotations_f32(const float* boxes, const float* labels,
                                        const int32_t* offsets, int32_t batch, int32_t m,
                                        float min_size, int32_t xywh, float* out, int32_t* kept,
                                        rh_stream_t stream) {
  RH_REQUIRE(batch >= 1 && m >= 1, RH_ESHAPE, "bad shape batch=%d m=%d", batch, m);
  RH_REQUIRE(offsets && out, RH_EINVAL, "null pointer");
  RH_REQUIRE(min_size == min_size, RH_EINVAL, "min_size is NaN");
  hipLaunchKernelGGL(sanitize_collate_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, boxes,
                     labels, offsets, out, kept, m, min_size, xywh ? 1 : 0);
  RH_CHECK_LAUNCH("sanitize_collate_kernel");
  return RH_OK;
}

int rh_match_detections(const float* det_boxes, const int64_t* det_labels, const int32_t* det_count,
                        const float* annotations, int32_t batch, int32_t d, int32_t m,
                        double iou_threshold, uint8_t* tp, rh_stream_t stream) {
  RH_REQUIRE(batch >= 1 && d >= 0 && m >= 0, RH_ESHAPE, "bad shape batch=%d d=%d m=%d", batch, d, m);
  if (d == 0) return RH_OK;
  RH_REQUIRE(det_boxes && det_labels && det_count && tp && (annotations || m == 0), RH_EINVAL,
             "null pointer");
  RH_REQUIRE(aligned16(det_boxes), RH_EALIGN, "det_boxes must be 16-byte aligned");
  RH_REQUIRE(m <= 64 * 1024, RH_ESHAPE, "m=%d: at most 65536 annotation rows per image", m);
  const size_t lds = (size_t)(m > 0 ? m : 1);
  // (always the kernel's ceiling, m's limit above: a per-launch value could be lowered by another
  // host thread between this call and the launch)
  if (lds > 48 * 1024)
    RH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(match_detections_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipLaunchKernelGGL(match_detections_kernel, dim3(batch), dim3(64), lds, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(det_boxes),
                     reinterpret_cast<const long long*>(det_labels), det_count, annotations, tp, d, m,
                     iou_threshold);
  RH_CHECK_LAUNCH("match_detections_kernel");
  return RH_OK;
}

}  // extern "C"
