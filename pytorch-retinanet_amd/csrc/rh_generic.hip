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

}  // extern "C"
