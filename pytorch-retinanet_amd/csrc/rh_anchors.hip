// Anchor generation on gfx950.
//
// Reference behaviour: models/retinanet/anchors.py:22-41 (levels, feature-map
// shapes, concatenation order), :48-79 (the A = ratios x scales base boxes),
// :115-135 (shift by (x + 0.5) * stride, cell-major / anchor-minor order).
// The reference computes in float64 and narrows once to float32; so does this:
// base boxes are formed on the host in double, the per-cell sum on the device in
// double, and the only rounding to float32 is the final conversion.
#include <math.h>
#include "rh_common.h"

namespace rh {

constexpr int kMaxPerCell = RH_MAX_RATIOS * RH_MAX_SCALES;

struct LevelBase {
  double box[kMaxPerCell][4];
};

__global__ __launch_bounds__(256) void anchors_level_kernel(float4* __restrict__ out, LevelBase base,
                                                            int per_cell, int fw, long long count,
                                                            double stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const long long cell = i / per_cell;
  const int a = (int)(i - cell * per_cell);
  const long long y = cell / fw;
  const long long x = cell - y * fw;
  const double sx = ((double)x + 0.5) * stride;
  const double sy = ((double)y + 0.5) * stride;
  out[i] = make_float4((float)(base.box[a][0] + sx), (float)(base.box[a][1] + sy),
                       (float)(base.box[a][2] + sx), (float)(base.box[a][3] + sy));
}

static int check_cfg(const rh_anchor_cfg* cfg, int height, int width) {
  RH_REQUIRE(cfg, RH_EINVAL, "null anchor config");
  RH_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= RH_MAX_LEVELS, RH_ESHAPE,
             "num_levels=%d out of range", cfg->num_levels);
  RH_REQUIRE(cfg->num_ratios >= 1 && cfg->num_ratios <= RH_MAX_RATIOS, RH_ESHAPE,
             "num_ratios=%d out of range", cfg->num_ratios);
  RH_REQUIRE(cfg->num_scales >= 1 && cfg->num_scales <= RH_MAX_SCALES, RH_ESHAPE,
             "num_scales=%d out of range", cfg->num_scales);
  RH_REQUIRE(height >= 0 && width >= 0, RH_ESHAPE, "negative image size %dx%d", height, width);
  for (int i = 0; i < cfg->num_levels; ++i)
    RH_REQUIRE(cfg->levels[i] >= 0 && cfg->levels[i] < 31, RH_ESHAPE, "pyramid level %d out of range",
               cfg->levels[i]);
  return RH_OK;
}

static void level_shape(int level, int height, int width, long long* fh, long long* fw) {
  const long long s = 1ll << level;   // anchors.py:26: (shape + 2^l - 1) // 2^l
  *fh = ((long long)height + s - 1) / s;
  *fw = ((long long)width + s - 1) / s;
}

}  // namespace rh

using namespace rh;

extern "C" {

int64_t rh_anchor_count(const rh_anchor_cfg* cfg, int32_t height, int32_t width) {
  if (int e = check_cfg(cfg, height, width)) return e;
  long long total = 0;
  for (int i = 0; i < cfg->num_levels; ++i) {
    long long fh, fw;
    level_shape(cfg->levels[i], height, width, &fh, &fw);
    total += fh * fw * cfg->num_ratios * cfg->num_scales;
  }
  return total;
}

int rh_anchors_f32(const rh_anchor_cfg* cfg, int32_t height, int32_t width, float* out, int64_t n,
                   rh_stream_t stream) {
  const int64_t want = rh_anchor_count(cfg, height, width);
  if (want < 0) return (int)want;
  RH_REQUIRE(n == want, RH_ESHAPE, "output holds %lld anchors, image %dx%d needs %lld", (long long)n,
             height, width, (long long)want);
  if (n == 0) return RH_OK;
  RH_REQUIRE(out, RH_EINVAL, "null output");
  RH_REQUIRE(aligned16(out), RH_EALIGN, "output must be 16-byte aligned");
  const int per_cell = cfg->num_ratios * cfg->num_scales;
  long long offset = 0;
  for (int i = 0; i < cfg->num_levels; ++i) {
    // anchors.py:63-77, ratio-major / scale-minor
    LevelBase base;
    int a = 0;
    for (int r = 0; r < cfg->num_ratios; ++r) {
      for (int s = 0; s < cfg->num_scales; ++s, ++a) {
        const double side = cfg->sizes[i] * cfg->scales[s];
        const double area = side * side;
        const double w = sqrt(area / cfg->ratios[r]);
        const double h = w * cfg->ratios[r];
        base.box[a][0] = 0.0 - w * 0.5;
        base.box[a][1] = 0.0 - h * 0.5;
        base.box[a][2] = w - w * 0.5;
        base.box[a][3] = h - h * 0.5;
      }
    }
    long long fh, fw;
    level_shape(cfg->levels[i], height, width, &fh, &fw);
    const long long count = fh * fw * per_cell;
    if (count == 0) continue;
    RH_REQUIRE(fw < (1ll << 31) && (count + 255) / 256 < (1ll << 31), RH_ESHAPE,
               "level %d too large", cfg->levels[i]);
    hipLaunchKernelGGL(anchors_level_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<float4*>(out) + offset, base, per_cell,
                       (int)fw, count, cfg->strides[i]);
    RH_CHECK_LAUNCH("anchors_level_kernel");
    offset += count;
  }
  return RH_OK;
}

}  // extern "C"
