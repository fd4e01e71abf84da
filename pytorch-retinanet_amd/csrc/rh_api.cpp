// Version, error text and parameter defaults of libretinahead.so.
#include <string.h>
#include "rh_common.h"

namespace rh {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace rh

extern "C" {

int rh_version(void) { return RH_VERSION; }

size_t rh_last_error(char* buf, size_t cap) {
  size_t len = strlen(rh::g_error);
  if (buf && cap) {
    size_t n = len < cap - 1 ? len : cap - 1;
    memcpy(buf, rh::g_error, n);
    buf[n] = 0;
  }
  return len;
}

void rh_focal_params_default(rh_focal_params* p) {
  if (!p) return;
  p->alpha = 0.25f;
  p->neg_thresh = 0.4f;
  p->pos_thresh = 0.5f;
  p->grad_cls = 1.0f;
  p->grad_reg = 1.0f;
  p->batch_divisor = 0;
  p->from_logits = 0;
  p->normalizer = RH_NORM_PER_IMAGE;
  p->num_pos_total = NULL;
}

void rh_anchor_cfg_default(rh_anchor_cfg* cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof(*cfg));
  cfg->num_levels = 5;
  for (int i = 0; i < 5; ++i) {
    int l = 3 + i;
    cfg->levels[i] = l;
    cfg->strides[i] = (double)(1 << l);
    cfg->sizes[i] = (double)(1 << (l + 2));
  }
  cfg->num_ratios = 3;
  cfg->ratios[0] = 0.5;
  cfg->ratios[1] = 1.0;
  cfg->ratios[2] = 2.0;
  cfg->num_scales = 3;
  // 2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0) as Python evaluates them (IEEE pow in double)
  cfg->scales[0] = 1.0;
  cfg->scales[1] = 1.2599210498948732;
  cfg->scales[2] = 1.5874010519681994;
}

}  // extern "C"
