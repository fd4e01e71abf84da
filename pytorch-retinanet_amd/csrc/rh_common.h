// Internal helpers shared by the retinahead kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "retinahead.h"

namespace rh {

constexpr int kWave = 64;

// ---- error reporting (thread-local text, C return codes) -------------------
void set_error(const char* fmt, ...);

#define RH_FAIL(code, ...)            \
  do {                                \
    ::rh::set_error(__VA_ARGS__);     \
    return (code);                    \
  } while (0)

#define RH_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) RH_FAIL(code, __VA_ARGS__); \
  } while (0)

// Checks the launch itself (configuration errors); never synchronises.
#define RH_CHECK_LAUNCH(what)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess)                                                 \
      RH_FAIL(RH_ELAUNCH, "%s: %s", (what), hipGetErrorString(e__));        \
  } while (0)

#define RH_CHECK_HIP(expr)                                                 \
  do {                                                                     \
    hipError_t e__ = (expr);                                               \
    if (e__ != hipSuccess)                                                 \
      RH_FAIL(RH_ELAUNCH, "%s: %s", #expr, hipGetErrorString(e__));         \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves consecutive 256-byte aligned regions out of a caller-provided workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += round_up(count * sizeof(T), 256);
    return r;
  }
};

// ---- wave64 reductions ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// K sums at once, step by step: the K exchanges of a step are independent and go out together
// (one wait per step instead of one per exchange); each sum is the butterfly of wave_sum().
template <int K>
__device__ __forceinline__ void wave_sum_many(double (&v)[K]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double t[K];
#pragma unroll
    for (int a = 0; a < K; ++a) t[a] = __shfl_xor(v[a], o, 64);
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += t[a];
  }
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exact x / d for 0 <= x < 2^32 / d, with magic = ceil(2^32 / d)  (d >= 2).
// Proof: magic*d = 2^32 + e with 0 <= e < d, so x*magic/2^32 = x/d + x*e/(d*2^32),
// and the extra term is below 1/d whenever x*e < 2^32, which x*d < 2^32 guarantees.
__device__ __forceinline__ uint32_t fast_div(uint32_t x, uint32_t magic) { return __umulhi(x, magic); }
inline uint32_t div_magic(uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); }

// min / max / clamp-at-zero as single instructions.  fminf / fmaxf compile to the same
// instructions wrapped in NaN-quieting moves (v_max x, x, x) on every operand, which in the
// pair loops of the assignment and NMS-mask kernels more than doubles the instruction count;
// for the finite box coordinates those loops see, the results are identical.
__device__ __forceinline__ float vmin(const float a, const float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(const float a, const float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax0(const float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
// Minimum / maximum over the wave, as a wave-uniform value: inclusive scan by DPP row shifts
// (1, 2, 4, 8 lanes inside each row of 16), then the row broadcasts 15 -> rows 1, 3 and
// 31 -> rows 2, 3; lane 63 ends up with the result.  Lanes without a source keep `ident`.
template <bool kMax>
__device__ __forceinline__ float wave_extreme(float v) {
  const int ident = __float_as_int(kMax ? -__builtin_inff() : __builtin_inff());
#define RH_DPP_STEP(ctrl, rows)                                                               \
  {                                                                                           \
    const float t = __int_as_float(                                                           \
        __builtin_amdgcn_update_dpp(ident, __float_as_int(v), ctrl, rows, 0xf, false));       \
    v = kMax ? vmax(v, t) : vmin(v, t);                                                       \
  }
  RH_DPP_STEP(0x111, 0xf)
  RH_DPP_STEP(0x112, 0xf)
  RH_DPP_STEP(0x114, 0xf)
  RH_DPP_STEP(0x118, 0xf)
  RH_DPP_STEP(0x142, 0xa)
  RH_DPP_STEP(0x143, 0xc)
#undef RH_DPP_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float lane_value(const float v, const int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// float32 sigmoid of a head logit (models/retinanet/model.py:177, 193), shared by the loss
// stream and the score scan so that both see the same probability for the same logit.
// exp is the hardware exp2 (1 ulp; the rounding of x * log2(e) adds up to |x| * 6e-8 relative) and
// the division is the hardware reciprocal plus one Newton step.  The step matters: 1 - p is
// exact given p, so for p close to 1 the last bit of p decides several digits of a loss
// gradient, and a plain 1-ulp reciprocal would show there.  exp2 is capped so that 1 + e
// stays finite for the Newton step.
__device__ __forceinline__ float sigmoid_f32(const float x) {
  const float e = fminf(__builtin_amdgcn_exp2f(x * -1.442695040888963f), 1e30f);
  const float d = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

}  // namespace rh
