// Micro-benchmark: how fast does one wave (and 2, 4, 8 waves per SIMD) run chains of vector
// instructions of the kind the NMS pair loop uses?  Prints ns per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int kMode>
__global__ void body(float* out, long long* t, int iters, unsigned long long hm_in) {
  const int lane = threadIdx.x & 63;
  float4 mine = make_float4(lane, lane * 2.f, lane + 30.f, lane * 2.f + 30.f);
  float4 e = make_float4(lane * 0.5f, lane, lane * 0.5f + 20.f, lane + 25.f);
  float area = 900.f, earea = 500.f, acc = 0.f;
  unsigned long long hits = 0;
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    unsigned long long m = hm_in;
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      if (kMode == 0) {          // readlane body
        const float bx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), src));
        const float by1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.y), src));
        const float bx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.z), src));
        const float by2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.w), src));
        const float ba = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(earea), src));
        const float w = fminf(bx2, mine.z) - fmaxf(bx1, mine.x);
        const float h = fminf(by2, mine.w) - fmaxf(by1, mine.y);
        const float inter = fmaxf(w, 0.f) * h;
        const float denom = fmaxf((area + ba) - inter, 0.f);
        hits |= __ballot(inter > denom * 0.4999f);
      } else if (kMode == 1) {   // pure VALU chain of the same length, no cross-lane ops
        float w = fminf(e.z, mine.z) - fmaxf(e.x, mine.x) + (float)src;
        float h = fminf(e.w, mine.w) - fmaxf(e.y, mine.y);
        float inter = fmaxf(w, 0.f) * h;
        float denom = fmaxf((area + earea) - inter, 0.f);
        acc += inter > denom * 0.4999f ? 1.f : 0.f;
      } else {                    // only the scalar loop skeleton + one readlane
        acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), src));
      }
      e.x += 1e-6f;
    }
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)hits + e.x;
}

template <int kMode>
void run(const char* name, int blocks, int threads) {
  float* out;
  long long* t;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&t, sizeof(long long));
  const int iters = 200;
  const unsigned long long hm = 0x00FFFFFFFFFFFFFFull;   // 56 survivors
  hipLaunchKernelGGL(body<kMode>, dim3(blocks), dim3(threads), 0, 0, out, t, iters, hm);
  hipLaunchKernelGGL(body<kMode>, dim3(blocks), dim3(threads), 0, 0, out, t, iters, hm);
  hipDeviceSynchronize();
  long long ticks;
  hipMemcpy(&ticks, t, sizeof(ticks), hipMemcpyDeviceToHost);
  printf("%-28s blocks %5d x %4d threads: %7.2f ns per survivor iteration\n", name, blocks, threads,
         ticks * 10.0 / (iters * 56.0));
  hipFree(out);
  hipFree(t);
}

int main() {
  for (int wpb : {64, 256}) {
    for (int blocks : {256, 1024, 2048}) {
      run<0>("readlane body", blocks, wpb);
      run<1>("plain VALU chain", blocks, wpb);
      run<2>("loop skeleton + 1 readlane", blocks, wpb);
    }
  }
  return 0;
}
