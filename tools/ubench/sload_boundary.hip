// Do scalar loads (s_load_dwordx4, with and without glc) of records that the PREVIOUS kernel on the
// stream wrote with plain vector stores ever return older bytes -- when the same addresses were
// read by scalar loads in an earlier launch (so the scalar cache held those lines), and while a
// second stream keeps the chip busy?  Round r: writer kernel stores (r, i) records; reader kernel
// loads every record three ways (s_load, s_load glc, vector load) and counts the words that are
// not of round r.  Build: hipcc --offload-arch=gfx950 -O2 sload_boundary.hip -o sload_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void writer(uint4* rec, int n, unsigned round) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rec[i] = make_uint4(round, (unsigned)i, round ^ 0x5a5a5a5au, ~(unsigned)i);
}

// Each wave walks its share of the records with wave-uniform addresses.
__global__ void reader(const uint4* rec, int n, unsigned round, unsigned long long* bad) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves = (gridDim.x * blockDim.x) >> 6;
  unsigned long long b0 = 0, b1 = 0, b2 = 0;
  for (int i = wave; i < n; i += waves) {
    const uint4* p = rec + __builtin_amdgcn_readfirstlane(i);
    unsigned a0, a1, a2, a3, g0, g1, g2, g3;
    asm volatile("s_load_dwordx4 s[20:23], %4, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "v_mov_b32 %0, s20\n\tv_mov_b32 %1, s21\n\tv_mov_b32 %2, s22\n\tv_mov_b32 %3, s23"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "s"(p) : "s20", "s21", "s22", "s23", "memory");
    asm volatile("s_load_dwordx4 s[20:23], %4, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "v_mov_b32 %0, s20\n\tv_mov_b32 %1, s21\n\tv_mov_b32 %2, s22\n\tv_mov_b32 %3, s23"
                 : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "s"(p) : "s20", "s21", "s22", "s23", "memory");
    const volatile unsigned* pv = reinterpret_cast<const volatile unsigned*>(p);
    const uint4 v = make_uint4(pv[0], pv[1], pv[2], pv[3]);
    const unsigned e1 = (unsigned)i, e2 = round ^ 0x5a5a5a5au, e3 = ~(unsigned)i;
    b0 += (a0 != round) + (a1 != e1) + (a2 != e2) + (a3 != e3);
    b1 += (g0 != round) + (g1 != e1) + (g2 != e2) + (g3 != e3);
    b2 += (v.x != round) + (v.y != e1) + (v.z != e2) + (v.w != e3);
  }
  if ((threadIdx.x & 63) == 0 && (b0 | b1 | b2)) {
    atomicAdd(bad + 0, b0);
    atomicAdd(bad + 1, b1);
    atomicAdd(bad + 2, b2);
  }
}

__global__ void noise(float* x, int n) {   // a streaming kernel on another stream
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = x[i] * 1.0001f + 1.0f;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  const int sizes[] = {512, 4096, 65536};             // records of 16 bytes: 8 KB, 64 KB, 1 MB
  uint4* rec;
  unsigned long long* bad;
  float* junk;
  CHECK(hipMalloc(&rec, 65536 * sizeof(uint4)));
  CHECK(hipMalloc(&bad, 3 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&junk, 64 << 20));
  CHECK(hipMemset(junk, 0, 64 << 20));
  hipStream_t s, s2;
  CHECK(hipStreamCreate(&s));
  CHECK(hipStreamCreate(&s2));
  for (int with_noise = 0; with_noise < 2; ++with_noise)
    for (int n : sizes) {
      CHECK(hipMemset(bad, 0, 3 * sizeof(unsigned long long)));
      for (int r = 1; r <= rounds; ++r) {
        if (with_noise && (r & 7) == 0) hipLaunchKernelGGL(noise, dim3(2048), dim3(256), 0, s2, junk, 16 << 20);
        hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s, rec, n, (unsigned)r);
        hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, s, rec, n, (unsigned)r, bad);
      }
      CHECK(hipDeviceSynchronize());
      unsigned long long h[3];
      CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
      printf("records %6d  second stream busy %d  rounds %d  stale words: s_load %llu  s_load glc %llu  vector %llu\n",
             n, with_noise, rounds, h[0], h[1], h[2]);
    }
  return 0;
}
