// Inference half of the dense-head path on gfx950: box decode / clip, per-anchor
// class max + score threshold + candidate compaction, per-image sort, bitmask NMS
// and the greedy sweep; plus the generic elementwise focal BCE.
//
// Reference behaviour restated here: models/retinanet/utils.py:96-120 (decode),
// :133-142 (clip), models/retinanet/model.py:320-340 (post-processing) and the
// contract of torchvision.ops.nms called at model.py:336 (greedy; a box is dropped
// when its IoU with a kept, higher-scoring box is > threshold; ties in score keep
// the lower index first); models/focal_loss.py:14-25 (generic focal BCE).
//
// Compiled with -ffp-contract=off: box coordinates and IoU values decide integer
// keep lists that must match a float32 CPU evaluation bit for bit.
#include "rh_common.h"

namespace rh {

typedef unsigned long long u64;
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTileRows = 64;
constexpr int kSortThreads = 1024;
constexpr int kLdsSortMax = 16384;
constexpr int kCounterStride = 32;  // ints: per-image counters sit in separate 128-byte lines   // keys of 8 bytes: 128 KiB of the CU's 160 KiB LDS

// Order-preserving map float32 -> uint32 (larger float <=> larger key) and back.
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// 64-bit sort key: descending order = descending score, then ascending index.
__device__ __forceinline__ u64 make_key(float score, unsigned index) {
  return ((u64)f2key(score) << 32) | (u64)(0xFFFFFFFFu - index);
}
__device__ __forceinline__ unsigned key_index(u64 k) { return 0xFFFFFFFFu - (unsigned)k; }
__device__ __forceinline__ float key_score(u64 k) { return key2f((unsigned)(k >> 32)); }

struct BoxCoder {
  float mean[4], std[4];
};

// utils.py:98-116 for one box, float32, same operation order.
__device__ __forceinline__ float4 decode_box(const float4 a, const float4 d, const BoxCoder bc) {
  const float w = a.z - a.x, h = a.w - a.y;
  const float cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
  const float dx = d.x * bc.std[0] + bc.mean[0];
  const float dy = d.y * bc.std[1] + bc.mean[1];
  const float dw = d.z * bc.std[2] + bc.mean[2];
  const float dh = d.w * bc.std[3] + bc.mean[3];
  const float pcx = cx + dx * w, pcy = cy + dy * h;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  return make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
}
// utils.py:136-140
__device__ __forceinline__ float4 clip_box(float4 b, float width, float height) {
  b.x = fmaxf(b.x, 0.0f);
  b.y = fmaxf(b.y, 0.0f);
  b.z = fminf(b.z, width);
  b.w = fminf(b.w, height);
  return b;
}

__global__ __launch_bounds__(kThreads) void bbox_transform_kernel(const float4* __restrict__ boxes,
                                                                  const float4* __restrict__ deltas,
                                                                  float4* __restrict__ out,
                                                                  long long n, long long total,
                                                                  BoxCoder bc) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  out[i] = decode_box(boxes[i % n], deltas[i], bc);
}

__global__ __launch_bounds__(kThreads) void clip_boxes_kernel(float4* __restrict__ boxes,
                                                              long long rows, float width,
                                                              float height) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= rows) return;
  boxes[i] = clip_box(boxes[i], width, height);
}

// ---------------------------------------------------------------------------
// Candidate emission shared by the two decode kernels: anchor `n` of image `j`
// has class maximum `best` (key of (score, ~class)); model.py:323-333.
// The decoded, clipped box and the label are stored at the anchor's own slot of
// dense arrays, so the sort needs no payload: a key carries the anchor index.
// ---------------------------------------------------------------------------
struct CandOut {
  u64* keys;            // [B][key_stride]
  float4* dense_box;    // [B*N]
  int* dense_label;     // [B*N]
  int* cand_count;      // [B][kCounterStride]: one counter per 128-byte line
  int* overflow;        // [B][kCounterStride]
  int cap;
  int key_stride;
};

// Called by ALL 64 lanes of one wave (lane = tile row); rows of one tile belong to at
// most two images, so slots are reserved with one atomic per (wave, image).
__device__ __forceinline__ void emit_candidates_wave(const bool valid_row, const u64 best,
                                                     const int j, const int n, const int N,
                                                     const float4* __restrict__ anchors,
                                                     const float* __restrict__ reg,
                                                     const BoxCoder bc, const float width,
                                                     const float height, const float thresh,
                                                     const CandOut o) {
  const int lane = threadIdx.x & 63;
  const float score = key_score(best);
  const bool cand = valid_row && (score > thresh);
  const int j_first = __builtin_amdgcn_readfirstlane(j);
  u64 todo = __ballot(cand);
  int slot = -1;
  // at most two distinct images per wave: the first lane's, then whatever remains
  for (int pass = 0; pass < 2 && todo; ++pass) {
    const int jj = pass == 0 ? j_first : __builtin_amdgcn_readfirstlane(
        __shfl(j, __builtin_ctzll(todo), 64));
    const u64 grp = __ballot(cand && j == jj) & todo;
    if (grp) {
      int base = 0;
      if (lane == __builtin_ctzll(grp))
        base = atomicAdd(&o.cand_count[(size_t)jj * kCounterStride], __popcll(grp));
      base = __shfl(base, __builtin_ctzll(grp), 64);
      if (cand && j == jj) slot = base + __popcll(grp & ((1ull << lane) - 1ull));
      todo &= ~grp;
    }
  }
  if (!cand) return;
  const size_t row = (size_t)j * N + n;
  const float4 d = reinterpret_cast<const float4*>(reg)[row];
  o.dense_box[row] = clip_box(decode_box(anchors[n], d, bc), width, height);
  o.dense_label[row] = (int)key_index(best);
  if (slot < o.cap) {
    o.keys[(size_t)j * o.key_stride + slot] = make_key(score, (unsigned)n);
  } else {
    atomicAdd(&o.overflow[(size_t)j * kCounterStride], 1);
  }
}

// Tile kernel: 64 anchor rows per tile, float4 stream over the [B*N, C] matrix,
// per-row (max, first arg-max) through 64-bit LDS atomic max (order-independent).
// Host guarantees 4 <= C < 8192, N >= 64, 16-byte aligned pointers.
template <int U>
__global__ __launch_bounds__(kThreads) void decode_tile_kernel(
    const float* __restrict__ prob, const float* __restrict__ reg,
    const float4* __restrict__ anchors, CandOut o, int N, int C, int total_rows, int num_tiles,
    unsigned magic_c, BoxCoder bc, float width, float height, float thresh) {
  __shared__ u64 s_best[kTileRows];
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int row0 = tile * kTileRows;
    const int rows = min(kTileRows, total_rows - row0);
    __syncthreads();
    if (tid < kTileRows) s_best[tid] = 0ull;
    __syncthreads();
    const size_t base = (size_t)row0 * C;
    const v4f* __restrict__ p4 = reinterpret_cast<const v4f*>(prob + base);
    const int elems = rows * C;
    const int nvec = elems >> 2;
    for (int v0 = tid; v0 < nvec; v0 += kThreads * U) {
      v4f val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = v0 + u * kThreads;
        if (v < nvec) val[u] = __builtin_nontemporal_load(&p4[v]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = v0 + u * kThreads;
        if (v < nvec) {
          const unsigned x = (unsigned)v << 2;
          int r = (int)fast_div(x, magic_c);
          int col = (int)x - r * C;
          const float in[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
          u64 best = 0ull;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col >= C) {
              atomicMax(&s_best[r], best);
              best = 0ull;
              col -= C;
              r += 1;
            }
            const u64 k = make_key(in[e], (unsigned)col);
            best = k > best ? k : best;
            col += 1;
          }
          atomicMax(&s_best[r], best);
        }
      }
    }
    {
      const int x = (nvec << 2) + tid;   // < 4 trailing elements of the last tile
      if (x < elems) {
        const int r = x / C, col = x - r * C;
        atomicMax(&s_best[r], make_key(prob[base + x], (unsigned)col));
      }
    }
    __syncthreads();
    if (tid < kTileRows) {     // wave 0: one lane per row
      const int grow = min(row0 + tid, total_rows - 1);
      const int j = grow / N;
      emit_candidates_wave(tid < rows, s_best[tid], j, grow - j * N, N, anchors, reg, bc, width,
                           height, thresh, o);
    }
  }
}

// Generic kernel: one wave per anchor row, scalar loads; any N, C, alignment.
__global__ __launch_bounds__(kThreads) void decode_rows_kernel(
    const float* __restrict__ prob, const float* __restrict__ reg,
    const float* __restrict__ anchors, CandOut o, int N, int C, int total_rows, BoxCoder bc,
    float width, float height, float thresh) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  const int waves = (gridDim.x * kThreads) >> 6;
  for (int grow = wave; grow < total_rows; grow += waves) {
    const float* p = prob + (size_t)grow * C;
    u64 best = 0ull;
    for (int c = lane; c < C; c += 64) {
      const u64 k = make_key(p[c], (unsigned)c);
      best = k > best ? k : best;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const u64 other = __shfl_xor(best, s, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) {
      const int j = grow / N, n = grow - j * N;
      const float score = key_score(best);
      if (score > thresh) {
        const size_t row = (size_t)grow;
        const float4 a = make_float4(anchors[(size_t)n * 4], anchors[(size_t)n * 4 + 1],
                                     anchors[(size_t)n * 4 + 2], anchors[(size_t)n * 4 + 3]);
        const float4 d = make_float4(reg[row * 4], reg[row * 4 + 1], reg[row * 4 + 2],
                                     reg[row * 4 + 3]);
        o.dense_box[row] = clip_box(decode_box(a, d, bc), width, height);
        o.dense_label[row] = (int)key_index(best);
        const int slot = atomicAdd(&o.cand_count[(size_t)j * kCounterStride], 1);
        if (slot < o.cap) o.keys[(size_t)j * o.key_stride + slot] = make_key(score, (unsigned)n);
        else atomicAdd(&o.overflow[(size_t)j * kCounterStride], 1);
      }
    }
  }
}

// Per-image counters live one per 128-byte line; copy them out densely.
__global__ __launch_bounds__(kThreads) void export_counters_kernel(const int* __restrict__ cnt,
                                                                   const int* __restrict__ ovf,
                                                                   int* __restrict__ cnt_out,
                                                                   int* __restrict__ ovf_out, int B) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= B) return;
  if (cnt_out) cnt_out[j] = cnt[(size_t)j * kCounterStride];
  if (ovf_out) ovf_out[j] = ovf[(size_t)j * kCounterStride];
}

// Keys for a plain (boxes, scores) NMS call: every box is a candidate.
__global__ __launch_bounds__(kThreads) void make_keys_kernel(const float* __restrict__ scores,
                                                             u64* __restrict__ keys, int k,
                                                             int* __restrict__ count) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < k) keys[i] = make_key(scores[i], (unsigned)i);
  if (i == 0) *count = k;
}

// ---------------------------------------------------------------------------
// Sort, descending, of each image's candidate keys.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// After the sort: clamp to top-k, write the sorted keys back and gather the
// candidates' boxes and labels into sorted order.
__device__ __forceinline__ void store_sorted(const u64 key, const int i, const int j, const int N,
                                             u64* __restrict__ keys_out,
                                             const float4* __restrict__ dense_box,
                                             const int* __restrict__ dense_label,
                                             float4* __restrict__ sbox, int* __restrict__ slabel) {
  keys_out[i] = key;
  const size_t row = (size_t)j * N + key_index(key);
  sbox[i] = dense_box[row];
  if (slabel) slabel[i] = dense_label ? dense_label[row] : 0;
}

// One workgroup per image; the whole key array lives in LDS (bitonic network).
__global__ __launch_bounds__(kSortThreads) void sort_lds_kernel(
    u64* __restrict__ keys, const int* __restrict__ cand_count, int* __restrict__ keff, int cap,
    int key_stride, int topk, int N, const float4* __restrict__ dense_box,
    const int* __restrict__ dense_label, float4* __restrict__ sbox, int* __restrict__ slabel,
    int box_stride) {
  extern __shared__ __attribute__((aligned(16))) u64 s_keys[];
  const int j = blockIdx.x;
  const int tid = threadIdx.x;
  const int K = min(cand_count[(size_t)j * kCounterStride], cap);
  const int P = next_pow2(max(K, 1));
  u64* kj = keys + (size_t)j * key_stride;
  for (int i = tid; i < P; i += kSortThreads) s_keys[i] = i < K ? kj[i] : 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < (P >> 1); t += kSortThreads) {
        const int i = 2 * t - (t & (jj - 1));
        const int l = i + jj;
        const u64 a = s_keys[i], b = s_keys[l];
        const bool desc = (i & k) == 0;
        if ((a < b) == desc) {
          s_keys[i] = b;
          s_keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
  const int ke = topk > 0 ? min(K, topk) : K;
  for (int i = tid; i < ke; i += kSortThreads)
    store_sorted(s_keys[i], i, j, N, kj, dense_box, dense_label, sbox + (size_t)j * box_stride,
                 slabel ? slabel + (size_t)j * box_stride : nullptr);
  if (tid == 0) keff[(size_t)j * kCounterStride] = ke;
}

// Global-memory fallback for candidate lists that do not fit LDS.
__global__ __launch_bounds__(kThreads) void sort_pad_kernel(u64* __restrict__ keys,
                                                            const int* __restrict__ cand_count,
                                                            int cap, int key_stride, int P) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int K = min(cand_count[(size_t)j * kCounterStride], cap);
  if (i < P && i >= K) keys[(size_t)j * key_stride + i] = 0ull;
}
__global__ __launch_bounds__(kThreads) void sort_step_kernel(u64* __restrict__ keys, int key_stride,
                                                             int P, int k, int jj) {
  const int j = blockIdx.y;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= (P >> 1)) return;
  u64* kj = keys + (size_t)j * key_stride;
  const int i = 2 * t - (t & (jj - 1));
  const int l = i + jj;
  const u64 a = kj[i], b = kj[l];
  const bool desc = (i & k) == 0;
  if ((a < b) == desc) {
    kj[i] = b;
    kj[l] = a;
  }
}
__global__ __launch_bounds__(kThreads) void sort_gather_kernel(
    u64* __restrict__ keys, const int* __restrict__ cand_count, int* __restrict__ keff, int cap,
    int key_stride, int topk, int N, const float4* __restrict__ dense_box,
    const int* __restrict__ dense_label, float4* __restrict__ sbox, int* __restrict__ slabel,
    int box_stride) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int K = min(cand_count[(size_t)j * kCounterStride], cap);
  const int ke = topk > 0 ? min(K, topk) : K;
  u64* kj = keys + (size_t)j * key_stride;
  if (i < ke)
    store_sorted(kj[i], i, j, N, kj, dense_box, dense_label, sbox + (size_t)j * box_stride,
                 slabel ? slabel + (size_t)j * box_stride : nullptr);
  if (i == 0) keff[(size_t)j * kCounterStride] = ke;
}

// ---------------------------------------------------------------------------
// NMS bitmask.  Boxes are in sorted order; word (i, w) of an image's mask has bit
// c set when box w*64+c comes after box i and overlaps it by more than `thresh`
// (and, per class, carries the same label).  Only the upper triangle of 64x64
// tiles is produced; one wave per tile, the tile's column boxes staged in LDS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool nms_hit(const float4 a, const float area_a, const float4 b,
                                        const float area_b, const float thresh) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  if (!(inter > 0.0f) && !(thresh < 0.0f)) return false;   // IoU is 0 or NaN: never > thresh >= 0
  return inter / ((area_a + area_b) - inter) > thresh;
}

__global__ __launch_bounds__(kThreads) void nms_mask_kernel(const float4* __restrict__ sbox,
                                                            const int* __restrict__ slabel,
                                                            const int* __restrict__ keff,
                                                            u64* __restrict__ mask, int box_stride,
                                                            int wstride, float thresh,
                                                            int per_class) {
  __shared__ float4 s_box[kThreads / 64][64];
  __shared__ int s_lab[kThreads / 64][64];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = keff[(size_t)j * kCounterStride];
  const int W = (K + 63) >> 6;
  const long long T = (long long)W * (W + 1) / 2;
  const float4* bj = sbox + (size_t)j * box_stride;
  const int* lj = slabel ? slabel + (size_t)j * box_stride : nullptr;
  u64* mj = mask + (size_t)j * box_stride * wstride;
  for (long long id = (long long)blockIdx.x * (kThreads / 64) + wave; id < T;
       id += (long long)gridDim.x * (kThreads / 64)) {
    // tile id -> (rt, ct), rows of the upper triangle enumerated in order:
    // row rt starts at off(rt) = rt*W - rt*(rt-1)/2
    int rt = (int)(((2.0 * W + 1.0) - sqrt((2.0 * W + 1.0) * (2.0 * W + 1.0) - 8.0 * (double)id)) * 0.5);
    rt = max(0, min(rt, W - 1));
    while (rt > 0 && (long long)rt * W - (long long)rt * (rt - 1) / 2 > id) --rt;
    while ((long long)(rt + 1) * W - (long long)(rt + 1) * rt / 2 <= id) ++rt;
    const int ct = rt + (int)(id - ((long long)rt * W - (long long)rt * (rt - 1) / 2));
    const int i = rt * 64 + lane;
    const int c = ct * 64 + lane;
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
    int mylab = 0;
    if (i < K) {
      mine = bj[i];
      if (per_class) mylab = lj[i];
    }
    if (c < K) {
      s_box[wave][lane] = bj[c];
      if (per_class) s_lab[wave][lane] = lj[c];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes have landed
    const float area = (mine.z - mine.x) * (mine.w - mine.y);
    const int cmax = min(64, K - ct * 64);
    u64 bits = 0ull;
    for (int cc = 0; cc < cmax; ++cc) {
      const float4 o = s_box[wave][cc];
      bool hit = nms_hit(mine, area, o, (o.z - o.x) * (o.w - o.y), thresh);
      if (per_class) hit = hit && s_lab[wave][cc] == mylab;
      hit = hit && (ct * 64 + cc > i);
      bits |= hit ? (1ull << cc) : 0ull;
    }
    // column-major: word (column tile ct, row i); a wave writes 512 contiguous bytes
    if (i < K) mj[(size_t)ct * box_stride + i] = bits;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Greedy sweep: one wave per image walks the sorted boxes in chunks of 64.  For
// chunk c the wave first PULLS the chunk's "removed" word: the OR, over the kept
// boxes of all earlier chunks, of their mask word for column tile c (coalesced
// reads of the column-major mask, lane-wise masked by the kept bitmap, one wave
// OR-reduction).  The diagonal words then resolve the chunk itself: only rows that
// overlap a later row of the same chunk need a serial step.
// Outputs either the kept original indices (keep_idx, for rh_nms) or the
// (score, label, box) triples of model.py:338-340.
// ---------------------------------------------------------------------------
struct SweepOut {
  long long* keep_idx;   // [B][out_stride] or null
  float* scores;         // [B][out_stride] or null
  long long* labels;     // [B][out_stride] or null
  float4* boxes;         // [B][out_stride] or null
  int* count;            // [B]
  int out_stride;
  int max_keep;          // stop after this many (<= out_stride)
};

__device__ __forceinline__ u64 wave_or(u64 v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo |= (unsigned)__shfl_xor((int)lo, o, 64);
    hi |= (unsigned)__shfl_xor((int)hi, o, 64);
  }
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void nms_sweep_kernel(const u64* __restrict__ keys,
                                                       const float4* __restrict__ sbox,
                                                       const int* __restrict__ slabel,
                                                       const int* __restrict__ keff,
                                                       const u64* __restrict__ mask, int key_stride,
                                                       int box_stride, int wstride, SweepOut o) {
  extern __shared__ __attribute__((aligned(16))) u64 s_kept[];   // kept bitmap, one word per chunk
  const int j = blockIdx.x;
  const int lane = threadIdx.x;
  const int K = keff[(size_t)j * kCounterStride];
  const int W = (K + 63) >> 6;
  const u64* kj = keys + (size_t)j * key_stride;
  const float4* bj = sbox + (size_t)j * box_stride;
  const int* lj = slabel ? slabel + (size_t)j * box_stride : nullptr;
  const u64* mj = mask + (size_t)j * box_stride * wstride;
  int kept_total = 0;
  for (int c = 0; c < W && kept_total < o.max_keep; ++c) {
    const u64* col = mj + (size_t)c * box_stride;
    const int row = c * 64 + lane;
    const bool valid = row < K;
    const u64 diag = valid ? col[row] : 0ull;
    // pull: which rows of this chunk were suppressed by boxes kept earlier
    u64 acc = 0ull;
#pragma unroll 8
    for (int i = 0; i < c; ++i) {
      const u64 word = col[i * 64 + lane];
      acc |= ((s_kept[i] >> lane) & 1ull) ? word : 0ull;
    }
    const u64 removed = wave_or(acc);
    const u64 live_rows = (K - c * 64 >= 64) ? ~0ull : ((1ull << (K - c * 64)) - 1ull);
    u64 alive = ~removed & live_rows;
    // serial part: only rows that suppress something inside the chunk matter
    u64 todo = __ballot(diag != 0ull) & alive;
    const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
    while (todo) {
      const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
      const u64 d = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                    (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, b);
      alive &= ~d;               // d only has bits above b
      todo &= alive;
      todo &= ~((2ull << b) - 1ull);
    }
    const u64 kept = alive;
    if (lane == 0) s_kept[c] = kept;
    if ((kept >> lane) & 1ull) {
      const int pos = kept_total + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < o.max_keep) {
        const u64 key = kj[row];
        const size_t at = (size_t)j * o.out_stride + pos;
        if (o.keep_idx) o.keep_idx[at] = (long long)key_index(key);
        if (o.scores) o.scores[at] = key_score(key);
        if (o.labels) o.labels[at] = lj ? (long long)lj[row] : 0ll;
        if (o.boxes) o.boxes[at] = bj[row];
      }
    }
    kept_total += __popcll(kept);
    __syncthreads();   // s_kept[c] visible to the next chunk's pull
  }
  if (lane == 0) o.count[j] = min(kept_total, o.max_keep);
}

// ---------------------------------------------------------------------------
// Generic elementwise focal BCE (models/focal_loss.py:14-25), forward + backward.
// ---------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
static int host_next_pow2(long long v) {
  long long p = 1;
  while (p < v) p <<= 1;
  return (int)p;
}

static BoxCoder make_coder(const float* mean, const float* std) {
  BoxCoder bc;
  const float dm[4] = {0.f, 0.f, 0.f, 0.f}, ds[4] = {0.1f, 0.1f, 0.2f, 0.2f};
  for (int i = 0; i < 4; ++i) {
    bc.mean[i] = mean ? mean[i] : dm[i];
    bc.std[i] = std ? std[i] : ds[i];
  }
  return bc;
}

// Sort + mask + sweep over B candidate lists that are already keyed.
struct NmsPlan {
  int batch;
  int cap;          // candidates per image the buffers hold
  int key_stride;   // >= next_pow2(cap)
  int box_cap;      // boxes that reach the mask stage (min(cap, topk))
  int wstride;      // mask words per row
};

static NmsPlan make_plan(int batch, long long cap, int topk) {
  NmsPlan p;
  p.batch = batch;
  p.cap = (int)cap;
  p.key_stride = host_next_pow2(cap < 1 ? 1 : cap);
  p.box_cap = (topk > 0 && topk < cap) ? topk : (int)cap;
  if (p.box_cap < 1) p.box_cap = 1;
  p.wstride = (p.box_cap + 63) / 64;
  return p;
}

static int launch_sort(const NmsPlan& p, u64* keys, const int* cand_count, int* keff, int topk, int N,
                       const float4* dense_box, const int* dense_label, float4* sbox, int* slabel,
                       hipStream_t s) {
  if (p.key_stride <= kLdsSortMax) {
    const size_t lds = (size_t)p.key_stride * sizeof(u64);
    if (lds > 64 * 1024)   // above the default dynamic-LDS limit the kernel must opt in
      RH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sort_lds_kernel, dim3(p.batch), dim3(kSortThreads), lds, s, keys, cand_count,
                       keff, p.cap, p.key_stride, topk, N, dense_box, dense_label, sbox, slabel,
                       p.box_cap);
    RH_CHECK_LAUNCH("sort_lds_kernel");
    return RH_OK;
  }
  const int P = p.key_stride;
  dim3 gp((P + kThreads - 1) / kThreads, p.batch), gh((P / 2 + kThreads - 1) / kThreads, p.batch);
  hipLaunchKernelGGL(sort_pad_kernel, gp, dim3(kThreads), 0, s, keys, cand_count, p.cap,
                     p.key_stride, P);
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1)
      hipLaunchKernelGGL(sort_step_kernel, gh, dim3(kThreads), 0, s, keys, p.key_stride, P, k, jj);
  hipLaunchKernelGGL(sort_gather_kernel, dim3((p.box_cap + kThreads - 1) / kThreads, p.batch),
                     dim3(kThreads), 0, s, keys, cand_count, keff, p.cap, p.key_stride, topk, N,
                     dense_box, dense_label, sbox, slabel, p.box_cap);
  RH_CHECK_LAUNCH("global bitonic sort");
  return RH_OK;
}

static int launch_nms(const NmsPlan& p, const u64* keys, const float4* sbox, const int* slabel,
                      const int* keff, u64* mask, float thresh, int per_class, SweepOut out,
                      hipStream_t s) {
  const long long tiles = (long long)p.wstride * (p.wstride + 1) / 2;
  long long gx = (tiles + 3) / 4;
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)gx, p.batch), dim3(kThreads), 0, s, sbox,
                     per_class ? slabel : nullptr, keff, mask, p.box_cap, p.wstride, thresh,
                     per_class);
  RH_CHECK_LAUNCH("nms_mask_kernel");
  const size_t lds = (size_t)p.wstride * sizeof(u64);
  RH_REQUIRE(lds <= 160 * 1024, RH_ESHAPE, "%d boxes per image exceed the sweep's LDS bitmap",
             p.box_cap);
  if (lds > 64 * 1024)
    RH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(p.batch), dim3(64), lds, s, keys, sbox, slabel, keff,
                     mask, p.key_stride, p.box_cap, p.wstride, out);
  RH_CHECK_LAUNCH("nms_sweep_kernel");
  return RH_OK;
}

struct PostWorkspace {
  u64* keys;
  float4* dense_box;
  int* dense_label;
  int* counters;     // [3][B][kCounterStride]: cand_count, overflow, keff (zeroed each call)
  float4* sbox;
  int* slabel;
  u64* mask;
  size_t bytes;
};

static PostWorkspace carve_post(void* ws, const NmsPlan& p, long long dense_rows) {
  Carver cv(ws);
  PostWorkspace w;
  w.keys = cv.take<u64>((size_t)p.batch * p.key_stride);
  w.dense_box = cv.take<float4>((size_t)dense_rows);
  w.dense_label = cv.take<int>((size_t)dense_rows);
  w.counters = cv.take<int>((size_t)3 * p.batch * kCounterStride);
  w.sbox = cv.take<float4>((size_t)p.batch * p.box_cap);
  w.slabel = cv.take<int>((size_t)p.batch * p.box_cap);
  w.mask = cv.take<u64>((size_t)p.batch * p.box_cap * p.wstride);
  w.bytes = cv.off;
  return w;
}

}  // namespace rh

using namespace rh;

extern "C" {

int rh_bbox_transform_f32(const float* boxes, const float* deltas, int32_t batch, int64_t n,
                          const float* mean, const float* std, float* out, rh_stream_t stream) {
  RH_REQUIRE(batch >= 0 && n >= 0, RH_ESHAPE, "bad shape batch=%d n=%lld", batch, (long long)n);
  const long long total = (long long)batch * n;
  if (total == 0) return RH_OK;
  RH_REQUIRE(boxes && deltas && out, RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(boxes) && aligned16(deltas) && aligned16(out), RH_EALIGN,
             "box arrays must be 16-byte aligned");
  RH_REQUIRE((total + kThreads - 1) / kThreads < (1ll << 31), RH_ESHAPE, "too many boxes");
  hipLaunchKernelGGL(bbox_transform_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(boxes),
                     reinterpret_cast<const float4*>(deltas), reinterpret_cast<float4*>(out),
                     (long long)n, total, make_coder(mean, std));
  RH_CHECK_LAUNCH("bbox_transform_kernel");
  return RH_OK;
}

int rh_clip_boxes_f32(float* boxes, int64_t rows, float width, float height, rh_stream_t stream) {
  RH_REQUIRE(rows >= 0, RH_ESHAPE, "rows < 0");
  if (rows == 0) return RH_OK;
  RH_REQUIRE(boxes, RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(boxes), RH_EALIGN, "boxes must be 16-byte aligned");
  RH_REQUIRE((rows + kThreads - 1) / kThreads < (1ll << 31), RH_ESHAPE, "too many boxes");
  hipLaunchKernelGGL(clip_boxes_kernel, dim3((unsigned)((rows + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, reinterpret_cast<float4*>(boxes),
                     (long long)rows, width, height);
  RH_CHECK_LAUNCH("clip_boxes_kernel");
  return RH_OK;
}

size_t rh_nms_workspace_bytes(int64_t k) {
  if (k < 1 || k >= (1ll << 24)) return 0;
  const NmsPlan p = make_plan(1, k, 0);
  return carve_post(nullptr, p, 0).bytes;
}

int rh_nms(const float* boxes, const float* scores, const int32_t* labels, int64_t k,
           float iou_threshold, int64_t* keep, int32_t* keep_count, void* workspace,
           size_t workspace_bytes, rh_stream_t stream) {
  RH_REQUIRE(k >= 0 && k < (1ll << 24), RH_ESHAPE, "k=%lld out of range", (long long)k);
  RH_REQUIRE(keep_count, RH_EINVAL, "null keep_count");
  hipStream_t s = (hipStream_t)stream;
  if (k == 0) {
    RH_CHECK_HIP(hipMemsetAsync(keep_count, 0, sizeof(int), s));
    return RH_OK;
  }
  RH_REQUIRE(boxes && scores && keep, RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(boxes), RH_EALIGN, "boxes must be 16-byte aligned");
  const NmsPlan p = make_plan(1, k, 0);
  const size_t need = carve_post(nullptr, p, 0).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  const PostWorkspace w = carve_post(workspace, p, 0);
  int* cand_count = w.counters;
  int* keff = w.counters + 2 * kCounterStride;
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)((k + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, scores, w.keys, (int)k, cand_count);
  RH_CHECK_LAUNCH("make_keys_kernel");
  // the caller's arrays are the "dense" arrays: key index == row
  if (int e = launch_sort(p, w.keys, cand_count, keff, 0, 0, reinterpret_cast<const float4*>(boxes),
                          labels, w.sbox, w.slabel, s))
    return e;
  SweepOut out{};
  out.keep_idx = reinterpret_cast<long long*>(keep);
  out.count = keep_count;
  out.out_stride = (int)k;
  out.max_keep = (int)k;
  return launch_nms(p, w.keys, w.sbox, w.slabel, keff, w.mask, iou_threshold, labels != nullptr, out,
                    s);
}

static int check_post(int32_t batch, int64_t n, int32_t c, const rh_post_params* q) {
  RH_REQUIRE(q, RH_EINVAL, "null params");
  RH_REQUIRE(batch >= 1 && n >= 1 && c >= 1, RH_ESHAPE, "bad shape batch=%d n=%lld classes=%d", batch,
             (long long)n, c);
  RH_REQUIRE((long long)batch * n < (1ll << 31), RH_ESHAPE, "batch*n does not fit 31 bits");
  RH_REQUIRE(batch <= 65535, RH_ESHAPE, "batch=%d: at most 65535 images per call", batch);
  RH_REQUIRE(q->cand_cap >= 1 && q->cand_cap <= n, RH_EINVAL, "cand_cap=%d must be in [1, n]",
             q->cand_cap);
  RH_REQUIRE(q->pre_nms_topk >= 0 && q->max_det >= 0, RH_EINVAL, "negative top-k / max_det");
  return RH_OK;
}

size_t rh_postprocess_workspace_bytes(int32_t batch, int64_t n, int32_t num_classes,
                                      const rh_post_params* params) {
  if (check_post(batch, n, num_classes, params)) return 0;
  const NmsPlan p = make_plan(batch, params->cand_cap, params->pre_nms_topk);
  return carve_post(nullptr, p, (long long)batch * n).bytes;
}

int rh_postprocess(const float* classifications, const float* regressions, const float* anchors,
                   int32_t batch, int64_t n, int32_t num_classes, const rh_post_params* q,
                   float* out_scores, int64_t* out_labels, float* out_boxes, int32_t* out_count,
                   int32_t* cand_count, int32_t* overflow, void* workspace,
                   size_t workspace_bytes, int32_t stages, rh_stream_t stream) {
  if (int e = check_post(batch, n, num_classes, q)) return e;
  RH_REQUIRE(classifications && regressions && anchors && out_scores && out_labels && out_boxes &&
                 out_count,
             RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(anchors) && aligned16(regressions) && aligned16(out_boxes), RH_EALIGN,
             "anchors, regressions and out_boxes must be 16-byte aligned");
  RH_REQUIRE((stages & ~RH_POST_ALL) == 0 && stages != 0, RH_EINVAL, "bad stages mask %d", stages);
  const NmsPlan p = make_plan(batch, q->cand_cap, q->pre_nms_topk);
  const size_t need = carve_post(nullptr, p, (long long)batch * n).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const PostWorkspace w = carve_post(workspace, p, (long long)batch * n);
  int* cnt = w.counters;
  int* ovf = w.counters + (size_t)batch * kCounterStride;
  int* keff = w.counters + (size_t)2 * batch * kCounterStride;
  const int out_cap = q->max_det > 0 ? q->max_det : p.box_cap;

  if (stages & RH_POST_DECODE) {
    RH_CHECK_HIP(hipMemsetAsync(w.counters, 0, sizeof(int) * 3 * batch * kCounterStride, s));
    CandOut co{w.keys, w.dense_box, w.dense_label, cnt, ovf, p.cap, p.key_stride};
    const BoxCoder bc = make_coder(nullptr, nullptr);
    const int total_rows = (int)((long long)batch * n);
    const bool tile = n >= kTileRows && num_classes >= 4 && num_classes < 8192 &&
                      aligned16(classifications);
    if (tile) {
      const int num_tiles = (total_rows + kTileRows - 1) / kTileRows;
      const bool u5 = ((kTileRows * num_classes / 4 + kThreads - 1) / kThreads) % 5 == 0;
      auto kern = u5 ? decode_tile_kernel<5> : decode_tile_kernel<4>;
      hipLaunchKernelGGL(kern, dim3(num_tiles), dim3(kThreads), 0, s, classifications,
                         regressions, reinterpret_cast<const float4*>(anchors), co, (int)n,
                         num_classes, total_rows, num_tiles, div_magic((unsigned)num_classes), bc,
                         q->width, q->height, q->score_thresh);
      RH_CHECK_LAUNCH("decode_tile_kernel");
    } else {
      long long blocks = ((long long)total_rows + 3) / 4;
      if (blocks > 256 * 8) blocks = 256 * 8;
      hipLaunchKernelGGL(decode_rows_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s,
                         classifications, regressions, anchors, co, (int)n, num_classes, total_rows,
                         bc, q->width, q->height, q->score_thresh);
      RH_CHECK_LAUNCH("decode_rows_kernel");
    }
  }
  if (stages & RH_POST_SORT) {
    if (int e = launch_sort(p, w.keys, cnt, keff, q->pre_nms_topk, (int)n, w.dense_box,
                            w.dense_label, w.sbox, w.slabel, s))
      return e;
  }
  if (stages & RH_POST_NMS) {
    SweepOut out{};
    out.scores = out_scores;
    out.labels = reinterpret_cast<long long*>(out_labels);
    out.boxes = reinterpret_cast<float4*>(out_boxes);
    out.count = out_count;
    out.out_stride = out_cap;
    out.max_keep = out_cap;
    if (int e = launch_nms(p, w.keys, w.sbox, w.slabel, keff, w.mask, q->iou_thresh, q->per_class,
                           out, s))
      return e;
    if (cand_count || overflow) {
      hipLaunchKernelGGL(export_counters_kernel, dim3((batch + kThreads - 1) / kThreads),
                         dim3(kThreads), 0, s, cnt, ovf, cand_count, overflow, batch);
      RH_CHECK_LAUNCH("export_counters_kernel");
    }
  }
  return RH_OK;
}

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

}  // extern "C"
