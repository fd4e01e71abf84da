// Inference half of the dense-head path on gfx950: box decode / clip, per-anchor
// class max + score threshold + candidate compaction (on the flat [B,N,C] tensor or on
// the heads' per-level [B, A*C, H, W] outputs), per-image sort, bitmask NMS and the
// greedy sweep.
//
// Reference behaviour restated here: models/retinanet/utils.py:96-120 (decode),
// :133-142 (clip), models/retinanet/model.py:320-340 (post-processing) and the
// contract of torchvision.ops.nms called at model.py:336 (greedy; a box is dropped
// when its IoU with a kept, higher-scoring box is > threshold; ties in score keep
// the lower index first).
//
// Compiled with -ffp-contract=off: box coordinates and IoU values decide integer
// keep lists that must match a float32 CPU evaluation bit for bit.
#include "rh_common.h"
#include <atomic>
#include <mutex>

namespace rh {

typedef unsigned long long u64;
typedef float v4f __attribute__((ext_vector_type(4)));

// Build-time diagnostic (-DRH_TIMELINE): thread 0 of every block stores wall-clock stamps
// (100 MHz) at the marked places; rh_tl_read() copies them out.  Not part of the shipped library.
#ifdef RH_TIMELINE
constexpr int kTlBlocks = 8192, kTlSlots = 12;
__device__ long long rh_tl[3][kTlBlocks][kTlSlots];
#define RH_TL(kernel, slot)                                                                     \
  do {                                                                                          \
    if (threadIdx.x == 0)                                                                       \
      rh_tl[kernel][(blockIdx.y * gridDim.x + blockIdx.x) % kTlBlocks][slot] = wall_clock64();  \
  } while (0)
#define RH_TL_VALUE(kernel, slot, v)                                                            \
  do {                                                                                          \
    if (threadIdx.x == 0) rh_tl[kernel][(blockIdx.y * gridDim.x + blockIdx.x) % kTlBlocks][slot] = (v); \
  } while (0)
#define RH_TL_DO(...) __VA_ARGS__
#else
#define RH_TL(kernel, slot)
#define RH_TL_VALUE(kernel, slot, v)
#define RH_TL_DO(...)
#endif

constexpr int kThreads = 256;
constexpr int kTileRows = 64;
constexpr int kSortThreads = 1024;
constexpr int kLdsSortMax = 16384;
   // keys of 8 bytes: 128 KiB of the CU's 160 KiB LDS

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

// Geometry of the per-level head tensors (see score_scan_levels_kernel).
constexpr int kMaxLevels = 8;
constexpr int kLvlChunk = 1024;    // anchor-cells per workgroup of the level scan = 16 tile lists
struct LevelGeom {
  const float* cls[kMaxLevels];    // [B, A*C, H, W]
  const float* reg[kMaxLevels];    // [B, A*4, H, W]
  int hw[kMaxLevels];
  int anchor_off[kMaxLevels + 1];  // first anchor of each level, then N
  int chunk_begin[kMaxLevels + 1];
  int vec4[kMaxLevels];
  int num_levels;
  int a;                           // anchors per cell
  // anchor n -> level, anchor slot, cell
  __device__ __forceinline__ void locate(const int n, int& l, int& as, int& cell) const {
    l = 0;
    while (l + 1 < num_levels && n >= anchor_off[l + 1]) ++l;
    const int r = n - anchor_off[l];
    cell = r / a;
    as = r - cell * a;
  }
  // tile list (within the image) that holds anchor n's candidate
  __device__ __forceinline__ int tile_of(const int n) const {
    int l, as, cell;
    locate(n, l, as, cell);
    const int u = as * hw[l] + cell;
    return (chunk_begin[l] + u / kLvlChunk) * (kLvlChunk / 64) + (u % kLvlChunk) / 64;
  }
};

// ---------------------------------------------------------------------------
// Score scan: per anchor row the maximum over classes and its first arg-max
// (model.py:323), kept only where it exceeds the score threshold (model.py:325).
// Candidates are written to small per-tile lists: a tile (64 consecutive rows of the
// flat [B*N, C] matrix) owns 64 slots and a count, so no atomics and no zeroing
// are needed, and the scan reads the classification tensor only.  The lists are stored
// slot-major ([slot][tile]): a tile rarely holds more than a few candidates, so readers
// that walk consecutive tiles find the occupied slots densely packed in a few cache lines.
// ---------------------------------------------------------------------------
struct TileCands {
  int* cnt;   // [num_tiles]
  u64* key;   // [64][num_tiles]  make_key(score, global row = image*N + anchor)
  int* lab;   // [64][num_tiles]  arg-max class
  int num_tiles;
  // Flat layout with at least 65 rows per image (else null / 0): a tile holds rows of at most
  // two images; split[j] = how many candidates of the tile that also holds the first rows of
  // image j + 1 belong to image j (they come first in its list).  Written by every scan, so
  // that readers need not look at the keys of an image's first and last tile to tell.
  int* split;       // [batch]
  int rows_per_image;
  __device__ __forceinline__ size_t at(int tile, int slot) const {
    return (size_t)slot * num_tiles + tile;
  }
};

// Wave 0 of a block, lane = row of the tile; s_best[lane] = (max score, ~class) key or 0.
__device__ __forceinline__ void emit_tile(const int tile, const int rows, const int row0,
                                          const u64* s_best, const float thresh,
                                          const TileCands tc) {
  const int lane = threadIdx.x & 63;
  const u64 best = s_best[lane];
  const float score = key_score(best);
  const bool cand = lane < rows && best != 0ull && score > thresh;
  const u64 m = __ballot(cand);
  if (cand) {
    const size_t at = tc.at(tile, __popcll(m & ((1ull << lane) - 1ull)));
    tc.key[at] = make_key(score, (unsigned)(row0 + lane));
    tc.lab[at] = (int)key_index(best);
  }
  if (lane == 0) {
    tc.cnt[tile] = __popcll(m);
    if (tc.split) {
      const int j = row0 / tc.rows_per_image;
      const int b = (j + 1) * tc.rows_per_image - row0;     // rows of image j in this tile
      if (b < rows) tc.split[j] = __popcll(m & ((1ull << b) - 1ull));
    }
  }
}

// Tile kernel: float4 stream over the tile.  A row is a candidate only if its maximum
// exceeds the threshold, and then the maximum is one of the (rare) elements above it: only
// those elements touch the per-row (max, first arg-max) slots, through 64-bit LDS atomic
// max (order-independent); everything else is one compare per float4: a pure read stream.
// Host guarantees 4 <= C < 8192, 16-byte aligned classifications.
// kLogits: the elements are logits.  A cheap test against `pre` (a logit safely below the one
// whose sigmoid equals the threshold) filters the stream; only elements that pass it are
// turned into probabilities, and everything after works on those, so arg-max ties and the
// threshold test are decided on float32 probabilities exactly as in probability mode.
// At most four waves per SIMD: with the eight its register count allows, the kernel is slower
// (54.3 against 50.9 us) -- more tiles in flight than the memory system turns over in order.
// One tile per workgroup: with 2 / 4 / 8 consecutive tiles each (fewer workgroups to dispatch) the
// whole post-processing call takes 92.8 / 95.5 / 101.5 instead of 91.2 us.  With 2 / 4 tiles side by side in
// workgroups of 8 / 16 waves: 91.5 / 89.9 against 91.0 us per call, but 66.4 / 67.8 against 65.8 us
// per batch with three batches in flight.
template <int U, bool kLogits>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 4))) void score_scan_tile_kernel(
    const float* __restrict__ prob, TileCands tc, int C, int total_rows, unsigned magic_c,
    float thresh, float pre) {
  __shared__ u64 s_best[kTileRows];
  // The scan is the stage bound by HBM; when other batches' sort / pair / resolve workgroups
  // share its SIMDs (several batches in flight), its waves get to issue their loads first
  // (four streams: 68.0 -> 66.8 us per batch; alone it makes no difference).
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int row0 = tile * kTileRows;
  const int rows = min(kTileRows, total_rows - row0);
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
        const float m = fmaxf(fmaxf(val[u].x, val[u].y), fmaxf(val[u].z, val[u].w));
        if (m > (kLogits ? pre : thresh)) {
          const unsigned x = (unsigned)v << 2;
          int r = (int)fast_div(x, magic_c);
          int col = (int)x - r * C;
          const float in[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col >= C) {
              col -= C;
              r += 1;
            }
            if (kLogits) {
              if (in[e] > pre) {
                const float pv = sigmoid_f32(in[e]);
                if (pv > thresh) atomicMax(&s_best[r], make_key(pv, (unsigned)col));
              }
            } else if (in[e] > thresh) {
              atomicMax(&s_best[r], make_key(in[e], (unsigned)col));
            }
            col += 1;
          }
        }
      }
    }
  }
  {
    const int x = (nvec << 2) + tid;   // < 4 trailing elements of the last tile
    if (x < elems) {
      const int r = x / C, col = x - r * C;
      const float xv = prob[base + x];
      const float pv = kLogits ? (xv > pre ? sigmoid_f32(xv) : 0.0f) : xv;
      if (pv > thresh) atomicMax(&s_best[r], make_key(pv, (unsigned)col));
    }
  }
  __syncthreads();
  if (tid < kTileRows) emit_tile(tile, rows, row0, s_best, thresh, tc);
}

// Generic kernel: any C and alignment; each wave reduces 16 of the tile's rows, scalar loads.
template <bool kLogits>
__global__ __launch_bounds__(kThreads) void score_scan_rows_kernel(const float* __restrict__ prob,
                                                                   TileCands tc, int C,
                                                                   int total_rows, float thresh) {
  __shared__ u64 s_best[kTileRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int row0 = tile * kTileRows;
  const int rows = min(kTileRows, total_rows - row0);
  for (int r = wave; r < kTileRows; r += kThreads / 64) {
    u64 best = 0ull;
    if (r < rows) {
      const float* p = prob + (size_t)(row0 + r) * C;
      for (int c = lane; c < C; c += 64) {
        const u64 k = make_key(kLogits ? sigmoid_f32(p[c]) : p[c], (unsigned)c);
        best = k > best ? k : best;
      }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const u64 other = __shfl_xor(best, s, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) s_best[r] = best;
  }
  __syncthreads();
  if (threadIdx.x < kTileRows) emit_tile(tile, rows, row0, s_best, thresh, tc);
}

// ---------------------------------------------------------------------------
// Score scan on the classification head's own outputs: per pyramid level a tensor
// [B, A*C, H, W], which the reference permutes, reshapes and concatenates into [B, N, C]
// (model.py:196-202, 311-313) before taking the class maximum (model.py:323).  Work unit
// u = a*HW + cell of a level; a workgroup owns 1024 consecutive units of one image (4 per
// thread) and walks all C class planes of them, each a contiguous run of cells, keeping the
// running (maximum, first arg-max) in registers: no atomics.  Candidates go to the same kind
// of tile lists as in the flat scan, one list per 64 consecutive units, with the key built on
// the flat row index b*N + n (n = off_l + cell*A + a) so that everything downstream (ties by
// anchor index, decode, outputs) is identical to the flat path.
// kLogits as in score_scan_tile_kernel.
// ---------------------------------------------------------------------------
template <bool kLogits>
__global__ __launch_bounds__(kThreads) void score_scan_levels_kernel(LevelGeom lv, TileCands tc,
                                                                     int N, int C, float thresh,
                                                                     float pre) {
  __builtin_amdgcn_s_setprio(3);      // as in score_scan_tile_kernel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.y;
  int l = 0;
  while (l + 1 < lv.num_levels && (int)blockIdx.x >= lv.chunk_begin[l + 1]) ++l;
  const int hw = lv.hw[l], A = lv.a;
  const int units = A * hw;
  const int s0 = ((int)blockIdx.x - lv.chunk_begin[l]) * kLvlChunk;
  const bool v4 = lv.vec4[l] != 0;
  const float* cp = lv.cls[l] + (size_t)j * A * C * hw;

  int base_c[4], un[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = v4 ? s0 + tid * 4 + k : s0 + tid + k * kThreads;
    ok[k] = u < units;
    // dead units load from a valid stand-in (on the float4 path the last aligned group of four)
    const int uc = ok[k] ? u : (v4 ? units - 4 + k : units - 1);
    un[k] = uc;
    const int a = uc / hw;
    base_c[k] = a * C * hw + (uc - a * hw);
  }
  float best[4];
  int bcls[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    best[k] = -INFINITY;
    bcls[k] = 0;
  }
  constexpr int U = 8;
  for (int c0 = 0; c0 < C; c0 += U) {
    float in[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t co = (size_t)min(c0 + u, C - 1) * hw;
      if (v4) {
        const v4f val = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(cp + base_c[0] + co));
        in[u][0] = val.x; in[u][1] = val.y; in[u][2] = val.z; in[u][3] = val.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) in[u][k] = cp[base_c[k] + co];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u;
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // only values over the threshold can matter; strict > keeps the first maximum
          if (in[u][k] > (kLogits ? pre : thresh)) {
            const float pv = kLogits ? sigmoid_f32(in[u][k]) : in[u][k];
            if (pv > best[k]) {
              best[k] = pv;
              bcls[k] = c;
            }
          }
        }
      }
    }
  }
  // ---- emit: one list per 64 consecutive units
  const int tiles_per_image = lv.chunk_begin[lv.num_levels] * (kLvlChunk / 64);
  const int tile0 = j * tiles_per_image + (int)blockIdx.x * (kLvlChunk / 64);
  bool cand[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cand[k] = ok[k] && best[k] > thresh;
  if (v4) {
    // a thread's units are consecutive; 16 lanes make one tile
    const int mine = (int)cand[0] + (int)cand[1] + (int)cand[2] + (int)cand[3];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int o = __shfl_up(incl, d, 16);
      if ((lane & 15) >= d) incl += o;
    }
    const int tile = tile0 + wave * 4 + (lane >> 4);
    int slot = incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (cand[k]) {
        const int a = un[k] / hw;
        const unsigned n = (unsigned)(lv.anchor_off[l] + (un[k] - a * hw) * A + a);
        const size_t at = tc.at(tile, slot++);
        tc.key[at] = make_key(best[k], (unsigned)j * (unsigned)N + n);
        tc.lab[at] = bcls[k];
      }
    }
    if ((lane & 15) == 15) tc.cnt[tile] = incl;
  } else {
    // units strided by the block size: for each k a wave covers one tile
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u64 m = __ballot(cand[k]);
      const int tile = tile0 + k * (kThreads / 64) + wave;
      if (cand[k]) {
        const int a = un[k] / hw;
        const unsigned n = (unsigned)(lv.anchor_off[l] + (un[k] - a * hw) * A + a);
        const size_t at = tc.at(tile, __popcll(m & ((1ull << lane) - 1ull)));
        tc.key[at] = make_key(best[k], (unsigned)j * (unsigned)N + n);
        tc.lab[at] = bcls[k];
      }
      if (lane == 0) tc.cnt[tile] = __popcll(m);
    }
  }
}

// out = sigmoid_f32(x): the probabilities the logits-in modes work with.
__global__ __launch_bounds__(kThreads) void sigmoid_kernel(const float* __restrict__ x,
                                                           float* __restrict__ out, long long count) {
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count; i += stride)
    out[i] = sigmoid_f32(x[i]);
}

// Keys for a plain (boxes, scores) NMS call: every box is a candidate.
__global__ __launch_bounds__(kThreads) void make_keys_kernel(const float* __restrict__ scores,
                                                             u64* __restrict__ keys, int k) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < k) keys[i] = make_key(scores[i], (unsigned)i);
}

// ---------------------------------------------------------------------------
// Where a sort finds its keys and how a sorted candidate becomes a box.
//   from_tiles != 0 (post-processing): keys are gathered from the image's tile lists; the
//     box is decoded from (anchor, regression) and clipped (utils.py:96-142), the label comes
//     from the tile list.
//   from_tiles == 0 (plain NMS): `count` keys already sit in the key buffer; boxes and labels
//     are the caller's arrays, indexed by the key's index.
// ---------------------------------------------------------------------------
struct SortSrc {
  TileCands tc;
  const float4* anchors;
  const float4* reg;
  BoxCoder bc;
  float width, height;
  const float4* boxes;
  const int* labels;
  const float* scores;   // plain NMS: the caller's scores (the rank sort builds its keys itself)
  int N;
  int count;
  int from_tiles;
  // head layout (from_tiles != 0 and tiles_per_image > 0): image j owns the tile lists
  // [j*T, (j+1)*T), which are NOT in anchor order, and regressions are the per-level tensors
  // [B, A*4, H, W] of `lv`
  int tiles_per_image;
  LevelGeom lv;
  // from_tiles != 0 and list_key != nullptr: topk_select_kernel has already copied the
  // candidates that can reach the top-k out of the tile lists: image j's are
  // list_key / list_lab [j*cap .. j*cap + list_count[j]), in no particular order
  const u64* list_key;
  const int* list_lab;
  const int* list_count;
};

// First and last tile list of image j.
__device__ __forceinline__ void image_tiles(const SortSrc& src, const int j, int& t_lo, int& t_hi) {
  if (src.tiles_per_image > 0) {
    t_lo = j * src.tiles_per_image;
    t_hi = t_lo + src.tiles_per_image - 1;
  } else {
    const long long first = (long long)j * src.N;
    t_lo = (int)(first >> 6);
    t_hi = (int)((first + src.N - 1) >> 6);
  }
}

// Regression deltas of anchor n of image j.
__device__ __forceinline__ float4 load_deltas(const SortSrc& src, const int j, const unsigned n,
                                              const unsigned idx) {
  if (src.tiles_per_image == 0) return src.reg[idx];
  int l, a, cell;
  src.lv.locate((int)n, l, a, cell);
  const int hw = src.lv.hw[l];
  const float* r = src.lv.reg[l] + ((size_t)j * src.lv.a + a) * 4 * hw + cell;
  return make_float4(r[0], r[hw], r[2 * (size_t)hw], r[3 * (size_t)hw]);
}

__device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// All threads of the block: copies image j's candidate keys into dst[0..cap) (LDS or global)
// and returns their number (which may exceed cap: the surplus is dropped, order is arbitrary).
__device__ __forceinline__ int gather_image_keys(const int j, const SortSrc src, u64* dst,
                                                 const int cap, int* s_count) {
  const long long first = (long long)j * src.N, last = first + src.N - 1;
  int t_lo, t_hi;
  image_tiles(src, j, t_lo, t_hi);
  if (threadIdx.x == 0) *s_count = 0;
  __syncthreads();
  for (int t = t_lo + (int)threadIdx.x; t <= t_hi; t += (int)blockDim.x) {
    const int c = src.tc.cnt[t];
    if (c == 0) continue;
    const bool edge = t == t_lo || t == t_hi;   // such a tile also holds a neighbour image's rows
    int mine = c;
    if (edge) {
      mine = 0;
      for (int r = 0; r < c; ++r) {
        const long long g = key_index(src.tc.key[src.tc.at(t, r)]);
        mine += (g >= first && g <= last);
      }
    }
    if (mine == 0) continue;
    int at = atomicAdd(s_count, mine);          // LDS counter
    for (int r = 0; r < c; ++r) {
      const u64 k = src.tc.key[src.tc.at(t, r)];
      const long long g = key_index(k);
      if (!edge || (g >= first && g <= last)) {
        if (at < cap) dst[at] = k;
        ++at;
      }
    }
  }
  __syncthreads();
  return *s_count;
}

// Where the sort stage leaves its results, per image j.
//   keys [j*key_stride + r], sbox / slabel [j*box_stride + r]: candidate of rank r (descending
//     score, then ascending index), r < boxes passed on;
//   nz [(j*box_stride + r)*nzw + g]: bitmap of the non-zero predecessor words of box r
//     (bit w of word g: word g*64 + w), zeroed here and raised by the pair kernel;
//   cbox / cinfo [j*list_stride + q]: the same boxes in LIST order q (the order the candidates
//     had before sorting: ascending anchor index from the score scan, the caller's order for a
//     plain NMS), with (rank or -1 when the box was cut by the top-k, label, bits of the box's
//     float32 area, 0).  Neighbours in
//     that order are neighbours in the image, which is what the pair kernel's culling needs;
//   csum [(j*clusters + g)*3 ..]: for cluster g = list entries [64g, 64g + 64): the bounding box of
//     its boxes that passed, then (smallest area, largest area, 0, 0), then the bounding box of
//     their centres.
//     All three are null when the sort path does not produce them;
//   mask [j][w][r]: the predecessor words; the rank sort zeroes the words the NMS can touch.
struct SortOut {
  u64* keys;
  float4* sbox;
  int* slabel;
  u64* nz;
  float4* cbox;
  int4* cinfo;
  float4* csum;
  u64* mask;
  int key_stride, box_stride, list_stride, nzw, wstride, clusters;
};

// Bounding box and area range of the wave's boxes (lanes with ok == false take no part); every
// lane gets the result.  An empty cluster gets an inverted box that nothing intersects.
struct ClusterSummary {
  float x1, y1, x2, y2, amin, amax;
  float cx1, cy1, cx2, cy2;      // bounding box of the boxes' centres
};
constexpr int kSummaryVecs = 3;  // float4 per cluster: box, (area min, area max, 0, 0), centres' box
__device__ __forceinline__ ClusterSummary summarize_cluster(const float4 b, const bool ok) {
  const float inf = __builtin_inff();
  const float area = (b.z - b.x) * (b.w - b.y);
  ClusterSummary c;
  c.x1 = wave_extreme<false>(ok ? b.x : inf);
  c.y1 = wave_extreme<false>(ok ? b.y : inf);
  c.x2 = wave_extreme<true>(ok ? b.z : -inf);
  c.y2 = wave_extreme<true>(ok ? b.w : -inf);
  c.amin = wave_extreme<false>(ok ? area : inf);
  c.amax = wave_extreme<true>(ok ? area : -inf);
  const float cx = 0.5f * (b.x + b.z), cy = 0.5f * (b.y + b.w);
  c.cx1 = wave_extreme<false>(ok ? cx : inf);
  c.cy1 = wave_extreme<false>(ok ? cy : inf);
  c.cx2 = wave_extreme<true>(ok ? cx : -inf);
  c.cy2 = wave_extreme<true>(ok ? cy : -inf);
  return c;
}

// Writes sorted candidate i of image j: its key, box and label.
__device__ __forceinline__ void store_sorted(const u64 key, const int i, const int j,
                                             const SortSrc src, const SortOut so) {
  so.keys[(size_t)j * so.key_stride + i] = key;
  const size_t o = (size_t)j * so.box_stride + i;
  for (int g = 0; g < so.nzw; ++g) so.nz[o * so.nzw + g] = 0ull;
  const unsigned idx = key_index(key);
  if (src.from_tiles) {
    const unsigned n = idx - (unsigned)j * (unsigned)src.N;
    so.sbox[o] = clip_box(decode_box(src.anchors[n], load_deltas(src, j, n, idx), src.bc), src.width,
                          src.height);
    const unsigned t = src.tiles_per_image > 0
                           ? (unsigned)(j * src.tiles_per_image + src.lv.tile_of((int)n))
                           : idx >> 6;
    const int c = src.tc.cnt[t];
    int lab = 0;
    for (int r = 0; r < c; ++r)
      if (src.tc.key[src.tc.at((int)t, r)] == key) lab = src.tc.lab[src.tc.at((int)t, r)];
    so.slabel[o] = lab;
  } else {
    so.sbox[o] = src.boxes[idx];
    so.slabel[o] = src.labels ? src.labels[idx] : 0;
  }
}

// All compare-exchange steps of a bitonic network whose partner distance is below 64 stay
// inside 64 consecutive keys: a wave holds them one per lane and exchanges by shuffles, no LDS
// round trip or barrier.  Runs the steps (k, jj) for jj = jj_first, jj_first/2, ..., 1.
__device__ __forceinline__ u64 bitonic_wave_steps(u64 key, const int index, const int k,
                                                  const int jj_first) {
  const int lane = threadIdx.x & 63;
  for (int jj = jj_first; jj > 0; jj >>= 1) {
    const u64 other = __shfl_xor(key, jj, 64);
    const bool desc = (index & k) == 0;
    const bool lower = (lane & jj) == 0;
    const u64 hi = key > other ? key : other, lo = key > other ? other : key;
    key = (lower == desc) ? hi : lo;
  }
  return key;
}

// One workgroup per image; the whole key array lives in LDS (bitonic network, descending).
// counters: [3][B] = candidates found, candidates dropped for lack of room, boxes passed on.
__global__ __launch_bounds__(kSortThreads) void sort_lds_kernel(int* __restrict__ counters, int batch,
                                                                int cap, int topk, SortSrc src,
                                                                SortOut so) {
  const int key_stride = so.key_stride;
  // one dynamic LDS array: key_stride keys, then the gather counter (a separate static
  // __shared__ object could shift the dynamic base off its 16-byte alignment)
  extern __shared__ __attribute__((aligned(16))) u64 s_keys[];
  int* s_count = reinterpret_cast<int*>(&s_keys[key_stride]);
  const int j = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  u64* kj = so.keys + (size_t)j * key_stride;
  int found;
  if (src.from_tiles) {
    found = gather_image_keys(j, src, s_keys, cap, s_count);
  } else {
    found = src.count;
    for (int i = tid; i < min(found, cap); i += kSortThreads) s_keys[i] = kj[i];
    __syncthreads();
  }
  const int K = min(found, cap);
  const int P = max(next_pow2(max(K, 1)), 64);
  // head of the network: stages k = 2..64, entirely in registers
  for (int seg = wave * 64; seg < P; seg += kSortThreads) {
    const int i = seg + lane;
    u64 key = i < K ? s_keys[i] : 0ull;
    for (int k = 2; k <= 64; k <<= 1) key = bitonic_wave_steps(key, i, k, k >> 1);
    s_keys[i] = key;
  }
  __syncthreads();
  for (int k = 128; k <= P; k <<= 1) {
    for (int jj = k >> 1; jj >= 64; jj >>= 1) {      // partner in another 64-key segment: LDS
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
    for (int seg = wave * 64; seg < P; seg += kSortThreads) {   // jj = 32..1: in registers
      const int i = seg + lane;
      s_keys[i] = bitonic_wave_steps(s_keys[i], i, k, 32);
    }
    __syncthreads();
  }
  const int ke = topk > 0 ? min(K, topk) : K;
  for (int i = tid; i < ke; i += kSortThreads) store_sorted(s_keys[i], i, j, src, so);
  if (tid == 0) {
    counters[j] = found;
    counters[batch + j] = max(found - cap, 0);
    counters[2 * batch + j] = ke;
  }
}

// ---------------------------------------------------------------------------
// Top-k pre-selection (one workgroup per image), for calls that keep far fewer candidates than
// the score scan finds (pre_nms_topk = 1000 of ~5600 at 1024 px): ranking a key costs a pass
// over the whole list, so the list is first cut down to the keys that can reach the top-k.
// A histogram over the upper bits of the scores (bins of 2^16 float32 steps, 1/128 of a binade;
// the bin is a non-decreasing function of the score) gives the lowest bin b* such that the bins
// from b* up hold at least topk keys; every key below b* has a smaller score than topk others
// and cannot be among the top-k, whatever the tie-break.  The keys from b* up (topk of them
// plus the rest of bin b*) are copied out in tile order -- anchor order but for the order inside
// a tile, which is what the pair kernel's culling likes -- and their ranks among themselves are
// their ranks among all.
// counters [j] = candidates found, [batch + j] = selected keys that did not fit `cap`.
// ---------------------------------------------------------------------------
constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / 64;
constexpr int kSelBins = 640;            // scores 1/32 (0x3D000000) .. 1 (0x3F800000), bits >> 16
constexpr int kSelTiles = 3, kSelSlots = 8;
// The upper half of a key is the score in its order-preserving form (sign bit set for positive
// floats), so its top 16 bits order the keys like the scores; below 1/32 everything is bin 0.
__device__ __forceinline__ int score_bin(const u64 key) {
  const int b = (int)(key >> 48) - (int)((0x3D000000u ^ 0x80000000u) >> 16);
  return min(max(b, 0), kSelBins - 1);
}

// Wave w owns a contiguous run of the image's tiles; lane l holds tiles run + u*64 + l for
// u < kSelTiles (consecutive lanes, consecutive tiles: coalesced loads -- with a run of
// consecutive tiles per THREAD the one compute unit that serves an image spent 8 us issuing
// partly used cache lines), the first kSelSlots list entries of each, requested together with
// the tile counts (one trip to memory, whatever the counts turn out to be) and kept in
// registers for all three passes of the kernel.  The rare rest (longer tile lists, longer runs:
// images beyond 3072 tiles) is fetched again by each pass.
struct SelectHeld {
  int c[kSelTiles];
  u64 k[kSelTiles][kSelSlots];
  int l[kSelTiles][kSelSlots];
};

struct SelectRun {
  int wb, we;       // the wave's tiles [wb, we)
  long long first, last;
};

// fn(key, label) for the candidates of held tile u (this image's only).
template <typename F>
__device__ __forceinline__ void select_visit_held(const SelectHeld& h, const int u,
                                                  const TileCands& tc, const SelectRun& run,
                                                  F&& fn) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < kSelSlots; ++r) {
    const long long g = key_index(h.k[u][r]);
    if (r < h.c[u] && g >= run.first && g <= run.last) fn(h.k[u][r], h.l[u][r]);
  }
  for (int r = kSelSlots; r < h.c[u]; ++r) {
    const size_t o = tc.at(run.wb + u * 64 + lane, r);
    const u64 k = tc.key[o];
    const long long g = key_index(k);
    if (g >= run.first && g <= run.last) fn(k, tc.lab[o]);
  }
}

// ... and for the lane's tiles beyond the held ones.
template <typename F>
__device__ __forceinline__ void select_visit_rest(const TileCands& tc, const SelectRun& run, F&& fn) {
  for (int t = run.wb + kSelTiles * 64 + (int)(threadIdx.x & 63); t < run.we; t += 64) {
    const int c = tc.cnt[t];
    for (int r = 0; r < c; ++r) {
      const size_t o = tc.at(t, r);
      const u64 k = tc.key[o];
      const long long g = key_index(k);
      if (g >= run.first && g <= run.last) fn(k, tc.lab[o]);
    }
  }
}

__global__ __launch_bounds__(kSelThreads) void topk_select_kernel(
    int* __restrict__ counters, int batch, int cap, int topk, SortSrc src, u64* __restrict__ sel_key,
    int* __restrict__ sel_lab, int* __restrict__ sel_count) {
  __shared__ int s_hist[kSelBins];
  __shared__ int s_wtot[kSelWaves];
  __shared__ int s_bstar, s_kept;
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TileCands tc = src.tc;
  int t_lo, t_hi;
  image_tiles(src, j, t_lo, t_hi);
  SelectRun run;
  run.first = (long long)j * src.N;
  run.last = run.first + src.N - 1;
  const int chunk = ((t_hi - t_lo + kSelThreads) / kSelThreads) * 64;
  run.wb = min(t_lo + wave * chunk, t_hi + 1);
  run.we = min(run.wb + chunk, t_hi + 1);
  RH_TL(0, 5);
  SelectHeld h;
#pragma unroll
  for (int u = 0; u < kSelTiles; ++u) {
    const int t = min(run.wb + u * 64 + lane, t_hi);
    h.c[u] = tc.cnt[t];
#pragma unroll
    for (int r = 0; r < kSelSlots; ++r) {
      h.k[u][r] = tc.key[tc.at(t, r)];
      h.l[u][r] = tc.lab[tc.at(t, r)];
    }
  }
  for (int b = tid; b < kSelBins; b += kSelThreads) s_hist[b] = 0;
  if (tid == 0) {
    s_bstar = 0;
    s_kept = 0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kSelTiles; ++u)
    if (run.wb + u * 64 + lane >= run.we) h.c[u] = 0;
  RH_TL(0, 6);
  auto count_bin = [&](const u64 key, int) { atomicAdd(&s_hist[score_bin(key)], 1); };
#pragma unroll
  for (int u = 0; u < kSelTiles; ++u) select_visit_held(h, u, tc, run, count_bin);
  select_visit_rest(tc, run, count_bin);
  __syncthreads();
  RH_TL(0, 7);
  // S(b) = keys in bins >= b; b* = the highest b with S(b) >= topk (bin 0 when there are fewer)
  int suf = 0;
  if (wave < kSelBins / 64) {
    suf = s_hist[wave * 64 + lane];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_down(suf, d, 64);
      if (lane + d < 64) suf += o;
    }
    if (lane == 0) s_wtot[wave] = suf;
  }
  __syncthreads();
  int above = 0, found = 0;
#pragma unroll
  for (int w = 0; w < kSelBins / 64; ++w) {
    above += w > wave ? s_wtot[w] : 0;
    found += s_wtot[w];
  }
  if (wave < kSelBins / 64) {
    const unsigned long long reach = __ballot(above + suf >= topk);
    if (reach != 0ull && lane == 63 - __builtin_clzll(reach)) atomicMax(&s_bstar, wave * 64 + lane);
  }
  __syncthreads();
  const int bstar = s_bstar;
  if (wave == bstar / 64 && lane == bstar % 64) s_kept = above + suf;     // S(b*)
  RH_TL(0, 8);
  // List positions, in the order tile group 0 (lane 0..63), group 1, ..., then the rest per lane:
  // tile order as long as the held tiles cover the run.
  int kc[kSelTiles + 1], at[kSelTiles + 1];
#pragma unroll
  for (int u = 0; u <= kSelTiles; ++u) kc[u] = 0;
#pragma unroll
  for (int u = 0; u < kSelTiles; ++u)
    select_visit_held(h, u, tc, run, [&](const u64 key, int) { kc[u] += score_bin(key) >= bstar; });
  select_visit_rest(tc, run, [&](const u64 key, int) { kc[kSelTiles] += score_bin(key) >= bstar; });
  int wave_total = 0;
#pragma unroll
  for (int u = 0; u <= kSelTiles; ++u) {
    int incl = kc[u];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    at[u] = wave_total + incl - kc[u];
    wave_total += __builtin_amdgcn_readlane(incl, 63);
  }
  if (lane == 0) s_wtot[wave] = wave_total;      // (its earlier readers are past the barrier above)
  __syncthreads();
  int wave_base = 0;
#pragma unroll
  for (int w = 0; w < kSelWaves; ++w) wave_base += w < wave ? s_wtot[w] : 0;
  RH_TL(0, 9);
  u64* kj = sel_key + (size_t)j * cap;
  int* lj = sel_lab + (size_t)j * cap;
#pragma unroll
  for (int u = 0; u <= kSelTiles; ++u) {
    int pos = wave_base + at[u];
    auto put = [&](const u64 key, const int lab) {
      if (score_bin(key) >= bstar) {
        if (pos < cap) {
          kj[pos] = key;
          lj[pos] = lab;
        }
        ++pos;
      }
    };
    if (u < kSelTiles) select_visit_held(h, u < kSelTiles ? u : 0, tc, run, put);
    else select_visit_rest(tc, run, put);
  }
  RH_TL(0, 10);
  if (tid == 0) {
    const int kept = s_kept;
    sel_count[j] = kept;
    counters[j] = found;
    counters[batch + j] = max(kept - cap, 0);
  }
}

// ---------------------------------------------------------------------------
// Rank sort for candidate lists of up to kRankMax keys per image.  Keys are unique (the
// index is part of the key), so a key's position in descending order is the number of keys
// greater than it: no exchange network, no serial dependency, and the work spreads over the
// chip instead of one workgroup per image.
// The grid holds a fixed number of blocks per image (at most one block per compute unit in
// all).  Every block of an image first rebuilds the same candidate list in LDS (tile lists
// compacted in tile order by a block-wide prefix sum, so all blocks see one order, which is
// ascending anchor index) and a histogram of the scores' upper 16 bits with it; the bins, laid
// out from the highest score down, then receive the list positions of their entries
// (RankBuckets).  The image's clusters of 64 list entries are dealt evenly to its blocks; a
// wave takes a cluster, lane l its entry l, whose rank is the number of entries in the bins
// above plus the greater keys in its own bin -- a handful of comparisons.  The owner then
// decodes the box and writes (key, box, label) at its rank.
// When some bin is too full for that (equal scores, scores outside (1/32, 1]) the block counts
// over the whole list instead: each of its eight waves counts, for the block's entries, the
// greater keys in its eighth of the list (uniform addresses: LDS broadcast reads; because the
// list is in index order, "greater" is a 32-bit test on the score alone outside the block's
// own entries).
// ---------------------------------------------------------------------------
constexpr int kRankMax = 8192;   // 14 bytes of LDS per key: 112 KiB of the CU's 160
#ifndef RH_RANK_KPL
#define RH_RANK_KPL 2
#endif
constexpr int kRankKpl = RH_RANK_KPL;       // least keys per lane: clusters of 64 list entries per block
constexpr int kRankKeys = 64 * kRankKpl;
#ifndef RH_RANK_KPL_MAX
#define RH_RANK_KPL_MAX 4
#endif
// Most clusters one pass of a block handles (at most its 8 waves).  Measured with the bucketed
// ranks, whole call on one stream / three batches in flight: (kRankKpl, kRankKplMax) = (2, 4) 92.4 /
// 68.4 us; (4, 4) 92.6 / 68.0-72.4; (4, 8) 92.9 / 68.3; (8, 8) 94.7 / 68.4-72.6.  Requesting the
// first pass's anchors and regressions before the bucketing (with barriers that do not wait for
// global loads) changed nothing.
constexpr int kRankKplMax = RH_RANK_KPL_MAX;
#ifndef RH_RANK_THREADS
#define RH_RANK_THREADS 512
#endif
constexpr int kRankThreads = RH_RANK_THREADS;
constexpr int kRankWaves = kRankThreads / 64;
constexpr int kRankBatch = 3;   // groups of 64 tiles whose lists a wave loads together
#ifndef RH_RANK_SLOTS
#define RH_RANK_SLOTS 4
#endif
// list slots per tile loaded together.  Measured, whole call on one stream: 2 or 4 slots 97.0 /
// 97.4 us; 8 slots 106.4; 8 slots requesting only those a tile has filled 99.1
constexpr int kRankSlots = RH_RANK_SLOTS;

// mode 0: score >= mine, 1: score > mine, 2: full (score, index) key > mine; every list entry
// read is compared with the lane's kRankKpl keys
template <int kMode, int NC>
__device__ __forceinline__ void rank_count(const unsigned* s_sk, const unsigned* s_ix, int qa,
                                           const int qb, const unsigned (&mys)[NC],
                                           const unsigned (&myx)[NC], int (&rank)[NC]) {
  // eight entries per round as 16-byte LDS reads (uniform address: broadcast).  Every caller's
  // qa is a multiple of 8 whenever the range is not empty.
  for (; qa + 8 <= qb; qa += 8) {
    const uint4 s0 = *reinterpret_cast<const uint4*>(s_sk + qa);
    const uint4 s1 = *reinterpret_cast<const uint4*>(s_sk + qa + 4);
    const unsigned sk8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    unsigned ix8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (kMode == 2) {
      const uint4 i0 = *reinterpret_cast<const uint4*>(s_ix + qa);
      const uint4 i1 = *reinterpret_cast<const uint4*>(s_ix + qa + 4);
      ix8[0] = i0.x; ix8[1] = i0.y; ix8[2] = i0.z; ix8[3] = i0.w;
      ix8[4] = i1.x; ix8[5] = i1.y; ix8[6] = i1.z; ix8[7] = i1.w;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned sk = sk8[u];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (kMode == 0) rank[c] += sk >= mys[c];
        if (kMode == 1) rank[c] += sk > mys[c];
        if (kMode == 2) rank[c] += (((u64)sk << 32) | ix8[u]) > (((u64)mys[c] << 32) | myx[c]);
      }
    }
  }
  for (; qa < qb; ++qa) {
    const unsigned sk = s_sk[qa];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (kMode == 0) rank[c] += sk >= mys[c];
      if (kMode == 1) rank[c] += sk > mys[c];
      if (kMode == 2) rank[c] += (((u64)sk << 32) | s_ix[qa]) > (((u64)mys[c] << 32) | myx[c]);
    }
  }
}

__host__ __device__ inline int rank_lds_keys(int cap) { return (cap + 63) & ~63; }
__host__ __device__ inline size_t rank_lds_bytes(int cap) { return (size_t)rank_lds_keys(cap) * 14; }

// The list bucketed by score: bin b (rank_bin, monotone in the score; kSelBins of them) owns the
// slice [start[b], cursor[b]) of `idx`, list positions of its entries in no particular order, the
// bins with the higher scores first.  A key's rank is start[its bin] plus the greater keys of
// its own bin: a handful of comparisons instead of one with every entry of the list.
#ifndef RH_BUCKET_ROUND
#define RH_BUCKET_ROUND 1
#endif
// entries of a bin compared per round (whole call on one stream: 1: 92.6 us, 2: 94.1, 4: 93.5, 8: 93.4;
// bins hold a few entries each)
constexpr int kBucketRound = RH_BUCKET_ROUND;
struct RankBuckets {
  const int* start;
  const int* cursor;
  const unsigned short* idx;
  bool on;      // block-uniform; off when some bin is so full that counting over the list is cheaper
};
__device__ __forceinline__ int rank_bin(const unsigned score_key) {
  const int b = (int)(score_key >> 16) - (int)((0x3D000000u ^ 0x80000000u) >> 16);
  return min(max(b, 0), kSelBins - 1);
}

// (Measured and left as it is: zeroing the first pass's predecessor words before the bins are laid
// out, and requesting an entry's anchor and regression before the comparisons inside its bin --
// the whole call stays at 92.4 us.  Of a block's 11 us, 1.8 are the counting pass, 4.0 the
// gather, 1.1 laying out the bins, 0.5 filling them, 0.9 zeroing the words, 2.7 the ranks, boxes
// and stores.)
// One pass of a rank-sort block: counts the ranks of the NC clusters [c0, c0 + NC) of image j's
// list (in LDS: s_sk, s_ix, s_lab; K entries) and writes what the NMS stage reads for them.
template <int NC>
__device__ __forceinline__ void rank_pass(const int c0, const bool writes_counters, const int j,
                                          const int K, const int ke, const int found,
                                          const int cap, const int batch,
                                          const unsigned* s_sk, const unsigned* s_ix,
                                          const int* s_lab, int (*s_part)[64 * kRankKplMax],
                                          int* __restrict__ counters, const SortSrc& src,
                                          const SortOut& so, const RankBuckets bk) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p0 = c0 * 64;
  if (!bk.on) {
  // lane l counts for list entries p0 + c*64 + l, c < NC: one pass over the list serves them all
  unsigned mys[NC], myx[NC];
  int rank[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int pc = min(p0 + c * 64 + lane, max(K - 1, 0));
    mys[c] = s_sk[pc];
    myx[c] = s_ix[pc];
    rank[c] = 0;
  }
  // wave w counts in [q0, q1): equal shares of the list, multiples of 8 entries
  const int seg = ((K + 8 * kRankWaves - 1) / (8 * kRankWaves)) << 3;
  const int q0 = min(wave * seg, K), q1 = min(q0 + seg, K);
  if (src.from_tiles && (src.tiles_per_image > 0 || src.list_key)) {
    rank_count<2, NC>(s_sk, s_ix, q0, q1, mys, myx, rank);      // list not in anchor order: full keys
  } else {
    rank_count<0, NC>(s_sk, s_ix, q0, min(q1, p0), mys, myx, rank);
    rank_count<2, NC>(s_sk, s_ix, max(q0, p0), min(q1, p0 + NC * 64), mys, myx, rank);
    rank_count<1, NC>(s_sk, s_ix, max(q0, p0 + NC * 64), q1, mys, myx, rank);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) s_part[wave][c * 64 + lane] = rank[c];
  }
  // The pair kernel ORs bits into the predecessor words: zero the words of the ranks
  // [p0, p0 + NC * 64) (whichever keys end up there), for rank r the words 0 .. r/64, one
  // 512-byte row per store.
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int r = p0 + c * 64 + lane;
    if (r < ke)
      for (int w = wave; w <= c0 + c; w += kRankWaves)
        so.mask[((size_t)j * so.wstride + w) * so.box_stride + r] = 0ull;
  }
  if (!bk.on) __syncthreads();      // (block-uniform) the partial counts
  RH_TL(0, 3);
  if (wave < NC) {            // wave c finishes cluster c of the pass
    const int cluster = c0 + wave;
    const int p = p0 + wave * 64 + lane;
    const int pc = min(p, max(K - 1, 0));
    const unsigned ms = s_sk[pc], mx = s_ix[pc];
    int rk = 0;
    if (bk.on) {
      const int b = rank_bin(ms);
      const u64 mykey = ((u64)ms << 32) | (u64)mx;
      const int qe = bk.cursor[b];
      int q = p < K ? bk.start[b] : qe;
      rk = q;                                  // the entries of the bins above
      for (; q < qe; q += kBucketRound) {    // the reads of a round do not depend on each other
        int e[kBucketRound];
#pragma unroll
        for (int u = 0; u < kBucketRound; ++u) e[u] = bk.idx[min(q + u, qe - 1)];
        unsigned es[kBucketRound], ex[kBucketRound];
#pragma unroll
        for (int u = 0; u < kBucketRound; ++u) {
          es[u] = s_sk[e[u]];
          ex[u] = s_ix[e[u]];
        }
#pragma unroll
        for (int u = 0; u < kBucketRound; ++u)
          rk += q + u < qe && ((((u64)es[u]) << 32) | (u64)ex[u]) > mykey;
      }
    } else {
#pragma unroll
      for (int w = 0; w < kRankWaves; ++w) rk += s_part[w][wave * 64 + lane];
    }
    const bool passed = p < K && rk < ke;      // else beyond the list, or cut by the top-k
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < K) {
      const size_t q = (size_t)j * so.list_stride + p;
      if (passed) {
        const size_t o = (size_t)j * so.box_stride + rk;
        so.keys[(size_t)j * so.key_stride + rk] = ((u64)ms << 32) | (u64)mx;
        const unsigned idx = 0xFFFFFFFFu - mx;
        if (src.from_tiles) {
          const unsigned n = idx - (unsigned)j * (unsigned)src.N;
          box = clip_box(decode_box(src.anchors[n], load_deltas(src, j, n, idx), src.bc), src.width,
                         src.height);
        } else {
          box = src.boxes[idx];
        }
        so.sbox[o] = box;
        if (so.cbox) so.cbox[q] = box;
        so.slabel[o] = s_lab[p];
        for (int g = 0; g < so.nzw; ++g) so.nz[o * so.nzw + g] = 0ull;
      }
      if (so.cbox)
        so.cinfo[q] = make_int4(passed ? rk : -1, s_lab[p],
                                __float_as_int((box.z - box.x) * (box.w - box.y)), 0);
    }
    // each 64 list entries are one cluster of the pair kernel
    if (cluster < so.clusters) {
      const ClusterSummary cs = summarize_cluster(box, passed);
      if (lane == 0) {
        float4* out = so.csum + ((size_t)j * so.clusters + cluster) * kSummaryVecs;
        out[0] = make_float4(cs.x1, cs.y1, cs.x2, cs.y2);
        out[1] = make_float4(cs.amin, cs.amax, 0.f, 0.f);
        out[2] = make_float4(cs.cx1, cs.cy1, cs.cx2, cs.cy2);
      }
    }
    if (writes_counters && wave == 0 && lane == 0) {
      if (!src.list_key) {           // (the selection kernel has written these two)
        counters[j] = found;
        counters[batch + j] = max(found - cap, 0);
      }
      counters[2 * batch + j] = ke;
    }
    RH_TL(0, 4);
  }
}

__global__ __launch_bounds__(kRankThreads) void sort_rank_kernel(int* __restrict__ counters, int batch,
                                                                 int cap, int topk, SortSrc src,
                                                                 SortOut so) {
  // dynamic LDS, sized by the list capacity: score halves of the keys, index halves (~index),
  // labels
  extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
  unsigned* s_sk = s_dyn;
  unsigned* s_ix = s_dyn + rank_lds_keys(cap);
  int* s_lab = reinterpret_cast<int*>(s_dyn + 2 * rank_lds_keys(cap));
  __shared__ int s_wsum[kRankWaves];
  __shared__ int s_part[kRankWaves][64 * kRankKplMax];
  __shared__ int s_start[kSelBins], s_cursor[kSelBins], s_ctot[kSelBins / 64], s_big;
  // the image index is the fast grid dimension: blocks beyond an image's candidate count
  // (most of the grid when the lists are far from full) are dispatched last and exit at once
  const int j = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The grid holds gridDim.y blocks per image, sized by the host to the machine rather than to
  // the list capacity; the image's clusters are dealt to them evenly once the list's length is
  // known (at least kRankKpl per block, so short lists leave the later blocks without work).
  auto clusters_per_block = [&](const int k) {
    return max(kRankKpl, (((k + 63) >> 6) + (int)gridDim.y - 1) / (int)gridDim.y);
  };
  auto nothing_to_rank = [&](const int found_now) {     // block-uniform
    const int k = min(found_now, cap);
    return blockIdx.y != 0 && (int)blockIdx.y * clusters_per_block(k) * 64 >= k;
  };
  RH_TL(0, 0);
  // score histogram of the list (see RankBuckets), filled as the entries arrive in LDS; the
  // barriers of the list's construction order the zeroing before the first increment
  for (int b = tid; b < kSelBins; b += kRankThreads) s_cursor[b] = 0;
  if (tid == 0) s_big = 0;
  int found;
  if (src.from_tiles && !src.list_key) {
    const TileCands tc = src.tc;
    const long long first = (long long)j * src.N, last = first + src.N - 1;
    int t_lo, t_hi;
    image_tiles(src, j, t_lo, t_hi);
    // Wave w compacts a contiguous range of the image's tiles, 64 tiles (one per lane) at a
    // time, so list loads are coalesced.  Loads are unconditional (tile and slot indices
    // clamped) and issued kRankBatch x kRankSlots at a time ahead of their use: a chain of
    // guarded, dependent loads would cost one memory latency per tile and per key.
    // Rows of a tile are listed in ascending order, so in the image's first tile the previous
    // image's candidates come first (skip_lo of them) and in its last tile the next image's
    // come last.
    const int chunk = ((t_hi - t_lo + kRankThreads) / kRankThreads) * 64;
    const int wb = min(t_lo + wave * chunk, t_hi + 1), we = min(wb + chunk, t_hi + 1);
    // Does the image's first / last tile also hold rows of the previous / next image?  With
    // the scan's split counts (or tile lists per image) the counting pass needs no keys.
    const bool own_tiles = src.tiles_per_image > 0;
    const bool lo_shared = !own_tiles && (first & 63) != 0;
    const bool hi_shared = !own_tiles && ((last + 1) & 63) != 0 && j + 1 < batch;
    int split_lo = 0, split_hi = 0;
    if (tc.split) {
      if (lo_shared) split_lo = tc.split[j - 1];
      if (hi_shared) split_hi = tc.split[j];
    }
    // a wave whose tiles fit one batch of loads keeps their counts for the second pass
    const bool one_batch = we - wb <= 64 * kRankBatch;
    int craw[kRankBatch];
    int mine = 0, skip_lo = 0;
    for (int i0 = wb; i0 < we; i0 += 64 * kRankBatch) {
      int c[kRankBatch];
#pragma unroll
      for (int u = 0; u < kRankBatch; ++u) c[u] = tc.cnt[min(i0 + u * 64 + lane, t_hi)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kRankBatch; ++u) {
        const int t = i0 + u * 64 + lane;
        craw[u] = c[u];
        if (t >= we) c[u] = 0;
        if (own_tiles || tc.split) {
          // (more than 64 rows per image: its first and last tile differ)
          if (t == t_lo && lo_shared) {
            skip_lo = min(split_lo, c[u]);
            c[u] -= skip_lo;
          }
          if (t == t_hi && hi_shared) c[u] = min(split_hi, c[u]);
        } else if (c[u] != 0 && (t == t_lo || t == t_hi)) {
          int m = 0, before = 0;
          for (int r0 = 0; r0 < c[u]; r0 += 8) {
            u64 k[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) k[r] = tc.key[tc.at(t, min(r0 + r, kTileRows - 1))];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const long long g = key_index(k[r]);
              m += (r0 + r < c[u] && g >= first && g <= last);
              before += (r0 + r < c[u] && g < first);
            }
          }
          c[u] = m;
          if (t == t_lo) skip_lo = before;
        }
        mine += c[u];
      }
    }
    mine = wave_sum(mine);
    if (lane == 0) s_wsum[wave] = mine;
    __syncthreads();
    int run = 0;            // list position of this wave's next entry
    found = 0;
#pragma unroll
    for (int w = 0; w < kRankWaves; ++w) {
      run += w < wave ? s_wsum[w] : 0;
      found += s_wsum[w];
    }
    RH_TL(0, 1);
    if (nothing_to_rank(found)) return;
    for (int i0 = wb; i0 < we; i0 += 64 * kRankBatch) {
      int c[kRankBatch], lo[kRankBatch], base[kRankBatch];
      if (one_batch) {
#pragma unroll
        for (int u = 0; u < kRankBatch; ++u) c[u] = craw[u];
      } else {
#pragma unroll
        for (int u = 0; u < kRankBatch; ++u) c[u] = tc.cnt[min(i0 + u * 64 + lane, t_hi)];
        __builtin_amdgcn_sched_barrier(0);
      }
      int cm = 0;
#pragma unroll
      for (int u = 0; u < kRankBatch; ++u) {
        const int t = i0 + u * 64 + lane;
        if (t >= we) c[u] = 0;
        lo[u] = t == t_lo ? min(skip_lo, c[u]) : 0;
        // entries this tile adds (too many for the image's last tile, after which nothing follows)
        const int eff = c[u] - lo[u];
        int incl = eff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        base[u] = run + incl - eff;
        run += __builtin_amdgcn_readlane(incl, 63);
        cm = max(cm, c[u]);
      }
      for (int r0 = 0; __ballot(r0 < cm) != 0ull; r0 += kRankSlots) {
        u64 k[kRankBatch][kRankSlots];
        int l[kRankBatch][kRankSlots];
#pragma unroll
        for (int u = 0; u < kRankBatch; ++u) {
          const int t = min(i0 + u * 64 + lane, t_hi);
#pragma unroll
          for (int r = 0; r < kRankSlots; ++r) {
            const size_t o = tc.at(t, min(r0 + r, kTileRows - 1));
            k[u][r] = tc.key[o];
            l[u][r] = tc.lab[o];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kRankBatch; ++u) {
#pragma unroll
          for (int r = 0; r < kRankSlots; ++r) {
            const int rr = r0 + r;
            const long long g = key_index(k[u][r]);
            const int pos = base[u] + rr - lo[u];
            if (rr >= lo[u] && rr < c[u] && g <= last && pos < cap) {
              s_sk[pos] = (unsigned)(k[u][r] >> 32);
              atomicAdd(&s_cursor[rank_bin((unsigned)(k[u][r] >> 32))], 1);
              s_ix[pos] = (unsigned)k[u][r];
              s_lab[pos] = l[u][r];
            }
          }
        }
      }
    }
  } else if (src.list_key) {
    found = src.list_count[j];
    if (nothing_to_rank(found)) return;
    __syncthreads();                      // the histogram is zero
    for (int i = tid; i < min(found, cap); i += kRankThreads) {
      const u64 k = src.list_key[(size_t)j * cap + i];
      s_sk[i] = (unsigned)(k >> 32);
      atomicAdd(&s_cursor[rank_bin((unsigned)(k >> 32))], 1);
      s_ix[i] = (unsigned)k;
      s_lab[i] = src.list_lab[(size_t)j * cap + i];
    }
  } else {
    found = src.count;
    if (nothing_to_rank(found)) return;
    __syncthreads();                      // the histogram is zero
    for (int i = tid; i < min(found, cap); i += kRankThreads) {
      const u64 k = make_key(src.scores[i], (unsigned)i);
      s_sk[i] = (unsigned)(k >> 32);
      atomicAdd(&s_cursor[rank_bin((unsigned)(k >> 32))], 1);
      s_ix[i] = (unsigned)k;
      s_lab[i] = src.labels ? src.labels[i] : 0;
    }
  }
  __syncthreads();
  RH_TL(0, 2);
  const int K = min(found, cap);
  const int ke = topk > 0 ? min(K, topk) : K;
  // ---- bucket the list by score: counts per bin, bins laid out from the highest down, then
  // every entry's list position into its bin's slice
  unsigned short* b_idx = reinterpret_cast<unsigned short*>(s_dyn + 3 * rank_lds_keys(cap));
  {
    const int limit = max(64, K >> 3);      // a fuller bin: counting over the whole list is cheaper
    constexpr int kRounds = (kSelBins / 64 + kRankWaves - 1) / kRankWaves;
    int h[kRounds], incl[kRounds];
#pragma unroll
    for (int u = 0; u < kRounds; ++u) {
      const int ch = wave + u * kRankWaves;                       // 64 bins, from the top
      h[u] = ch < kSelBins / 64 ? s_cursor[kSelBins - 1 - (ch * 64 + lane)] : 0;
      incl[u] = h[u];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl[u], d, 64);
        if (lane >= d) incl[u] += o;
      }
      if (ch < kSelBins / 64 && lane == 63) s_ctot[ch] = incl[u];
      if (h[u] > limit) s_big = 1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRounds; ++u) {
      const int ch = wave + u * kRankWaves;
      if (ch < kSelBins / 64) {
        int off = 0;
        for (int c = 0; c < ch; ++c) off += s_ctot[c];
        const int b = kSelBins - 1 - (ch * 64 + lane);
        s_start[b] = off + incl[u] - h[u];
        s_cursor[b] = off + incl[u] - h[u];
      }
    }
  }
  const bool bucketed = s_big == 0;
  __syncthreads();
  RH_TL(0, 5);
  if (bucketed) {
    for (int i = tid; i < K; i += kRankThreads)
      b_idx[atomicAdd(&s_cursor[rank_bin(s_sk[i])], 1)] = (unsigned short)i;
    __syncthreads();
  }
  RH_TL(0, 6);
  const RankBuckets bk{s_start, s_cursor, b_idx, bucketed};
  const int kpl = clusters_per_block(K);
  const int c_end = min(((int)blockIdx.y + 1) * kpl, (K + 63) >> 6);
  bool first = true;
  for (int c0 = (int)blockIdx.y * kpl; first || c0 < c_end; c0 += kRankKplMax) {
    const bool wc = first && blockIdx.y == 0;
    const int nc = max(min(kRankKplMax, c_end - c0), 1);      // block-uniform
    if (!first) __syncthreads();                              // s_part is reused
#define RH_RANK_PASS(n)                                                                       \
  case n:                                                                                     \
    if constexpr (n <= kRankKplMax)                                                           \
      rank_pass<n>(c0, wc, j, K, ke, found, cap, batch, s_sk, s_ix, s_lab, s_part, counters,  \
                   src, so, bk);                                                              \
    break;
    switch (nc) {
      RH_RANK_PASS(1) RH_RANK_PASS(2) RH_RANK_PASS(3) RH_RANK_PASS(4)
      RH_RANK_PASS(5) RH_RANK_PASS(6) RH_RANK_PASS(7) RH_RANK_PASS(8)
    }
#undef RH_RANK_PASS
    first = false;
  }
}

// Global-memory fallback for candidate lists that do not fit LDS.
__global__ __launch_bounds__(kSortThreads) void sort_gather_keys_kernel(
    u64* __restrict__ keys, int* __restrict__ counters, int batch, int cap, int key_stride, int P,
    SortSrc src) {
  __shared__ int s_count;
  const int j = blockIdx.x;
  u64* kj = keys + (size_t)j * key_stride;
  int found = src.count;
  if (src.from_tiles) found = gather_image_keys(j, src, kj, cap, &s_count);
  const int K = min(found, cap);
  for (int i = K + threadIdx.x; i < P; i += kSortThreads) kj[i] = 0ull;
  if (threadIdx.x == 0) {
    counters[j] = found;
    counters[batch + j] = max(found - cap, 0);
  }
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
__global__ __launch_bounds__(kThreads) void sort_finish_kernel(int* __restrict__ counters, int batch,
                                                               int cap, int topk, SortSrc src,
                                                               SortOut so) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int K = min(counters[j], cap);
  const int ke = topk > 0 ? min(K, topk) : K;
  if (i < ke) store_sorted(so.keys[(size_t)j * so.key_stride + i], i, j, src, so);
  if (i == 0) counters[2 * batch + j] = ke;
}

// ---------------------------------------------------------------------------
// NMS predecessor bitmask.  Boxes are in sorted order; word (w, i) has bit c set when
// box w*64+c comes BEFORE box i and overlaps it by more than `thresh` (and, per class,
// carries the same label): the boxes that would suppress box i if they are kept.
// Only tiles with w <= i/64 exist; one wave per 64x64 tile, the tile's earlier boxes
// staged in LDS.  Layout [w][i] ("column-major"): a wave writes 512 contiguous bytes.
// ---------------------------------------------------------------------------
// Exact test "float32 IoU > thresh" with torchvision's arithmetic (division included).
__device__ __forceinline__ bool nms_hit_exact(const float inter, const float denom,
                                              const float thresh) {
  return inter / denom > thresh;
}

__global__ __launch_bounds__(kThreads) void nms_mask_general_kernel(const float4* __restrict__ sbox,
                                                            const int* __restrict__ slabel,
                                                            const int* __restrict__ keff,
                                                            u64* __restrict__ mask, int box_stride,
                                                            int wstride, float thresh,
                                                            int per_class) {
  __shared__ float4 s_box[kThreads / 64][64];
  __shared__ float s_area[kThreads / 64][64];
  __shared__ int s_lab[kThreads / 64][64];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = keff[j];
  const int W = (K + 63) >> 6;
  const long long T = (long long)W * (W + 1) / 2;
  const float4* bj = sbox + (size_t)j * box_stride;
  const int* lj = slabel ? slabel + (size_t)j * box_stride : nullptr;
  u64* mj = mask + (size_t)j * box_stride * wstride;
  // "inter / denom > thresh" is decided without dividing wherever inter is clearly above or
  // below thresh*denom (a 2^-21 relative band covers the rounding of the product and of the
  // quotient); the few pairs inside the band, and degenerate ones, are divided after the loop.
  const float band_hi = 1.0f + 4.8e-7f, band_lo = 1.0f - 4.8e-7f;
  for (long long id = (long long)blockIdx.x * (kThreads / 64) + wave; id < T;
       id += (long long)gridDim.x * (kThreads / 64)) {
    // tile id -> (rt, ct) with rt <= ct; row rt of the triangle starts at rt*W - rt*(rt-1)/2
    int rt = (int)(((2.0 * W + 1.0) - sqrt((2.0 * W + 1.0) * (2.0 * W + 1.0) - 8.0 * (double)id)) * 0.5);
    rt = max(0, min(rt, W - 1));
    while (rt > 0 && (long long)rt * W - (long long)rt * (rt - 1) / 2 > id) --rt;
    while ((long long)(rt + 1) * W - (long long)(rt + 1) * rt / 2 <= id) ++rt;
    const int ct = rt + (int)(id - ((long long)rt * W - (long long)rt * (rt - 1) / 2));
    const int i = ct * 64 + lane;       // my box (the later one)
    const int e = rt * 64 + lane;       // the earlier box this lane stages
    const float4 mine = bj[min(i, K - 1)];
    const int mylab = per_class ? lj[min(i, K - 1)] : 0;
    {
      const float4 eb = bj[min(e, K - 1)];
      s_box[wave][lane] = eb;
      s_area[wave][lane] = (eb.z - eb.x) * (eb.w - eb.y);
      if (per_class) s_lab[wave][lane] = lj[min(e, K - 1)];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes have landed
    const float area = (mine.z - mine.x) * (mine.w - mine.y);
    const int emax = min(64, K - rt * 64);
    // earlier boxes only: bits below my own position in the diagonal tile, all bits otherwise
    const u64 allowed = (rt == ct ? ((1ull << lane) - 1ull) : ~0ull) &
                        (emax >= 64 ? ~0ull : ((1ull << emax) - 1ull));
    u64 sure = 0ull, unsure = 0ull;
#pragma unroll 8
    for (int cc = 0; cc < 64; ++cc) {
      const float4 b = s_box[wave][cc];
      const float w = fmaxf(0.0f, fminf(mine.z, b.z) - fmaxf(mine.x, b.x));
      const float h = fmaxf(0.0f, fminf(mine.w, b.w) - fmaxf(mine.y, b.y));
      const float inter = w * h;
      const float denom = (area + s_area[wave][cc]) - inter;
      const float td = thresh * denom;
      const bool same = !per_class || s_lab[wave][cc] == mylab;
      const bool yes = inter > td * band_hi && td > 0.0f;
      const bool no = inter < td * band_lo && td > 0.0f;
      const bool maybe = !yes && !no && (inter > 0.0f || thresh < 0.0f);
      sure |= (yes && same) ? (1ull << cc) : 0ull;
      unsure |= (maybe && same) ? (1ull << cc) : 0ull;
    }
    sure &= allowed;
    unsure &= allowed;
    while (unsure) {                      // rare: exact division for the pairs inside the band
      const int cc = __builtin_ctzll(unsure);
      unsure &= unsure - 1ull;
      const float4 b = s_box[wave][cc];
      const float w = fmaxf(0.0f, fminf(mine.z, b.z) - fmaxf(mine.x, b.x));
      const float h = fmaxf(0.0f, fminf(mine.w, b.w) - fmaxf(mine.y, b.y));
      const float inter = w * h;
      if (nms_hit_exact(inter, (area + s_area[wave][cc]) - inter, thresh)) sure |= 1ull << cc;
    }
    if (i < K) mj[(size_t)rt * box_stride + i] = sure;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Pair kernel (0 < thresh): finds the overlapping pairs without testing most pairs, and records
// each as one bit of the later box's predecessor words.
//
// Boxes are grouped into CLUSTERS of 64 that are consecutive in list order (the order the
// candidates had before sorting: ascending anchor index, i.e. neighbouring cells of one pyramid
// level, similar sizes); the sort stage leaves each cluster's bounding box and area range.  Two
// boxes can only have IoU > thresh > 0 if they intersect and if the smaller area is more than
// thresh times the larger (the intersection is at most the smaller area, the union at least the
// larger one; in float32 the computed quotient stays below min/max * (1 + 1e-6), and the test
// here leaves a margin of 1e-4).  So:
//   * a pair of clusters whose bounding boxes do not intersect, or whose area ranges are too far
//     apart, is skipped as a whole (one compare per cluster pair, 64 cluster pairs per ballot);
//   * inside a remaining pair (A, B) the wave holds A's boxes one per lane, loads B's boxes one
//     per lane, and drops with ONE ballot every box of B that misses A's bounding box or area
//     range; the survivors are broadcast from their lanes one at a time (v_readlane) and tested
//     against the wave's 64 boxes:
//       w = min(x2, X2) - max(x1, X1); h likewise; inter = max(w, 0) * h   (h <= 0 gives inter <= 0)
//       denom = max((area + AREA) - inter, 0)
//       possible = inter > (area + AREA) * thresh / (1 + thresh) * (1 - 2^-19)  -> IoU > thresh
//                  cannot be excluded (inter > thresh * denom solved for inter, loosened)
//       sure     = inter > denom * (thresh * (1 + 2^-21))   -> IoU > thresh for certain
//     inter > 0 implies both boxes have positive width and height and denom >= inter > 0, so the
//     clamp of denom only matters for degenerate boxes, where it makes the test fail as it must
//     (IoU is then 0 or NaN).  min / max are single instructions (the compiler would wrap each
//     in NaN-quieting moves).
//   * possible & ~sure (a 2^-20 relative band around the threshold) is settled with the exact
//     float32 division of torchvision's arithmetic.  All of it is symmetric in the two boxes, so
//     every unordered pair is looked at once (cluster pairs A <= B; inside A == B the lower lane
//     against the higher).
// A hit sets bit (earlier rank % 64) of word (earlier rank / 64) of the LATER box by an atomic OR
// (the sort stage zeroed the words) and raises that word's bit in the later box's nz bitmap.
// Without spatial order in the list (a plain NMS call on shuffled boxes) every pair is tested, at
// about the cost of a dense 64 x 64 tiling.
// Grid clusters x slices x images: the waves of the `slices` blocks of cluster A share the
// clusters B >= A that survive the cluster test, half a cluster pair at a time, round robin.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float vmin_s(const float uniform, const float v) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(v));
  return r;
}
__device__ __forceinline__ float vmax_s(const float uniform, const float v) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(v));
  return r;
}

// One box of a cluster as a lane holds it.
struct PairLane {
  float4 box;
  int rank;    // -1: no box here
  int lab;
  float area;
};

#ifndef RH_PAIR_THREADS
#define RH_PAIR_THREADS 512
#endif
constexpr int kPairThreads = RH_PAIR_THREADS;
template <bool kPerClass>
__global__ __launch_bounds__(kPairThreads) void nms_pairs_kernel(
    const float4* __restrict__ cbox, const int4* __restrict__ cinfo,
    const float4* __restrict__ csum, const int* __restrict__ counters,
    const int* __restrict__ list_len, int batch, u64* __restrict__ mask, u64* __restrict__ nz,
    int cap, int list_stride, int clusters, int box_stride, int wstride, int nzw, int slices,
    float thresh, int ranked) {
  // one-dimensional grid, the image index fastest: the workgroups of clusters beyond an image's
  // list (most of the grid when the lists are far from full) come last and exit at once
  const int j = blockIdx.x % batch;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // A block's units (cluster A, slice) are blockIdx.x / batch, then every gridDim.x / batch
  // further: the grid need not be as large as the list capacity would allow.
  const int unit_step = (int)(gridDim.x / batch);
  int unit = (int)(blockIdx.x / batch);
  int A = unit / slices, slice = unit % slices;
  RH_TL(1, 0);
  const float4* bj = cbox + (size_t)j * list_stride;
  const int4* ij = cinfo + (size_t)j * list_stride;
  const float4* sj = csum + (size_t)j * clusters * kSummaryVecs;
  u64* mj = mask + (size_t)j * box_stride * wstride;
  u64* nj = nz + (size_t)j * box_stride * nzw;
  // My cluster's boxes and the summaries go out together with the counters (all inside their
  // arrays whatever the counters say; what lies beyond the list is discarded below).
  auto load_lane = [&](const int cluster) -> PairLane {
    PairLane r;
    const int q = min(cluster * 64 + lane, list_stride - 1);
    r.box = bj[q];
    const int4 info = ij[q];
    r.rank = info.x;
    r.lab = info.y;
    r.area = __int_as_float(info.z);
    return r;
  };
  PairLane me = load_lane(min(A, clusters - 1));
  float4 a_box = sj[min(A, clusters - 1) * kSummaryVecs], a_rng = sj[min(A, clusters - 1) * kSummaryVecs + 1];   // uniform
  float4 a_ctr = sj[min(A, clusters - 1) * kSummaryVecs + 2];
  int first_b = min(A + lane, clusters - 1);
  float4 b_box0 = sj[first_b * kSummaryVecs], b_rng0 = sj[first_b * kSummaryVecs + 1];
  float4 b_ctr0 = sj[first_b * kSummaryVecs + 2];
  __builtin_amdgcn_sched_barrier(0);
  // ranked: the list is what the rank sort saw; else the boxes are in rank order themselves
  // (list_len: the entries the rank sort was given -- all candidates found, or those the
  // top-k pre-selection kept)
  const int klist = ranked ? min(list_len[j], cap) : counters[2 * batch + j];
  const int C = (klist + 63) >> 6;
  RH_TL(1, 1);
  const float f = fminf(thresh, 1.0f) * 0.9999f;
  // From a threshold of one half on, an overlapping pair also has each box's centre inside the
  // other box: IoU > 1/2 means the intersection covers more than half of either box's area, hence
  // more than half of its width and of its height, and an interval that covers more than half of
  // a segment contains the segment's midpoint.  The float32 quotient can exceed 1/2 when the
  // exact one is 1e-6 short of it, and the midpoints are rounded: `slack` (a thousandth of the
  // larger side of A's bounding box -- a box that overlaps one of A's by half is at most twice
  // as wide -- or of its largest coordinate, whose rounding the midpoints carry) covers both.
  const bool centre_rule = thresh >= 0.5f;
  float slack = 0.0f;
  const float c_hi = thresh * (1.0f + 4.8e-7f);
  const float k_filter = thresh / (1.0f + thresh) * (1.0f - 1.9e-6f);
  for (; A < C; unit += unit_step, A = unit / slices, slice = unit % slices) {
  if (unit != (int)(blockIdx.x / batch)) {       // (the first unit's records were requested above)
    me = load_lane(A);
    a_box = sj[A * kSummaryVecs];
    a_rng = sj[A * kSummaryVecs + 1];
    a_ctr = sj[A * kSummaryVecs + 2];
    first_b = min(A + lane, clusters - 1);
    b_box0 = sj[first_b * kSummaryVecs];
    b_rng0 = sj[first_b * kSummaryVecs + 1];
    b_ctr0 = sj[first_b * kSummaryVecs + 2];
  }
  if (A * 64 + lane >= klist) me.rank = -1;
  slack = 1e-3f * fmaxf(fmaxf(a_box.z - a_box.x, a_box.w - a_box.y),
                        fmaxf(fmaxf(fabsf(a_box.x), fabsf(a_box.y)), fmaxf(fabsf(a_box.z), fabsf(a_box.w))));

  const float4 mine = me.box;
  const float area = me.area;
  const bool valid = me.rank >= 0;
  const u64 lanes_ok = __ballot(valid);
  const int nwaves = slices * (kPairThreads / 64), mywave = slice * (kPairThreads / 64) + wave;

  // lanes of `mine` that may overlap box b (record info) by more than thresh (almost always
  // none).  The box is wave-uniform and sits in scalar registers.
  auto possible = [&](const float4 b, const int4 info, float& inter, float& sum) -> u64 {
    const float w = vmin_s(b.z, mine.z) - vmax_s(b.x, mine.x);
    const float h = vmin_s(b.w, mine.w) - vmax_s(b.y, mine.y);
    inter = vmax0(w) * h;
    if (kPerClass) inter = info.y == me.lab ? inter : -1.0f;
    sum = area + __int_as_float(info.z);
    // inter > thresh * (sum - inter), solved for inter and loosened by 2^-19 (this is only the
    // filter: whatever passes is decided exactly below, and a pair that the exact test accepts
    // passes here with room to spare)
    return __ballot(inter > sum * k_filter);
  };
  // decide and record the pairs (lane, box) of `hit`; src = the box's lane in cluster B;
  // inter, sum: the lane's values from possible()
  auto settle = [&](const float4 b, const int erank, const int src, const bool same, u64 hit,
                    const float inter, const float sum) {
    // inside one cluster each unordered pair once: only the lanes above the box's own
    hit &= same ? lanes_ok & ~((2ull << src) - 1ull) : lanes_ok;
    if (hit == 0ull) return;
    if ((hit >> lane) & 1ull) {
      // Clear of the 2^-20 band around the threshold the product test decides (a positive inter
      // means positive overlap in x and y, where it equals the clamped product the division
      // form uses); inside the band the float32 division of torchvision's arithmetic decides.
      bool sure = inter > fmaxf(sum - inter, 0.0f) * c_hi;
      if (__ballot(!sure) != 0ull) {
        const float w = fmaxf(0.0f, fminf(mine.z, b.z) - fmaxf(mine.x, b.x));
        const float h = fmaxf(0.0f, fminf(mine.w, b.w) - fmaxf(mine.y, b.y));
        const float exact_inter = w * h;
        sure = sure || nms_hit_exact(exact_inter, sum - exact_inter, thresh);
      }
      if (sure) {
        const int brank = __builtin_amdgcn_readlane(erank, src);
        const int first = min(me.rank, brank), later = max(me.rank, brank);
        const int word = first >> 6;
        atomicOr(&mj[(size_t)word * box_stride + later], 1ull << (first & 63));
        atomicOr(&nj[(size_t)later * nzw + (word >> 6)], 1ull << (word & 63));
      }
    }
  };

  int taken = 0;                                    // units seen so far
  RH_TL(1, 2);
  RH_TL_VALUE(1, 8, (long long)__builtin_amdgcn_s_memtime());
  RH_TL_DO(int tl_tiles = 0, tl_surv = 0, tl_nset = 0; long long tl_a = 0, tl_t = 0, tl_set = 0;)
  for (int b0 = A; b0 < C; b0 += 64) {
    // ---- cluster test, 64 clusters per ballot (the first 64 summaries were requested up front)
    const int bc = min(b0 + lane, C - 1);
    const float4 b_box = b0 == A ? b_box0 : sj[bc * kSummaryVecs], b_rng = b0 == A ? b_rng0 : sj[bc * kSummaryVecs + 1];
    const float4 b_ctr = b0 == A ? b_ctr0 : sj[bc * kSummaryVecs + 2];
    bool apart_c = (b_box.z <= a_box.x) | (b_box.x >= a_box.z) | (b_box.w <= a_box.y) |
                   (b_box.y >= a_box.w);
    // centre rule: some centre of B must lie in A's bounding box and some centre of A in B's
    if (centre_rule)
      apart_c |= (b_ctr.z < a_box.x - slack) | (b_ctr.x > a_box.z + slack) | (b_ctr.w < a_box.y - slack) |
                 (b_ctr.y > a_box.w + slack) | (a_ctr.z < b_box.x - slack) | (a_ctr.x > b_box.z + slack) |
                 (a_ctr.w < b_box.y - slack) | (a_ctr.y > b_box.w + slack);
    const bool off_c = (b_rng.x * f > a_rng.y) | (b_rng.y < a_rng.x * f);   // all too large / too small
    u64 bm = __ballot(b0 + lane < C && !apart_c && !off_c);
    // A unit of work is half a cluster pair (32 of B's boxes), dealt round robin over the waves
    // of cluster A's workgroups.
    u64 my_lo = 0ull, my_hi = 0ull;               // cluster pairs whose low / high half is mine
    for (u64 t = bm; t != 0ull; t &= t - 1ull) {      // nwaves is a power of two
      if ((taken & (nwaves - 1)) == mywave) my_lo |= t & (0ull - t);
      if (((taken + 1) & (nwaves - 1)) == mywave) my_hi |= t & (0ull - t);
      taken += 2;
    }
    u64 my_bm = my_lo | my_hi;
    if (my_bm == 0ull) continue;
    PairLane nxt = load_lane(b0 + __builtin_ctzll(my_bm));
    while (my_bm != 0ull) {
      const int bit = __builtin_ctzll(my_bm);
      const int B = b0 + bit;
      my_bm &= my_bm - 1ull;
      const u64 halves = (((my_lo >> bit) & 1ull) ? 0x00000000FFFFFFFFull : 0ull) |
                         (((my_hi >> bit) & 1ull) ? 0xFFFFFFFF00000000ull : 0ull);
      const PairLane e = nxt;
      if (my_bm != 0ull) nxt = load_lane(b0 + __builtin_ctzll(my_bm));   // on its way meanwhile
      // B's boxes one per lane, only to find the ones worth testing: a box that misses A's
      // bounding box or area range is out (a NaN coordinate is never culled, and never hits)
      const float ecx = 0.5f * (e.box.x + e.box.z), ecy = 0.5f * (e.box.y + e.box.w);
      bool apart = (e.box.z <= a_box.x) | (e.box.x >= a_box.z) | (e.box.w <= a_box.y) |
                   (e.box.y >= a_box.w);
      if (centre_rule) {      // the box's centre in A's bounding box, a centre of A in the box
        apart |= (ecx < a_box.x - slack) | (ecx > a_box.z + slack) | (ecy < a_box.y - slack) |
                 (ecy > a_box.w + slack) | (a_ctr.z < e.box.x - slack) | (a_ctr.x > e.box.z + slack) |
                 (a_ctr.w < e.box.y - slack) | (a_ctr.y > e.box.w + slack);
      }
      const bool off = (e.area * f > a_rng.y) | (e.area < a_rng.x * f);
      const bool there = B * 64 + lane < klist && e.rank >= 0;
      const u64 hm = __ballot(there && !apart && !off) & halves;
      RH_TL_DO(++tl_tiles; tl_surv += __popcll(hm); tl_t = wall_clock64();)
      // The survivors one at a time as wave-uniform values, broadcast from their lanes
      // (v_readlane), four per round.  Measured alternatives: testing first, per round, whether any
      // lane's box contains a survivor's centre (two broadcasts and five instructions per box
      // instead of the overlap arithmetic) rejects almost no round -- among 4 x 64 pairs of one
      // strip of the image some lane nearly always does -- and cost 2 us per call; packing them into LDS and reading
      // them back with 16-byte broadcast reads was bound by the LDS pipe, which all waves of a
      // compute unit share (a broadcast read costs the bandwidth of 64 lanes).  (A variant with
      // hand-written scalar loads of the same records once gave wrong keep lists; scalar loads of
      // data that the kernel before has written are sound on this chip -- 120 000 writer / reader
      // rounds of tools/ubench/sload_boundary.hip, with the lines resident in the scalar cache and
      // a second stream busy, returned no old word -- so the fault was in that variant's own
      // addressing, not in the cache; the lane broadcast is also the faster form.)
      for (u64 m = hm; m != 0ull;) {
        int src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          src[u] = m != 0ull ? __builtin_ctzll(m) : src[u > 0 ? u - 1 : 0];   // past the last: repeat it
          m &= m - 1ull;                                                      // 0 stays 0
        }
        float4 box[4];
        int4 info[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          box[u] = make_float4(lane_value(e.box.x, src[u]), lane_value(e.box.y, src[u]),
                               lane_value(e.box.z, src[u]), lane_value(e.box.w, src[u]));
          info[u] = make_int4(0, kPerClass ? __builtin_amdgcn_readlane(e.lab, src[u]) : 0,
                              __builtin_amdgcn_readlane(__float_as_int(e.area), src[u]), 0);
        }
        u64 h[4];
        float inter[4], sum[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = possible(box[u], info[u], inter[u], sum[u]);
        if ((h[0] | h[1] | h[2] | h[3]) != 0ull) {
          RH_TL_DO(const long long tl_s = wall_clock64(); ++tl_nset;)
#pragma unroll
          for (int u = 0; u < 4; ++u) settle(box[u], e.rank, src[u], B == A, h[u], inter[u], sum[u]);
          RH_TL_DO(tl_set += wall_clock64() - tl_s;)
        }
      }
      RH_TL_DO(tl_a += wall_clock64() - tl_t;)
    }
  }
  RH_TL_DO(if (unit + unit_step < C * slices) continue;)   // (stamps: the block's last unit)
  RH_TL(1, 3);
  RH_TL_VALUE(1, 9, (long long)__builtin_amdgcn_s_memtime());
  RH_TL_VALUE(1, 4, tl_tiles);
  RH_TL_VALUE(1, 5, tl_surv);
  RH_TL_VALUE(1, 6, tl_a);
  RH_TL_VALUE(1, 7, tl_set * 1000 + tl_nset);
  }
}

// Cluster summaries, list records and zeroed mask words for sort paths that do not provide them
// (lists beyond the rank sort's capacity): clusters are 64 consecutive ranks.  Grid (clusters,
// images), one wave per cluster.
__global__ __launch_bounds__(64) void cluster_prepare_kernel(const float4* __restrict__ sbox,
                                                             const int* __restrict__ slabel,
                                                             const int* __restrict__ keff,
                                                             float4* __restrict__ csum,
                                                             int4* __restrict__ cinfo,
                                                             u64* __restrict__ mask, int box_stride,
                                                             int wstride, int clusters) {
  const int j = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  const int K = keff[j];
  if (g * 64 >= K) return;
  const int i = g * 64 + lane;
  const float4 box = sbox[(size_t)j * box_stride + min(i, K - 1)];
  const ClusterSummary cs = summarize_cluster(box, i < K);
  if (lane == 0) {
    float4* out = csum + ((size_t)j * clusters + g) * kSummaryVecs;
    out[0] = make_float4(cs.x1, cs.y1, cs.x2, cs.y2);
    out[1] = make_float4(cs.amin, cs.amax, 0.f, 0.f);
    out[2] = make_float4(cs.cx1, cs.cy1, cs.cx2, cs.cy2);
  }
  if (i < K) {
    cinfo[(size_t)j * box_stride + i] =
        make_int4(i, slabel[(size_t)j * box_stride + i],
                  __float_as_int((box.z - box.x) * (box.w - box.y)), 0);
    for (int w = 0; w <= g; ++w) mask[((size_t)j * wstride + w) * box_stride + i] = 0ull;
  }
}

// nz bitmap for the dense mask kernel: every word up to the box's own chunk may be non-zero.
__global__ __launch_bounds__(kThreads) void nz_dense_kernel(const int* __restrict__ keff,
                                                            u64* __restrict__ nz, int box_stride,
                                                            int nzw) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= keff[j]) return;
  const int last = i >> 6;                               // words 0 .. last
  for (int g = 0; g < nzw; ++g) {
    const int lo = g * 64;
    u64 bits = 0ull;
    if (last >= lo + 63) bits = ~0ull;
    else if (last >= lo) bits = (2ull << (last - lo)) - 1ull;
    nz[((size_t)j * box_stride + i) * nzw + g] = bits;
  }
}

// ---------------------------------------------------------------------------
// Pair search by a hash of the box centres (thresholds from one half on, boxes inside a known
// image): the same predecessor words as nms_pairs_kernel leaves, found another way.
//
// Two boxes with IoU > thresh >= 1/2 contain each other's centre and differ in area by less than
// a factor two (the intersection is at most the smaller area, the union at least the larger one;
// the float32 quotient stays within a few ulp of that, the bounds below carry a margin of 1e-5).
// So the box with the SMALLER area (equal areas: the lower rank) owns the pair, and all it has to
// look at are boxes of its own area class (a class is a binade of the area) or the next one,
// whose centres lie inside it.  Boxes are binned by (area class, cell of the centre), cells of
// class c being about as large as the boxes of class c, so a box covers a handful of cells per
// class and a cell holds a handful of boxes -- about 18 candidates per box at the benchmark's
// density instead of the ~170 that survive the cluster tests of nms_pairs_kernel.
//
// Every workgroup of an image builds the image's table in LDS: a counting sort of the boxes by
// cell (LDS atomics for the counts and for the places inside a cell -- their order is left to
// chance, the result is a set of bits), which leaves the boxes themselves in cell order, their
// ranks beside them, and for every rank the place of its box.  Then the workgroup handles its
// share of the owners (ranks g, g + G, ..: a split by place in the table would differ from
// workgroup to workgroup), four or eight threads per owner.  The cells of one row of one class
// that an owner reaches are consecutive, so their entries form one run; the owner's runs
// (typically four, two or three rows in each of two classes) are walked as a single sequence
// of which its threads take every fourth (eighth) entry.  Candidates that own nothing, lie outside the area
// window or have their centre outside the owner's box (with the slack of nms_pairs_kernel's
// centre rule) are dropped; the rest get torchvision's float32 division.
// A box without area (or with a NaN coordinate) is in no cell and owns nothing: its IoU with
// anything is 0 or NaN, never above a threshold.
// 608 px, 16 images of ~2 000 boxes: 11.3 us against 15.1 us for nms_pairs_kernel.  Larger
// workgroups are faster alone (640 / 768 / 1024 threads: 10.8 / 10.6 / 10.2 us, the call on
// one stream 88.6 instead of 90.2 us: the table allows one workgroup per compute unit either
// way, and 256 groups of four threads take a full list in one pass) but cost the pipelined
// form a tenth of its rate (three batches in flight 70.5 instead of 64.2 us per batch from 640
// threads on: a workgroup of more than two waves per SIMD has to wait for room among the other
// batches' score-scan waves), so 512 it is.  All with 512 threads: the walk with one loop per
// run 13.1, with the boxes looked up through their ranks 13.6, with the runs dealt out to the
// four threads 13.6, two / eight threads for every owner 13.8 / 12.4; runs taken eight / four
// at a time 11.6 / 11.8.
// ---------------------------------------------------------------------------
#ifndef RH_HASH_THREADS
#define RH_HASH_THREADS 512
#endif
constexpr int kHashThreads = RH_HASH_THREADS;
constexpr int kHashMaxBoxes = 4096;
constexpr size_t kHashLdsLimit = 150 * 1024;        // dynamic LDS of the table, at most
constexpr int kHashMaxClasses = 20;
#ifndef RH_HASH_RUNS
#define RH_HASH_RUNS 6
#endif
constexpr int kHashRuns = RH_HASH_RUNS;             // runs of entries taken together
struct HashGeom {
  int classes;                    // area classes: class c holds areas in [2^(e0 + c), 2^(e0 + c + 1)),
  int e0;                         // the first one everything below, the last one everything above
  int cells;                      // all classes together
  float inv_g[kHashMaxClasses];   // 1 / cell size
  int nx[kHashMaxClasses], ny[kHashMaxClasses], base[kHashMaxClasses];
};

__host__ __device__ inline size_t hash_lds_bytes(int box_cap, int cells) {
  return (size_t)box_cap * sizeof(float4) + ((size_t)cells + 1) * sizeof(unsigned) +
         (size_t)box_cap * 2 * sizeof(unsigned short);
}

__device__ __forceinline__ int hash_class(const float area, const int e0, const int classes) {
  const int e = (int)((__float_as_uint(area) >> 23) & 255u) - 127;      // floor(log2(area))
  return min(max(e - e0, 0), classes - 1);
}

template <bool kPerClass>
__global__ __launch_bounds__(kHashThreads) void nms_pairs_hash_kernel(
    const float4* __restrict__ sbox, const int* __restrict__ slabel, const int* __restrict__ keff,
    u64* __restrict__ mask, u64* __restrict__ nz, int box_stride, int wstride, int nzw,
    float thresh, const HashGeom geo) {
  // the table: the boxes in cell order, where each cell's boxes start, the boxes' ranks, and
  // where the box of a rank is
  extern __shared__ __attribute__((aligned(16))) float4 s_cbox[];              // [box_stride]
  unsigned* s_off = reinterpret_cast<unsigned*>(s_cbox + box_stride);          // [cells + 1]
  unsigned short* s_rank = reinterpret_cast<unsigned short*>(s_off + geo.cells + 1);   // [box_stride]
  unsigned short* s_where = s_rank + box_stride;                               // [box_stride], 0xFFFF: nowhere
  __shared__ float s_inv[kHashMaxClasses];
  __shared__ int s_nx[kHashMaxClasses], s_ny[kHashMaxClasses], s_base[kHashMaxClasses];
  __shared__ unsigned s_wtot[kHashThreads / 64];
  const int j = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* bj = sbox + (size_t)j * box_stride;
  RH_TL(1, 0);
  // the boxes are requested first of all (rows past the list's end are read and not used)
  constexpr int kPer = (kHashMaxBoxes + kHashThreads - 1) / kHashThreads;
  float4 mine[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) mine[u] = bj[min(tid + u * kHashThreads, box_stride - 1)];
  const int K = min(keff[j], box_stride);
  const int cells = geo.cells;
  for (int i = tid; i <= cells; i += kHashThreads) s_off[i] = 0u;
  if (tid < kHashMaxClasses) {
    s_inv[tid] = geo.inv_g[tid];
    s_nx[tid] = geo.nx[tid];
    s_ny[tid] = geo.ny[tid];
    s_base[tid] = geo.base[tid];
  }
  __syncthreads();
  unsigned cell[kPer], pos[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int p = tid + u * kHashThreads;
    cell[u] = 0xFFFFu;
    pos[u] = 0u;
    if (p < K) {
      const float4 b = mine[u];
      const float area = (b.z - b.x) * (b.w - b.y);
      if (area > 0.0f && area < __builtin_inff()) {    // (false for NaN)
        const int c = hash_class(area, geo.e0, geo.classes);
        const float cx = (b.x + b.z) * 0.5f, cy = (b.y + b.w) * 0.5f;
        const int ix = min(max((int)(cx * s_inv[c]), 0), s_nx[c] - 1);
        const int iy = min(max((int)(cy * s_inv[c]), 0), s_ny[c] - 1);
        cell[u] = (unsigned)(s_base[c] + iy * s_nx[c] + ix);
        pos[u] = atomicAdd(&s_off[cell[u]], 1u);
      }
    }
  }
  __syncthreads();
  RH_TL(1, 1);
  // exclusive prefix over the cells' counts, in place: a run of cells per thread, the runs'
  // totals by wave scans
  {
    const int run = (cells + kHashThreads - 1) / kHashThreads;
    const int lo = min(tid * run, cells), hi = min(lo + run, cells);
    unsigned tot = 0;
    for (int i = lo; i < hi; ++i) tot += s_off[i];
    unsigned incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    unsigned before = incl - tot;
    for (int w = 0; w < wave; ++w) before += s_wtot[w];
    for (int i = lo; i < hi; ++i) {
      const unsigned c = s_off[i];
      s_off[i] = before;
      before += c;
    }
    if (tid == kHashThreads - 1) s_off[cells] = before;      // the last run's end: all boxes in cells
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int p = tid + u * kHashThreads;
    unsigned at = 0xFFFFu;
    if (cell[u] != 0xFFFFu) {
      at = s_off[cell[u]] + pos[u];
      s_cbox[at] = mine[u];
      s_rank[at] = (unsigned short)p;
    }
    if (p < box_stride) s_where[p] = (unsigned short)at;
  }
  __syncthreads();
  RH_TL(1, 2);

  // ---- the owners of this workgroup: ranks g, g + G, .. (by rank and not by place in the table:
  // the order inside a cell differs from workgroup to workgroup)
  const int G = gridDim.x, g = blockIdx.x;
  u64* mj = mask + (size_t)j * box_stride * wstride;
  u64* nj = nz + (size_t)j * box_stride * nzw;
  const int* lj = kPerClass ? slabel + (size_t)j * box_stride : nullptr;
  // `share` threads per owner (a constant of each instance: with the number in a register the
  // kernel took 0.4 us longer)
  auto query = [&](auto share_constant) {
  constexpr int share = decltype(share_constant)::value;
  constexpr int groups = kHashThreads / share;        // owners per pass; threads past groups * share idle
  const int group = tid / share;
  const unsigned sub = tid - group * share;
  for (int i = group;; i += groups) {
    const int p = group < groups ? g + i * G : K;
    if (__ballot(p < K) == 0ull) break;
    if (p >= K) continue;
    const unsigned own = s_where[p];
    if (own == 0xFFFFu) continue;
    const float4 a = s_cbox[own];
    const float wa = a.z - a.x, ha = a.w - a.y;
    const float area = wa * ha;
    const float a_max = area * 2.00002f;
    // slack of the centre test: a thousandth of the box's larger side or of its largest
    // coordinate (whose rounding the midpoints carry)
    const float slack = 1e-3f * fmaxf(fmaxf(wa, ha), fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)),
                                                          fmaxf(fabsf(a.z), fabsf(a.w))));
    const float x_lo = a.x - slack, x_hi = a.z + slack, y_lo = a.y - slack, y_hi = a.w + slack;
    // The owner's class and the next one (twice the area: the next binade).  A box of the class
    // after that has at least 2^(k+2) of area against less than 2^(k+1) here, and the float32
    // quotient of an intersection below 2^(k+1) by a union of 2^(k+2) (1 - 2^-24) or more is at
    // most one half -- however close to 2^(k+1) the owner's area is.
    const int c_lo = hash_class(area, geo.e0, geo.classes), c_hi = hash_class(area * 2.0f, geo.e0, geo.classes);
    const int my_lab = kPerClass ? lj[p] : 0;
    // The runs of entries the box reaches (a run: the consecutive cells of one row of one class),
    // numbered through both classes.
    int first[2], span[2], pitch[2], rows[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = k ? c_hi : c_lo;
      const float inv = s_inv[c];
      const int nx = s_nx[c], ny = s_ny[c];
      const int ix0 = min(max((int)(x_lo * inv), 0), nx - 1), ix1 = min(max((int)(x_hi * inv), 0), nx - 1);
      const int iy0 = min(max((int)(y_lo * inv), 0), ny - 1), iy1 = min(max((int)(y_hi * inv), 0), ny - 1);
      first[k] = s_base[c] + iy0 * nx + ix0;
      span[k] = ix1 - ix0 + 1;
      pitch[k] = nx;
      rows[k] = iy1 - iy0 + 1;
    }
    if (c_hi == c_lo) rows[1] = 0;
    const int runs = rows[0] + rows[1];
    // The entries of the runs are walked as one sequence, kHashRuns runs at a time: where the
    // runs start and how many entries come before each go to registers first (independent reads),
    // then entry t of the sequence is found by compares alone, and the owner's threads take
    // every share-th entry.  So the lanes of a wave step in unison however the entries are
    // spread over the runs, and a lane's number of steps is its owner's entries over the share.
    auto test = [&](const float4 b, const int q) {
      const float area_b = (b.z - b.x) * (b.w - b.y);
      // the smaller area owns the pair (equal areas: the lower rank)
      const bool owned = area_b > area || (area_b == area && q > p);
      const float cx = (b.x + b.z) * 0.5f, cy = (b.y + b.w) * 0.5f;
      if (!owned || area_b > a_max || cx < x_lo || cx > x_hi || cy < y_lo || cy > y_hi) return;
      if (kPerClass && lj[q] != my_lab) return;
      const float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
      const float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
      const float inter = w * h;
      if (nms_hit_exact(inter, (area + area_b) - inter, thresh)) {
        const int first = min(p, q), later = max(p, q);
        const int word = first >> 6;
        atomicOr(&mj[(size_t)word * box_stride + later], 1ull << (first & 63));
        atomicOr(&nj[(size_t)later * nzw + (word >> 6)], 1ull << (word & 63));
      }
    };
    for (int r0 = 0; r0 < runs; r0 += kHashRuns) {
      unsigned shift[kHashRuns], before[kHashRuns + 1];       // entry t of run k is s_cbox[t + shift[k]]
      before[0] = 0u;
#pragma unroll
      for (int k = 0; k < kHashRuns; ++k) {
        const int r = r0 + k;
        const bool real = r < runs, hi = r >= rows[0];
        // (selects and 24-bit products: no branches among the reads)
        const int row = hi ? r - rows[0] : r;
        int at = (hi ? first[1] : first[0]) + __mul24(row, hi ? pitch[1] : pitch[0]);
        at = real ? at : 0;
        const unsigned from = s_off[at], to = s_off[at + (hi ? span[1] : span[0])];
        shift[k] = from - before[k];
        before[k + 1] = before[k] + (real ? to - from : 0u);
      }
      const unsigned all = before[kHashRuns];
      for (unsigned t = sub; t < all; t += share) {
        unsigned e = t + shift[0];
#pragma unroll
        for (int k = 1; k < kHashRuns; ++k) e = t >= before[k] ? t + shift[k] : e;
        test(s_cbox[e], s_rank[e]);
      }
    }
  }
  };
  // Four threads per owner, eight when the workgroup has threads enough for that.  (The code of
  // every instance costs the kernel time, it seems in instruction fetches: with instances for
  // three, two and one thread as well -- which save a workgroup the second pass when it has a
  // few owners more than 128 -- the kernel took 12.2 instead of 11.6 us; the prefix kept in
  // registers between its two passes, 16 unrolled cells a thread, cost what it saved.)
  const int owners = K > g ? (K - g + G - 1) / G : 0;
  if (owners <= kHashThreads / 8) query(std::integral_constant<int, 8>{});
  else query(std::integral_constant<int, 4>{});
  RH_TL(1, 3);
}

// ---------------------------------------------------------------------------
// Resolve kernel: which boxes survive the greedy pass, from the sparse predecessor words.
// Box i is dropped iff a KEPT box that precedes it overlaps it.  One workgroup per image keeps two
// bitmaps over the boxes in LDS, `lo` (surely kept) and `hi` (possibly kept), starting from
// lo = nothing, hi = everything, and every round each open box looks at its predecessor words:
//   a predecessor in lo  -> the box is dropped (leaves hi);
//   no predecessor in hi -> the box is kept (enters lo).
// lo only grows and hi only shrinks, lo <= kept <= hi holds throughout, and a round settles at
// least every box whose predecessors are settled (the first open box always is one), so the
// loop ends with lo == hi == the greedy result after as many rounds as the longest chain of
// undecided overlaps is deep -- a handful in practice, against one serial step per 64 boxes for
// a sweep in rank order.  A box without predecessor words (three in four) is kept in the first
// round.  Lane l of wave v handles boxes (s*16 + v)*64 + l, s = 0, 1, ..: a wave's ballot over a
// slot is that chunk's bitmap word.  With kSlots > 0 a thread has at most kSlots boxes and
// keeps the first two predecessor words of each in registers (boxes with more re-read the rest
// from memory each round); kSlots == 0 handles any number of boxes without that cache.
// Outputs either the kept original indices (keep_idx, for rh_nms) or the
// (score, label, box) triples of model.py:338-340, in rank order.
// ---------------------------------------------------------------------------
struct SweepOut {
  long long* keep_idx;   // [B][out_stride] or null
  float* scores;         // [B][out_stride] or null
  long long* labels;     // [B][out_stride] or null
  float4* boxes;         // [B][out_stride] or null
  int* count;            // [B]
  int out_stride;
  int max_keep;          // stop after this many (<= out_stride)
  const int* counters;   // [3][B] of the sort stage, or null
  int* cand_out;         // [B] candidates found, or null
  int* ovf_out;          // [B] candidates dropped for lack of room, or null
  int batch;
};

constexpr int kResolveWaves = 16;
constexpr int kResolveThreads = 64 * kResolveWaves;
constexpr size_t kResolveLdsLimit = 160 * 1024;     // the whole LDS of a compute unit

__host__ __device__ inline size_t resolve_lds_bytes(int wstride) {
  // lo and hi bitmaps, kept-before-chunk counts, three "still open" flags
  return (size_t)wstride * 2 * sizeof(u64) + ((size_t)wstride + 1 + 3) * sizeof(int);
}

// A box's predecessor words as the cached form holds them.
struct BoxPreds {
  u64 w0, w1;      // first two predecessor words (0 when absent)
  int i0, i1;      // their word indices (0 when absent)
  bool more;       // further words exist: those are read from memory every round
};

// Bits of mask word w that can belong to box `row`: boxes that precede it in rank.  The pair
// kernel only sets such bits; masking them here as well means that whatever the words hold, the
// relation the rounds work on has no cycle, so the rounds always end.
__device__ __forceinline__ u64 earlier_bits(const int w, const int row) {
  const int c = row >> 6;
  return w < c ? ~0ull : (w == c ? (1ull << (row & 63)) - 1ull : 0ull);
}

// Is a predecessor of box `row` surely kept (d != 0) / possibly kept (m != 0)?  Walks the box's
// non-zero words beyond the first `skip` of them, eight loads in flight at a time, and stops as
// soon as the box is known to be dropped.
__device__ __forceinline__ void scan_words(const u64* s_lo, const u64* s_hi,
                                           const u64* __restrict__ nzrow, const int nzw,
                                           const u64* __restrict__ mj, const int box_stride,
                                           const int row, int skip, u64& d, u64& m) {
  for (int g = 0; g < nzw && d == 0ull; ++g) {
    u64 bits = nzrow[g];
    for (; skip > 0 && bits != 0ull; --skip) bits &= bits - 1ull;
    while (bits != 0ull && d == 0ull) {
      int w[8];
      u64 word[8];
      int n = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool has = bits != 0ull;
        w[u] = has ? g * 64 + __builtin_ctzll(bits) : (u > 0 ? w[u - 1] : 0);
        n += has;
        bits &= bits - 1ull;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) word[u] = mj[(size_t)w[u] * box_stride + row];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < n) {
          const u64 pw = word[u] & earlier_bits(w[u], row);
          d |= pw & s_lo[w[u]];
          m |= pw & s_hi[w[u]];
        }
      }
    }
  }
}

template <int kSlots>
__global__ __launch_bounds__(kResolveThreads) void nms_resolve_kernel(
    const u64* __restrict__ keys, const float4* __restrict__ sbox, const int* __restrict__ slabel,
    const int* __restrict__ keff, const u64* __restrict__ mask, const u64* __restrict__ nz,
    int key_stride, int box_stride, int wstride, int nzw, SweepOut o) {
  extern __shared__ __attribute__((aligned(16))) u64 s_lo[];      // [wstride] surely kept
  u64* s_hi = s_lo + wstride;                                     // [wstride] possibly kept
  int* s_total = reinterpret_cast<int*>(s_hi + wstride);          // [wstride + 1] kept before chunk c
  int* s_open = s_total + wstride + 1;                            // [3] a chunk still has open boxes
  const int j = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  RH_TL(2, 0);
  RH_TL_VALUE(2, 8, (long long)__builtin_amdgcn_s_memtime());
  const int K = keff[j];
  const int W = (K + 63) >> 6;
  const u64* kj = keys + (size_t)j * key_stride;
  const float4* bj = sbox + (size_t)j * box_stride;
  const int* lj = slabel ? slabel + (size_t)j * box_stride : nullptr;
  const u64* mj = mask + (size_t)j * box_stride * wstride;
  const u64* nj = nz + (size_t)j * box_stride * nzw;
  constexpr int kCached = kSlots > 0 ? kSlots : 1;
  const int nslots = kSlots > 0 ? kSlots : (W + kResolveWaves - 1) / kResolveWaves;

  if (threadIdx.x == 0) s_open[0] = 0;
  // ---- start: lo = nothing, hi = every box
  for (int s = 0; s < nslots; ++s) {
    const int c = s * kResolveWaves + wave;
    if (c < W && lane == 0) {
      const int left = K - c * 64;
      s_lo[c] = 0ull;
      s_hi[c] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    }
  }
  // ---- cached form (one nz word per box: at most 64 words): the nz words of all slots go out
  // together, then the first two predecessor words of every box, unconditionally (indices
  // clamped into the mask; a word that is not there is loaded from word 0 and discarded)
  BoxPreds pr[kCached];
  u64 okey[kCached];        // what the outputs need, requested now and used after the rounds
  float4 obox[kCached];
  int olab[kCached];
  if (kSlots > 0) {
    u64 bits[kCached];
    // (no load below sits behind a condition of its own -- labels are read through a pointer
    // that is always valid, absent words from word 0 and masked out afterwards: a load inside a
    // branch is followed by a wait for EVERYTHING in flight where the branch ends, which made
    // the slots' requests take a memory latency each instead of one together)
    const int* lj_any = lj ? lj : reinterpret_cast<const int*>(kj);
#pragma unroll
    for (int s = 0; s < kCached; ++s) {
      // The first two slots (2 048 boxes) are requested beside the load of K, rows clamped by
      // the arrays' size; the further ones only where the list reaches them, once K is known
      // (lists are mostly well below the capacity the slots are sized for).
      const int c = s * kResolveWaves + wave;
      const int row = min(c * 64 + lane, box_stride - 1);
      if (s < 2 || c < W) {
        bits[s] = nj[(size_t)row * nzw];
        okey[s] = kj[row];
        obox[s] = bj[row];
        olab[s] = lj_any[row];
      } else {
        bits[s] = 0ull;
        okey[s] = 0ull;
        obox[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        olab[s] = 0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u64 x[kCached], y[kCached];
    bool has0[kCached], has1[kCached];
#pragma unroll
    for (int s = 0; s < kCached; ++s) {
      const int c = s * kResolveWaves + wave;
      const int row = min(c * 64 + lane, box_stride - 1);
      const bool live = c < W && c * 64 + lane < K;
      u64 b = live ? bits[s] : 0ull;
      // a box without predecessor words is kept: the first round is done here
      const u64 free_now = __ballot(live && b == 0ull);
      if (c < W && lane == 0) s_lo[c] = free_now;
      has0[s] = b != 0ull;
      pr[s].i0 = has0[s] ? min((int)__builtin_ctzll(b), wstride - 1) : 0;
      b &= b - 1ull;
      has1[s] = b != 0ull;
      pr[s].i1 = has1[s] ? min((int)__builtin_ctzll(b), wstride - 1) : 0;
      b &= b - 1ull;
      pr[s].more = b != 0ull;
      x[s] = y[s] = 0ull;
      if (s < 2 || c < W) {
        x[s] = mj[(size_t)pr[s].i0 * box_stride + row];
        y[s] = mj[(size_t)pr[s].i1 * box_stride + row];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < kCached; ++s) {
      const int c = s * kResolveWaves + wave;
      // (a mask, not a conditional: a value that is only used under a condition lets the compiler
      // move its load under that condition too)
      pr[s].w0 = x[s] & earlier_bits(pr[s].i0, c * 64 + lane) & (has0[s] ? ~0ull : 0ull);
      pr[s].w1 = y[s] & earlier_bits(pr[s].i1, c * 64 + lane) & (has1[s] ? ~0ull : 0ull);
    }
  }
  __syncthreads();

  // one round of one chunk; returns whether the chunk still has open boxes
  auto round_of = [&](const int c, const BoxPreds& p) -> bool {
    const u64 lo = s_lo[c], hi = s_hi[c];
    const u64 open = hi & ~lo;
    if (open == 0ull) return false;               // wave-uniform: this chunk is settled
    const bool mine_open = (open >> lane) & 1ull;
    u64 d = 0ull, m = 0ull;
    if (mine_open) {
      const int row = c * 64 + lane;
      if (kSlots > 0) {
        d = (p.w0 & s_lo[p.i0]) | (p.w1 & s_lo[p.i1]);
        m = (p.w0 & s_hi[p.i0]) | (p.w1 & s_hi[p.i1]);
        // the further words only matter once the cached ones neither drop the box nor hold a
        // predecessor that is still open (while one is, the box stays open whatever they say)
        if (p.more && d == 0ull && m == 0ull)
          scan_words(s_lo, s_hi, nj + (size_t)row * nzw, nzw, mj, box_stride, row, 2, d, m);
      } else {
        scan_words(s_lo, s_hi, nj + (size_t)row * nzw, nzw, mj, box_stride, row, 0, d, m);
      }
    }
    const u64 now_kept = __ballot(mine_open && m == 0ull);      // no predecessor can still be kept
    const u64 now_dropped = __ballot(mine_open && d != 0ull);   // a predecessor is surely kept
    const u64 nlo = lo | now_kept, nhi = hi & ~now_dropped;
    if (lane == 0) {
      s_lo[c] = nlo;
      s_hi[c] = nhi;
    }
    return (nhi & ~nlo) != 0ull;
  };

  // ---- rounds
  RH_TL(2, 1);
  RH_TL_DO(int tl_rounds = 0;)
  bool gave_up = false;
  for (int round = 0;; ++round) {
    RH_TL_DO(++tl_rounds;)
    // every round settles at least the first open box, so K rounds always suffice; the bound
    // only keeps a logic error from hanging the device (the image's count is then -1)
    if (round > K + 1) {
      gave_up = true;
      break;
    }
    int open_any = 0;
    if (kSlots > 0) {
#pragma unroll
      for (int s = 0; s < kCached; ++s) {
        const int c = s * kResolveWaves + wave;
        if (c < W) open_any |= round_of(c, pr[s]);
      }
    } else {
      for (int c = wave; c < W; c += kResolveWaves) open_any |= round_of(c, pr[0]);
    }
    // "does any chunk still have an open box?" through a flag in LDS, one barrier per round
    // (the library's workgroup-wide OR costs well over a microsecond for 16 waves).  Three flags
    // in rotation: the one cleared here was last read two barriers ago.
    if (threadIdx.x == 0) s_open[(round + 1) % 3] = 0;
    if (open_any && lane == 0) s_open[round % 3] = 1;
    __syncthreads();
    const int again = s_open[round % 3];
    RH_TL_DO(if (round == 0) RH_TL(2, 4); if (round == 1) RH_TL(2, 6); if (round == 2) RH_TL(2, 7);)
    if (!again) break;
  }

  // ---- kept boxes before each chunk (wave 0), then the outputs
  RH_TL(2, 2);
  RH_TL_VALUE(2, 9, (long long)__builtin_amdgcn_s_memtime());
  RH_TL_VALUE(2, 5, tl_rounds);
  if (wave == 0) {
    int run = 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
      const int c = c0 + lane;
      const int n = c < W ? __popcll(s_lo[c]) : 0;
      int incl = n;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const int t = __shfl_up(incl, dd, 64);
        if (lane >= dd) incl += t;
      }
      if (c < W) s_total[c] = run + incl - n;
      run += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      s_total[W] = run;
      o.count[j] = gave_up ? -1 : min(run, o.max_keep);
    }
  }
  __syncthreads();
  auto emit = [&](const int c, const u64 key, const float4 box, const int lab) {
    const u64 kept = s_lo[c];
    if ((kept >> lane) & 1ull) {
      const int pos = s_total[c] + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < o.max_keep) {
        const size_t at = (size_t)j * o.out_stride + pos;
        if (o.keep_idx) o.keep_idx[at] = (long long)key_index(key);
        if (o.scores) o.scores[at] = key_score(key);
        if (o.labels) o.labels[at] = lj ? (long long)lab : 0ll;
        if (o.boxes) o.boxes[at] = box;
      }
    }
  };
  if (kSlots > 0) {
#pragma unroll
    for (int s = 0; s < kCached; ++s) {
      const int c = s * kResolveWaves + wave;
      if (c < W) emit(c, okey[s], obox[s], olab[s]);
    }
  } else {
    for (int c = wave; c < W; c += kResolveWaves) {
      const int row = min(c * 64 + lane, K - 1);
      emit(c, kj[row], bj[row], lj ? lj[row] : 0);
    }
  }
  // (at the end: their counters come by dependent scalar loads, which at the start of the
  // kernel held wave 0 back from its first requests)
  if (threadIdx.x == kResolveThreads - 1) {
    if (o.cand_out) o.cand_out[j] = o.counters[j];
    if (o.ovf_out) o.ovf_out[j] = o.counters[o.batch + j];
  }
  RH_TL(2, 3);
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

// Buffer geometry of sort + pair + resolve over B candidate lists.
struct NmsPlan {
  int batch;
  int cap;          // candidates per image the key buffers hold
  int key_stride;   // >= next_pow2(cap), >= 64
  int box_cap;      // boxes that reach the NMS (min(cap, topk))
  int wstride;      // mask words per box
  int nzw;          // words of a box's bitmap of non-zero mask words
  bool ranked;      // the rank sort runs: list-order arrays exist
  bool select;      // a top-k pre-selection cuts the candidate lists down first
  int clusters;     // clusters of 64 list entries (ranked) or 64 ranks (not ranked)
  bool hash;        // the pair search will be the centre hash: the rank sort leaves out what only
                    // the cluster form reads (boxes and ranks in list order, cluster summaries)
};

static NmsPlan make_plan(int batch, long long cap, int topk) {
  NmsPlan p;
  p.batch = batch;
  p.cap = (int)cap;
  p.key_stride = host_next_pow2(cap < 64 ? 64 : cap);   // the LDS sort works on whole 64-key segments
  p.box_cap = (topk > 0 && topk < cap) ? topk : (int)cap;
  if (p.box_cap < 1) p.box_cap = 1;
  p.wstride = (p.box_cap + 63) / 64;
  p.nzw = (p.wstride + 63) / 64;
  p.ranked = p.cap <= kRankMax;
  // worth its launch when at most half the room can pass (measured at 1024 px, 5600 candidates,
  // top-1000: the rank sort alone 36 us, selection + rank sort of the ~1100 kept 16 us)
  p.select = p.ranked && topk > 0 && 2ll * topk <= cap;
  // (Not for every call: with nothing to cut, the selection's one workgroup per image is just a
  // slower gather -- 608 px, 2000 candidates: 108 against 100 us alone, 74.6 against 69.7 us per
  // batch with four in flight.)
  p.clusters = ((p.ranked ? p.cap : p.box_cap) + 63) / 64;
  p.hash = false;     // (set by the callers that know the image's size: hash_geometry())
  return p;
}

// Limits of the NMS stages, checked before anything is launched.
static int check_plan(const NmsPlan& p) {
  RH_REQUIRE(resolve_lds_bytes(p.wstride) <= kResolveLdsLimit, RH_ESHAPE,
             "%d boxes per image exceed the LDS bitmaps of the NMS (at most %d): lower cand_cap or "
             "set pre_nms_topk", p.box_cap, (int)((kResolveLdsLimit - 16) / 20) * 64);
  return RH_OK;
}

struct PostWorkspace {
  int* tile_cnt;
  u64* tile_key;
  int* tile_lab;
  u64* keys;
  int* counters;     // [3][B]: candidates found, dropped, passed to NMS (written every call)
  float4* sbox;
  int* slabel;
  u64* mask;         // [B][wstride][box_cap] predecessor words; only non-zero words are written
  u64* nz;           // [B][box_cap][nzw] which words those are
  float4* cbox;      // [B][cap] boxes in list order (rank sort only)
  int4* cinfo;       // [B][cap] (rank or -1, label) in list order (rank sort only)
  float4* csum;      // [B][clusters][3] bounding box, area range and centres' box of each cluster
  u64* sel_key;      // [B][cap] keys kept by the top-k pre-selection, their labels, their number
  int* sel_lab;
  int* sel_count;
  int* split;        // [B] see TileCands
  size_t bytes;
};

static PostWorkspace carve_post(void* ws, const NmsPlan& p, long long num_tiles) {
  Carver cv(ws);
  PostWorkspace w;
  w.tile_cnt = cv.take<int>((size_t)num_tiles);
  w.tile_key = cv.take<u64>((size_t)num_tiles * kTileRows);
  w.tile_lab = cv.take<int>((size_t)num_tiles * kTileRows);
  w.keys = cv.take<u64>((size_t)p.batch * p.key_stride);
  w.counters = cv.take<int>((size_t)3 * p.batch);
  w.sbox = cv.take<float4>((size_t)p.batch * p.box_cap);
  w.slabel = cv.take<int>((size_t)p.batch * p.box_cap);
  w.mask = cv.take<u64>((size_t)p.batch * p.box_cap * p.wstride);
  w.nz = cv.take<u64>((size_t)p.batch * p.box_cap * p.nzw);
  w.cbox = cv.take<float4>(p.ranked ? (size_t)p.batch * p.cap : 0);
  w.cinfo = cv.take<int4>((size_t)p.batch * (p.ranked ? p.cap : p.box_cap));
  w.csum = cv.take<float4>((size_t)p.batch * p.clusters * kSummaryVecs);
  w.sel_key = cv.take<u64>(p.select ? (size_t)p.batch * p.cap : 0);
  w.sel_lab = cv.take<int>(p.select ? (size_t)p.batch * p.cap : 0);
  w.sel_count = cv.take<int>(p.select ? (size_t)p.batch : 0);
  w.split = cv.take<int>((size_t)p.batch);
  w.bytes = cv.off;
  return w;
}

static SortOut sort_out(const NmsPlan& p, const PostWorkspace& w) {
  SortOut so;
  so.keys = w.keys;
  so.sbox = w.sbox;
  so.slabel = w.slabel;
  so.nz = w.nz;
  so.cbox = p.ranked && !p.hash ? w.cbox : nullptr;    // (null: the rank sort writes neither cbox nor cinfo)
  so.cinfo = w.cinfo;
  so.csum = w.csum;
  so.mask = w.mask;
  so.key_stride = p.key_stride;
  so.box_stride = p.box_cap;
  so.list_stride = p.cap;
  so.nzw = p.nzw;
  so.wstride = p.wstride;
  so.clusters = p.ranked && p.hash ? 0 : p.clusters;   // (0: no cluster summaries)
  return so;
}

// Kernels that use more dynamic LDS than the default limit must opt in.  The limit is a property
// of the kernel, not of the launch, so it is always set to `ceiling`, the most the kernel is
// ever launched with: set to each launch's own size, a host thread with a small launch could
// lower it between another thread's call here and that thread's larger launch.
template <typename Kernel>
static int allow_lds(Kernel kernel, size_t bytes, size_t ceiling) {
  RH_REQUIRE(bytes <= ceiling, RH_ESHAPE, "%zu bytes of LDS asked for, the kernel's limit is %zu", bytes,
             ceiling);
  if (bytes > 48 * 1024)
    RH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)ceiling));
  return RH_OK;
}

// Compute units of the current device (asked once per device).
static int compute_units(int* out) {
  constexpr int kMaxDevices = 64;
  static std::atomic<int> cached[kMaxDevices];
  int device = -1;
  RH_CHECK_HIP(hipGetDevice(&device));
  RH_REQUIRE(device >= 0 && device < kMaxDevices, RH_EINVAL, "device %d out of range", device);
  int n = cached[device].load(std::memory_order_relaxed);
  if (n == 0) {
    RH_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    RH_REQUIRE(n > 0, RH_ELAUNCH, "device reports %d compute units", n);
    cached[device].store(n, std::memory_order_relaxed);
  }
  *out = n;
  return RH_OK;
}

static int launch_sort(const NmsPlan& p, const PostWorkspace& w, int topk, const SortSrc& src,
                       hipStream_t s) {
  const SortOut so = sort_out(p, w);
  if (p.ranked) {
    SortSrc from = src;
    if (p.select && src.from_tiles) {
      hipLaunchKernelGGL(topk_select_kernel, dim3(p.batch), dim3(kSelThreads), 0, s, w.counters,
                         p.batch, p.cap, topk, src, w.sel_key, w.sel_lab, w.sel_count);
      RH_CHECK_LAUNCH("topk_select_kernel");
      from.list_key = w.sel_key;
      from.list_lab = w.sel_lab;
      from.list_count = w.sel_count;
    }
    const size_t lds = rank_lds_bytes(p.cap);
    if (int e = allow_lds(sort_rank_kernel, lds, rank_lds_bytes(kRankMax))) return e;
    // Blocks per image: as many as the capacity could need at kRankKpl clusters each, but no
    // more than gives every compute unit one block -- a second block on a unit (even one that
    // finds nothing to rank) slows the first: 16 images x 32 blocks, half of them without work,
    // 16.6 us; 16 x 16 blocks 13.0 us.  Longer lists then take more clusters per block.
    int cus = 0;
    if (int e = compute_units(&cus)) return e;
    const int per_image = std::min((p.cap + kRankKeys - 1) / kRankKeys, std::max(1, cus / p.batch));
    hipLaunchKernelGGL(sort_rank_kernel, dim3(p.batch, per_image),
                       dim3(kRankThreads), lds, s, w.counters, p.batch, p.cap, topk, from, so);
    RH_CHECK_LAUNCH("sort_rank_kernel");
    return RH_OK;
  }
  if (!src.from_tiles) {
    hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)((src.count + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, s, src.scores, w.keys, src.count);
    RH_CHECK_LAUNCH("make_keys_kernel");
  }
  if (p.key_stride <= kLdsSortMax) {
    const size_t lds = (size_t)p.key_stride * sizeof(u64) + 16;
    if (int e = allow_lds(sort_lds_kernel, lds, (size_t)kLdsSortMax * sizeof(u64) + 16)) return e;
    hipLaunchKernelGGL(sort_lds_kernel, dim3(p.batch), dim3(kSortThreads), lds, s, w.counters, p.batch,
                       p.cap, topk, src, so);
    RH_CHECK_LAUNCH("sort_lds_kernel");
    return RH_OK;
  }
  const int P = p.key_stride;
  hipLaunchKernelGGL(sort_gather_keys_kernel, dim3(p.batch), dim3(kSortThreads), 0, s, w.keys,
                     w.counters, p.batch, p.cap, p.key_stride, P, src);
  dim3 gh((P / 2 + kThreads - 1) / kThreads, p.batch);
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1)
      hipLaunchKernelGGL(sort_step_kernel, gh, dim3(kThreads), 0, s, w.keys, p.key_stride, P, k, jj);
  hipLaunchKernelGGL(sort_finish_kernel, dim3((p.box_cap + kThreads - 1) / kThreads, p.batch),
                     dim3(kThreads), 0, s, w.counters, p.batch, p.cap, topk, src, so);
  RH_CHECK_LAUNCH("global bitonic sort");
  return RH_OK;
}

#ifndef RH_PAIR_HASH
#define RH_PAIR_HASH 1
#endif
// Geometry of the centre hash for boxes inside [0, width] x [0, height]; false when the hash
// form does not apply (the cluster form then runs).
static bool hash_geometry(const NmsPlan& p, float thresh, float width, float height, HashGeom* out) {
  if (!RH_PAIR_HASH || !(thresh >= 0.5f) || !(thresh < 1e30f) || p.box_cap > kHashMaxBoxes) return false;
  // short lists: the table costs more than the cluster form's tests (four 1024 px images, the
  // best 1 000 boxes of each: 7.5 against 6.8 us)
  if (p.box_cap < 1536) return false;
  if (!(width >= 1.0f) || !(height >= 1.0f) || !(width <= 16384.0f) || !(height <= 16384.0f)) return false;
  HashGeom g{};
  g.e0 = 8;
  int top = 0;
  while (top < 40 && std::ldexp(1.0f, top + 1) <= width * height) ++top;       // floor(log2(W * H))
  g.classes = std::min(std::max(top - g.e0 + 1, 1), kHashMaxClasses);
  int cells = 0;
  for (int c = 0; c < g.classes; ++c) {
    const float size = std::max(16.0f, std::sqrt(std::ldexp(1.0f, g.e0 + c)));
    g.inv_g[c] = 1.0f / size;
    g.nx[c] = (int)(width * g.inv_g[c]) + 1;
    g.ny[c] = (int)(height * g.inv_g[c]) + 1;
    g.base[c] = cells;
    cells += g.nx[c] * g.ny[c];
  }
  g.cells = cells;
  if (cells >= 65535 || hash_lds_bytes(p.box_cap, cells) > kHashLdsLimit) return false;
  *out = g;
  return true;
}

// width, height: the image the boxes were clipped to (0: unknown)
static int launch_nms(const NmsPlan& p, const PostWorkspace& w, float thresh, int per_class,
                      SweepOut out, hipStream_t s, float width = 0.0f, float height = 0.0f) {
  if (int e = check_plan(p)) return e;
  const int* keff = w.counters + 2 * p.batch;
  if (thresh > 0.0f && thresh < 1e30f && !p.ranked) {
    // (sort paths other than the rank sort: this kernel zeroes the predecessor words -- both
    // pair kernels OR into them -- and leaves the cluster summaries of the cluster form)
    hipLaunchKernelGGL(cluster_prepare_kernel, dim3(p.clusters, p.batch), dim3(64), 0, s, w.sbox,
                       w.slabel, keff, w.csum, w.cinfo, w.mask, p.box_cap, p.wstride, p.clusters);
    RH_CHECK_LAUNCH("cluster_prepare_kernel");
  }
  HashGeom geo;
  if (p.hash) {
    RH_REQUIRE(hash_geometry(p, thresh, width, height, &geo), RH_EINVAL,
               "the plan asks for the hash pair search, which these arguments do not allow");
    int cus = 0;
    if (int e = compute_units(&cus)) return e;
    // (workgroups per image, 608 px, ~2 000 boxes, 8 images: 16: 10.9 us, 32 or more: 9.2; one image:
    // 16: 10.2, 32: 8.6, 64: 8.1)
    const int per_image = std::max(1, std::min(64, cus / p.batch));
    const size_t lds = hash_lds_bytes(p.box_cap, geo.cells);
    auto kern = per_class ? nms_pairs_hash_kernel<true> : nms_pairs_hash_kernel<false>;
    if (int e = allow_lds(kern, lds, kHashLdsLimit)) return e;
    hipLaunchKernelGGL(kern, dim3(per_image, p.batch), dim3(kHashThreads), lds, s, w.sbox, w.slabel, keff,
                       w.mask, w.nz, p.box_cap, p.wstride, p.nzw, thresh, geo);
    RH_CHECK_LAUNCH("nms_pairs_hash_kernel");
  } else if (thresh > 0.0f && thresh < 1e30f) {
    // `slices` blocks (kPairThreads / 64 waves each) share the cluster pairs of one cluster (the bench workload
    // above with 1 / 2 / 4 / 8 slices: 96.0 / 93.3 / 95.9 / 99.7 us per call)
    int slices = 1;                                   // a power of two (the kernel masks with it)
    while (slices < 8 && slices * 64 <= p.clusters) slices *= 2;
    slices = std::max(1, slices * kThreads / kPairThreads);
    RH_REQUIRE((long long)p.clusters * slices * p.batch < (1ll << 31), RH_ESHAPE, "too many NMS workgroups");
    auto kern = per_class ? nms_pairs_kernel<true> : nms_pairs_kernel<false>;
    // Blocks per image: enough for three quarters of the list capacity at one unit (cluster,
    // slice) each; the blocks of a longer list take a second unit.  Lists are mostly well below
    // the capacity, and blocks that find no unit still cost their dispatch, above all while other
    // batches' kernels are being dispatched too.  608 px, 16 images, about 2 000 of 4 096
    // candidates, whole call on one stream / three batches in flight, 4-wave blocks: all units
    // 93.4 / 68.4 us, three quarters 92.7 / 67.8, half 93.6 / 67.3, a quarter 97.5 / 70; 8-wave
    // blocks (half as many): all units 92.3 / 65.9, three quarters 92.4 / 65.5; 16-wave blocks
    // 94.7 / 76.4; two 8-wave blocks per cluster 94.4 / 68-72.
    // (Only for grids of more than 512 blocks: for two images with full lists the second
    // units cost the pair kernel 31 instead of 20 us.)
    const int all_units = p.clusters * slices;
#ifndef RH_PAIR_GRID_LIMIT
#define RH_PAIR_GRID_LIMIT 512
#endif
    const int units = (long long)all_units * p.batch > RH_PAIR_GRID_LIMIT ? (all_units * 3 + 3) / 4 : all_units;
    hipLaunchKernelGGL(kern, dim3((unsigned)(units * p.batch)), dim3(kPairThreads), 0, s,
                       p.ranked ? w.cbox : w.sbox, w.cinfo, w.csum, w.counters,
                       p.select ? w.sel_count : w.counters, p.batch, w.mask, w.nz,
                       p.cap, p.ranked ? p.cap : p.box_cap, p.clusters, p.box_cap, p.wstride, p.nzw,
                       slices, thresh, p.ranked ? 1 : 0);
    RH_CHECK_LAUNCH("nms_pairs_kernel");
  } else {
    // thresh <= 0 (or not a number): every pair may count, the dense mask is the right form
    const long long tiles = (long long)p.wstride * (p.wstride + 1) / 2;
    long long gx = (tiles + 3) / 4;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(nms_mask_general_kernel, dim3((unsigned)gx, p.batch), dim3(kThreads), 0, s,
                       w.sbox, per_class ? w.slabel : nullptr, keff, w.mask, p.box_cap, p.wstride,
                       thresh, per_class);
    hipLaunchKernelGGL(nz_dense_kernel, dim3((p.box_cap + kThreads - 1) / kThreads, p.batch),
                       dim3(kThreads), 0, s, keff, w.nz, p.box_cap, p.nzw);
    RH_CHECK_LAUNCH("nms_mask_general_kernel");
  }
  const size_t lds = resolve_lds_bytes(p.wstride);
  auto kern = p.box_cap <= 1024   ? nms_resolve_kernel<1>
              : p.box_cap <= 2048 ? nms_resolve_kernel<2>
              : p.box_cap <= 4096 ? nms_resolve_kernel<4>
                                  : nms_resolve_kernel<0>;
  if (int e = allow_lds(kern, lds, kResolveLdsLimit)) return e;
  hipLaunchKernelGGL(kern, dim3(p.batch), dim3(kResolveThreads), lds, s, w.keys, w.sbox, w.slabel, keff,
                     w.mask, w.nz, p.key_stride, p.box_cap, p.wstride, p.nzw, out);
  RH_CHECK_LAUNCH("nms_resolve_kernel");
  return RH_OK;
}

#ifdef RH_TIMELINE
extern "C" int rh_tl_read(long long* out, size_t count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rh_tl), count * sizeof(long long));
}
#endif

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

int rh_sigmoid_f32(const float* x, float* out, int64_t count, rh_stream_t stream) {
  RH_REQUIRE(count >= 0, RH_ESHAPE, "count < 0");
  if (count == 0) return RH_OK;
  RH_REQUIRE(x && out, RH_EINVAL, "null pointer");
  long long blocks = (count + kThreads - 1) / kThreads;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                     x, out, (long long)count);
  RH_CHECK_LAUNCH("sigmoid_kernel");
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
  if (int e = check_plan(p)) return e;
  const size_t need = carve_post(nullptr, p, 0).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  const PostWorkspace w = carve_post(workspace, p, 0);
  SortSrc src{};
  src.scores = scores;
  src.boxes = reinterpret_cast<const float4*>(boxes);   // the caller's arrays: key index == row
  src.labels = labels;
  src.count = (int)k;
  src.from_tiles = 0;
  if (int e = launch_sort(p, w, 0, src, s)) return e;
  SweepOut out{};
  out.keep_idx = reinterpret_cast<long long*>(keep);
  out.count = keep_count;
  out.out_stride = (int)k;
  out.max_keep = (int)k;
  return launch_nms(p, w, iou_threshold, labels != nullptr, out, s);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// rh_nms_static: rh_nms for callers that cannot allocate.  The workspace rh_nms takes from
// its caller (sort keys, counters, sorted boxes and labels, predecessor mask) lives in a
// file-scope device array instead.  Each GPU holds its own copy of the array at its own
// address; in host code the array's name is only a handle, so the address is looked up once
// per device with hipGetSymbolAddress and that pointer is what the kernels receive.
// ---------------------------------------------------------------------------
#define RH_NMS_STATIC_BYTES (3u << 20)
static __device__ __align__(256) unsigned char scratch[RH_NMS_STATIC_BYTES];

namespace {
constexpr int kStaticMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kStaticMaxDevices];   // per device: address of `scratch` there, or null

// Address of the current device's copy of `scratch` (looked up on the first call per device).
int static_scratch(void** out, hipStream_t stream) {
  int device = -1;
  RH_CHECK_HIP(hipGetDevice(&device));
  RH_REQUIRE(device >= 0 && device < kStaticMaxDevices, RH_EINVAL, "device %d out of range", device);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    // the look-up is a runtime call of its own, not a stream operation: it cannot be captured
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess)
      RH_REQUIRE(cap == hipStreamCaptureStatusNone, RH_EINVAL,
                 "rh_nms_static: the first call on a device looks the scratch address up and must be "
                 "made outside stream capture (call it once, on any input, before capturing)");
    void* ptr = nullptr;
    RH_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    RH_REQUIRE(ptr != nullptr, RH_ELAUNCH, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return RH_OK;
}
}  // namespace

extern "C" {

int rh_nms_static(const float* boxes, const float* scores, const int32_t* labels, int64_t k,
                  float iou_threshold, int64_t* keep, int32_t* keep_count, rh_stream_t stream) {
  RH_REQUIRE(k >= 0, RH_ESHAPE, "k=%lld is negative", (long long)k);
  RH_REQUIRE(k <= RH_NMS_STATIC_MAX_BOXES, RH_ESHAPE,
             "k=%lld exceeds the %d boxes rh_nms_static has scratch for (use rh_nms with a "
             "caller-owned workspace)", (long long)k, RH_NMS_STATIC_MAX_BOXES);
  const size_t need = k > 0 ? rh_nms_workspace_bytes(k) : 0;
  RH_REQUIRE(need <= RH_NMS_STATIC_BYTES, RH_EWORKSPACE,
             "internal: %zu bytes needed for k=%lld, %u reserved", need, (long long)k,
             (unsigned)RH_NMS_STATIC_BYTES);
  void* ptr = nullptr;
  if (int e = static_scratch(&ptr, (hipStream_t)stream)) return e;
  // Nothing in the scratch has to start at zero: every stage writes what the next one reads,
  // for this call's k, before it is read.
  return rh_nms(boxes, scores, labels, k, iou_threshold, keep, keep_count, ptr,
                RH_NMS_STATIC_BYTES, stream);
}

// A logit whose sigmoid is certainly below `thresh` in float32 (and -inf / +inf when every /
// no probability can exceed the threshold): elements at or below it need no sigmoid.
static float logit_prefilter(float thresh) {
  if (!(thresh > 0.0f)) return -INFINITY;
  if (thresh >= 1.0f) return INFINITY;
  const double t = (double)thresh;
  return (float)(log(t / (1.0 - t)) - 1e-3);
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
  return carve_post(nullptr, p, ((long long)batch * n + kTileRows - 1) / kTileRows).bytes;
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
  NmsPlan p = make_plan(batch, q->cand_cap, q->pre_nms_topk);
  if (int e = check_plan(p)) return e;
  HashGeom unused;
  p.hash = hash_geometry(p, q->iou_thresh, q->width, q->height, &unused);
  const int total_rows = (int)((long long)batch * n);
  const int num_tiles = (total_rows + kTileRows - 1) / kTileRows;
  const size_t need = carve_post(nullptr, p, num_tiles).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const PostWorkspace w = carve_post(workspace, p, num_tiles);
  const bool splits = n > kTileRows;
  const TileCands tc{w.tile_cnt, w.tile_key, w.tile_lab, num_tiles, splits ? w.split : nullptr,
                     splits ? (int)n : 0};
  const int out_cap = q->max_det > 0 ? q->max_det : p.box_cap;

  if (stages & RH_POST_DECODE) {
    const bool tile = num_classes >= 4 && num_classes < 8192 && aligned16(classifications);
    if (tile) {
      const bool u5 = ((kTileRows * num_classes / 4 + kThreads - 1) / kThreads) % 5 == 0;
      auto kern = q->from_logits
                      ? (u5 ? score_scan_tile_kernel<5, true> : score_scan_tile_kernel<4, true>)
                      : (u5 ? score_scan_tile_kernel<5, false> : score_scan_tile_kernel<4, false>);
      hipLaunchKernelGGL(kern, dim3(num_tiles), dim3(kThreads), 0, s, classifications, tc,
                         num_classes, total_rows, div_magic((unsigned)num_classes),
                         q->score_thresh, logit_prefilter(q->score_thresh));
      RH_CHECK_LAUNCH("score_scan_tile_kernel");
    } else {
      auto kern = q->from_logits ? score_scan_rows_kernel<true> : score_scan_rows_kernel<false>;
      hipLaunchKernelGGL(kern, dim3(num_tiles), dim3(kThreads), 0, s, classifications, tc,
                         num_classes, total_rows, q->score_thresh);
      RH_CHECK_LAUNCH("score_scan_rows_kernel");
    }
  }
  if (stages & RH_POST_SORT) {
    SortSrc src{};
    src.tc = tc;
    src.anchors = reinterpret_cast<const float4*>(anchors);
    src.reg = reinterpret_cast<const float4*>(regressions);
    src.bc = make_coder(nullptr, nullptr);
    src.width = q->width;
    src.height = q->height;
    src.N = (int)n;
    src.from_tiles = 1;
    if (int e = launch_sort(p, w, q->pre_nms_topk, src, s)) return e;
  }
  if (stages & RH_POST_NMS) {
    SweepOut out{};
    out.scores = out_scores;
    out.labels = reinterpret_cast<long long*>(out_labels);
    out.boxes = reinterpret_cast<float4*>(out_boxes);
    out.count = out_count;
    out.out_stride = out_cap;
    out.max_keep = out_cap;
    out.counters = w.counters;
    out.cand_out = cand_count;
    out.ovf_out = overflow;
    out.batch = batch;
    if (int e = launch_nms(p, w, q->iou_thresh, q->per_class, out, s, q->width, q->height)) return e;
  }
  return RH_OK;
}

static int make_level_geom(int num_levels, const int32_t* heights, const int32_t* widths, int a,
                           LevelGeom* g, long long* n_out) {
  RH_REQUIRE(num_levels >= 1 && num_levels <= kMaxLevels, RH_ESHAPE, "num_levels=%d not in [1, %d]",
             num_levels, kMaxLevels);
  RH_REQUIRE(heights && widths, RH_EINVAL, "null level shapes");
  RH_REQUIRE(a >= 1 && a <= 65535, RH_ESHAPE, "anchors_per_cell=%d out of range", a);
  long long cells = 0;
  int chunks = 0;
  for (int l = 0; l < num_levels; ++l) {
    RH_REQUIRE(heights[l] >= 1 && widths[l] >= 1, RH_ESHAPE, "level %d has shape %d x %d", l,
               heights[l], widths[l]);
    const long long hw = (long long)heights[l] * widths[l];
    RH_REQUIRE(hw * a < (1ll << 31) && (cells + hw) * a < (1ll << 31), RH_ESHAPE, "levels too large");
    g->hw[l] = (int)hw;
    g->anchor_off[l] = (int)(cells * a);
    g->chunk_begin[l] = chunks;
    chunks += (int)((hw * a + kLvlChunk - 1) / kLvlChunk);
    cells += hw;
  }
  g->anchor_off[num_levels] = (int)(cells * a);
  g->chunk_begin[num_levels] = chunks;
  g->num_levels = num_levels;
  g->a = a;
  *n_out = cells * a;
  return RH_OK;
}

size_t rh_postprocess_levels_workspace_bytes(int32_t batch, int32_t num_levels,
                                             const int32_t* heights, const int32_t* widths,
                                             int32_t anchors_per_cell, int32_t num_classes,
                                             const rh_post_params* params) {
  LevelGeom g{};
  long long n = 0;
  if (make_level_geom(num_levels, heights, widths, anchors_per_cell, &g, &n)) return 0;
  if (check_post(batch, n, num_classes, params)) return 0;
  const NmsPlan p = make_plan(batch, params->cand_cap, params->pre_nms_topk);
  return carve_post(nullptr, p, (long long)batch * g.chunk_begin[num_levels] * (kLvlChunk / 64)).bytes;
}

int rh_postprocess_levels(const float* const* cls_levels, const float* const* reg_levels,
                          const float* anchors, int32_t batch, int32_t num_levels,
                          const int32_t* heights, const int32_t* widths, int32_t anchors_per_cell,
                          int32_t num_classes, const rh_post_params* q, float* out_scores,
                          int64_t* out_labels, float* out_boxes, int32_t* out_count,
                          int32_t* cand_count, int32_t* overflow, void* workspace,
                          size_t workspace_bytes, int32_t stages, rh_stream_t stream) {
  LevelGeom g{};
  long long n = 0;
  if (int e = make_level_geom(num_levels, heights, widths, anchors_per_cell, &g, &n)) return e;
  if (int e = check_post(batch, n, num_classes, q)) return e;
  RH_REQUIRE(cls_levels && reg_levels && anchors && out_scores && out_labels && out_boxes && out_count,
             RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(anchors) && aligned16(out_boxes), RH_EALIGN,
             "anchors and out_boxes must be 16-byte aligned");
  RH_REQUIRE((stages & ~RH_POST_ALL) == 0 && stages != 0, RH_EINVAL, "bad stages mask %d", stages);
  for (int l = 0; l < num_levels; ++l) {
    RH_REQUIRE(cls_levels[l] && reg_levels[l], RH_EINVAL, "null pointer for level %d", l);
    // offsets inside one image's [A*C, H*W] block are 32-bit in the kernel
    RH_REQUIRE((long long)anchors_per_cell * num_classes * g.hw[l] < (1ll << 31), RH_ESHAPE,
               "level %d: A*C*H*W does not fit 31 bits", l);
    g.cls[l] = cls_levels[l];
    g.reg[l] = reg_levels[l];
    g.vec4[l] = g.hw[l] % 4 == 0 && aligned16(cls_levels[l]);
  }
  const int chunks = g.chunk_begin[num_levels];
  const int tiles_per_image = chunks * (kLvlChunk / 64);
  const long long num_tiles = (long long)batch * tiles_per_image;
  RH_REQUIRE(num_tiles < (1ll << 31), RH_ESHAPE, "too many tile lists");
  NmsPlan p = make_plan(batch, q->cand_cap, q->pre_nms_topk);
  if (int e = check_plan(p)) return e;
  HashGeom unused;
  p.hash = hash_geometry(p, q->iou_thresh, q->width, q->height, &unused);
  const size_t need = carve_post(nullptr, p, num_tiles).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const PostWorkspace w = carve_post(workspace, p, num_tiles);
  const TileCands tc{w.tile_cnt, w.tile_key, w.tile_lab, (int)num_tiles, nullptr, 0};
  const int out_cap = q->max_det > 0 ? q->max_det : p.box_cap;

  if (stages & RH_POST_DECODE) {
    auto kern = q->from_logits ? score_scan_levels_kernel<true> : score_scan_levels_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(chunks, batch), dim3(kThreads), 0, s, g, tc, (int)n, num_classes,
                       q->score_thresh, logit_prefilter(q->score_thresh));
    RH_CHECK_LAUNCH("score_scan_levels_kernel");
  }
  if (stages & RH_POST_SORT) {
    SortSrc src{};
    src.tc = tc;
    src.anchors = reinterpret_cast<const float4*>(anchors);
    src.bc = make_coder(nullptr, nullptr);
    src.width = q->width;
    src.height = q->height;
    src.N = (int)n;
    src.from_tiles = 1;
    src.tiles_per_image = tiles_per_image;
    src.lv = g;
    if (int e = launch_sort(p, w, q->pre_nms_topk, src, s)) return e;
  }
  if (stages & RH_POST_NMS) {
    SweepOut out{};
    out.scores = out_scores;
    out.labels = reinterpret_cast<long long*>(out_labels);
    out.boxes = reinterpret_cast<float4*>(out_boxes);
    out.count = out_count;
    out.out_stride = out_cap;
    out.max_keep = out_cap;
    out.counters = w.counters;
    out.cand_out = cand_count;
    out.ovf_out = overflow;
    out.batch = batch;
    if (int e = launch_nms(p, w, q->iou_thresh, q->per_class, out, s, q->width, q->height)) return e;
  }
  return RH_OK;
}

}  // extern "C"
