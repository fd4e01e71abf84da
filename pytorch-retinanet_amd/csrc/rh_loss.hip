// Training half of the dense-head path on gfx950: GT compaction, anchor x GT IoU
// assignment, and the fused focal + smooth-L1 forward/backward stream (on the flat
// [B,N,C] / [B,N,4] tensors or on the heads' per-level [B, A*C, H, W] outputs).
//
// Reference behaviour restated here: models/retinanet/losses.py:6-23 (IoU),
// :48-49 (GT filter), :57 (clamp), :59-61 (max/argmax), :67-94 (targets, focal
// loss, normaliser), :98-136 (box targets, smooth-L1), :139-141 (batch mean).
//
// Everything is HBM-bound float32 streaming; there is no contraction, so no MFMA.
// The file is compiled with -ffp-contract=off: IoU values feed thresholds and an
// arg-max whose integer results must equal the reference's bit for bit.
#include "rh_common.h"

namespace rh {

// One valid ground-truth box, as compacted by assign_kernel (32 bytes).
struct __attribute__((aligned(16))) GtRec {
  float x1, y1, x2, y2;
  float area;
  int label;   // truncated class, or -1 when outside [0, C)
  int pad0, pad1;
};

// Packed per-anchor assignment that the stream kernel consumes.
//   >= 0 : positive, (gt_index << 16) | label   (label 0xFFFF: class out of range)
//   -1   : negative (all-zero target row)
//   -2   : ignored (no loss, no gradient)
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kCodeNeg = -1;
constexpr int kCodeIgn = -2;
constexpr int kNoLabel = 0xFFFF;

constexpr int kTileRows = 64;      // anchors per stream tile
constexpr int kThreads = 256;
constexpr int kGenericBlocks = 64; // blocks per image of the generic row kernel

// ---------------------------------------------------------------------------
// IoU of one anchor with one GT, float32, operation order of losses.py:9-21.
// Returns 0 without dividing when the boxes do not overlap: the reference clamps
// iw and ih at 0, so its intersection is 0 and 0 / ua == 0 exactly.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float iou_pair(const float4 a, const float area_a, const float4 b,
                                          const float area_b) {
  const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  const float ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  if (!(fminf(iw, ih) > 0.0f)) return 0.0f;
  const float inter = iw * ih;
  float ua = (area_a + area_b) - inter;
  ua = fmaxf(ua, 1e-8f);
  return inter / ua;
}

// ---------------------------------------------------------------------------
// Assignment, one launch: GT filter + IoU + max / first arg-max + thresholds + labels.
//
// A wave works on its own: no LDS staging, no barriers before the final count.  Lane r holds
// annotation row r of the image (64 rows at a time); rows with label != -1 are the valid GT
// boxes, in order (losses.py:48-49; areas losses.py:7; labels truncated like .long(),
// losses.py:79).  The wave first forms the bounding box of all its anchors (consecutive
// anchors lie in neighbouring cells, so on the fine pyramid levels the box is small): a GT box
// that does not reach into it has an empty intersection with every anchor of the wave -- for
// each one min(ax2, bx2) - max(ax1, bx1) <= 0 or the same in y, exactly, in float32 -- so its
// IoU is 0 and it can neither raise the maximum nor move the first arg-max.  One ballot over
// the rows therefore leaves the few boxes the wave has to look at; they are broadcast from
// their lanes one at a time and compared with kAssignPer anchors per thread (strided by the
// block size, so loads and stores stay coalesced).  Most (anchor, GT) pairs never reach the
// vector ALU.  Wave 0 of block 0 of each image also publishes the compacted records, the valid
// count and the status word for the kernels that follow.  Positives are counted per block
// into blk_pos[image][block]: no atomics and nothing to zero beforehand.
// Max / first arg-max (losses.py:61), thresholds (:70-72), label lookup (:76-79), count (:74).
// ---------------------------------------------------------------------------
#ifndef RH_ASSIGN_PER
#define RH_ASSIGN_PER 2
#endif
constexpr int kAssignPer = RH_ASSIGN_PER;   // anchors per thread
constexpr int kAssignBlock = kThreads * kAssignPer;   // anchors per block
// Long annotation lists take the split form of the kernel: the four waves of a block share one
// wave's worth of anchors and each scans every fourth 64-row chunk of the list, so the serial
// walk over the boxes that survive the cull -- on the coarse pyramid levels nearly all of them --
// is a quarter as long; the waves' maxima are merged through LDS (larger IoU first, then lower GT
// index: the first arg-max).  "Long" is more than kSplitRows rows when the maximum IoU, arg-max or
// state of every anchor is wanted (rh_assign); when only the loss's packed codes are, boxes that
// cannot reach the lower threshold with any anchor of the wave are culled as well, few survive
// even on the coarse levels, and the plain form stays the faster one up to kSplitRowsCodes rows
// (4 x 196 416 anchors, 256 boxes per image: assignment 18.7 us split against 13.1 plain, the
// step that carries it 111.6 against 96.6 us; 600 boxes: 113 against 110; 1500: 79 against 89).
#ifndef RH_SPLIT_ROWS
#define RH_SPLIT_ROWS 128
#endif
#ifndef RH_SPLIT_ROWS_CODES
#define RH_SPLIT_ROWS_CODES 1024
#endif
constexpr int kSplitRows = RH_SPLIT_ROWS;
constexpr int kSplitRowsCodes = RH_SPLIT_ROWS_CODES;
// Anchors per thread in the split form: six make a third as many workgroups as two, each paying
// the walk over the annotation rows once (256 boxes per image: 18.7 -> 16.2 us, the carrying step
// 111.6 -> 100.3 us; eight spill inside the stream kernel's 96 registers).
#ifndef RH_SPLIT_PER
#define RH_SPLIT_PER 6
#endif
constexpr int kSplitPer = RH_SPLIT_PER;
constexpr int kSplitAnchors = 64 * kSplitPer;
static inline bool assign_split(int m, bool codes_only) {
  return m > (codes_only ? kSplitRowsCodes : kSplitRows);
}
static inline int assign_blocks(long long n, bool split) {
  const int per = split ? kSplitAnchors : kAssignBlock;
  return (int)((n + per - 1) / per);
}

// Where the packed codes go.  num_levels == 0: code[j*N + n], the order of the flat [B,N,C]
// tensors.  Otherwise the head-layout kernel's order: within pyramid level l the codes of
// anchor a form a plane of H*W cells, code[j*N + off_l + a*hw_l + cell], so that a workgroup
// that walks the cells of one (level, anchor) reads its codes as one contiguous run.
constexpr int kMaxLevels = 8;
struct CodeLayout {
  int num_levels;
  int a;                           // anchors per cell
  int anchor_off[kMaxLevels + 1];  // first anchor of each level, then N
  int hw[kMaxLevels];
};

struct AnnRow {
  float x1, y1, x2, y2, lab;
};
__device__ __forceinline__ AnnRow load_row(const float* __restrict__ aj, const int m, const int M) {
  const int mc = min(m, M - 1);       // clamped: all five values go out at once
  AnnRow r;
  r.x1 = aj[mc * 5 + 0];
  r.y1 = aj[mc * 5 + 1];
  r.x2 = aj[mc * 5 + 2];
  r.y2 = aj[mc * 5 + 3];
  r.lab = aj[mc * 5 + 4];
  return r;
}

// Build-time diagnostic (-DRH_TIMELINE): thread 0 of every block stores wall-clock stamps at
// the marked places; rh_dbg_read() copies them out.  Not part of the shipped library.
#ifdef RH_TIMELINE
__device__ long long rh_dbg[8 * 65536];
#define RH_MARK(slot) do { if (threadIdx.x == 0) rh_dbg[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define RH_MARK(slot)
#endif
// Epochs tag the partial records of one loss call (see StreamFinal): odd, and mixed with the
// clock so that a workspace whose epoch word was overwritten (the buffer served another shape in
// between, or is fresh and zero like the last fresh one) does not walk the same sequence again.
__device__ __forceinline__ unsigned next_epoch(const unsigned e) {
  return ((e ^ (unsigned)wall_clock64()) * 1664525u + 1013904223u) | 1u;
}

// Everything the assignment of one batch reads and writes.
struct AssignArgs {
  const float4* anchors;
  const float* ann;
  GtRec* gt;
  int* gt_count;
  int* status;
  int* code;
  unsigned char* state;
  int* argmax;
  int* label;
  float* iou_max;
  int* blk_pos;
  int batch;
  unsigned* epoch;   // of the loss call that follows (null: none): advanced here
  // record atoms of the loss call that follows (the tile partials, then the per-image results):
  // the assignment stage clears them, so that no byte the workspace held before -- another
  // caller's data, records of another shape -- can be taken for a record of that call
  unsigned long long* clear_a;
  long long clear_a_count;
  unsigned long long* clear_b;
  int clear_b_count;
  int N, M, C;
  float neg_thresh, pos_thresh;
  CodeLayout lay;
};

// The work of block bx (of nblk) of image j.  Blocks are best started from the last one down:
// the blocks of the coarse pyramid levels, whose large anchors meet most GT boxes, are the long
// ones.
template <bool kSplit>
__device__ __forceinline__ void assign_block(const AssignArgs& aa, const int j, const int bx,
                                             const int nblk) {
  __shared__ int s_cnt[kThreads / 64];
  constexpr int kPer = kSplit ? kSplitPer : kAssignPer;     // anchors per thread
  constexpr int kShare = kSplit ? kSplitAnchors : 1;
  __shared__ float s_best[kThreads / 64][kShare];
  __shared__ int s_bidx[kThreads / 64][kShare];
  __shared__ int s_blab[kThreads / 64][kShare];
  const float4* __restrict__ anchors = aa.anchors;
  const float* __restrict__ ann = aa.ann;
  GtRec* __restrict__ gt = aa.gt;
  int* __restrict__ gt_count = aa.gt_count;
  int* __restrict__ status = aa.status;
  int* __restrict__ code = aa.code;
  unsigned char* __restrict__ state = aa.state;
  int* __restrict__ argmax = aa.argmax;
  int* __restrict__ label = aa.label;
  float* __restrict__ iou_max = aa.iou_max;
  int* __restrict__ blk_pos = aa.blk_pos;
  const int N = aa.N, M = aa.M, C = aa.C;
  const float neg_thresh = aa.neg_thresh, pos_thresh = aa.pos_thresh;
  const CodeLayout& lay = aa.lay;
  RH_MARK(0);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // anchors of thread t: base + k * stride.  Split form: every wave holds the same anchors.
  constexpr int kStride = kSplit ? 64 : kThreads;
  const int base = kSplit ? bx * kSplitAnchors + lane : bx * kAssignBlock + (int)threadIdx.x;
  const float* aj = ann + (size_t)j * M * 5;
  const bool publisher = bx == 0 && wave == 0;

  float4 a[kPer];
  float area_a[kPer], best[kPer];
  int best_idx[kPer], best_lab[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    // lanes past the last anchor repeat it: their results are not stored, and the wave's
    // bounding box stays what the real anchors make it
    a[k] = anchors[min(base + k * kStride, N - 1)];
  }
  AnnRow next = M > 0 ? load_row(aj, lane, M) : AnnRow{0.f, 0.f, 0.f, 0.f, -1.0f};
  // (after the requests above, not before: the clearing's own kernel arguments arrive by scalar
  // loads that would otherwise stand between the start of the block and its first request)
  __builtin_amdgcn_sched_barrier(0);
  if (aa.clear_a) {       // this block's share of the record atoms, interleaved by thread
    const long long step = (long long)aa.batch * nblk * kThreads;
    for (long long i = ((long long)j * nblk + bx) * kThreads + threadIdx.x; i < aa.clear_a_count; i += step)
      aa.clear_a[i] = 0ull;
    if (bx == 0 && j == 0)
      for (int i = threadIdx.x; i < aa.clear_b_count; i += kThreads) aa.clear_b[i] = 0ull;
  }
  float lo_x = a[0].x, lo_y = a[0].y, hi_x = a[0].z, hi_y = a[0].w;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    area_a[k] = (a[k].z - a[k].x) * (a[k].w - a[k].y);
    best[k] = 0.0f;   // every IoU is >= 0, so "first maximum" starts at (0, index 0)
    best_idx[k] = 0;
    best_lab[k] = -2; // "label of valid GT 0", resolved below if nothing beats IoU 0
    if (k > 0) {
      lo_x = vmin(lo_x, a[k].x);
      lo_y = vmin(lo_y, a[k].y);
      hi_x = vmax(hi_x, a[k].z);
      hi_y = vmax(hi_y, a[k].w);
    }
  }
  const float ux1 = wave_extreme<false>(lo_x), uy1 = wave_extreme<false>(lo_y);
  const float ux2 = wave_extreme<true>(hi_x), uy2 = wave_extreme<true>(hi_y);
  // When only the packed codes are wanted (the loss), a pair matters only if its IoU can reach
  // the lower of the two thresholds t: below it the anchor's state does not depend on the pair
  // (an anchor whose other IoUs all stay below t is a negative either way), and a maximum at or
  // above t is never one of the pairs left out, so state, arg-max and label are unchanged.  The
  // IoU is at most the ratio of the smaller area to the larger (the float32 quotient stays
  // within 1e-6 of that: both areas and the intersection are products of rounded differences,
  // and rounding is monotone), so a box whose area is below t times the smallest anchor area of
  // the wave, or above the largest divided by t, is dropped with the same ballot as the boxes
  // that miss the wave's bounding box.  The maximum IoU itself (rh_assign's output) needs them all.
  float a_lo = 0.0f, a_hi = __builtin_inff();     // a box is kept when a_lo <= its area <= a_hi
  float t_side = -1.0f;                           // ... and t_side of each of its sides reaches into the wave's box
  if (!state && !argmax && !label && !iou_max) {
    float amin = area_a[0], amax = area_a[0];
#pragma unroll
    for (int k = 1; k < kPer; ++k) {
      amin = vmin(amin, area_a[k]);
      amax = vmax(amax, area_a[k]);
    }
    const float t = fminf(neg_thresh, pos_thresh) * 0.9999f;
    if (t > 0.0f) {
      a_lo = wave_extreme<false>(amin) * t;
      a_hi = wave_extreme<true>(amax) / t;
      t_side = t * 0.999f;
    }
  }

  RH_MARK(1);
  int cnt = 0;        // valid GT rows seen so far
  int bad = 0;
  int lab0 = -1;      // label of the image's first valid GT (arg-max of an all-zero IoU row)
  for (int m0 = 0; m0 < M; m0 += 64) {
    const AnnRow r = next;
    if (m0 + 64 < M) next = load_row(aj, m0 + 64 + lane, M);   // in flight during this chunk
    const bool valid = m0 + lane < M && r.lab != -1.0f;
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) continue;
    const float rarea = (r.x2 - r.x1) * (r.y2 - r.y1);
    int li = (int)r.lab;
    if (!(r.lab > -1.0f && r.lab < (float)C)) {   // also catches NaN
      li = -1;
      bad |= valid ? RH_STATUS_LABEL_RANGE : 0;
    }
    // a NaN or infinite coordinate: x - x is 0 exactly for finite x only
    const float span = (r.x1 - r.x1) + (r.y1 - r.y1) + (r.x2 - r.x2) + (r.y2 - r.y2);
    if (valid && !(span == 0.0f)) bad |= RH_STATUS_NONFINITE_BOX;
    const int before = __popcll(vm & ((1ull << lane) - 1ull));   // valid rows in lower lanes
    if (publisher && valid) {
      GtRec g;
      g.x1 = r.x1;
      g.y1 = r.y1;
      g.x2 = r.x2;
      g.y2 = r.y2;
      g.area = rarea;
      g.label = li;
      g.pad0 = g.pad1 = 0;
      gt[(size_t)j * M + cnt + before] = g;
    }
    if (cnt == 0) lab0 = __builtin_amdgcn_readlane(li, __builtin_ctzll(vm));
    if (kSplit && ((m0 >> 6) & (kThreads / 64 - 1)) != wave) {   // another wave's chunk
      cnt += __popcll(vm);
      continue;
    }
    // rows whose box reaches into the wave's bounding box; a NaN coordinate is never culled
    // (codes only, as above: IoU >= t needs the intersection to cover t of the box's own area,
    // hence t of its width and of its height; an anchor's overlap with the box is within the
    // overlap of the wave's bounding box with it)
    const float ovx = vmin(r.x2, ux2) - vmax(r.x1, ux1), ovy = vmin(r.y2, uy2) - vmax(r.y1, uy1);
    const bool apart = (r.x2 <= ux1) | (r.x1 >= ux2) | (r.y2 <= uy1) | (r.y1 >= uy2) |
                       (rarea < a_lo) | (rarea > a_hi) | (ovx < (r.x2 - r.x1) * t_side) |
                       (ovy < (r.y2 - r.y1) * t_side);
    unsigned long long hm = __ballot(valid & !apart);
    while (hm) {
      const int src = __builtin_ctzll(hm);
      hm &= hm - 1ull;
      const float4 b = make_float4(lane_value(r.x1, src), lane_value(r.y1, src),
                                   lane_value(r.x2, src), lane_value(r.y2, src));
      const int gi = cnt + __popcll(vm & ((1ull << src) - 1ull));
      // overlap test for the thread's anchors first; one (rare) branch per GT box
      float iw[kPer], ih[kPer], mm[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        iw[k] = vmin(a[k].z, b.z) - vmax(a[k].x, b.x);
        ih[k] = vmin(a[k].w, b.w) - vmax(a[k].y, b.y);
        mm[k] = vmin(iw[k], ih[k]);
      }
      float any = mm[0];
#pragma unroll
      for (int k = 1; k < kPer; ++k) any = vmax(any, mm[k]);
      if (any > 0.0f) {
        const float ab = lane_value(rarea, src);
        const int lb = __builtin_amdgcn_readlane(li, src);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          if (mm[k] > 0.0f) {
            // losses.py:12-21 (iw, ih > 0 here, so the clamps at 0 are no-ops)
            const float inter = iw[k] * ih[k];
            float ua = (area_a[k] + ab) - inter;
            ua = fmaxf(ua, 1e-8f);
            // The correctly rounded quotient can exceed `best` only if inter > best*ua*(1 - 2^-23);
            // test that first (with margin) and divide only for the few boxes that can win.
            if (inter >= (best[k] * ua) * 0.999999f) {
              const float v = inter / ua;
              if (v > best[k]) {
                best[k] = v;
                best_idx[k] = gi;
                best_lab[k] = lb;
              }
            }
          }
        }
      }
    }
    cnt += __popcll(vm);
  }
  RH_MARK(2);

  if (kSplit) {
    // merge the waves' results: larger IoU wins, equal IoU goes to the lower GT index (a wave
    // that found nothing holds IoU 0 at index 0, which only a positive IoU displaces)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      s_best[wave][lane + k * 64] = best[k];
      s_bidx[wave][lane + k * 64] = best_idx[k];
      s_blab[wave][lane + k * 64] = best_lab[k];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
#pragma unroll
      for (int w = 1; w < kThreads / 64; ++w) {
        const float v = s_best[w][lane + k * 64];
        const int vi = s_bidx[w][lane + k * 64];
        if (v > best[k] || (v == best[k] && vi < best_idx[k])) {
          best[k] = v;
          best_idx[k] = vi;
          best_lab[k] = s_blab[w][lane + k * 64];
        }
      }
    }
  }

  int npos = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int n = base + k * kStride;
    int st, cd, lb = -1;
    if (cnt == 0) {
      st = RH_IGNORE;      // losses.py:51-55: the image contributes nothing
      cd = kCodeIgn;
    } else if (best[k] >= pos_thresh) {
      st = RH_POSITIVE;
      lb = best_lab[k] == -2 ? lab0 : best_lab[k];
      cd = (best_idx[k] << 16) | (lb < 0 ? kNoLabel : lb);
    } else if (best[k] < neg_thresh) {
      st = RH_NEGATIVE;
      cd = kCodeNeg;
    } else {
      st = RH_IGNORE;
      cd = kCodeIgn;
    }
    if (n < N) {
      const size_t o = (size_t)j * N + n;
      if (code) {
        // (plane order scatters a wave's stores over 64 cache lines; staging the block's codes
        // in LDS and writing them plane by plane was measured and is slower, 15.4 against 13.2 us)
        size_t oc = o;
        if (lay.num_levels) {
          int l = 0;
          while (l + 1 < lay.num_levels && n >= lay.anchor_off[l + 1]) ++l;
          const int r = n - lay.anchor_off[l];
          const int cell = r / lay.a;
          oc = (size_t)j * N + lay.anchor_off[l] + (size_t)(r - cell * lay.a) * lay.hw[l] + cell;
        }
        code[oc] = cd;
      }
      if (state) state[o] = (unsigned char)st;
      if (argmax) argmax[o] = best_idx[k];
      if (label) label[o] = lb;
      if (iou_max) iou_max[o] = best[k];
      npos += st == RH_POSITIVE;
    }
  }
  npos = wave_sum(npos);
  if (kSplit) {             // only wave 0 is left, and it holds the block's anchors
    if (lane == 0) blk_pos[(size_t)j * nblk + bx] = npos;
  } else {
    if (lane == 0) s_cnt[wave] = npos;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; ++w) t += s_cnt[w];
      blk_pos[(size_t)j * nblk + bx] = t;
    }
  }
  if (publisher) {          // this wave has seen every row of the image
    const int bits = (__ballot(bad & RH_STATUS_LABEL_RANGE) != 0ull ? RH_STATUS_LABEL_RANGE : 0) |
                     (__ballot(bad & RH_STATUS_NONFINITE_BOX) != 0ull ? RH_STATUS_NONFINITE_BOX : 0);
    if (lane == 0) {
      gt_count[j] = cnt;
      if (status) status[j] = bits;
      if (aa.epoch && j == 0) *aa.epoch = next_epoch(*aa.epoch);   // the loss call that follows starts a new epoch
    }
  }
  RH_MARK(3);
}

// Grid (image, block), blocks taken from the last one down, for every image at once.  (A
// one-dimensional grid that also puts the B workgroups reading the same 512 anchors on one XCD,
// to share its L2, was measured: 12.0 against 11.5 us.)
// (at most six waves per SIMD: 11.4 against 11.7 us with eight)
template <bool kSplit>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 6))) void assign_kernel(
    const AssignArgs aa) {
  assign_block<kSplit>(aa, blockIdx.x, (int)gridDim.y - 1 - (int)blockIdx.y, gridDim.y);
}
#ifdef RH_TIMELINE
extern "C" int rh_dbg_read(long long* out, size_t count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rh_dbg), count * sizeof(long long));
}
#endif

// Sum of an image's per-block positive counts; all threads of the block call it and get the
// same value.  s_tmp: kThreads / 64 ints of LDS.
__device__ __forceinline__ int sum_blk_pos(const int* __restrict__ blk_pos, const int nblk,
                                           const int j, int* s_tmp) {
  int v = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) v += blk_pos[(size_t)j * nblk + i];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_tmp[w];
  return t;
}

__global__ __launch_bounds__(kThreads) void sum_blk_pos_kernel(const int* __restrict__ blk_pos,
                                                               int nblk,
                                                               int* __restrict__ num_pos) {
  __shared__ int s_tmp[kThreads / 64];
  const int t = sum_blk_pos(blk_pos, nblk, blockIdx.x, s_tmp);
  if (threadIdx.x == 0) num_pos[blockIdx.x] = t;
}

// Batch-wide normaliser: per-image positive counts and their total as a float (exact below
// 2^24 positives).  One workgroup; image j is summed by wave j % 4, lane-strided, so the result
// does not depend on timing.
__global__ __launch_bounds__(kThreads) void count_positives_kernel(const int* __restrict__ blk_pos,
                                                                   int nblk, int B,
                                                                   int* __restrict__ num_pos,
                                                                   float* __restrict__ total) {
  __shared__ int s_tot[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int acc = 0;
  for (int j = wave; j < B; j += kThreads / 64) {
    int v = 0;
    for (int i = lane; i < nblk; i += 64) v += blk_pos[(size_t)j * nblk + i];
    v = wave_sum(v);
    if (lane == 0 && num_pos) num_pos[j] = v;
    acc += v;
  }
  if (lane == 0) s_tot[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_tot[w];
    *total = (float)t;
  }
}

// Full IoU matrix (losses.py:6-23), one thread per output element.
__global__ __launch_bounds__(kThreads) void calc_iou_kernel(const float4* __restrict__ a,
                                                            const float4* __restrict__ b,
                                                            float* __restrict__ out, long long n,
                                                            int m) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n * m) return;
  const long long r = i / m;
  const int c = (int)(i - r * m);
  const float4 x = a[r], y = b[c];
  out[i] = iou_pair(x, (x.z - x.x) * (x.w - x.y), y, (y.z - y.x) * (y.w - y.y));
}

// ---------------------------------------------------------------------------
// Element maths shared by the two stream kernels.
// ---------------------------------------------------------------------------
struct FocalConst {
  float alpha;       // weight of target-1 elements; 1-alpha for target-0 (losses.py:81-83)
  float lo, hi;      // clamp bounds (losses.py:57)
};

constexpr float kLn2 = 0.693147180559945f;

// The stream is VALU-sensitive (one log and one divide per element), so the element
// maths uses the hardware's 1-ulp log2 and reciprocal (v_log_f32, v_rcp_f32) with the
// constant factors folded: a handful of ulps (< 1e-6 relative), against a 1e-5 bar.
// Arguments stay in [1e-4, 1 - 1e-4]: no denormal or range handling is needed.
//
//   target 1 (u = 1-p, w = p):    L = a u^2 (-ln w)       dL/dp =  a [2u ln w - u^2 / w]
//   target 0 (u = p, w = 1-p):    L = (1-a) u^2 (-ln w)   dL/dp = -(1-a) [2u ln w - u^2 / w]
// Outside [lo, hi] the clamp blocks the gradient (inclusive bounds pass it): the clamp
// changes p exactly when p is outside, so "pc == p" is the pass condition.
//
// kl = -a ln2 and kg = +-a * scale are formed by the caller (per target kind).
__device__ __forceinline__ float focal_core(const float p, const float pc, const float u,
                                            const float w, const float kl, const float kg,
                                            float& grad) {
  const float lg = __builtin_amdgcn_logf(w);          // log2(w)
  const float uu = u * u;
  const float t = uu * __builtin_amdgcn_rcpf(w);
  const float g = kg * __builtin_fmaf(u, lg * (2.0f * kLn2), -t);
  grad = (pc == p) ? g : 0.0f;
  return uu * (lg * kl);
}

// Target-0 element (the overwhelmingly common case).  k0l = -(1-a) ln2, k0g = -(1-a)*scale.
__device__ __forceinline__ float focal_zero(const float p, const FocalConst k, const float k0l,
                                            const float k0g, float& grad) {
  const float pc = __builtin_amdgcn_fmed3f(p, k.lo, k.hi);
  return focal_core(p, pc, pc, 1.0f - pc, k0l, k0g, grad);
}

// Either target.  scale multiplies the gradient.
__device__ __forceinline__ float focal_any(const float p, const bool one, const FocalConst k,
                                           const float scale, float& grad) {
  const float pc = __builtin_amdgcn_fmed3f(p, k.lo, k.hi);
  const float q = 1.0f - pc;
  const float a = one ? k.alpha : 1.0f - k.alpha;
  return focal_core(p, pc, one ? q : pc, one ? pc : q, -a * kLn2, (one ? a : -a) * scale, grad);
}

// Logits-in mode: the head's nn.Sigmoid (model.py:177, 193) is folded into the stream.  The
// element is a logit x; p = sigmoid_f32(x), the loss is the probability-mode loss of p and
// the gradient flows through the sigmoid as autograd does it: dL/dx = dL/dp * p * (1 - p).
// The stream has room for about a dozen more vector instructions per element before it
// stops being bound by HBM, which is what the hardware exp2 / reciprocal form of sigmoid_f32
// costs; an expf call and an IEEE division took the kernel from 122 to 164 us.
template <bool kLogits>
__device__ __forceinline__ float focal_zero_in(const float x, const FocalConst k, const float k0l,
                                               const float k0g, float& grad) {
  if (!kLogits) return focal_zero(x, k, k0l, k0g, grad);
  const float p = sigmoid_f32(x);
  float gp;
  const float l = focal_zero(p, k, k0l, k0g, gp);
  grad = gp * ((1.0f - p) * p);
  return l;
}
template <bool kLogits>
__device__ __forceinline__ float focal_any_in(const float x, const bool one, const FocalConst k,
                                              const float scale, float& grad) {
  if (!kLogits) return focal_any(x, one, k, scale, grad);
  const float p = sigmoid_f32(x);
  float gp;
  const float l = focal_any(p, one, k, scale, gp);
  grad = gp * ((1.0f - p) * p);
  return l;
}

// Smooth-L1 of one regression coordinate of a positive anchor (losses.py:101-133).
// k = 0..3 selects dx, dy, dw, dh.  Returns the loss; *grad = d loss / d regression.
__device__ __forceinline__ float smooth_l1_coord(const int k, const float4 a, const GtRec r,
                                                 const float rv, float* grad) {
  const float aw = a.z - a.x, ah = a.w - a.y;
  const float gw = r.x2 - r.x1, gh = r.y2 - r.y1;
  float t;
  if (k == 0) {
    t = ((r.x1 + 0.5f * gw) - (a.x + 0.5f * aw)) / aw;
  } else if (k == 1) {
    t = ((r.y1 + 0.5f * gh) - (a.y + 0.5f * ah)) / ah;
  } else if (k == 2) {
    t = logf(fmaxf(gw, 1.0f) / aw);
  } else {
    t = logf(fmaxf(gh, 1.0f) / ah);
  }
  t = t / (k < 2 ? 0.1f : 0.2f);
  const float diff = t - rv;
  const float d = fabsf(diff);
  const bool small = d <= (float)(1.0 / 9.0);
  const float gmag = small ? 9.0f * d : 1.0f;
  // d/d rv of |t - rv| is -sign(t - rv); sign(0) = 0
  *grad = diff > 0.0f ? -gmag : (diff < 0.0f ? gmag : 0.0f);
  return small ? 4.5f * (d * d) : d - (float)(0.5 / 9.0);
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---------------------------------------------------------------------------
// Final reduction: fixed order, double accumulation.  One WAVE sums the partial records of one
// image (lane-strided, then a butterfly), applies the normalisers of losses.py:94 and :134 and
// writes the image's outputs; the batch means of losses.py:139-141 are then formed from the
// images in ascending order.  Two callers share this code, so their results are the same bits:
// finalize_kernel (one workgroup, wave w takes images w, w+16, ...) and the last workgroup of
// the tile stream kernel to finish an image (below).
// tile_mode > 0: partials are the tile kernel's records, tile_mode x 32 bytes per tile (four
// (value, tag) atoms, below); 0: the row / level kernels' [B][blocks][2].
// kCoherent: the records are being written during the SAME launch by other workgroups, as
// device-scope atomic stores; they are read as device-scope atomic loads, again and again until
// each carries the call's tag.
// ---------------------------------------------------------------------------
constexpr int kFinalBatch = 8;
#ifndef RH_FINAL_KERNEL_BATCH
#define RH_FINAL_KERNEL_BATCH 12
#endif
constexpr int kFinalKernelBatch = RH_FINAL_KERNEL_BATCH;   // finalize_kernel has registers for more loads in flight
constexpr int kFinishSpins = 1 << 21;
constexpr int kMaxFinishers = 128;      // finishing workgroups inside a stream launch   // about two seconds of polling before giving up
struct ImageLoss {
  float cl, rl;
  int pos;
};
struct FinalIn {
  const float* partials;
  const int* blk_pos;
  int nblk;
  const int* gt_count;
  float* per_image;
  int* num_pos_out;
  int B, N, tile_mode, generic_blocks;
  const float* pos_total;
  unsigned tag;    // kCoherent readers: the tag this call's records carry
  int* status;     // [B] per-image status words (kCoherent readers report a reduction that gave up)
};

// A tile record is four 8-byte atoms, (value, tag): classification and regression sums of the
// tile's first image, then of its second.  An atom is written by one aligned 64-bit store, so a
// reader sees it whole.
__device__ __forceinline__ unsigned long long make_atom(const float v, const unsigned tag) {
  return (unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32);
}
__device__ __forceinline__ unsigned long long load_atom_coherent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_atom_coherent(unsigned long long* p, const unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The image's sums are defined over kFinalShares interleaved shares of its records: share s holds
// the records base + 64 s + lane + 256 i (i = 0, 1, ..), each lane adds its records in ascending
// order in double, a butterfly adds the 64 lanes, and the shares are added as (S0 + S1) + (S2 + S3).
// One wave can work through all shares (kWaves = 1: consecutive loads of a lane belong to
// consecutive shares) or four waves one share each (kWaves = 4, wave w share w, combined through
// s_fin): the same additions in the same order, so the same bits.
// kBatch (a multiple of kFinalShares): records requested together per lane; 16 for the polling
// reader would not fit the stream kernel's 96 registers (spills, and a slower kernel for everyone).
constexpr int kFinalShares = 4;
template <bool kCoherent, int kBatch, int kWaves>
__device__ __forceinline__ ImageLoss finalize_image(const FinalIn& f, const int j, double* s_fin) {
  static_assert(kBatch % kFinalShares == 0 && (kWaves == 1 || kWaves == kFinalShares), "shares");
  const int lane = threadIdx.x & 63;
  const int wave = kWaves == 1 ? 0 : (int)(threadIdx.x >> 6);
  const int nblk = f.nblk, N = f.N, tile_mode = f.tile_mode;
  constexpr int kAcc = kWaves == 1 ? kFinalShares : 1;
  double c[kAcc], r[kAcc];
#pragma unroll
  for (int a = 0; a < kAcc; ++a) c[a] = r[a] = 0.0;
  bool timed_out = false;
  // the small loads go out with the first batch of records, not after the last
  int npv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) npv[k] = f.blk_pos[(size_t)j * nblk + min(lane + k * 64, nblk - 1)];
  const int gtc = f.gt_count[j];
  const float total = f.pos_total ? *f.pos_total : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  if (tile_mode) {
    const long long first = (long long)j * N, last = first + N - 1;
    const int t_lo = (int)(first / kTileRows), t_hi = (int)(last / kTileRows);
    // only the image's first tile can also hold rows of the previous image; there this
    // image is the tile's second one
    const bool shared_first = (first % kTileRows) != 0;
    const int u_shared = (t_lo + 1) * tile_mode;      // records below it belong to tile t_lo
    // (kBatch loads in flight per lane, only the half of each record that belongs to
    // this image, and the small loads above issued with them.)
    const int u_end = (t_hi + 1) * tile_mode;
    const unsigned long long* atoms = reinterpret_cast<const unsigned long long*>(f.partials);
#pragma nounroll
    for (int u0 = t_lo * tile_mode + wave * 64 + lane; u0 < u_end; u0 += 64 * kWaves * kBatch) {
      unsigned long long qc[kBatch], qr[kBatch];   // this image's two atoms of each record
      for (int spin = 0;; ++spin) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const int u = min(u0 + k * 64 * kWaves, u_end - 1);
          const size_t at = 4 * (size_t)u + 2 * (shared_first && u < u_shared);
          if (kCoherent) {
            qc[k] = load_atom_coherent(atoms + at);
            qr[k] = load_atom_coherent(atoms + at + 1);
          } else {
            qc[k] = atoms[at];
            qr[k] = atoms[at + 1];
          }
        }
        // every load of the round is requested before the first one is used (left to itself the
        // scheduler pairs each load with its additions to save registers: a memory latency per
        // pair instead of one per round)
        __builtin_amdgcn_sched_barrier(0);
        if (!kCoherent) {
          // (the tag halves are not read here; were they dead, the register allocator would hand
          // them to the next load's address arithmetic, which then waits for the load in flight
          // into them: a memory latency per pair of loads.  The empty statement keeps them alive.)
#pragma unroll
          for (int k = 0; k < kBatch; ++k) asm volatile("" ::"v"(qc[k]), "v"(qr[k]));
          break;
        }
        // written during this launch: wait until every atom carries the call's tag
        bool all_in = true;
#pragma unroll
        for (int k = 0; k < kBatch; ++k)      // (a load past the end repeats the image's last record,
          all_in &= u0 + k * 64 * kWaves >= u_end ||      // which its owner may have consumed by now)
                    ((unsigned)(qc[k] >> 32) == f.tag && (unsigned)(qr[k] >> 32) == f.tag);
        if (__ballot(!all_in) == 0ull) break;
        if (spin >= kFinishSpins) {          // (a logic error: report it, do not hang)
          timed_out = true;
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if (u0 + k * 64 * kWaves < u_end) {
          c[kWaves == 1 ? k % kFinalShares : 0] += __uint_as_float((unsigned)qc[k]);
          r[kWaves == 1 ? k % kFinalShares : 0] += __uint_as_float((unsigned)qr[k]);
          if (kCoherent) {      // consumed: tag 0 again, which no call expects
            const int u = u0 + k * 64 * kWaves;
            const size_t at = 4 * (size_t)u + 2 * (shared_first && u < u_shared);
            unsigned long long* w = const_cast<unsigned long long*>(atoms);
            store_atom_coherent(w + at, qc[k] & 0xFFFFFFFFull);
            store_atom_coherent(w + at + 1, qr[k] & 0xFFFFFFFFull);
          }
        }
      }
    }
  } else {
    const float2* pj = reinterpret_cast<const float2*>(f.partials) + (size_t)j * f.generic_blocks;
#pragma nounroll
    for (int t0 = wave * 64 + lane; t0 < f.generic_blocks; t0 += 64 * kWaves * kBatch) {
      float2 q[kBatch];        // (the loads of a round go out together instead of one latency each)
#pragma unroll
      for (int k = 0; k < kBatch; ++k) q[k] = pj[min(t0 + k * 64 * kWaves, f.generic_blocks - 1)];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if (t0 + k * 64 * kWaves < f.generic_blocks) {
          c[kWaves == 1 ? k % kFinalShares : 0] += q[k].x;
          r[kWaves == 1 ? k % kFinalShares : 0] += q[k].y;
        }
      }
    }
  }
  int np = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) np += lane + k * 64 < nblk ? npv[k] : 0;
  for (int i = lane + 256; i < nblk; i += 64) np += f.blk_pos[(size_t)j * nblk + i];
  const bool empty = gtc == 0;
  np = wave_sum(np);
  double ct, rt;
  if (kWaves == 1) {
    double both[2 * kAcc];
#pragma unroll
    for (int a = 0; a < kAcc; ++a) {
      both[a] = c[a];
      both[kAcc + a] = r[a];
    }
    wave_sum_many(both);
#pragma unroll
    for (int a = 0; a < kAcc; ++a) {
      c[a] = both[a];
      r[a] = both[kAcc + a];
    }
    ct = (c[0] + c[kAcc > 1 ? 1 : 0]) + (c[kAcc > 2 ? 2 : 0] + c[kAcc > 3 ? 3 : 0]);
    rt = (r[0] + r[kAcc > 1 ? 1 : 0]) + (r[kAcc > 2 ? 2 : 0] + r[kAcc > 3 ? 3 : 0]);
  } else {
    // one share per wave: combined through LDS (s_fin: 2 * kFinalShares + 1 doubles)
    double both[2] = {c[0], r[0]};
    wave_sum_many(both);
    const double cw = both[0], rw = both[1];
    const bool any_out = __ballot(timed_out) != 0ull;
    __syncthreads();
    if (lane == 0) {
      s_fin[wave] = cw;
      s_fin[kFinalShares + wave] = rw;
    }
    if (threadIdx.x == 0) s_fin[2 * kFinalShares] = 0.0;
    __syncthreads();
    if (any_out && lane == 0) s_fin[2 * kFinalShares] = 1.0;
    __syncthreads();
    ct = (s_fin[0] + s_fin[1]) + (s_fin[2] + s_fin[3]);
    rt = (s_fin[kFinalShares] + s_fin[kFinalShares + 1]) + (s_fin[kFinalShares + 2] + s_fin[kFinalShares + 3]);
    timed_out = s_fin[2 * kFinalShares] != 0.0;
  }
  // batch-wide normaliser: every image is divided by the one total (an image's share of
  // the batch loss); otherwise by its own count, losses.py:94 and :134
  const float npf = f.pos_total ? total : (float)np;
  ImageLoss out;
  out.cl = empty ? 0.f : (float)ct / fmaxf(npf, 1.f);
  out.rl = (empty || np == 0) ? 0.f : (float)rt / (4.0f * npf);
  out.pos = np;
  if (timed_out) out.cl = out.rl = __builtin_nanf("");
  if (lane == 0 && wave == 0) {
    if (timed_out && kCoherent && f.status) atomicOr(f.status + j, (int)RH_STATUS_REDUCTION_GAVE_UP);
    if (f.per_image) {
      f.per_image[j] = out.cl;
      f.per_image[f.B + j] = out.rl;
    }
    if (f.num_pos_out) f.num_pos_out[j] = np;
  }
  return out;
}

// One thread: the call's five output floats from the batch sums.
__device__ __forceinline__ void store_batch_losses(float* losses, const double bc, const double br,
                                                   const long long bp, const int B,
                                                   const float divisor) {
  const float lc = (float)(bc / (double)divisor), lr = (float)(br / (double)divisor);
  losses[0] = lc;
  losses[1] = lr;
  // what data-parallel ranks exchange beside the two losses, ready in the same 16 bytes
  losses[2] = (float)bp;
  losses[3] = (float)B;
  losses[4] = lc + lr;   // what train.py:185 forms from the two; saves the caller two launches
}

constexpr int kFinalThreads = 1024;

__global__ __launch_bounds__(kFinalThreads) void finalize_kernel(const FinalIn f, float* __restrict__ losses,
                                                                float divisor) {
  __shared__ float s_img[2][kFinalThreads / 64];
  __shared__ int s_pos[kFinalThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double bc = 0.0, br = 0.0;     // thread 0 only: batch sums, images visited in ascending order
  long long bp = 0;              // positives of the whole call
  for (int j0 = 0; j0 < f.B; j0 += kFinalThreads / 64) {
    const int j = j0 + wave;
    ImageLoss il{0.f, 0.f, 0};
    if (j < f.B) il = finalize_image<false, kFinalKernelBatch, 1>(f, j, nullptr);
    __syncthreads();
    if (lane == 0) {
      s_img[0][wave] = il.cl;
      s_img[1][wave] = il.rl;
      s_pos[wave] = il.pos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 0; w < kFinalThreads / 64; ++w) {
        bc += s_img[0][w];
        br += s_img[1][w];
        bp += s_pos[w];
      }
    }
  }
  if (threadIdx.x == 0) store_batch_losses(losses, bc, br, bp, f.B, divisor);
}

// The same inside the tile stream kernel, when the call runs both stages: one finishing
// workgroup per image sits at the end of the stream's part of the grid, polls the image's records
// until they are all in, and reduces them; the one that completes the call's last image forms the
// batch means.  No launch, kernel boundary or trip through memory is left after the stream, and
// with the next step's assignment riding behind, the last image's reduction hides under it.
// Writers never wait: a record atom is (value, tag) in one aligned 64-bit device-scope store,
// the tag being the workspace's EPOCH, an odd number that the assignment stage and the end of
// every such call advance (mixed with the clock), so records of earlier calls -- or whatever a
// fresh workspace holds -- do not look finished; a stream stage run on its own writes tag 0, and
// a finisher puts tag 0 back on every atom it has consumed.  Ordering a ticket after
// the record instead (store, wait for the acknowledgement, atomic add) was measured: the wait
// also covers the wave's streamed gradient stores and doubled the kernel's time; a release
// fence, whose L2 write-back sits behind 370 MB of such stores, took it from 0.12 to 0.95 ms.
struct StreamFinal {
  int enabled;
  unsigned* epoch;   // [1] the workspace's epoch
  float* img_out;    // [B] x three atoms: classification loss, regression loss, positives
  float* losses;
  float divisor;
  FinalIn in;
};

// The finishing workgroup of image j (all its waves, a share of the records each): reduces the
// image as its records arrive and publishes the image's results the same way, as three tagged
// atoms; wave 0 of the finisher of image 0 then collects all images (polling, like the records)
// and forms the batch means.
__device__ __forceinline__ void stream_finish_image(const StreamFinal& fin, const int j) {
  __shared__ double s_fin[2 * kFinalShares + 1];
  const int lane = threadIdx.x & 63;
  FinalIn in = fin.in;
  const unsigned tag = *fin.epoch;
  in.tag = tag;
  const ImageLoss il = finalize_image<true, kFinalBatch, kFinalShares>(in, j, s_fin);
  if (threadIdx.x >= 64) return;
  unsigned long long* out = reinterpret_cast<unsigned long long*>(fin.img_out);
  if (lane == 0) {
    store_atom_coherent(out + 3 * (size_t)j, make_atom(il.cl, tag));
    store_atom_coherent(out + 3 * (size_t)j + 1, make_atom(il.rl, tag));
    store_atom_coherent(out + 3 * (size_t)j + 2, make_atom(__int_as_float(il.pos), tag));
  }
  if (j != 0) return;
  // the batch means, images in ascending order (as finalize_kernel sums them)
  double bc = 0.0, br = 0.0;
  long long bp = 0;
  bool timed_out = false;
  for (int i0 = 0; i0 < in.B; i0 += 64) {
    const int i = min(i0 + lane, in.B - 1);
    unsigned long long a0, a1, a2;
    for (int spin = 0;; ++spin) {
      a0 = load_atom_coherent(out + 3 * (size_t)i);
      a1 = load_atom_coherent(out + 3 * (size_t)i + 1);
      a2 = load_atom_coherent(out + 3 * (size_t)i + 2);
      const bool in_ = (unsigned)(a0 >> 32) == tag && (unsigned)(a1 >> 32) == tag &&
                       (unsigned)(a2 >> 32) == tag;
      if (__ballot(!in_) == 0ull) break;
      if (spin >= kFinishSpins) {
        timed_out = true;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    if (i0 + lane < in.B) {       // consumed
      store_atom_coherent(out + 3 * (size_t)i, a0 & 0xFFFFFFFFull);
      store_atom_coherent(out + 3 * (size_t)i + 1, a1 & 0xFFFFFFFFull);
      store_atom_coherent(out + 3 * (size_t)i + 2, a2 & 0xFFFFFFFFull);
    }
    const int cnt = min(64, in.B - i0);
    for (int w = 0; w < cnt; ++w) {
      bc += (double)__uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)a0, w));
      br += (double)__uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)a1, w));
      bp += __builtin_amdgcn_readlane((int)(unsigned)a2, w);
    }
  }
  if (lane == 0) {
    *fin.epoch = next_epoch(tag);        // this call's records are history now
    if (timed_out) bc = br = (double)__builtin_nanf("");
    store_batch_losses(fin.losses, bc, br, bp, in.B, fin.divisor);
  }
}

// ---------------------------------------------------------------------------
// Tile stream kernel.  The [B*N, C] probability matrix is cut into tiles of 64
// consecutive rows; a tile starts at a multiple of 64*C floats, so it is 16-byte
// aligned for every C and is read / written as float4 regardless of row length.
// Each float4's (row, column) comes from one multiply-high.  Per-row assignment
// codes sit in LDS.  A tile touches at most two images (host guarantees N >= 64):
// loss partial sums are kept per (tile, image slot) and written, never atomically
// added, so the final sums are bitwise reproducible.
//
// Requirements checked on the host: 4 <= C < 8192, N >= 64, all pointers 16-byte
// aligned, B*N < 2^31.
// ---------------------------------------------------------------------------
// (Double-buffering the loads of wide tiles inside one workgroup was measured at C = 601 and
// changed nothing, 886 against 887 us; cutting them into parts, tile_parts() below, did.)
// (Issuing the first batch of data loads ahead of the code and count loads, which helped the
// head-layout kernel, gains under 1 % here and costs a spill: not done.)
// Five waves per SIMD (at most 96 vector registers; the compiler's own choice was 99, i.e. four
// waves): measured 120.0 against 121.5 us.  Six waves need spills and take 156 us.
//
// kAhead != 0: the launch also carries the IoU assignment of the NEXT step (its annotations are known
// before its predictions are).  Its workgroups are the last of the grid, image-major from the
// long blocks down like the stand-alone kernel's: they take the workgroup slots that the stream
// leaves empty as its last tiles drain, and end about 6 us after it, against 11 us plus a kernel
// boundary for a launch of their own.  (Measured at B=16, N=69354, C=80, step time with the
// assignment placed: in its own launch 142.5 us; at the front of the grid, where its 2176
// workgroups fill every slot of the chip and the stream starts only as they finish, 140.6 us;
// one in every six workgroups over the first 70 % of the grid 146.5 us -- the stream is bound by
// the loads its resident workgroups keep in flight, so every slot lent out costs its share; at
// the end of the grid 137.2 us.)
struct AheadArgs {
  AssignArgs aa;
  int batch;
  int nblk;
};

// kAhead: 0 no assignment rides along, 1 the plain form, 2 the split form (long annotation lists)
template <int U, bool kLogits, int kAhead>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5))) void focal_stream_kernel(
    const float* __restrict__ prob, float* __restrict__ dprob, const float* __restrict__ reg,
    float* __restrict__ dreg, const float4* __restrict__ anchors, const GtRec* __restrict__ gt,
    const int* __restrict__ code, const int* __restrict__ blk_pos, int nblk,
    float* __restrict__ partials, int N, int C, int M, int total_rows, int num_tiles,
    unsigned magic_c, FocalConst fc, float gcls, float greg, float divisor,
    const float* __restrict__ pos_total, int parts, int part_vecs, const AheadArgs ahead,
    const StreamFinal fin) {
  // Grid: the stream's workgroups, then (both stages in one call) one finishing workgroup per
  // image, then (kAhead) the next step's assignment.
  const int units = num_tiles * parts;
  if ((int)blockIdx.x >= units) {
    const int extra = (int)blockIdx.x - units;
    const int finishers = fin.enabled ? fin.in.B : 0;
    if (extra < finishers) {
      stream_finish_image(fin, extra);
      return;
    }
    if (kAhead) {
      const int at = extra - finishers;
      const int b = at / ahead.batch;
      assign_block<kAhead == 2>(ahead.aa, at - b * ahead.batch, ahead.nblk - 1 - b, ahead.nblk);
    }
    return;
  }
  __shared__ int s_code[kTileRows];
  __shared__ float4 s_red[kThreads / 64];
  __shared__ int s_np[kThreads / 64][2];
  __shared__ int s_special[2];
  const int tid = threadIdx.x;
  if (tid == 0) s_special[0] = s_special[1] = 0;
  int turn = 0;
  const float k0l = -(1.0f - fc.alpha) * kLn2;
  // unroll of the general (row-tracking) path: keeps registers low (3: no change; 5, one float4
  // at a time behind scheduling barriers: more than 96 registers, 131 against 119 us)
  constexpr int UG = 2;

  // A workgroup takes one part of one tile: with many classes a tile's classification block is
  // cut into `parts` runs of part_vecs float4 (a multiple of the block size), so that every
  // workgroup streams about as much as it does at C = 80 -- short-lived workgroups whose loads
  // all go out at once, and enough of them to fill the chip evenly.  Part 0 also does the
  // tile's regression rows.
  for (int unit = blockIdx.x; unit < units; unit += units) {
    const int tile = parts == 1 ? unit : unit / parts;
    const int part = unit - tile * parts;
    const int row0 = tile * kTileRows;
    const int rows = min(kTileRows, total_rows - row0);
    const int j0 = row0 / N;
    const int split = (j0 + 1) * N - row0;            // first tile-local row of image j0+1
    const bool two = split < rows;
    // the tile's codes and the positive counts go out together: one memory latency, not two
    const int my_code_raw = code[min(row0 + min(tid, kTileRows - 1), total_rows - 1)];
    // positives of the tile's image(s): sum of the assignment kernel's per-block counts
    // (with a batch-wide normaliser every image uses the one total instead)
    const float total = pos_total ? *pos_total : 0.f;
    int v0 = 0, v1 = 0;
    if (!pos_total)
      for (int i = tid; i < nblk; i += kThreads) {
        v0 += blk_pos[(size_t)j0 * nblk + i];
        if (two) v1 += blk_pos[(size_t)(j0 + 1) * nblk + i];
      }
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);

    __syncthreads();   // previous tile's readers of s_code / s_red / s_np are done
    if ((tid & 63) == 0) {
      s_np[tid >> 6][0] = v0;
      s_np[tid >> 6][1] = v1;
    }
    const int my_code = tid < rows ? my_code_raw : kCodeNeg;
    if (tid < kTileRows) s_code[tid] = tid < rows ? my_code : kCodeIgn;
    // any positive or ignored row in this tile?  An LDS flag and the barrier that publishes
    // s_code and s_np anyway (the library's workgroup OR costs a reduction through LDS with
    // barriers of its own); two flags take turns so that the next tile can clear its own.
    if (__ballot(my_code != kCodeNeg) != 0ull && (tid & 63) == 0) s_special[turn] = 1;
    __syncthreads();
    const int special = s_special[turn];
    turn ^= 1;
    if (tid == 0) s_special[turn] = 0;
    const float np0 = pos_total ? total : (float)((s_np[0][0] + s_np[1][0]) + (s_np[2][0] + s_np[3][0]));
    const float np1 = pos_total ? total : (float)((s_np[0][1] + s_np[1][1]) + (s_np[2][1] + s_np[3][1]));
    const float norm0 = fmaxf(np0, 1.f), norm1 = fmaxf(np1, 1.f);
    const float sc0 = gcls / (norm0 * divisor), sc1 = gcls / (norm1 * divisor);

    float cl0 = 0.f, cl1 = 0.f, rl0 = 0.f, rl1 = 0.f;

    // ---- regression: 4 lanes per anchor, one coordinate each -----------------
    if (part == 0) {
      const int r = tid >> 2, k = tid & 3;
      if (r < rows) {
        const int grow = row0 + r;
        float g = 0.f;
        if (special) {
          const int cd = s_code[r];
          if (cd >= 0) {
            const bool second = r >= split;
            const int j = second ? j0 + 1 : j0;
            const float cnt4 = 4.0f * (second ? np1 : np0);
            const float rv = reg[(size_t)grow * 4 + k];
            float gr;
            const float l = smooth_l1_coord(k, anchors[grow - j * N],
                                            gt[(size_t)j * M + (cd >> 16)], rv, &gr);
            g = gr * (greg / (cnt4 * divisor));
            if (second) rl1 = l; else rl0 = l;
          }
        }
        __builtin_nontemporal_store(g, &dreg[(size_t)grow * 4 + k]);
      }
    }

    // ---- classification stream ----------------------------------------------
    const size_t base = (size_t)row0 * C;
    const v4f* __restrict__ p4 = reinterpret_cast<const v4f*>(prob + base);
    v4f* __restrict__ d4 = reinterpret_cast<v4f*>(dprob + base);
    const int elems = rows * C;
    const int nvec = elems >> 2;
    const int v_begin = part * part_vecs;
    const int v_end = min(nvec, v_begin + part_vecs);

    if (!special && !two) {
      // Every row is a negative of one image: pure elementwise stream, no index maths.
      const float k0g = -(1.0f - fc.alpha) * sc0;
      for (int v0 = v_begin + tid; v0 < v_end; v0 += kThreads * U) {
        v4f val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = v0 + u * kThreads;
          if (v < v_end) val[u] = __builtin_nontemporal_load(&p4[v]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = v0 + u * kThreads;
          if (v < v_end) {
            float g0, g1, g2, g3;
            cl0 += focal_zero_in<kLogits>(val[u].x, fc, k0l, k0g, g0);
            cl0 += focal_zero_in<kLogits>(val[u].y, fc, k0l, k0g, g1);
            cl0 += focal_zero_in<kLogits>(val[u].z, fc, k0l, k0g, g2);
            cl0 += focal_zero_in<kLogits>(val[u].w, fc, k0l, k0g, g3);
            v4f o;
            o.x = g0; o.y = g1; o.z = g2; o.w = g3;
            __builtin_nontemporal_store(o, &d4[v]);
          }
        }
      }
    } else {
      for (int v0 = v_begin + tid; v0 < v_end; v0 += kThreads * UG) {
        v4f val[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          const int v = v0 + u * kThreads;
          if (v < v_end) val[u] = __builtin_nontemporal_load(&p4[v]);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          const int v = v0 + u * kThreads;
          if (v < v_end) {
            const unsigned x = (unsigned)v << 2;
            int r = (int)fast_div(x, magic_c);
            int col = (int)x - r * C;
            const float in[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
            float out[4];
            int cd = s_code[r];
            // Most float4 of a tile with a few positive or ignored rows still lie in plain
            // negative rows of one image: those take the cheap element path as well.
            const int r_end = col + 3 >= C ? r + 1 : r;   // C >= 4: a float4 spans at most two rows
            const bool second4 = r >= split;
            if (cd == kCodeNeg && s_code[r_end] == kCodeNeg && (r_end >= split) == second4) {
              const float k0g = -(1.0f - fc.alpha) * (second4 ? sc1 : sc0);
              float l = focal_zero_in<kLogits>(in[0], fc, k0l, k0g, out[0]);
              l += focal_zero_in<kLogits>(in[1], fc, k0l, k0g, out[1]);
              l += focal_zero_in<kLogits>(in[2], fc, k0l, k0g, out[2]);
              l += focal_zero_in<kLogits>(in[3], fc, k0l, k0g, out[3]);
              cl0 += second4 ? 0.f : l;
              cl1 += second4 ? l : 0.f;
            } else
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (col >= C) {            // C >= 4: at most one row change per float4
                col -= C;
                r += 1;
                cd = s_code[r];
              }
              const bool one = cd >= 0 && (cd & 0xFFFF) == col;
              const bool second = r >= split;
              const bool live = cd != kCodeIgn;
              float g;
              float l = focal_any_in<kLogits>(in[e], one, fc, second ? sc1 : sc0, g);
              l = live ? l : 0.f;
              out[e] = live ? g : 0.f;
              cl0 += second ? 0.f : l;
              cl1 += second ? l : 0.f;
              col += 1;
            }
            v4f o;
            o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
            __builtin_nontemporal_store(o, &d4[v]);
          }
        }
      }
    }
    // last tile only: up to 3 trailing elements when B*N*C is not a multiple of 4
    if (part == parts - 1) {
      const int x = (nvec << 2) + tid;
      if (x < elems) {
        const int r = x / C, col = x - r * C;
        const int cd = s_code[r];
        const bool one = cd >= 0 && (cd & 0xFFFF) == col;
        const bool second = r >= split;
        const bool live = cd != kCodeIgn;
        float g;
        float l = focal_any_in<kLogits>(prob[base + x], one, fc, second ? sc1 : sc0, g);
        dprob[base + x] = live ? g : 0.f;
        l = live ? l : 0.f;
        cl0 += second ? 0.f : l;
        cl1 += second ? l : 0.f;
      }
    }

    // ---- per-tile partial sums (fixed reduction tree) ------------------------
    cl0 = wave_sum(cl0);
    cl1 = wave_sum(cl1);
    rl0 = wave_sum(rl0);
    rl1 = wave_sum(rl1);
    if ((tid & 63) == 0) s_red[tid >> 6] = make_float4(cl0, cl1, rl0, rl1);
    __syncthreads();
    if (tid == 0) {
      const float4 a = s_red[0], b = s_red[1], c = s_red[2], d = s_red[3];
      // record: (classification, regression) of the tile's first image, then of its second
      const float4 rec = make_float4((a.x + b.x) + (c.x + d.x), (a.z + b.z) + (c.z + d.z),
                                     (a.y + b.y) + (c.y + d.y), (a.w + b.w) + (c.w + d.w));
      // four (value, tag) atoms; tag 0 (never an epoch) when the reduction is a later launch
      unsigned long long* atoms = reinterpret_cast<unsigned long long*>(partials) + 4 * (size_t)unit;
      if (!fin.enabled) {
        atoms[0] = make_atom(rec.x, 0u);
        atoms[1] = make_atom(rec.y, 0u);
        atoms[2] = make_atom(rec.z, 0u);
        atoms[3] = make_atom(rec.w, 0u);
      } else {
        const unsigned tag = *fin.epoch;
        store_atom_coherent(atoms + 0, make_atom(rec.x, tag));
        store_atom_coherent(atoms + 1, make_atom(rec.y, tag));
        store_atom_coherent(atoms + 2, make_atom(rec.z, tag));
        store_atom_coherent(atoms + 3, make_atom(rec.w, tag));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Generic row kernel: any N, C and alignment.  Grid (kGenericBlocks, B); one wave
// per anchor row at a time, scalar loads.  Used for shapes the tile kernel does
// not take (C < 4, C >= 8192, N < 64, unaligned tensors).
// partials layout: [B][kGenericBlocks][2] (classification, regression).
// ---------------------------------------------------------------------------
template <bool kLogits>
__global__ __launch_bounds__(kThreads) void focal_rows_kernel(
    const float* __restrict__ prob, float* __restrict__ dprob, const float* __restrict__ reg,
    float* __restrict__ dreg, const float* __restrict__ anchors, const GtRec* __restrict__ gt,
    const int* __restrict__ code, const int* __restrict__ blk_pos, int nblk,
    float* __restrict__ partials, int N, int C, int M, FocalConst fc, float gcls, float greg,
    float divisor, const float* __restrict__ pos_total) {
  __shared__ float s_red[4];
  __shared__ int s_tmp[kThreads / 64];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const float np = pos_total ? *pos_total : (float)sum_blk_pos(blk_pos, nblk, j, s_tmp);
  const float sc = gcls / (fmaxf(np, 1.f) * divisor);
  const float rsc = greg / ((4.0f * np) * divisor);
  float cl = 0.f, rl = 0.f;
  for (int n = blockIdx.x * (kThreads / 64) + wave; n < N; n += gridDim.x * (kThreads / 64)) {
    const size_t row = (size_t)j * N + n;
    const int cd = code[row];
    const bool live = cd != kCodeIgn;
    const float* p = prob + row * C;
    float* d = dprob + row * C;
    for (int c = lane; c < C; c += 64) {
      float g;
      const float l = focal_any_in<kLogits>(p[c], cd >= 0 && (cd & 0xFFFF) == c, fc, sc, g);
      d[c] = live ? g : 0.f;
      cl += live ? l : 0.f;
    }
    if (lane < 4) {
      float g = 0.f;
      if (cd >= 0) {
        const float4 a = make_float4(anchors[(size_t)n * 4], anchors[(size_t)n * 4 + 1],
                                     anchors[(size_t)n * 4 + 2], anchors[(size_t)n * 4 + 3]);
        float gr;
        rl += smooth_l1_coord(lane, a, gt[(size_t)j * M + (cd >> 16)], reg[row * 4 + lane], &gr);
        g = gr * rsc;
      }
      dreg[row * 4 + lane] = g;
    }
  }
  const float t0 = block_sum_256(cl, s_red);
  const float t1 = block_sum_256(rl, s_red);
  if (threadIdx.x == 0) {
    float* o = partials + ((size_t)j * gridDim.x + blockIdx.x) * 2;
    o[0] = t0;
    o[1] = t1;
  }
}

// ---------------------------------------------------------------------------
// Head-layout stream kernel.  The heads produce, per pyramid level, a tensor [B, A*C, H, W]
// (classification) and [B, A*4, H, W] (regression); the reference turns them into the flat
// [B, N, C] / [B, N, 4] tensors by permute + contiguous + view + cat (model.py:152-154,
// 196-202, 311-313), one full copy of the N x C data each way.  This kernel reads the level
// tensors where they lie and writes the gradients in the same layout (what the head
// convolutions' backward consumes).  Anchor index of (level l, cell s = y*W + x, anchor a):
// n = off_l + s*A + a; class c of it is channel a*C + c.
//
// Work unit u = a*HW + cell of a level ("anchor-cell": the order of the code planes the
// assignment kernel writes for this layout, so a unit's code is code[off_l + u]).  A
// workgroup owns 1024 consecutive units of one level of one image (4 per thread) and one
// share of the C class planes (blockIdx.y; share 0 also does the four regression planes); it
// walks its planes U at a time, each a contiguous run of cells: float4 per thread when H*W is
// a multiple of 4 and the tensors are 16-byte aligned (then a thread's four units share their
// anchor), else one unit per access.  Small levels fill workgroups as well as large ones,
// which matters: a workgroup lives for a few memory latencies whatever it carries.
// partials layout: [B][gridDim.x * gridDim.y][2].
// ---------------------------------------------------------------------------
constexpr int kLvlPlanes = 5;          // class planes loaded together
constexpr int kLvlThreads = 256;
constexpr int kLvlChunk = 4 * kLvlThreads;   // units per workgroup: 4 per thread

// Sum over the workgroup (any multiple of 64 threads), fixed order.
__device__ __forceinline__ float block_sum_any(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}

struct LevelTable {
  const float* cls[kMaxLevels];
  float* dcls[kMaxLevels];
  const float* reg[kMaxLevels];
  float* dreg[kMaxLevels];
  int hw[kMaxLevels];            // cells of the level
  int anchor_off[kMaxLevels];    // index of the level's first anchor
  int chunk_begin[kMaxLevels + 1];
  int vec4[kMaxLevels];          // float4 path usable
  int num_levels;
};

template <bool kLogits>
__global__ __launch_bounds__(kLvlThreads) void focal_levels_kernel(
    LevelTable lt, const float4* __restrict__ anchors, const GtRec* __restrict__ gt,
    const int* __restrict__ code, const int* __restrict__ blk_pos, int nblk,
    float* __restrict__ partials, int N, int A, int C, int M, FocalConst fc, float gcls, float greg,
    float divisor, const float* __restrict__ pos_total) {
  __shared__ float s_red[kLvlThreads / 64];
  __shared__ int s_tmp[kLvlThreads / 64];
  const int tid = threadIdx.x;
  const int part = blockIdx.y, csplit = gridDim.y, j = blockIdx.z;
  const int cper = (C + csplit - 1) / csplit;
  const int c_lo = part * cper, c_hi = min(C, c_lo + cper);
  int l = 0;
  while (l + 1 < lt.num_levels && (int)blockIdx.x >= lt.chunk_begin[l + 1]) ++l;
  const int hw = lt.hw[l];
  const int units = A * hw;
  const int s0 = ((int)blockIdx.x - lt.chunk_begin[l]) * kLvlChunk;
  const bool v4 = lt.vec4[l] != 0;

  // my four units: consecutive (float4 path) or strided by the block size (scalar path)
  int cell[4], base_c[4], base_r[4], ucl[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = v4 ? s0 + tid * 4 + k : s0 + tid + k * kLvlThreads;
    ok[k] = u < units;                         // a dead unit stores nothing, and loads from
    // a valid stand-in: the level's last unit, or on the float4 path (units is a multiple of
    // 4 there) the last aligned group of four, so that no access leaves the tensor
    const int uc = ok[k] ? u : (v4 ? units - 4 + k : units - 1);
    ucl[k] = uc;
    const int a = uc / hw;
    cell[k] = uc - a * hw;
    // offsets of the unit inside image j's [A*C, HW] / [A*4, HW] blocks, class / coordinate 0
    base_c[k] = a * C * hw + cell[k];
    base_r[k] = a * 4 * hw + cell[k];
  }
  const float* cp = lt.cls[l] + (size_t)j * A * C * hw;
  float* dcp = lt.dcls[l] + (size_t)j * A * C * hw;
  constexpr int U = kLvlPlanes;
  // The class planes are loaded U at a time, and the first batch goes out before anything else:
  // nothing it needs depends on the codes or the positive count, and a workgroup only lives
  // for a few memory latencies, so putting those loads behind the data loads instead of in
  // front of them takes a whole latency off its life.
  float in[U][4];
  auto load_planes = [&](const int c0) {
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
  };
  load_planes(c_lo);
  __builtin_amdgcn_sched_barrier(0);

  const int* cplane = code + (size_t)j * N + lt.anchor_off[l];
  int cd[4];
  bool special = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cd[k] = cplane[ucl[k]];
    if (!ok[k]) cd[k] = kCodeNeg;
    special |= cd[k] != kCodeNeg;
  }
  const float np = pos_total ? *pos_total : (float)sum_blk_pos(blk_pos, nblk, j, s_tmp);
  const float sc = gcls / (fmaxf(np, 1.f) * divisor);
  const float rsc = greg / ((4.0f * np) * divisor);
  // decide per wave, not per lane: a wave with a single positive or ignored unit would
  // otherwise run both element paths one after the other
  special = __ballot(special) != 0ull;
  float cl = 0.f, rl = 0.f;

  // ---- regression planes
  if (part == 0) {
    const float* rp = lt.reg[l] + (size_t)j * A * 4 * hw;
    float* dp = lt.dreg[l] + (size_t)j * A * 4 * hw;
    float g[4][4];   // [coordinate][unit]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q][k] = 0.f;
      if (cd[k] >= 0) {
        const int a = base_r[k] / (4 * hw);
        const float4 an = anchors[lt.anchor_off[l] + cell[k] * A + a];
        const GtRec r = gt[(size_t)j * M + (cd[k] >> 16)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float gr;
          rl += smooth_l1_coord(q, an, r, rp[(size_t)base_r[k] + (size_t)q * hw], &gr);
          g[q][k] = gr * rsc;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (v4) {
        if (ok[0]) {
          v4f o;
          o.x = g[q][0]; o.y = g[q][1]; o.z = g[q][2]; o.w = g[q][3];
          __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dp + (size_t)base_r[0] + (size_t)q * hw));
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ok[k]) dp[(size_t)base_r[k] + (size_t)q * hw] = g[q][k];
      }
    }
  }

  // ---- classification planes c_lo .. c_hi-1 of every unit
  const float k0l = -(1.0f - fc.alpha) * kLn2, k0g = -(1.0f - fc.alpha) * sc;
  for (int c0 = c_lo; c0 < c_hi; c0 += U) {
    float out[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u;
      if (!special) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lk = focal_zero_in<kLogits>(in[u][k], fc, k0l, k0g, out[u][k]);
          cl += (ok[k] && c < c_hi) ? lk : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool one = cd[k] >= 0 && (cd[k] & 0xFFFF) == c;
          const bool live = cd[k] != kCodeIgn && ok[k] && c < c_hi;
          float gk;
          const float lk = focal_any_in<kLogits>(in[u][k], one, fc, sc, gk);
          out[u][k] = live ? gk : 0.f;
          cl += live ? lk : 0.f;
        }
      }
    }
    // next batch on its way before this one is stored
    if (c0 + U < c_hi) load_planes(c0 + U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u;
      if (c < c_hi) {
        const size_t co = (size_t)c * hw;
        if (v4) {
          if (ok[0]) {
            v4f o;
            o.x = out[u][0]; o.y = out[u][1]; o.z = out[u][2]; o.w = out[u][3];
            __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dcp + base_c[0] + co));
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ok[k]) dcp[base_c[k] + co] = out[u][k];
        }
      }
    }
  }
  const float t0 = block_sum_any(cl, s_red);
  const float t1 = block_sum_any(rl, s_red);
  if (tid == 0) {
    const size_t nb = (size_t)gridDim.x * gridDim.y;
    float* o = partials + ((size_t)j * nb + (size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    o[0] = t0;
    o[1] = t1;
  }
}

// x *= (*sx) / assumed and y *= (*sy) / assumed in one launch; each array is left alone
// (no traffic) when its scale equals `assumed`.  A null scale pointer means "zero the array".
__global__ __launch_bounds__(kThreads) void apply_upstream_kernel(
    float* __restrict__ x, long long nx, const float* __restrict__ sx, float* __restrict__ y,
    long long ny, const float* __restrict__ sy, float assumed) {
  const long long stride = (long long)gridDim.x * kThreads;
  const long long t0 = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (x) {
    const float s = sx ? *sx : 0.0f;
    if (s != assumed) {
      const float f = s / assumed;
      for (long long i = t0; i < nx; i += stride) x[i] = sx ? x[i] * f : 0.0f;
    }
  }
  if (y) {
    const float s = sy ? *sy : 0.0f;
    if (s != assumed) {
      const float f = s / assumed;
      for (long long i = t0; i < ny; i += stride) y[i] = sy ? y[i] * f : 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct LossWorkspace {
  GtRec* gt;
  int* gt_count;
  int* blk_pos;      // [B][assign blocks per image]
  int nblk;
  bool split;        // form of the assignment kernel that fills blk_pos (nblk follows from it)
  int* code;
  float* partials;
  long long partial_count;   // floats
  float* pos_total;  // [1] positives of the whole call (batch-wide normaliser only)
  unsigned* epoch;   // [1] tag of the records of the call in flight (in-launch final reduction)
  float* img_out;    // [B][6] per-image results on their way to the batch means (three atoms)
  size_t bytes;
};

static bool tile_path_ok(long long n, int c) { return n >= kTileRows && c >= 4 && c < 8192; }

// Parts a tile's classification block is cut into.  Up to 20 float4 per thread the tile is cut
// into runs of about 10 (one or two batches of loads); wider tiles into runs of about 5, one
// batch, as at C = 80.  Measured on 8 x 120 087 x 601: 878 us unsplit, 840 us with runs of 10,
// 802 us with runs of 5; runs of 5 at C = 91..100 (two half-empty parts) cost 20 % instead.
static int tile_parts(int c) {
  const int per_thread = (kTileRows * c / 4 + kThreads - 1) / kThreads;
  const int run = per_thread <= 20 ? 10 : 5;
  return (per_thread + run - 1) / run;
}

static long long partial_floats(int batch, long long n, int c) {
  const long long tiles = ((long long)batch * n + kTileRows - 1) / kTileRows;
  const long long a = tiles * 8 * tile_parts(c), b = (long long)batch * kGenericBlocks * 2;
  return a > b ? a : b;
}

static LossWorkspace carve_loss(void* ws, int batch, long long n, int c, int m) {
  Carver cv(ws);
  LossWorkspace w{};
  w.gt = cv.take<GtRec>((size_t)batch * (m > 0 ? m : 1));
  w.gt_count = cv.take<int>(batch);
  w.split = assign_split(m, true);      // the loss wants the codes only
  w.nblk = assign_blocks(n, w.split);
  w.blk_pos = cv.take<int>((size_t)batch * w.nblk);
  w.code = cv.take<int>((size_t)batch * n);
  w.partial_count = partial_floats(batch, n, c);
  w.partials = cv.take<float>((size_t)w.partial_count);
  w.pos_total = cv.take<float>(1);
  w.epoch = cv.take<unsigned>(1);
  w.img_out = cv.take<float>((size_t)batch * 6);
  w.bytes = cv.off;
  return w;
}

static int check_assign_shapes(int batch, long long n, int m, int c) {
  RH_REQUIRE(batch >= 1 && n >= 1 && m >= 0 && c >= 1, RH_ESHAPE,
             "bad shape: batch=%d n=%lld m=%d classes=%d", batch, n, m, c);
  RH_REQUIRE((long long)batch * n < (1ll << 31), RH_ESHAPE, "batch*n=%lld does not fit 31 bits",
             (long long)batch * n);
  RH_REQUIRE(m < 32768, RH_ESHAPE, "m=%d: at most 32767 annotation rows per image", m);
  RH_REQUIRE(c < 65535, RH_ESHAPE, "classes=%d: at most 65534", c);
  RH_REQUIRE(batch <= 65535, RH_ESHAPE, "batch=%d: at most 65535 images per call", batch);
  return RH_OK;
}

static AssignArgs assign_args(const float* anchors, const float* annotations, int batch, long long n, int m,
                              int c, float neg_t, float pos_t, const LossWorkspace& w, int* code,
                              unsigned char* state, int* argmax, int* label, float* iou_max,
                              int* status, const CodeLayout& lay) {
  AssignArgs aa;
  aa.anchors = reinterpret_cast<const float4*>(anchors);
  aa.ann = annotations;
  aa.gt = w.gt;
  aa.gt_count = w.gt_count;
  aa.status = status;
  aa.code = code;
  aa.state = state;
  aa.argmax = argmax;
  aa.label = label;
  aa.iou_max = iou_max;
  aa.blk_pos = w.blk_pos;
  aa.batch = batch;
  aa.epoch = w.epoch;
  aa.clear_a = w.epoch ? reinterpret_cast<unsigned long long*>(w.partials) : nullptr;
  aa.clear_a_count = w.partial_count / 2;
  aa.clear_b = reinterpret_cast<unsigned long long*>(w.img_out);
  aa.clear_b_count = w.img_out ? batch * 3 : 0;
  aa.N = (int)n;
  aa.M = m;
  aa.C = c;
  aa.neg_thresh = neg_t;
  aa.pos_thresh = pos_t;
  aa.lay = lay;
  return aa;
}

static int launch_assign(const float* anchors, const float* annotations, int batch, long long n,
                         int m, int c, float neg_t, float pos_t, const LossWorkspace& w, int* code,
                         unsigned char* state, int* argmax, int* label, float* iou_max, int* status,
                         hipStream_t s, const CodeLayout& lay = CodeLayout{}) {
  RH_REQUIRE(w.nblk <= 65535, RH_ESHAPE, "n=%lld with m=%d: more than 65535 assignment blocks per image", n, m);
  dim3 grid((unsigned)batch, (unsigned)w.nblk);
  hipLaunchKernelGGL(w.split ? assign_kernel<true> : assign_kernel<false>, grid, dim3(kThreads), 0, s,
                     assign_args(anchors, annotations, batch, n, m, c, neg_t, pos_t, w, code, state, argmax,
                                 label, iou_max, status, lay));
  RH_CHECK_LAUNCH("assign_kernel");
  return RH_OK;
}


// ---- head-layout entry points ---------------------------------------------------------------
struct LevelPlan {
  long long n;        // anchors per image: A * sum of cells
  int chunks;         // workgroups along x
  int chunk_begin[kMaxLevels + 1];
  int anchor_off[kMaxLevels];
  int hw[kMaxLevels];
};

static int make_level_plan(int num_levels, const int32_t* heights, const int32_t* widths, int a,
                           LevelPlan* p) {  // chunks: workgroups of kLvlChunk anchor-cells
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
    RH_REQUIRE(hw < (1ll << 31) && (cells + hw) * a < (1ll << 31), RH_ESHAPE, "levels too large");
    p->hw[l] = (int)hw;
    p->anchor_off[l] = (int)(cells * a);
    p->chunk_begin[l] = chunks;
    chunks += (int)((hw * a + kLvlChunk - 1) / kLvlChunk);
    cells += hw;
  }
  p->chunk_begin[num_levels] = chunks;
  p->chunks = chunks;
  p->n = cells * a;
  return RH_OK;
}

// Workgroups that share the class planes of one (image, chunk of units): about 10 planes each,
// two batches of kLvlPlanes (measured best at C = 80 among 5, 8, 10, 16 and 32 planes).
constexpr int kMaxCsplit = 16;
static int level_csplit(int c) {
  int s = c / 10;
  return s < 1 ? 1 : (s > kMaxCsplit ? kMaxCsplit : s);
}

static LossWorkspace carve_levels(void* ws, int batch, long long n, int m, long long blocks) {
  Carver cv(ws);
  LossWorkspace w{};
  w.gt = cv.take<GtRec>((size_t)batch * (m > 0 ? m : 1));
  w.gt_count = cv.take<int>(batch);
  w.split = assign_split(m, true);
  w.nblk = assign_blocks(n, w.split);
  w.blk_pos = cv.take<int>((size_t)batch * w.nblk);
  w.code = cv.take<int>((size_t)batch * n);
  w.partials = cv.take<float>((size_t)batch * blocks * 2);
  w.pos_total = cv.take<float>(1);
  w.bytes = cv.off;
  return w;
}

// How the call normalises: per image (the reference) or by one batch-wide positive count.
struct Normaliser {
  const float* total;   // device f32 [1]; null = per image
  float divisor;        // images the batch mean runs over (1 with a batch-wide count)
};

static int resolve_normaliser(const rh_focal_params* p, int batch, const LossWorkspace& w,
                              Normaliser* out) {
  RH_REQUIRE(p->batch_divisor >= 0, RH_EINVAL, "batch_divisor < 0");
  RH_REQUIRE(p->normalizer == RH_NORM_PER_IMAGE || p->normalizer == RH_NORM_GLOBAL, RH_EINVAL,
             "normalizer=%d is neither RH_NORM_PER_IMAGE nor RH_NORM_GLOBAL", p->normalizer);
  if (p->normalizer == RH_NORM_GLOBAL) {
    out->total = p->num_pos_total ? p->num_pos_total : w.pos_total;
    out->divisor = 1.0f;
  } else {
    RH_REQUIRE(!p->num_pos_total, RH_EINVAL, "num_pos_total given without RH_NORM_GLOBAL");
    out->total = nullptr;
    out->divisor = (float)(p->batch_divisor > 0 ? p->batch_divisor : batch);
  }
  return RH_OK;
}

static FinalIn final_in(const LossWorkspace& w, float* per_image, int* num_pos, int batch,
                        long long n, int tile_mode, const Normaliser& nrm) {
  FinalIn f;
  f.partials = w.partials;
  f.blk_pos = w.blk_pos;
  f.nblk = w.nblk;
  f.gt_count = w.gt_count;
  f.per_image = per_image;
  f.num_pos_out = num_pos;
  f.B = batch;
  f.N = (int)n;
  f.tile_mode = tile_mode;
  f.generic_blocks = kGenericBlocks;
  f.pos_total = nrm.total;
  f.tag = 0u;
  f.status = nullptr;
  return f;
}

static int launch_count(const LossWorkspace& w, int batch, int* num_pos, hipStream_t s) {
  hipLaunchKernelGGL(count_positives_kernel, dim3(1), dim3(kThreads), 0, s, w.blk_pos, w.nblk, batch,
                     num_pos, w.pos_total);
  RH_CHECK_LAUNCH("count_positives_kernel");
  return RH_OK;
}

}  // namespace rh

using namespace rh;

extern "C" {

int rh_calc_iou_f32(const float* a, int64_t n, const float* b, int64_t m, float* out,
                    rh_stream_t stream) {
  RH_REQUIRE(n >= 0 && m >= 0 && m < (1ll << 31), RH_ESHAPE, "bad shape n=%lld m=%lld",
             (long long)n, (long long)m);
  if (n == 0 || m == 0) return RH_OK;
  RH_REQUIRE(a && b && out, RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(a) && aligned16(b), RH_EALIGN, "box arrays must be 16-byte aligned");
  const long long total = (long long)n * m;
  RH_REQUIRE((total + kThreads - 1) / kThreads < (1ll << 31), RH_ESHAPE, "n*m too large");
  hipLaunchKernelGGL(calc_iou_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(b), out, (long long)n, (int)m);
  RH_CHECK_LAUNCH("calc_iou_kernel");
  return RH_OK;
}

static LossWorkspace carve_assign(void* ws, int batch, long long n, int m) {
  Carver cv(ws);
  LossWorkspace w{};
  w.gt = cv.take<GtRec>((size_t)batch * (m > 0 ? m : 1));
  w.gt_count = cv.take<int>(batch);
  w.split = assign_split(m, false);
  w.nblk = assign_blocks(n, w.split);
  w.blk_pos = cv.take<int>((size_t)batch * w.nblk);
  w.bytes = cv.off;
  return w;
}

size_t rh_assign_workspace_bytes(int32_t batch, int64_t n, int32_t m) {
  if (batch < 1 || n < 1 || m < 0) return 0;
  return carve_assign(nullptr, batch, n, m).bytes;
}

int rh_assign(const float* anchors, const float* annotations, int32_t batch, int64_t n, int32_t m,
              int32_t num_classes, float neg_thresh, float pos_thresh, uint8_t* state,
              int32_t* argmax, int32_t* label, float* iou_max, int32_t* num_pos, int32_t* status,
              void* workspace, size_t workspace_bytes, rh_stream_t stream) {
  if (int e = check_assign_shapes(batch, n, m, num_classes)) return e;
  RH_REQUIRE(anchors && (annotations || m == 0) && num_pos, RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(anchors), RH_EALIGN, "anchors must be 16-byte aligned");
  RH_REQUIRE(workspace && workspace_bytes >= rh_assign_workspace_bytes(batch, n, m), RH_EWORKSPACE,
             "workspace too small: need %zu bytes", rh_assign_workspace_bytes(batch, n, m));
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  const LossWorkspace w = carve_assign(workspace, batch, n, m);
  hipStream_t s = (hipStream_t)stream;
  if (int e = launch_assign(anchors, annotations, batch, n, m, num_classes, neg_thresh, pos_thresh,
                            w, nullptr, state, argmax, label, iou_max, status, s))
    return e;
  hipLaunchKernelGGL(sum_blk_pos_kernel, dim3(batch), dim3(kThreads), 0, s, w.blk_pos, w.nblk,
                     num_pos);
  RH_CHECK_LAUNCH("sum_blk_pos_kernel");
  return RH_OK;
}

size_t rh_focal_loss_workspace_bytes(int32_t batch, int64_t n, int32_t num_classes, int32_t m) {
  if (batch < 1 || n < 1 || num_classes < 1 || m < 0) return 0;
  return carve_loss(nullptr, batch, n, num_classes, m).bytes;
}

int rh_focal_loss_fwd_bwd(const float* classifications, const float* regressions,
                          const float* anchors, const float* annotations, int32_t batch, int64_t n,
                          int32_t num_classes, int32_t m, const rh_focal_params* params,
                          float* d_classifications, float* d_regressions, float* losses,
                          float* per_image, int32_t* num_pos, int32_t* status, void* workspace,
                          size_t workspace_bytes, int32_t stages, rh_stream_t stream) {
  return rh_focal_loss_fwd_bwd_ahead(classifications, regressions, anchors, annotations, batch, n,
                                     num_classes, m, params, d_classifications, d_regressions,
                                     losses, per_image, num_pos, status, workspace, workspace_bytes,
                                     stages, nullptr, stream);
}

int rh_focal_loss_fwd_bwd_ahead(const float* classifications, const float* regressions,
                                const float* anchors, const float* annotations, int32_t batch,
                                int64_t n, int32_t num_classes, int32_t m,
                                const rh_focal_params* params, float* d_classifications,
                                float* d_regressions, float* losses, float* per_image,
                                int32_t* num_pos, int32_t* status, void* workspace,
                                size_t workspace_bytes, int32_t stages, const rh_assign_ahead* next,
                                rh_stream_t stream) {
  if (int e = check_assign_shapes(batch, n, m, num_classes)) return e;
  LossWorkspace wn{};
  if (next) {
    RH_REQUIRE(stages & RH_STAGE_STREAM, RH_EINVAL,
               "the next step's assignment rides on the stream stage: stages=%d has none", stages);
    if (int e = check_assign_shapes(batch, n, next->m, num_classes)) return e;
    RH_REQUIRE(next->annotations || next->m == 0, RH_EINVAL, "null pointer");
    const size_t need_next = rh_focal_loss_workspace_bytes(batch, n, num_classes, next->m);
    RH_REQUIRE(next->workspace && next->workspace_bytes >= need_next, RH_EWORKSPACE,
               "next step's workspace too small: need %zu bytes, got %zu", need_next,
               next->workspace_bytes);
    RH_REQUIRE(aligned16(next->workspace), RH_EALIGN, "workspace must be 16-byte aligned");
    RH_REQUIRE(next->workspace != workspace, RH_EINVAL,
               "the next step needs a workspace of its own: this step's is still being read");
    wn = carve_loss(next->workspace, batch, n, num_classes, next->m);
  }
  RH_REQUIRE((stages & ~RH_STAGE_ALL) == 0 && stages != 0, RH_EINVAL, "bad stages mask %d", stages);
  // The assignment stage reads the anchors and the annotations only, so it can be run ahead of
  // the step that owns the predictions (into that step's workspace); the other stages need the
  // prediction and gradient tensors (the final reduction for their alignment, which selects the
  // kernel whose partial sums it reads).
  RH_REQUIRE(anchors && (annotations || m == 0) && params, RH_EINVAL, "null pointer");
  if (stages & (RH_STAGE_STREAM | RH_STAGE_FINALIZE))
    RH_REQUIRE(classifications && regressions && d_classifications && d_regressions && losses,
               RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(anchors), RH_EALIGN, "anchors must be 16-byte aligned");
  const size_t need = rh_focal_loss_workspace_bytes(batch, n, num_classes, m);
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");

  hipStream_t s = (hipStream_t)stream;
  const LossWorkspace w = carve_loss(workspace, batch, n, num_classes, m);
  Normaliser nrm;
  if (int e = resolve_normaliser(params, batch, w, &nrm)) return e;
  const float divisor = nrm.divisor;
  const FocalConst fc{params->alpha, 1e-4f, (float)(1.0 - 1e-4)};
  const bool tile = tile_path_ok(n, num_classes) && aligned16(classifications) &&
                    aligned16(d_classifications);
  const int total_rows = (int)((long long)batch * n);
  const int num_tiles = (total_rows + kTileRows - 1) / kTileRows;

  if (stages & RH_STAGE_ASSIGN) {
    if (int e = launch_assign(anchors, annotations, batch, n, m, num_classes, params->neg_thresh,
                              params->pos_thresh, w, w.code, nullptr, nullptr, nullptr, nullptr,
                              status, s))
      return e;
    if (params->normalizer == RH_NORM_GLOBAL)
      if (int e = launch_count(w, batch, num_pos, s)) return e;
  }
  bool next_assigned = false, finalized = false;
  if (stages & RH_STAGE_STREAM) {
    if (tile) {
      // float4 slots per thread and part: pick the unroll that wastes the fewest
      const int parts = tile_parts(num_classes);
      const int tile_vecs = kTileRows * num_classes / 4;
      // (Equal parts instead -- 1204 float4 each at C = 601 rather than seven of 1280 and a
      // remainder of 656 -- were measured: 844 against 798 us at 8 x 120 087 x 601.)
      const int part_vecs = (int)round_up((tile_vecs + parts - 1) / parts, kThreads);
      const int per_thread = part_vecs / kThreads;
      const bool u5 = (per_thread + 4) / 5 * 5 < (per_thread + 3) / 4 * 4;
      RH_REQUIRE((long long)num_tiles * parts < (1ll << 31), RH_ESHAPE, "too many workgroups");
      // One tile per workgroup: the hardware dispatcher balances the 17k+ tiles better than a
      // grid-stride loop did (measured 125 us against 142 us at B=16, N=69354, C=80), and
      // non-temporal loads and stores (the data is touched once) were worth another 8 %.
      // The next step's assignment joins this launch in its plain form (short annotation lists);
      // otherwise it follows as a launch of its own below.
      // Both stages in one call: the stream kernel's workgroups reduce each image as its last
      // record arrives and no finalize launch follows.
      StreamFinal fin{};
      fin.in = final_in(w, per_image, num_pos, batch, n, parts, nrm);
      // (The finishers wait for records of other workgroups of the same grid.  They are the last
      // workgroups of the grid, and HIP dispatches a grid in index order in practice but does not
      // promise to; with no more finishers than the chip has compute units they could not keep
      // the stream's workgroups from running even if they all started first -- larger batches
      // reduce in a launch of their own.  The wait is bounded either way: a reduction that gives
      // up leaves NaN losses and RH_STATUS_REDUCTION_GAVE_UP in the image's status word.)
      if ((stages & RH_STAGE_FINALIZE) && batch <= kMaxFinishers) {
        fin.in.status = status;
        fin.enabled = 1;
        fin.epoch = w.epoch;
        fin.img_out = w.img_out;
        fin.losses = losses;
        fin.divisor = divisor;
        finalized = true;
      }
      AheadArgs ahead{};
      int ahead_blocks = 0;
      const bool ride = next &&
                        (long long)num_tiles * parts + (long long)batch * wn.nblk < (1ll << 31);
      if (ride) {
        ahead.aa = assign_args(anchors, next->annotations, batch, n, next->m, num_classes,
                               params->neg_thresh, params->pos_thresh, wn, wn.code, nullptr, nullptr,
                               nullptr, nullptr, next->status, CodeLayout{});
        ahead.batch = batch;
        ahead.nblk = wn.nblk;
        ahead_blocks = batch * wn.nblk;
        next_assigned = true;
      }
      const int form = !ride ? 0 : (wn.split ? 2 : 1);
      typedef decltype(&focal_stream_kernel<5, false, 0>) StreamKernel;
      static const StreamKernel table[2][2][3] = {
          {{focal_stream_kernel<4, false, 0>, focal_stream_kernel<4, false, 1>, focal_stream_kernel<4, false, 2>},
           {focal_stream_kernel<4, true, 0>, focal_stream_kernel<4, true, 1>, focal_stream_kernel<4, true, 2>}},
          {{focal_stream_kernel<5, false, 0>, focal_stream_kernel<5, false, 1>, focal_stream_kernel<5, false, 2>},
           {focal_stream_kernel<5, true, 0>, focal_stream_kernel<5, true, 1>, focal_stream_kernel<5, true, 2>}}};
      const StreamKernel kern = table[u5 ? 1 : 0][params->from_logits ? 1 : 0][form];
      hipLaunchKernelGGL(kern, dim3(num_tiles * parts + (fin.enabled ? batch : 0) + ahead_blocks),
                         dim3(kThreads), 0, s,
                         classifications, d_classifications, regressions, d_regressions,
                         reinterpret_cast<const float4*>(anchors), w.gt, w.code, w.blk_pos, w.nblk,
                         w.partials, (int)n, num_classes, m, total_rows, num_tiles,
                         div_magic((unsigned)num_classes), fc, params->grad_cls, params->grad_reg,
                         divisor, nrm.total, parts, part_vecs, ahead, fin);
      RH_CHECK_LAUNCH("focal_stream_kernel");
    } else {
      auto kern = params->from_logits ? focal_rows_kernel<true> : focal_rows_kernel<false>;
      hipLaunchKernelGGL(kern, dim3(kGenericBlocks, batch), dim3(kThreads), 0, s,
                         classifications, d_classifications, regressions, d_regressions, anchors,
                         w.gt, w.code, w.blk_pos, w.nblk, w.partials, (int)n, num_classes, m, fc,
                         params->grad_cls, params->grad_reg, divisor, nrm.total);
      RH_CHECK_LAUNCH("focal_rows_kernel");
    }
  }
  if ((stages & RH_STAGE_FINALIZE) && !finalized) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kFinalThreads), 0, s,
                       final_in(w, per_image, num_pos, batch, n, tile ? tile_parts(num_classes) : 0, nrm),
                       losses, divisor);
    RH_CHECK_LAUNCH("finalize_kernel");
  }
  if (next) {
    if (!next_assigned)
      if (int e = launch_assign(anchors, next->annotations, batch, n, next->m, num_classes,
                                params->neg_thresh, params->pos_thresh, wn, wn.code, nullptr,
                                nullptr, nullptr, nullptr, next->status, s))
        return e;
    if (params->normalizer == RH_NORM_GLOBAL)
      if (int e = launch_count(wn, batch, next->num_pos, s)) return e;
  }
  return RH_OK;
}

size_t rh_focal_loss_levels_workspace_bytes(int32_t batch, int32_t num_levels, const int32_t* heights,
                                            const int32_t* widths, int32_t anchors_per_cell,
                                            int32_t m) {
  LevelPlan lp;
  if (batch < 1 || m < 0 || make_level_plan(num_levels, heights, widths, anchors_per_cell, &lp))
    return 0;
  // sized for the largest split so that the query does not depend on the class count
  return carve_levels(nullptr, batch, lp.n, m, (long long)lp.chunks * kMaxCsplit).bytes;
}

int rh_focal_loss_levels_fwd_bwd(const float* const* cls_levels, const float* const* reg_levels,
                                 const float* anchors, const float* annotations, int32_t batch,
                                 int32_t num_levels, const int32_t* heights, const int32_t* widths,
                                 int32_t anchors_per_cell, int32_t num_classes, int32_t m,
                                 const rh_focal_params* params, float* const* d_cls_levels,
                                 float* const* d_reg_levels, float* losses, float* per_image,
                                 int32_t* num_pos, int32_t* status, void* workspace,
                                 size_t workspace_bytes, int32_t stages, rh_stream_t stream) {
  LevelPlan lp;
  if (int e = make_level_plan(num_levels, heights, widths, anchors_per_cell, &lp)) return e;
  if (int e = check_assign_shapes(batch, lp.n, m, num_classes)) return e;
  RH_REQUIRE(cls_levels && reg_levels && d_cls_levels && d_reg_levels && anchors &&
                 (annotations || m == 0) && losses && params,
             RH_EINVAL, "null pointer");
  RH_REQUIRE(aligned16(anchors), RH_EALIGN, "anchors must be 16-byte aligned");
  const int csplit = level_csplit(num_classes);
  const long long blocks = (long long)lp.chunks * csplit;
  const size_t need = carve_levels(nullptr, batch, lp.n, m,
                                   (long long)lp.chunks * kMaxCsplit).bytes;
  RH_REQUIRE(workspace && workspace_bytes >= need, RH_EWORKSPACE,
             "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  RH_REQUIRE(aligned16(workspace), RH_EALIGN, "workspace must be 16-byte aligned");
  RH_REQUIRE((stages & ~RH_STAGE_ALL) == 0 && stages != 0, RH_EINVAL, "bad stages mask %d", stages);
  RH_REQUIRE(blocks < (1ll << 24), RH_ESHAPE, "too many workgroups per image");

  LevelTable lt{};
  lt.num_levels = num_levels;
  for (int l = 0; l < num_levels; ++l) {
    // offsets inside one image's [A*C, H*W] block are 32-bit in the kernel
    RH_REQUIRE((long long)anchors_per_cell * num_classes * lp.hw[l] < (1ll << 31), RH_ESHAPE,
               "level %d: A*C*H*W does not fit 31 bits", l);
    RH_REQUIRE(cls_levels[l] && reg_levels[l] && d_cls_levels[l] && d_reg_levels[l], RH_EINVAL,
               "null pointer for level %d", l);
    lt.cls[l] = cls_levels[l];
    lt.dcls[l] = d_cls_levels[l];
    lt.reg[l] = reg_levels[l];
    lt.dreg[l] = d_reg_levels[l];
    lt.hw[l] = lp.hw[l];
    lt.anchor_off[l] = lp.anchor_off[l];
    lt.chunk_begin[l] = lp.chunk_begin[l];
    lt.vec4[l] = lp.hw[l] % 4 == 0 && aligned16(cls_levels[l]) && aligned16(d_cls_levels[l]) &&
                 aligned16(reg_levels[l]) && aligned16(d_reg_levels[l]);
  }
  lt.chunk_begin[num_levels] = lp.chunks;

  hipStream_t s = (hipStream_t)stream;
  const LossWorkspace w = carve_levels(workspace, batch, lp.n, m,
                                       (long long)lp.chunks * kMaxCsplit);
  Normaliser nrm;
  if (int e = resolve_normaliser(params, batch, w, &nrm)) return e;
  const float divisor = nrm.divisor;
  const FocalConst fc{params->alpha, 1e-4f, (float)(1.0 - 1e-4)};
  if (stages & RH_STAGE_ASSIGN) {
    CodeLayout lay{};
    lay.num_levels = num_levels;
    lay.a = anchors_per_cell;
    for (int l = 0; l < num_levels; ++l) {
      lay.anchor_off[l] = lp.anchor_off[l];
      lay.hw[l] = lp.hw[l];
    }
    lay.anchor_off[num_levels] = (int)lp.n;
    if (int e = launch_assign(anchors, annotations, batch, lp.n, m, num_classes, params->neg_thresh,
                              params->pos_thresh, w, w.code, nullptr, nullptr, nullptr, nullptr,
                              status, s, lay))
      return e;
    if (params->normalizer == RH_NORM_GLOBAL)
      if (int e = launch_count(w, batch, num_pos, s)) return e;
  }
  if (stages & RH_STAGE_STREAM) {
    auto kern = params->from_logits ? focal_levels_kernel<true> : focal_levels_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(lp.chunks, csplit, batch), dim3(kLvlThreads), 0, s, lt,
                       reinterpret_cast<const float4*>(anchors), w.gt, w.code, w.blk_pos, w.nblk,
                       w.partials, (int)lp.n, anchors_per_cell, num_classes, m, fc,
                       params->grad_cls, params->grad_reg, divisor, nrm.total);
    RH_CHECK_LAUNCH("focal_levels_kernel");
  }
  if (stages & RH_STAGE_FINALIZE) {
    FinalIn f = final_in(w, per_image, num_pos, batch, lp.n, 0, nrm);
    f.generic_blocks = (int)blocks;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kFinalThreads), 0, s, f, losses, divisor);
    RH_CHECK_LAUNCH("finalize_kernel");
  }
  return RH_OK;
}

int rh_apply_upstream_f32(float* x, int64_t nx, const float* scale_x, float* y, int64_t ny,
                          const float* scale_y, float assumed, rh_stream_t stream) {
  RH_REQUIRE(nx >= 0 && ny >= 0, RH_ESHAPE, "negative count");
  RH_REQUIRE(assumed != 0.0f, RH_EINVAL, "assumed scale must be non-zero");
  if (nx == 0) x = nullptr;
  if (ny == 0) y = nullptr;
  if (!x && !y) return RH_OK;
  const long long most = nx > ny ? nx : ny;
  long long blocks = (most + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(apply_upstream_kernel, dim3((unsigned)blocks), dim3(kThreads), 0,
                     (hipStream_t)stream, x, (long long)nx, scale_x, y, (long long)ny, scale_y,
                     assumed);
  RH_CHECK_LAUNCH("apply_upstream_kernel");
  return RH_OK;
}

}  // extern "C"
