// Wave-level device helpers of the training kernels (gfx950, wave64): lane shuffles of 64-bit values, the
// data-parallel-primitive prefix sum, and the split search of ONE feature held by one wave, shared by the
// growth-round kernels (kernels.hip) and the whole-tree kernel (tree_kernel.hip).  Everything is
// __forceinline__: including translation units get their own copies.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <limits>
#include "../common.hpp"
#include "kernels.hpp"

namespace lga {
namespace k {
namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
  int lo = __shfl((int)(uint32_t)((uint64_t)v & 0xffffffffull), src, kWave);
  int hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src, kWave);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int d) {
  int lo = __shfl_up((int)(uint32_t)((uint64_t)v & 0xffffffffull), d, kWave);
  int hi = __shfl_up((int)(uint32_t)((uint64_t)v >> 32), d, kWave);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int m) {
  int lo = __shfl_xor((int)(uint32_t)((uint64_t)v & 0xffffffffull), m, kWave);
  int hi = __shfl_xor((int)(uint32_t)((uint64_t)v >> 32), m, kWave);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  return __longlong_as_double(shfl_xor_i64(__double_as_longlong(v), m));
}
// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive moves (no LDS round trip per
// step): shifts by 1, 2, 4, 8 inside each row of 16 lanes, then the row totals are carried on with the row
// broadcasts (lane 15 of rows 0 / 2 to rows 1 / 3, lane 31 to rows 2 and 3).  Lanes without a source add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t dpp_move_i64(int64_t v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)v & 0xffffffffull), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)v >> 32), CTRL, ROW_MASK, 0xF, false);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ int64_t wave_prefix_sum_i64(int64_t v) {
  v += dpp_move_i64<0x111, 0xF>(v);   // row_shr:1
  v += dpp_move_i64<0x112, 0xF>(v);   // row_shr:2
  v += dpp_move_i64<0x114, 0xF>(v);   // row_shr:4
  v += dpp_move_i64<0x118, 0xF>(v);   // row_shr:8
  v += dpp_move_i64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move_i64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int64_t read_lane_i64(int64_t v, int src_lane) {   // src_lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v & 0xffffffffull), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src_lane);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_i64(v, m);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}

// ---------------------------------------------------------------------------
// split search of one feature
// ---------------------------------------------------------------------------
__device__ __forceinline__ void take_better(double& g, uint64_t& key, int64_t& lg, int64_t& lh, double g2,
                                            uint64_t k2, int64_t lg2, int64_t lh2) {
  if (cand_better(g2, k2, g, key)) { g = g2; key = k2; lg = lg2; lh = lh2; }
}

// A wave holds one feature's histogram row, `bpl` bins per lane (bins_per_lane: as few as cover the feature's bins):
// lane l has bins l * bpl .. l * bpl + bpl - 1 in entries 0 .. bpl - 1; further entries are zero.  A feature with
// up to 64 bins costs a quarter of the gain arithmetic of one with 256.
__device__ __forceinline__ int bins_per_lane(int num_bin) {
  const int b = (num_bin + kWave - 1) / kWave;
  return b < 1 ? 1 : (b > 4 ? 4 : b);
}
__device__ __forceinline__ void load_feat_cells(const int64_t* __restrict__ blk, int lane, int bpl, int64_t (&g)[4], int64_t (&h)[4]) {
  if (bpl == 4) {
    const longlong2* gs = reinterpret_cast<const longlong2*>(blk + lane * 4);
    const longlong2* hs = reinterpret_cast<const longlong2*>(blk + kMaxBins + lane * 4);
    const longlong2 g01 = gs[0], g23 = gs[1], h01 = hs[0], h23 = hs[1];
    g[0] = g01.x; g[1] = g01.y; g[2] = g23.x; g[3] = g23.y;
    h[0] = h01.x; h[1] = h01.y; h[2] = h23.x; h[3] = h23.y;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = j < bpl ? blk[lane * bpl + j] : 0;
      h[j] = j < bpl ? blk[kMaxBins + lane * bpl + j] : 0;
    }
  }
}
__device__ __forceinline__ void store_feat_cells(int64_t* __restrict__ blk, int lane, int bpl, const int64_t (&g)[4], const int64_t (&h)[4]) {
  if (bpl == 4) {
    longlong2* go = reinterpret_cast<longlong2*>(blk + lane * 4);
    longlong2* ho = reinterpret_cast<longlong2*>(blk + kMaxBins + lane * 4);
    go[0] = make_longlong2(g[0], g[1]); go[1] = make_longlong2(g[2], g[3]);
    ho[0] = make_longlong2(h[0], h[1]); ho[1] = make_longlong2(h[2], h[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < bpl) {
        blk[lane * bpl + j] = g[j];
        blk[kMaxBins + lane * bpl + j] = h[j];
      }
    }
  }
}

// Combines the lanes' running bests: on return every lane holds the wave's winner.  The candidates travel by
// data-parallel-primitive moves - shifts inside the rows of 16 lanes, then the row broadcasts, as in
// wave_prefix_sum_i64 - which leaves the winner in lane 63; one read-lane hands it to everybody.  (A butterfly of
// ds_bpermute shuffles costs several times as much: 8 dwords per step through the LDS crossbar.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_take_better(double& bg, uint64_t& bkey, int64_t& blg, int64_t& blh) {
  // lanes without a source keep their own candidate (bound_ctrl off: the old value is the lane's own)
  const int g_lo = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)__double_as_longlong(bg) & 0xffffffffull),
                                               (int)(uint32_t)((uint64_t)__double_as_longlong(bg) & 0xffffffffull), CTRL, ROW_MASK, 0xF, false);
  const int g_hi = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)__double_as_longlong(bg) >> 32),
                                               (int)(uint32_t)((uint64_t)__double_as_longlong(bg) >> 32), CTRL, ROW_MASK, 0xF, false);
  const int k_lo = __builtin_amdgcn_update_dpp((int)(uint32_t)(bkey & 0xffffffffull), (int)(uint32_t)(bkey & 0xffffffffull), CTRL, ROW_MASK, 0xF, false);
  const int k_hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(bkey >> 32), (int)(uint32_t)(bkey >> 32), CTRL, ROW_MASK, 0xF, false);
  const int a_lo = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)blg & 0xffffffffull), (int)(uint32_t)((uint64_t)blg & 0xffffffffull), CTRL, ROW_MASK, 0xF, false);
  const int a_hi = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)blg >> 32), (int)(uint32_t)((uint64_t)blg >> 32), CTRL, ROW_MASK, 0xF, false);
  const int b_lo = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)blh & 0xffffffffull), (int)(uint32_t)((uint64_t)blh & 0xffffffffull), CTRL, ROW_MASK, 0xF, false);
  const int b_hi = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)blh >> 32), (int)(uint32_t)((uint64_t)blh >> 32), CTRL, ROW_MASK, 0xF, false);
  const double g2 = __longlong_as_double((long long)(((uint64_t)(uint32_t)g_hi << 32) | (uint64_t)(uint32_t)g_lo));
  const uint64_t k2 = ((uint64_t)(uint32_t)k_hi << 32) | (uint64_t)(uint32_t)k_lo;
  const int64_t lg2 = (int64_t)(((uint64_t)(uint32_t)a_hi << 32) | (uint64_t)(uint32_t)a_lo);
  const int64_t lh2 = (int64_t)(((uint64_t)(uint32_t)b_hi << 32) | (uint64_t)(uint32_t)b_lo);
  take_better(bg, bkey, blg, blh, g2, k2, lg2, lh2);
}
__device__ __forceinline__ void wave_best(double& bg, uint64_t& bkey, int64_t& blg, int64_t& blh) {
  dpp_take_better<0x111, 0xF>(bg, bkey, blg, blh);   // row_shr:1
  dpp_take_better<0x112, 0xF>(bg, bkey, blg, blh);   // row_shr:2
  dpp_take_better<0x114, 0xF>(bg, bkey, blg, blh);   // row_shr:4
  dpp_take_better<0x118, 0xF>(bg, bkey, blg, blh);   // row_shr:8: lane 15 of every row holds its row's winner
  dpp_take_better<0x142, 0xA>(bg, bkey, blg, blh);   // row_bcast:15 into rows 1 and 3
  dpp_take_better<0x143, 0xC>(bg, bkey, blg, blh);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
  bg = __longlong_as_double(read_lane_i64(__double_as_longlong(bg), 63));
  bkey = (uint64_t)read_lane_i64((int64_t)bkey, 63);
  blg = read_lane_i64(blg, 63);
  blh = read_lane_i64(blh, 63);
}

// Best split of ONE feature of one leaf from the feature's histogram held by one wave (g, h: bpl bins per lane, see
// load_feat_cells; overwritten).  Every lane folds ITS candidates of this feature into its own running best; the
// lanes' bests are combined once per leaf (wave_best), not once per feature.  The per-feature gain adjustment is
// monotone in the gain, so it may be applied to a lane's best candidate of the feature.
__device__ __forceinline__ void scan_cells_wave(int f, int lane, int bpl, int64_t (&g)[4], int64_t (&h)[4], const FeatureMeta m,
                                                const LeafScanCtx& ctx, const SplitParams& sp, double& bg, uint64_t& bkey,
                                                int64_t& blg, int64_t& blh, uint32_t key) {
  const int nb = m.num_bin, mb = m.missing_bin;
  const int vb = mb >= 0 ? nb - 1 : nb;
  const double ninf = -std::numeric_limits<double>::infinity();
  double lbg = ninf;
  uint64_t lkey = ~0ull;
  int64_t llg = 0, llh = 0;
  const bool cat = m.categorical != 0;
  // numeric features: thresholds 0 .. last, the left child holds the bins up to the threshold (prefix sums) and - second
  // candidate per threshold - also the missing-value bin.  Categorical ones: one category against the rest, the left
  // child is a single value bin.
  int64_t mg = 0, mh = 0, eg = 0, eh = 0;
  if (!cat) {
    if (mb >= 0) {
      // stats of the missing-value bin, broadcast from the lane that holds it
      const int ml = mb / bpl, j = mb - ml * bpl;
      int64_t og = j == 0 ? g[0] : j == 1 ? g[1] : j == 2 ? g[2] : g[3];
      int64_t oh = j == 0 ? h[0] : j == 1 ? h[1] : j == 2 ? h[2] : h[3];
      mg = read_lane_i64(og, ml);
      mh = read_lane_i64(oh, ml);
    }
    // inclusive prefix inside the lane, exclusive prefix across lanes (entries beyond bpl are zero)
    g[1] += g[0]; g[2] += g[1]; g[3] += g[2];
    h[1] += h[0]; h[2] += h[1]; h[3] += h[2];
    eg = wave_prefix_sum_i64(g[3]) - g[3];
    eh = wave_prefix_sum_i64(h[3]) - h[3];
  }
  const int last = cat ? vb - 1 : (mb >= 0 ? vb - 1 : vb - 2);
  const int only = extra_trees_threshold(sp, key, f, last + 1);
  const int n_dir = (!cat && mb >= 0) ? 2 : 1;
  const int mono = cat ? 0 : m.monotone;
  // ONE copy of the gain arithmetic (it is a few hundred fp64 instructions): the lane's entries pass through entry 0
  int64_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
#pragma unroll 1
  for (int j = 0; j < bpl; ++j) {
    const int thr = lane * bpl + j;
    const bool in_range = thr <= last && (only < 0 || thr == only);
#pragma unroll 1
    for (int d = 0; d < n_dir; ++d) {
      const bool valid = in_range && (d == 0 || thr <= vb - 2);
      if (__any(valid)) {
        const int64_t lg = eg + g0 + (d ? mg : 0), lh = eh + h0 + (d ? mh : 0);
        const double gain = valid ? split_gain(lg, lh, ctx, sp, mono) : ninf;
        if (valid) take_better(lbg, lkey, llg, llh, gain, cand_key(f, thr, d), lg, lh);
      }
    }
    g0 = g1; g1 = g2; g2 = g3;
    h0 = h1; h1 = h2; h2 = h3;
  }
  // with a gain adjustment the feature's winner is chosen on the raw gains first, as the host engine does
  if (sp.feature_scale) wave_best(lbg, lkey, llg, llh);
  take_better(bg, bkey, blg, blh, adjust_feature_gain(lbg, f, ctx.total_cnt, sp), lkey, llg, llh);
}
__device__ __forceinline__ void write_split(SplitInfo* o, double bg, uint64_t bkey, int64_t blg, int64_t blh, int64_t tc,
                                            int64_t tg, int64_t th, const QuantScale& q, const SplitParams& sp) {
  const double ninf = -std::numeric_limits<double>::infinity();
  o->total_cnt = tc; o->total_g = tg; o->total_h = th; o->pad0 = 0;
  if (bg == ninf) {
    o->gain = ninf; o->feature = -1; o->threshold = 0; o->default_left = 0; o->left_g = 0; o->left_h = 0; o->left_cnt = 0;
  } else {
    o->gain = bg;
    o->feature = (int32_t)(bkey >> 16);
    o->threshold = (int32_t)((bkey >> 1) & 0x7fff);
    o->default_left = (int32_t)(bkey & 1);
    o->left_g = blg; o->left_h = blh;
    o->left_cnt = est_count(blh, make_scan_ctx(tg, th, tc, unit_of(q.kg), unit_of(q.kh), sp));
  }
}

__device__ __forceinline__ LeafScanCtx node_scan_ctx(int64_t tg, int64_t th, int64_t tc, const QuantScale& q,
                                                     const SplitParams& sp, int node) {
  const double kInf = std::numeric_limits<double>::infinity();
  const double kNaN = std::numeric_limits<double>::quiet_NaN();
  return make_scan_ctx(tg, th, tc, unit_of(q.kg), unit_of(q.kh), sp,
                       sp.node_bounds ? sp.node_bounds[kNodeInfoStride * node] : -kInf,
                       sp.node_bounds ? sp.node_bounds[kNodeInfoStride * node + 1] : kInf,
                       sp.node_bounds ? sp.node_bounds[kNodeInfoStride * node + 2] : kNaN);
}

// Hand-over between the workgroups of one launch.  Every wave has drained its own stores / atomics
// (s_waitcnt) before the barrier; one lane then releases at agent scope and draws a ticket.
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned release_and_ticket(unsigned* counter) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void acquire_after_ticket() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace
}  // namespace k
}  // namespace lga
