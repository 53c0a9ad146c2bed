// Whole-tree kernel for gfx950 (kernels.hpp: TreeArgs): one workgroup of 8 waves grows a complete leaf-wise tree.
//
// Small data - the 37 k x 6 folds of r/gridsearchCV.R:26-27,108-115 - cannot fill the device with one tree: a
// growth round there is a handful of tiny launches and a host round trip, ~20 times per tree.  Here the whole
// best-first loop runs inside one workgroup, synchronised by workgroup barriers only, so a tree costs one launch
// and one hand-over, and the device is filled by running MANY trees at once - one workgroup each (the fold
// boosters of a CV, the rows of a grid) - instead of by spreading one tree over many workgroups.
//
// Data layout: the bag's rows are gathered once per tree into record arenas (row id | packed 16-bit gradients |
// bin bytes) that are partitioned physically, ping-pong between two buffers like the row-index arenas of the
// growth rounds.  Every pass over a leaf therefore reads consecutive memory - no dependent row-id -> data
// gathers - and a split of a small leaf is one coalesced load, LDS atomics, one coalesced store.
//
// The arithmetic is the growth rounds' (wave_scan.hpp, ../common.hpp): integer histograms, the same candidate
// order, the same tie rules; the tree equals the one the host replay of grower.cpp builds.
#include "kernels.hpp"
#include "device.hpp"
#include "wave_scan.hpp"

namespace lga {
namespace k {
namespace {

constexpr int kTreeWaves = kTreeThreads / kWave;
constexpr int kMaskWords = kTreeMaxRows / kWave;
constexpr int kWordsPerThread = kMaskWords / kTreeThreads;
static_assert(kWordsPerThread * kTreeThreads == kMaskWords, "mask words must divide over the threads");
constexpr int kNodeCap = 512;   // kTreeMaxNodes rounded up

template <int BW>
__device__ __forceinline__ void load_words(const uint32_t* __restrict__ p, unsigned (&w)[BW]) {
  if constexpr (BW == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x; w[1] = v.y;
  } else if constexpr (BW == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
    const uint4 v0 = reinterpret_cast<const uint4*>(p)[0], v1 = reinterpret_cast<const uint4*>(p)[1];
    w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
  }
}
template <int BW>
__device__ __forceinline__ void store_words(uint32_t* __restrict__ p, const unsigned (&w)[BW]) {
  if constexpr (BW == 2) {
    *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
  } else if constexpr (BW == 4) {
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}
// Byte `f` of a row's bin words; f may differ per lane (selects, no indexed register file).
template <int BW>
__device__ __forceinline__ unsigned bin_byte(const unsigned (&w)[BW], unsigned f) {
  const unsigned q = f >> 2;
  unsigned d = w[0];
#pragma unroll
  for (int i = 1; i < BW; ++i) d = q == (unsigned)i ? w[i] : d;
  return (d >> ((f & 3u) * 8u)) & 255u;
}

// Adds one row to the LDS histogram.  Lane l starts at feature l mod nf and walks on from there, so the lanes of a
// wave work on different features at any moment - a feature with five distinct values does not make 64 lanes
// queue on five words - and cell (f, bin) sits at f * 256 + ((bin + 8 f) & 255), which keeps equal bins of
// different features in different banks.
template <int BW>
__device__ __forceinline__ void hist_add(unsigned long long* lds, const unsigned (&w)[BW], unsigned long long pk,
                                         int nf, unsigned f_start, unsigned enabled) {
  unsigned f = f_start;
#pragma unroll
  for (int j = 0; j < BW * 4; ++j) {
    if (j < nf) {
      if ((enabled >> f) & 1u) {
        const unsigned bn = bin_byte<BW>(w, f);
        atomicAdd(&lds[f * kMaxBins + ((bn + 8u * f) & 255u)], pk);
      }
      f = f + 1u == (unsigned)nf ? 0u : f + 1u;
    }
  }
}

// Lane l's cells (bins l * bpl .. l * bpl + bpl - 1, see load_feat_cells) of feature f in the LDS histogram, unpacked.
__device__ __forceinline__ void lds_feat_cells(const unsigned long long* lds, int f, int lane, int bpl, int64_t (&g)[4], int64_t (&h)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned long long v = j < bpl ? lds[f * kMaxBins + ((lane * bpl + j + 8 * f) & 255)] : 0ull;
    g[j] = (int64_t)(int32_t)(v >> 32);
    h[j] = (int64_t)(v & 0xffffffffull);
  }
}

// Wave-uniform values that come out of LDS or vector loads live in vector registers unless told otherwise: everything
// uniform is moved to scalar registers (the fp64 gain code needs the vector ones).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ long long uni(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

struct TreeShared {
  unsigned long long smask[kMaskWords];
  int word_prefix[kMaskWords + 1];
  int wave_tot[kTreeWaves];
  long long red[2][kTreeWaves];
  double ngain[kNodeCap];          // gain of an open leaf that may be split; -inf otherwise
  long long ntot[kNodeCap][3];     // rows, sum g, sum h
  int nbegin[kNodeCap], ncount[kNodeCap];
  unsigned nkey[kNodeCap];
  short nslot[kNodeCap], ndepth[kNodeCap];
  signed char nbuf[kNodeCap], nsplit[kNodeCap];   // arena of the node's rows; 1 once the node has been split
  short split_node[kTreeMaxSplits + 1];
  Cand wbest[kTreeWaves];
  FeatureMeta meta[kTileFeatures];
  // the tree as a walkable table (boosting chunks): a split node's decision word - feature | threshold << 8 |
  // default_left << 16 | categorical << 17 | (missing bin + 1) << 18 - and left child (the right one follows it),
  // every node's parent, a final leaf's output
  int ndec[kNodeCap];
  int nbest[kNodeCap];            // a leaf's best split, for the pass that applies it: feature | threshold << 8 |
                                  // default_left << 16 | (the left child is the smaller one) << 17
  short nleft[kNodeCap], nparent[kNodeCap];
  double nvalue[kNodeCap];
  long long ticks[8], t_mark;     // phase profile (thread 0)
  float fmax[2][kTreeWaves];
  double msum[4][2];
  int bag_n;
};

// Exclusive prefix over the go-left mask words S.smask[0 .. n_words): S.word_prefix, total in S.word_prefix[kMaskWords].
__device__ __forceinline__ void mask_prefix(TreeShared& S, int n_words) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int pcs[kWordsPerThread];
  int pc = 0;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    const int wi = tid * kWordsPerThread + j;
    pcs[j] = wi < n_words ? __popcll(S.smask[wi]) : 0;
    pc += pcs[j];
  }
  int incl = pc;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int o = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += o;
  }
  if (lane == kWave - 1) S.wave_tot[wave] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += S.wave_tot[w];
  int run = off + incl - pc;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    S.word_prefix[tid * kWordsPerThread + j] = run;
    run += pcs[j];
  }
  if (tid == kTreeThreads - 1) S.word_prefix[kMaskWords] = run;
  __syncthreads();
}

// Grows one tree over the rows bag[0 .. n); leaves the node tables in S and every node's best split in a.best.
// feature_mask: features the tree may use, on top of the descriptor's own enabled flags.  ticks: phase profile
// (thread 0).  Returns the number of splits.
template <int BW>
__device__ __forceinline__ int grow_one(const TreeArgs& a, const int32_t* __restrict__ bag, int n, unsigned feature_mask,
                                        const SplitParams& sp, int kg, int kh, unsigned long long* lds, TreeShared& S) {
  constexpr int FC = BW * 4;
  constexpr int PPW = (2 * FC + kTreeWaves - 1) / kTreeWaves;   // (child, feature) pairs per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nf = a.F;
  const double ninf = -std::numeric_limits<double>::infinity();
  const QuantScale q{kg, kh};
  if (tid < nf) {
    FeatureMeta m = a.meta[tid];
    if (!((feature_mask >> tid) & 1u)) m.enabled = 0;
    S.meta[tid] = m;
  }
  for (int i = tid; i < kNodeCap; i += kTreeThreads) { S.ngain[i] = ninf; S.nsplit[i] = 0; }
  for (int i = tid; i < nf * kMaxBins; i += kTreeThreads) lds[i] = 0ull;
  __syncthreads();
  unsigned enabled = 0u;
  for (int f = 0; f < nf; ++f) enabled |= S.meta[f].enabled ? (1u << f) : 0u;
  const unsigned f_start = (unsigned)(lane % nf);
  // optional phase profile: thread 0 reads the 100 MHz wall clock at the phase boundaries
  if (a.timing && tid == 0) S.t_mark = (long long)wall_clock64();
  auto lap = [&](int phase) {
    if (a.timing && tid == 0) {
      const long long now = (long long)wall_clock64();
      S.ticks[phase] += now - S.t_mark;
      S.t_mark = now;
    }
  };

  // ---- root: gather the bag into arena 0, quantise on the way, histogram in passes of kMaxRowsPerPass rows ----
  long long sg = 0, sh = 0;
  int64_t* slot0 = a.pool;
  for (int seg = 0; seg < n; seg += kMaxRowsPerPass) {
    const int seg_hi = min(n, seg + kMaxRowsPerPass);
    constexpr int U = 4;
    for (int base = seg; base < seg_hi; base += U * kTreeThreads) {
      int row[U];
      unsigned w[U][BW];
      float gv[U], hv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pos = base + u * kTreeThreads + tid;
        row[u] = pos < seg_hi ? bag[pos] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (row[u] >= 0) {
          load_words<BW>(reinterpret_cast<const uint32_t*>(a.bins_rt + (int64_t)row[u] * kTileFeatures), w[u]);
          gv[u] = a.g[row[u]];
          hv[u] = a.h[row[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (row[u] >= 0) {
          const int pos = base + u * kTreeThreads + tid;
          const unsigned long long pk = (unsigned long long)pack_gh(lga::quantize(gv[u], kg), lga::quantize(hv[u], kh));
          a.idx[0][pos] = row[u];
          a.gh[0][pos] = pk;
          store_words<BW>(a.bins[0] + (int64_t)pos * BW, w[u]);
          sg += (long long)(int32_t)(pk >> 32);
          sh += (long long)(pk & 0xffffffffull);
          hist_add<BW>(lds, w[u], pk, nf, f_start, enabled);
        }
      }
    }
    __syncthreads();
    // this pass's cells go to slot 0 (first pass: stored; later ones: added - every cell has one owner thread)
    for (int e = tid; e < nf * kMaxBins; e += kTreeThreads) {
      const int fe = e >> 8, b = e & 255;
      const int at = (fe << 8) + ((b + 8 * fe) & 255);
      const unsigned long long v = lds[at];
      lds[at] = 0ull;
      int64_t* blk = slot0 + (int64_t)fe * (2 * kMaxBins);
      const int64_t vg = (int64_t)(int32_t)(v >> 32), vh = (int64_t)(v & 0xffffffffull);
      if (seg == 0) { blk[b] = vg; blk[kMaxBins + b] = vh; }
      else { blk[b] += vg; blk[kMaxBins + b] += vh; }
    }
    __syncthreads();
  }
  sg = wave_sum_i64(sg);
  sh = wave_sum_i64(sh);
  if (lane == 0) { S.red[0][wave] = sg; S.red[1][wave] = sh; }
  __syncthreads();
  if (tid == 0) {
    long long tg = 0, th = 0;
    for (int w = 0; w < kTreeWaves; ++w) { tg += S.red[0][w]; th += S.red[1][w]; }
    S.ntot[0][0] = n; S.ntot[0][1] = tg; S.ntot[0][2] = th;
    S.nbegin[0] = 0; S.ncount[0] = n; S.nbuf[0] = 0; S.nslot[0] = 0; S.ndepth[0] = 0; S.nkey[0] = kRootNodeKey;
    S.nparent[0] = -1;
    a.qs_out->kg = kg; a.qs_out->kh = kh;
  }
  __syncthreads();
  // root search: one wave per feature
  {
    const LeafScanCtx ctx = node_scan_ctx(S.ntot[0][1], S.ntot[0][2], S.ntot[0][0], q, sp, 0);
    double bg = ninf;
    uint64_t bkey = ~0ull;
    int64_t blg = 0, blh = 0;
    for (int f = wave; f < nf; f += kTreeWaves) {
      const FeatureMeta m = S.meta[f];
      if (!m.enabled) continue;
      int64_t g[4], h[4];
      const int bpl = bins_per_lane(m.num_bin);
      load_feat_cells(slot0 + (int64_t)f * (2 * kMaxBins), lane, bpl, g, h);
      if (node_feature_on(sp, kRootNodeKey, f, nf) && m.categorical != kCategoricalSorted)
        scan_cells_wave(f, lane, bpl, g, h, m, ctx, sp, bg, bkey, blg, blh, kRootNodeKey);
    }
    wave_best(bg, bkey, blg, blh);
    if (lane == 0) S.wbest[wave] = Cand{bg, bkey, blg, blh};
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kTreeWaves; ++w) take_better(bg, bkey, blg, blh, S.wbest[w].gain, S.wbest[w].key, S.wbest[w].lg, S.wbest[w].lh);
      write_split(&a.best[0], bg, bkey, blg, blh, S.ntot[0][0], S.ntot[0][1], S.ntot[0][2], q, sp);
      const bool open = bg != ninf && isfinite(bg) && true;   // max_depth >= 1 never blocks the root
      S.ngain[0] = open ? bg : ninf;
      const SplitInfo& w = a.best[0];
      S.nbest[0] = (w.feature & 255) | ((w.threshold & 255) << 8) | ((w.default_left & 1) << 16) |
                   ((w.left_cnt <= w.total_cnt - w.left_cnt ? 1 : 0) << 17);
    }
    __syncthreads();
  }

  lap(0);
  // ---- best-first loop ----
  int n_splits = 0;
  for (int k = 0; k < a.budget; ++k) {
    const int n_nodes = 2 * k + 1;
    // every wave finds the open leaf with the largest gain (ties: the smaller node id, the replay's push order)
    double pg = ninf;
    int pn = 0x7fffffff;
    for (int i = lane; i < n_nodes; i += kWave) {
      const double gi = S.ngain[i];
      if (gi > pg) { pg = gi; pn = i; }     // ascending i: the first of equal gains stays
    }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
      const double g2 = shfl_xor_f64(pg, msk);
      const int n2 = __shfl_xor(pn, msk, kWave);
      if (g2 > pg || (g2 == pg && n2 < pn)) { pg = g2; pn = n2; }
    }
    if (!(pg > ninf)) break;
    lap(1);
    const int P = uni(pn), Lc = 2 * k + 1, Rc = 2 * k + 2;
    const int pb = uni(S.nbest[P]);
    const int sf = pb & 255, thr = (pb >> 8) & 255;
    const int dl = uni((int)S.meta[sf].categorical) ? kCategoricalTask : ((pb >> 16) & 1);
    const int mb = uni(S.meta[sf].missing_bin);
    const bool hist_left = ((pb >> 17) & 1) != 0;
    const int begin = uni(S.nbegin[P]), cnt = uni(S.ncount[P]), sb = uni((int)S.nbuf[P]);
    const int pslot = uni((int)S.nslot[P]);
    const int64_t* pblk_base = a.pool + (int64_t)pslot * a.slot_words;
    int64_t* nblk_base = a.pool + (int64_t)(k + 1) * a.slot_words;
    const int32_t* src_idx = a.idx[sb] + begin;
    const unsigned long long* src_gh = a.gh[sb] + begin;
    const uint32_t* src_bins = a.bins[sb] + (int64_t)begin * BW;
    int32_t* dst_idx = a.idx[1 - sb] + begin;
    unsigned long long* dst_gh = a.gh[1 - sb] + begin;
    uint32_t* dst_bins = a.bins[1 - sb] + (int64_t)begin * BW;
    const bool multi = cnt > kMaxRowsPerPass;
    constexpr int U = 4;
    const bool small = cnt <= U * kTreeThreads;   // the leaf is one trip: its records stay in registers until they move

    // The parent's histogram rows come from global memory: the waves that derive the sibling ask for them now and
    // meet them again after the pass over the rows (only while they fit the registers: up to two features per wave).
    constexpr bool kPrefetch = PPW <= 2;
    int64_t pre_g[kPrefetch ? PPW : 1][4], pre_h[kPrefetch ? PPW : 1][4];
    if constexpr (kPrefetch) {
      if (wave & 1) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          const int fe = (wave + kTreeWaves * j) >> 1;
          if (fe < nf && ((enabled >> fe) & 1u))
            load_feat_cells(pblk_base + (int64_t)fe * (2 * kMaxBins), lane, bins_per_lane(S.meta[fe].num_bin), pre_g[j], pre_h[j]);
        }
      }
    }

    // phase A: sides of all rows (ballot words in LDS), histogram of the rows of the smaller child
    sg = 0; sh = 0;
    int rid0[U];
    unsigned long long pk0[U];
    unsigned w0[U][BW];
    // one trip over positions base + u * 512 + tid: all loads first (bins, gradients - and the row ids when the
    // records are kept), then sides and ballots, then the LDS adds
    auto trip = [&](int base, int seg_hi, bool keep, auto& rid, auto& pk, auto& w) {
      constexpr int N = (int)(sizeof(pk) / sizeof(pk[0]));     // positions per thread and trip
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const int pos = base + u * kTreeThreads + tid;
        if (pos < seg_hi) {
          load_words<BW>(src_bins + (int64_t)pos * BW, w[u]);
          pk[u] = src_gh[pos];
          if (keep) rid[u] = src_idx[pos];
        }
      }
      bool take[N];
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const int pos = base + u * kTreeThreads + tid;
        bool left = false;
        if (pos < seg_hi) {
          const int b = (int)bin_byte<BW>(w[u], (unsigned)sf);
          left = dl == kCategoricalTask ? b == thr : (b == mb ? dl != 0 : b <= thr);
        }
        take[u] = pos < seg_hi && left == hist_left;
        if (base + u * kTreeThreads + wave * kWave < seg_hi) {   // wave-uniform
          const unsigned long long m = __ballot(left);
          if (lane == 0) S.smask[(base >> 6) + u * kTreeWaves + wave] = m;
        }
      }
#pragma unroll
      for (int u = 0; u < N; ++u) {
        if (take[u]) {
          sg += (long long)(int32_t)(pk[u] >> 32);
          sh += (long long)(pk[u] & 0xffffffffull);
          hist_add<BW>(lds, w[u], pk[u], nf, f_start, enabled);
        }
      }
    };
    constexpr int UL = 4;   // (8 positions per thread and trip cost more in spilled registers than the loads in flight gain)
    if (small) {
      trip(0, cnt, true, rid0, pk0, w0);
    } else {
      for (int seg = 0; seg < cnt; seg += kMaxRowsPerPass) {
        const int seg_hi = min(cnt, seg + kMaxRowsPerPass);
        for (int base = seg; base < seg_hi; base += UL * kTreeThreads) {
          int rid[UL];
          unsigned long long pk[UL];
          unsigned w[UL][BW];
          trip(base, seg_hi, false, rid, pk, w);
        }
        if (multi) {
          __syncthreads();
          for (int e = tid; e < nf * kMaxBins; e += kTreeThreads) {
            const int fe = e >> 8, b = e & 255;
            const int at = (fe << 8) + ((b + 8 * fe) & 255);
            const unsigned long long v = lds[at];
            lds[at] = 0ull;
            int64_t* blk = nblk_base + (int64_t)fe * (2 * kMaxBins);
            const int64_t vg = (int64_t)(int32_t)(v >> 32), vh = (int64_t)(v & 0xffffffffull);
            if (seg == 0) { blk[b] = vg; blk[kMaxBins + b] = vh; }
            else { blk[b] += vg; blk[kMaxBins + b] += vh; }
          }
          __syncthreads();   // the next pass adds to cells that are zero again
        }
      }
    }
    sg = wave_sum_i64(sg);
    sh = wave_sum_i64(sh);
    if (lane == 0) { S.red[0][wave] = sg; S.red[1][wave] = sh; }
    __syncthreads();
    lap(2);

    // exclusive prefix of the mask words' popcounts.  One trip: at most 32 words - every wave scans them itself and
    // keeps the prefixes of its own four words in registers (no further barrier)
    int lefts;
    int wp0[U];
    if (small) {
      const int n_words = (cnt + kWave - 1) / kWave;
      const int pc = lane < n_words ? __popcll(S.smask[lane]) : 0;
      int incl = pc;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const int o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
      }
      lefts = uni(__shfl(incl, 31, kWave));
#pragma unroll
      for (int u = 0; u < U; ++u) wp0[u] = uni(__shfl(incl - pc, u * kTreeWaves + wave, kWave));
    } else {
      mask_prefix(S, (cnt + kWave - 1) / kWave);
      lefts = uni(S.word_prefix[kMaskWords]);
    }
    // totals of the two children (every thread computes them: no further barrier needed for them)
    long long htg = 0, hth = 0;
#pragma unroll
    for (int w = 0; w < kTreeWaves; ++w) { htg += S.red[0][w]; hth += S.red[1][w]; }
    htg = uni(htg);
    hth = uni(hth);
    const int Hn = hist_left ? Lc : Rc, Sn = hist_left ? Rc : Lc;
    const long long hcnt = hist_left ? lefts : cnt - lefts;
    const long long ptc = uni(S.ntot[P][0]), ptg = uni(S.ntot[P][1]), pth = uni(S.ntot[P][2]);
    const int cdepth = uni((int)S.ndepth[P]) + 1;
    const unsigned pkey = uni(S.nkey[P]);

    // a one-trip leaf moves its records now, out of the registers (the stores need not be waited for, and the
    // registers are free for the search)
    if (small) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pos = u * kTreeThreads + tid;
        if (pos < cnt) {
          const unsigned long long m = S.smask[u * kTreeWaves + wave];
          const bool left = (m >> lane) & 1ull;
          const int lrank = wp0[u] + __popcll(m & ((1ull << lane) - 1ull));
          const int d = left ? lrank : lefts + (pos - lrank);
          dst_idx[d] = rid0[u];
          dst_gh[d] = pk0[u];
          store_words<BW>(dst_bins + (int64_t)d * BW, w0[u]);
        }
      }
    }
    lap(3);
    // search: wave w takes the pairs (feature, child) w, w + 8, ... - child = parity of the wave (0: the child whose
    // histogram was built, 1: its sibling = parent - child, stored over the parent's slot)
    {
      const int c = wave & 1;
      const int node = c == 0 ? Hn : Sn;
      const int side = node == Lc ? 0 : 1;
      const unsigned key = child_node_key(pkey, side);
      const long long tc = c == 0 ? hcnt : ptc - hcnt, tg = c == 0 ? htg : ptg - htg, th = c == 0 ? hth : pth - hth;
      const LeafScanCtx ctx = node_scan_ctx(tg, th, tc, q, sp, node);
      double bg = ninf;
      uint64_t bkey = ~0ull;
      int64_t blg = 0, blh = 0;
      // a child that cannot be split - too few rows for two leaves, or at the depth limit - only needs its histogram
      const bool searched = tc >= 2 * (long long)sp.min_data_in_leaf && !(sp.max_depth > 0 && cdepth >= sp.max_depth);
#pragma unroll 1
      for (int j = 0; j < PPW; ++j) {
        const int pi = wave + kTreeWaves * j;
        const int fe = pi >> 1;
        if (fe < nf) {
          const FeatureMeta m = S.meta[fe];
          if (m.enabled) {
            int64_t g[4], h[4];
            int64_t* nblk = nblk_base + (int64_t)fe * (2 * kMaxBins);
            const int bpl = bins_per_lane(m.num_bin);
            if (multi) load_feat_cells(nblk, lane, bpl, g, h);
            else lds_feat_cells(lds, fe, lane, bpl, g, h);
            if (c == 0) {
              if (!multi) store_feat_cells(nblk, lane, bpl, g, h);
            } else {
              int64_t* pblk = const_cast<int64_t*>(pblk_base) + (int64_t)fe * (2 * kMaxBins);
              if constexpr (kPrefetch) {
                // (j is not a compile-time index here: pick the prefetched row by hand)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int64_t og = (PPW > 1 && j == 1) ? pre_g[PPW > 1 ? 1 : 0][i] : pre_g[0][i];
                  const int64_t oh = (PPW > 1 && j == 1) ? pre_h[PPW > 1 ? 1 : 0][i] : pre_h[0][i];
                  g[i] = og - g[i];
                  h[i] = oh - h[i];
                }
              } else {
                int64_t og[4], oh[4];
                load_feat_cells(pblk, lane, bpl, og, oh);
#pragma unroll
                for (int i = 0; i < 4; ++i) { g[i] = og[i] - g[i]; h[i] = oh[i] - h[i]; }
              }
              store_feat_cells(pblk, lane, bpl, g, h);
            }
            if (searched && node_feature_on(sp, key, fe, nf) && m.categorical != kCategoricalSorted)
              scan_cells_wave(fe, lane, bpl, g, h, m, ctx, sp, bg, bkey, blg, blh, key);
          }
        }
      }
      wave_best(bg, bkey, blg, blh);
      if (lane == 0) S.wbest[wave] = Cand{bg, bkey, blg, blh};
    }

    lap(4);
    // phase B for larger leaves: move the records - lefts to the front, rights behind, in order
    if (!small) {
      for (int base = 0; base < cnt; base += UL * kTreeThreads) {
        int rid[UL];
        unsigned long long pk[UL];
        unsigned w[UL][BW];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int pos = base + u * kTreeThreads + tid;
          if (pos < cnt) {
            rid[u] = src_idx[pos];
            pk[u] = src_gh[pos];
            load_words<BW>(src_bins + (int64_t)pos * BW, w[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int pos = base + u * kTreeThreads + tid;
          if (pos < cnt) {
            const int wi = pos >> 6;
            const unsigned long long m = S.smask[wi];
            const bool left = (m >> lane) & 1ull;
            const int lrank = S.word_prefix[wi] + __popcll(m & ((1ull << lane) - 1ull));
            const int d = left ? lrank : lefts + (pos - lrank);
            dst_idx[d] = rid[u];
            dst_gh[d] = pk[u];
            store_words<BW>(dst_bins + (int64_t)d * BW, w[u]);
          }
        }
      }
    }
    __syncthreads();
    lap(5);
    // the LDS histogram is free again; two threads finish the children
    if (!multi)
      for (int i = tid; i < nf * kMaxBins; i += kTreeThreads) lds[i] = 0ull;
    if (tid == 0 || tid == kWave) {
      const int c = tid == 0 ? 0 : 1;
      const int node = c == 0 ? Hn : Sn;
      double bg = ninf;
      uint64_t bkey = ~0ull;
      int64_t blg = 0, blh = 0;
      for (int w = c; w < kTreeWaves; w += 2) take_better(bg, bkey, blg, blh, S.wbest[w].gain, S.wbest[w].key, S.wbest[w].lg, S.wbest[w].lh);
      const long long tc = c == 0 ? hcnt : ptc - hcnt, tg = c == 0 ? htg : ptg - htg, th = c == 0 ? hth : pth - hth;
      write_split(&a.best[node], bg, bkey, blg, blh, tc, tg, th, q, sp);
      {
        const SplitInfo& w = a.best[node];
        S.nbest[node] = (w.feature & 255) | ((w.threshold & 255) << 8) | ((w.default_left & 1) << 16) |
                        ((w.left_cnt <= w.total_cnt - w.left_cnt ? 1 : 0) << 17);
      }
      S.ntot[node][0] = tc; S.ntot[node][1] = tg; S.ntot[node][2] = th;
      const bool is_left = node == Lc;
      S.nbegin[node] = is_left ? begin : begin + lefts;
      S.ncount[node] = is_left ? lefts : cnt - lefts;
      S.nbuf[node] = (signed char)(1 - sb);
      S.nslot[node] = (short)(c == 0 ? k + 1 : pslot);
      S.ndepth[node] = (short)cdepth;
      S.nkey[node] = child_node_key(pkey, is_left ? 0 : 1);
      const bool open = bg != ninf && isfinite(bg) && !(sp.max_depth > 0 && cdepth >= sp.max_depth);
      S.ngain[node] = open ? bg : ninf;
      S.nparent[node] = (short)P;
      if (c == 0) {
        S.ngain[P] = ninf; S.nsplit[P] = 1; S.split_node[k] = (short)P;
        S.nleft[P] = (short)Lc;
        S.ndec[P] = sf | (thr << 8) | ((dl == kCategoricalTask ? 0 : (dl & 1)) << 16) | ((dl == kCategoricalTask ? 1 : 0) << 17) | ((mb + 1) << 18);
      }
    }
    n_splits = k + 1;
    __syncthreads();
    lap(6);
  }

  return n_splits;
}

// A final leaf's output as the host assembles it (grower.cpp, gbdt.cpp: renew_leaf_values_exact): the Newton step of
// its side of the parent's best split on the quantised sums, replaced - leaf-value renewal - by the step on the
// 36-bit sums of its rows when those hold any hessian; then the learning rate.
__device__ __forceinline__ double leaf_value_of(const SplitInfo& ps, bool is_left, long long wide_g, long long wide_h, bool renew,
                                                int kg, int kh, const SplitParams& sp, double learning_rate) {
  const double gunit = unit_of(kg), hunit = unit_of(kh);
  const double sg = (double)(is_left ? ps.left_g : ps.total_g - ps.left_g) * gunit;
  const double sh = (double)(is_left ? ps.left_h : ps.total_h - ps.left_h) * hunit;
  double v = leaf_output(sg, sh, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
  if (renew) {
    const double G = (double)wide_g * unit_of(kg + kWideExtraBits), H = (double)wide_h * unit_of(kh + kWideExtraBits);
    if (H > 0.0) v = leaf_output(G, H, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
  }
  return v * learning_rate;
}

// Writes a grown tree's result block (everything but the completion flag).  boost: also work out the leaves' outputs
// (S.nvalue) and add them to the training scores of the leaves' rows.  The leaves' rows are visited as ONE flat range
// of arena positions (their ranges tile [0, bag_n)), every thread several positions per trip with all loads of a
// trip in flight - a leaf-by-leaf walk would be a chain of dependent row-id -> gradient loads per leaf.
// scratch: at least 3 * kNodeCap 8-byte words of LDS nobody else uses during the call (the histogram).
template <bool BOOST>
__device__ __forceinline__ void tree_handover(const TreeArgs& a, char* out, int n_splits, int kg, int kh, int bag_n,
                                              const SplitParams& sp, double learning_rate, double* scores, TreeShared& S,
                                              unsigned long long* scratch) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int n_nodes = 2 * n_splits + 1;
  TreeHeader* hd = reinterpret_cast<TreeHeader*>(out);
  int64_t* wide = reinterpret_cast<int64_t*>(out + a.lay.off_wide);
  if (a.renew || BOOST) {
    unsigned long long* acc = scratch;                                   // [node][2]: 36-bit sums
    int* lbegin = reinterpret_cast<int*>(scratch + 2 * kNodeCap);        // non-empty final leaves sorted by range begin
    short* lnode = reinterpret_cast<short*>(lbegin + kNodeCap);
    __shared__ int n_live;
    if (tid == 0) n_live = 0;
    for (int i = tid; i < 2 * n_nodes; i += kTreeThreads) acc[i] = 0ull;
    __syncthreads();
    for (int node = tid; node < n_nodes; node += kTreeThreads) {
      if (S.nsplit[node] || S.ncount[node] == 0) continue;
      const int b = S.nbegin[node];
      int rank = 0;
      for (int o = 0; o < n_nodes; ++o) rank += (!S.nsplit[o] && S.ncount[o] > 0 && S.nbegin[o] < b) ? 1 : 0;
      lbegin[rank] = b;
      lnode[rank] = (short)node;
      atomicAdd(&n_live, 1);
    }
    __syncthreads();
    const int live = n_live;
    auto leaf_of = [&](int p) {      // the last live leaf whose range begins at or before p
      int lo = 0, hi = live - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (lbegin[mid] <= p) lo = mid; else hi = mid - 1;
      }
      return (int)lnode[lo];
    };
    constexpr int U = 4;
    if (a.renew) {
      const int wkg = kg + kWideExtraBits, wkh = kh + kWideExtraBits;
      for (int p0 = tid; p0 < bag_n; p0 += U * kTreeThreads) {
        int node[U], r[U];
        float gv[U], hv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = p0 + u * kTreeThreads;
          node[u] = p < bag_n ? leaf_of(p) : -1;
          if (node[u] >= 0) r[u] = a.idx[S.nbuf[node[u]]][p];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (node[u] >= 0) { gv[u] = a.g[r[u]]; hv[u] = a.h[r[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // a wave's 64 positions mostly lie in one leaf: one add per wave then
          long long wg = node[u] >= 0 ? quantize_wide(gv[u], wkg) : 0, wh = node[u] >= 0 ? quantize_wide(hv[u], wkh) : 0;
          const int first = __builtin_amdgcn_readfirstlane(node[u]);
          if (__all(node[u] == first)) {
            if (first >= 0) {
              wg = wave_sum_i64(wg);
              wh = wave_sum_i64(wh);
              if (lane == 0) { atomicAdd(&acc[2 * first], (unsigned long long)wg); atomicAdd(&acc[2 * first + 1], (unsigned long long)wh); }
            }
          } else if (node[u] >= 0) {
            atomicAdd(&acc[2 * node[u]], (unsigned long long)wg);
            atomicAdd(&acc[2 * node[u] + 1], (unsigned long long)wh);
          }
        }
      }
      __syncthreads();
    }
    for (int node = tid; node < n_nodes; node += kTreeThreads) {
      if (S.nsplit[node]) continue;
      const long long wg = (long long)acc[2 * node], wh = (long long)acc[2 * node + 1];
      if (a.renew) { wide[2 * node] = wg; wide[2 * node + 1] = wh; }
      if (BOOST && n_splits > 0) {
        const int p = S.nparent[node];
        S.nvalue[node] = leaf_value_of(a.best[p], S.nleft[p] == node, wg, wh, a.renew != 0, kg, kh, sp, learning_rate);
      }
    }
    if (BOOST && n_splits > 0) {
      __syncthreads();
      for (int p0 = tid; p0 < bag_n; p0 += U * kTreeThreads) {
        int node[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = p0 + u * kTreeThreads;
          node[u] = p < bag_n ? leaf_of(p) : -1;
          if (node[u] >= 0) r[u] = a.idx[S.nbuf[node[u]]][p];
        }
        double sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (node[u] >= 0) sc[u] = scores[r[u]];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (node[u] >= 0) scores[r[u]] = sc[u] + S.nvalue[node[u]];
      }
    }
  }
  {
    const int words = n_nodes * (int)(sizeof(SplitInfo) / 8);
    const int64_t* res = reinterpret_cast<const int64_t*>(a.best);
    int64_t* dst = reinterpret_cast<int64_t*>(out + a.lay.off_best);
    NodeRange* ranges = reinterpret_cast<NodeRange*>(out + a.lay.off_range);
    int32_t* splits = reinterpret_cast<int32_t*>(out + a.lay.off_split);
    for (int i = tid; i < words; i += kTreeThreads) dst[i] = res[i];
    for (int i = tid; i < n_nodes; i += kTreeThreads) {
      const NodeRange r{S.nbegin[i], S.ncount[i], (int32_t)S.nbuf[i], 0};
      ranges[i] = r;
      a.nodes[i] = r;
    }
    for (int i = tid; i < n_splits; i += kTreeThreads) splits[i] = S.split_node[i];
    if (tid == 0) { hd->n_splits = n_splits; hd->kg = kg; hd->kh = kh; hd->bag_n = bag_n; }
  }
}

// Adds the tree in S to the scores of rows[first .. n) (rows == nullptr: the rows themselves) by walking it over a
// feature-major bin matrix; four rows per thread walk together so that their bin reads overlap.
__device__ __forceinline__ void walk_add(const TreeShared& S, const uint8_t* __restrict__ bins_fm, int64_t N,
                                         const int32_t* __restrict__ rows, int first, int n, double* __restrict__ scores) {
  constexpr int U = 4;
  for (int i0 = first + (int)threadIdx.x; i0 < n; i0 += U * kTreeThreads) {
    int64_t r[U];
    int node[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kTreeThreads;
      r[u] = i < n ? (rows ? (int64_t)rows[i] : (int64_t)i) : -1;
      node[u] = 0;
    }
    bool any = true;
    while (any) {
      any = false;
      int b[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        d[u] = (r[u] >= 0 && S.nsplit[node[u]]) ? S.ndec[node[u]] : -1;
        if (d[u] >= 0) b[u] = bins_fm[(int64_t)(d[u] & 255) * N + r[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (d[u] >= 0) {
          const int thr = (d[u] >> 8) & 255, mb = (d[u] >> 18) - 1;
          const bool left = (d[u] & (1 << 17)) ? (b[u] == thr) : ((b[u] == mb) ? ((d[u] >> 16) & 1) != 0 : (b[u] <= thr));
          node[u] = left ? S.nleft[node[u]] : S.nleft[node[u]] + 1;
          any = true;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r[u] >= 0) scores[r[u]] += S.nvalue[node[u]];
  }
}

// One tree per workgroup.
template <int BW>
__device__ __forceinline__ void tree_one(const TreeArgs& a, unsigned long long* lds, TreeShared& S) {
  const int tid = threadIdx.x;
  const int kg = quant_exponent(__uint_as_float(a.maxbits[0]));
  const int kh = quant_exponent(__uint_as_float(a.maxbits[1]));
  if (tid < 8) S.ticks[tid] = 0;
  const long long c0 = a.timing ? (long long)clock64() : 0;
  const int n_splits = grow_one<BW>(a, a.bag, a.n, ~0u, a.sp, kg, kh, lds, S);
  tree_handover<false>(a, a.out, n_splits, kg, kh, a.n, a.sp, 1.0, nullptr, S, lds);
  TreeHeader* hd = reinterpret_cast<TreeHeader*>(a.out);
  if (a.timing && tid == 0) {
    S.ticks[7] += (long long)wall_clock64() - S.t_mark;
    for (int i = 0; i < 8; ++i) hd->phase_ticks[i] = S.ticks[i];
    hd->shader_cycles = (long long)clock64() - c0;
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) *reinterpret_cast<volatile unsigned*>(&hd->flag) = a.seq;
}

// A chunk of boosting iterations of one booster (kernels.hpp: BoostArgs).
template <int BW>
__device__ __forceinline__ void boost_one(const BoostArgs& b, unsigned long long* lds, TreeShared& S) {
  const TreeArgs& a = b.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_base = b.n_base;
  const int32_t* bag = a.bag;
  if (tid == 0) S.bag_n = a.n;
  __syncthreads();
  for (int it = 0; it < b.n_iters; ++it) {
    const BoostIter& I = b.iters[it];
    char* out = a.out + (int64_t)it * a.lay.bytes;
    if (tid < 8) S.ticks[tid] = 0;
    // ---- gradients of the objective at the current scores, and their maxima ----
    {
      float mg = 0.f, mh = 0.f;
      constexpr int U = 4;
      for (int i0 = tid; i0 < n_base; i0 += U * kTreeThreads) {
        int r[U];
        double sc[U];
        float lb[U], wt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kTreeThreads;
          r[u] = i < n_base ? b.base_rows[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (r[u] >= 0) { sc[u] = b.scores[r[u]]; lb[u] = b.label[r[u]]; wt[u] = b.weight ? b.weight[r[u]] : 1.0f; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (r[u] >= 0) {
            float gg, hh;
            grad_hess(b.obj, sc[u], lb[u], wt[u], &gg, &hh);
            b.g[r[u]] = gg;
            b.h[r[u]] = hh;
            mg = fmaxf(mg, fabsf(gg));
            mh = fmaxf(mh, hh);
          }
        }
      }
      mg = wave_max_f32(mg);
      mh = wave_max_f32(mh);
      if (lane == 0) { S.fmax[0][wave] = mg; S.fmax[1][wave] = mh; }
    }
    // ---- the bag of this iteration ----
    if (b.bagging && I.redraw) {
      // keep bits of the training rows in order, then a stable split: the kept rows first
      for (int base = 0; base < n_base; base += kTreeThreads) {
        const int i = base + tid;
        bool keep = false;
        if (i < n_base) {
          const int32_t row = b.base_rows[i];
          const float frac = (b.bagging == 3 && !(b.label[row] > 0.0f)) ? b.neg_fraction : b.bag_fraction;
          keep = rng_uniform(b.bag_seed, I.bag_iter, (uint64_t)(uint32_t)row) < frac;
        }
        if (base + wave * kWave < n_base) {
          const unsigned long long m = __ballot(keep);
          if (lane == 0) S.smask[(base >> 6) + wave] = m;
        }
      }
      __syncthreads();
      mask_prefix(S, (n_base + kWave - 1) / kWave);
      const int kept = S.word_prefix[kMaskWords];
      for (int i = tid; i < n_base; i += kTreeThreads) {
        const int wi = i >> 6;
        const unsigned long long m = S.smask[wi];
        const bool keep = (m >> lane) & 1ull;
        const int rank = S.word_prefix[wi] + __popcll(m & ((1ull << lane) - 1ull));
        b.bag_rows[keep ? rank : kept + (i - rank)] = b.base_rows[i];
      }
      if (tid == 0) S.bag_n = kept;
      bag = b.bag_rows;
    }
    __syncthreads();
    float vmg = 0.f, vmh = 0.f;
#pragma unroll
    for (int w = 0; w < kTreeWaves; ++w) { vmg = fmaxf(vmg, S.fmax[0][w]); vmh = fmaxf(vmh, S.fmax[1][w]); }
    if (tid == 0) { b.maxbits[0] = __float_as_uint(vmg); b.maxbits[1] = __float_as_uint(vmh); }
    const int kg = uni(quant_exponent(vmg)), kh = uni(quant_exponent(vmh));
    const int bag_n = uni(S.bag_n);
    // ---- the tree ----
    const SplitParams& sp = I.sp;
    int n_splits = 0;
    if (bag_n > 0) n_splits = grow_one<BW>(a, bag, bag_n, I.enabled, sp, kg, kh, lds, S);
    else {
      // an empty bag: a tree without a split over no rows
      if (tid == 0) {
        S.nsplit[0] = 0; S.nbegin[0] = 0; S.ncount[0] = 0; S.nbuf[0] = 0;
        write_split(&a.best[0], -std::numeric_limits<double>::infinity(), ~0ull, 0, 0, 0, 0, 0, QuantScale{kg, kh}, sp);
      }
      __syncthreads();
    }
    // ---- leaf outputs, in-bag scores, result block ----
    tree_handover<true>(a, out, n_splits, kg, kh, bag_n, sp, b.learning_rate, b.scores, S, lds);
    __syncthreads();
    TreeHeader* hd = reinterpret_cast<TreeHeader*>(out);
    if (n_splits > 0) {
      // out-of-bag training rows and the validation rows walk the tree
      walk_add(S, b.bins_fm, b.N, bag, bag_n, n_base, b.scores);
      walk_add(S, b.valid_bins_fm, b.valid_N, b.valid_rows, 0, b.n_valid, b.valid_scores);
    }
    __syncthreads();
    // ---- metric sums over the validation rows, in the order of metric_sum_kernel: virtual workgroups of 256
    // threads (one element each here), a wave-butterfly and a four-wave sum per workgroup, the workgroups' partial
    // sums lane-strided over 64 lanes, one more butterfly ----
    if (b.n_valid > 0 && b.n_metrics > 0) {
      double* part = reinterpret_cast<double*>(lds);      // [virtual workgroup][2]; the histogram is not in use here
      const int n = b.n_valid;
      const int B = (n + 255) / 256;                       // <= 256: n_valid <= kTreeMaxRows
      for (int m = 0; m < b.n_metrics; ++m) {
        for (int vb = wave; vb < B; vb += kTreeWaves) {
          double t0 = 0.0, t1 = 0.0;
#pragma unroll
          for (int vw = 0; vw < 4; ++vw) {
            const int i = vb * 256 + vw * 64 + lane;
            double v0 = 0.0, v1 = 0.0;
            if (i < n) {
              const int64_t r = b.valid_rows ? (int64_t)b.valid_rows[i] : (int64_t)i;
              const double w = b.valid_weight ? (double)b.valid_weight[r] : 1.0;
              const double raw = b.valid_scores[r];
              const double pred = transform_score(b.obj.kind, b.obj.sigmoid, raw);
              v0 += w * point_loss(b.metrics[m], pred, raw, b.valid_label[r], b.obj);
              v1 += w;
            }
            t0 += wave_sum_f64(v0);
            t1 += wave_sum_f64(v1);
          }
          if (lane == 0) { part[2 * vb] = t0; part[2 * vb + 1] = t1; }
        }
        __syncthreads();
        if (wave == 0) {
          double a0 = 0.0, a1 = 0.0;
          for (int vb = lane; vb < B; vb += 64) { a0 += part[2 * vb]; a1 += part[2 * vb + 1]; }
          a0 = wave_sum_f64(a0);
          a1 = wave_sum_f64(a1);
          if (lane == 0) { hd->metric[m][0] = a0; hd->metric[m][1] = a1; }
        }
        __syncthreads();
      }
    }
    if (a.timing && tid == 0) {
      S.ticks[7] += (long long)wall_clock64() - S.t_mark;    // everything around the tree: gradients, bag, scores, metrics
      for (int i = 0; i < 8; ++i) hd->phase_ticks[i] = S.ticks[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) *reinterpret_cast<volatile unsigned*>(&reinterpret_cast<TreeHeader*>(a.out)->flag) = a.seq;
}

template <int BW>
__global__ __launch_bounds__(kTreeThreads) void boost_kernel(const BoostArgs* __restrict__ descs) {
  __shared__ unsigned long long lds[BW * 4 * kMaxBins];
  __shared__ TreeShared S;
  boost_one<BW>(descs[blockIdx.x], lds, S);
}

template <int BW>
__global__ __launch_bounds__(kTreeThreads) void tree_kernel(const TreeArgs* __restrict__ descs) {
  __shared__ unsigned long long lds[BW * 4 * kMaxBins];
  __shared__ TreeShared S;
  tree_one<BW>(descs[blockIdx.x], lds, S);   // a uniform address: the descriptor's fields arrive through scalar loads
}

}  // namespace

void grow_trees(const TreeArgs* descs, int n_trees, int max_features, hipStream_t s) {
  if (n_trees <= 0) return;
  if (max_features <= 8) hipLaunchKernelGGL(tree_kernel<2>, dim3(n_trees), dim3(kTreeThreads), 0, s, descs);
  else if (max_features <= 16) hipLaunchKernelGGL(tree_kernel<4>, dim3(n_trees), dim3(kTreeThreads), 0, s, descs);
  else hipLaunchKernelGGL(tree_kernel<8>, dim3(n_trees), dim3(kTreeThreads), 0, s, descs);
  LGA_HIP_CHECK(hipGetLastError());
}

void boost_chunks(const BoostArgs* descs, int n_boosters, int max_features, hipStream_t s) {
  if (n_boosters <= 0) return;
  if (max_features <= 8) hipLaunchKernelGGL(boost_kernel<2>, dim3(n_boosters), dim3(kTreeThreads), 0, s, descs);
  else if (max_features <= 16) hipLaunchKernelGGL(boost_kernel<4>, dim3(n_boosters), dim3(kTreeThreads), 0, s, descs);
  else hipLaunchKernelGGL(boost_kernel<8>, dim3(n_boosters), dim3(kTreeThreads), 0, s, descs);
  LGA_HIP_CHECK(hipGetLastError());
}

}  // namespace k
}  // namespace lga
