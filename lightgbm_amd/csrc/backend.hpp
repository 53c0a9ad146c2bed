// Interface between the host-side tree grower and whatever executes the
// row-level work (the OpenMP host backend or the HIP backend).  A backend
// works in ROUNDS: every call splits a whole batch of leaves - partition rows,
// build the smaller children's histograms, derive the siblings by
// subtraction, scan all children for their best split - so that the number
// of host<->device synchronisations per tree is the number of rounds
// (about the tree depth) and not the number of leaves.
#pragma once
#include "common.hpp"
#include <cstddef>
#include <vector>

namespace lga {

// Collective operations used by the distributed learners.  The data-parallel
// learner sums integer histograms (exact, order independent); the
// feature-parallel learner gathers per-rank split candidates.
class Comm {
 public:
  virtual ~Comm() {}
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // true: the communicator only takes host buffers (device engines then stage through pinned memory)
  virtual bool host_only() const { return false; }
  // In-place sum over ranks.  `device` tells whether buf is device memory.
  virtual void allreduce_sum_i64(int64_t* buf, size_t n, bool device, void* stream) = 0;
  virtual void allreduce_sum_f64(double* buf, size_t n, bool device, void* stream) = 0;
  virtual void allreduce_max_f32(float* buf, size_t n, bool device, void* stream) = 0;
  // In-place reduce-scatter of size() blocks of `block` words each: on return rank r's block
  // (buf + r * block) holds the sum over ranks; the other blocks are unspecified.  The default sums everything.
  virtual void reduce_scatter_sum_i64(int64_t* buf, size_t block, bool device, void* stream) {
    allreduce_sum_i64(buf, block * (size_t)size(), device, stream);
  }
  // out must hold size()*nbytes; rank r's block lands at out + r*nbytes.
  virtual void allgather_bytes(const void* in, void* out, size_t nbytes, bool device, void* stream) = 0;
};

struct LeafOutput {
  int32_t node;    // backend node id of a final leaf
  int32_t pad;
  double value;    // already multiplied by the learning rate
};

class TreeBackend {
 public:
  virtual ~TreeBackend() {}
  // Node/slot capacity the grower may use for one tree.
  virtual int node_capacity() const = 0;
  // Start a tree over the current bag with the current quantised gradients.
  // meta[f].enabled carries this tree's feature_fraction mask.
  virtual void begin_tree(const std::vector<FeatureMeta>& meta, const SplitParams& params) = 0;
  // Histogram + scan of the root (node 0, slot 0).
  virtual void root_round(SplitInfo* out) = 0;
  // Split every task's leaf; infos[2i] / infos[2i+1] are the best splits of
  // task i's left / right child.  The backend fills the chunk fields of the tasks.
  virtual void split_round(std::vector<SplitTask>& tasks, SplitInfo* infos) = 0;
  virtual QuantScale quant_scale() const = 0;
  // The tree's last round has been waited for; nothing of the tree's histograms is read any more (the host goes on to
  // assemble the tree: a backend may use the time).
  virtual void tree_rounds_done() {}
  // Monotone constraints / path smoothing: the output range of a node and the node's own output, announced
  // before the round that creates the node is submitted (node 0 is unbounded, its output the plain optimum).
  // Never called when neither is active.
  virtual void set_node_bounds(int node, double lo, double hi, double out) = 0;
  // Interaction constraints: the features (bits over inner features, mask_words(F) words) that may split a
  // node; node 0's mask is announced between begin_tree and root_round.  Never called without constraints.
  virtual void set_node_mask(int node, const uint64_t* words) = 0;
  // Category-set splits (SplitInfo::default_left == kCategoricalSetTask): the bins (kCatSetWords words) that
  // go left in the best split reported for `node`.
  virtual void node_cat_set(int node, uint64_t* words) const = 0;
  // Local row count of a node (after the round that created it).
  virtual int64_t node_local_count(int node) const = 0;

  // Optional: the backend grows the WHOLE tree by itself, strictly best-first (ties between equal gains go to the
  // node created first), after begin_tree.  whole_tree_begin prepares that and says whether this tree can be
  // grown so (false: grow it in rounds); whole_tree_launch starts the build unless a batched launch already did;
  // whole_tree_finish waits and reports, for the k-th split in order, the node that was split - its children are
  // nodes 2k+1 and 2k+2 - and every node's best split.  The node table then holds the row ranges.
  struct WholeTree {
    std::vector<int32_t> split_node;
    std::vector<SplitInfo> best;   // 2 * splits + 1 entries
  };
  virtual bool whole_tree_begin(int max_splits) { (void)max_splits; return false; }
  virtual void whole_tree_launch() {}
  virtual void whole_tree_finish(WholeTree* out) { (void)out; }

  // Asynchronous form: up to max_in_flight() rounds may be submitted before the first one is waited
  // for, so one round's latency-bound tail (split scan, result hand-over, host replay) overlaps with
  // another round's row-level work.  Rounds in flight never share a leaf; a round may depend on the
  // results of any round that has been waited for.  The default runs every round at submission.
  virtual int max_in_flight() const { return 1; }
  virtual int submit_round(std::vector<SplitTask>& tasks) {
    int ticket = 0;
    while (ticket < (int)held_.size() && held_[(size_t)ticket].busy) ++ticket;
    if (ticket == (int)held_.size()) held_.emplace_back();
    Held& h = held_[(size_t)ticket];
    h.busy = true;
    h.infos.assign(tasks.size() * 2, SplitInfo{});
    split_round(tasks, h.infos.data());
    return ticket;
  }
  // Has the round finished?  The grower takes finished rounds first; a backend whose rounds must be replayed in
  // submission order (ranks that have to make the same decisions in the same order) always says yes.
  virtual bool round_done(int ticket) { (void)ticket; return true; }
  virtual void wait_round(int ticket, SplitInfo* infos) {
    Held& h = held_[(size_t)ticket];
    for (size_t i = 0; i < h.infos.size(); ++i) infos[i] = h.infos[i];
    h.busy = false;
  }

 private:
  struct Held {
    bool busy = false;
    std::vector<SplitInfo> infos;
  };
  std::vector<Held> held_;
};

}  // namespace lga
