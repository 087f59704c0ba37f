// bs_drain.cpp — the batched scheduling cycle, driven natively above the C ABI: score the whole queue, release the first
// ready gang, assume its pods, patch the resident state, score again — until no gang is ready.
//
// What one turn of the loop stands for in the reference: the scheduling queue hands pods to PreFilter one at a time
// (core.go:88-167); a pod that passes is placed by the default scheduler, assumed on its node, and waits in Permit
// (core.go:268-309) until its gang reaches the quorum of core.go:303; then StartBatchSchedule (batchscheduler.go:254-344)
// releases the gang and PostBind (core.go:312-362) counts its pods into Status.Scheduled.  bs_batch_run answers PreFilter
// (and Filter) for EVERY pending pod against a frozen snapshot, so `group_ready` is a pre-screen, not a reservation
// (DESIGN.md section 2): after a gang is released the node requests and group counters move, and the other gangs have to be
// scored again.  This driver does exactly that at GANG granularity with the queue, the nodes and the groups all resident on
// the device: bs_nodes_assume + bs_groups_apply + bs_pods_apply (three small launches, nothing re-uploaded) + bs_batch_run.
//
// Node choice is not the plugin's business (upstream's predicates / priorities pick among the nodes Filter lets through); the
// driver states its own rule (the CPU replay the tests compare it with restates the same one):
//   FIRST FIT in list order over nodes that are schedulable (no BS_NODE_* flag), fit the pod's class (checkFit bit), pass the
//   plugin's Filter when the Filter stage is on, and hold the request: lane j in {cpu, mem, eph} binds when the pod asks for it
//   (request > 0: request <= allocatable - requested), the pods lane always (requested + 1 <= allocatable), a scalar the pod
//   asks for needs the allocatable key and request <= allocatable - requested.
//   Assume (NodeInfo.AddPod ‡): requested += request on every lane the pod has, pods lane + 1.
// ALL OR NOTHING per gang: a ready gang is released only when EVERY member of it that passes in this batch finds a node; a gang
// whose pods cannot all be placed is rolled back and skipped for the rest of the drain ("stuck").  That is this driver's rule, not
// the reference's: core.go:303 turns true as soon as matched reaches MinMember - Scheduled, so the reference (and bs_seq_run, the
// pod-by-pod pass on the device) releases an over-subscribed gang — more pending pods than its quorum — when the quorum's worth
// of pods is placed and leaves the rest pending, and lets partial gangs hold what they assumed.  Use bs_seq_run for the
// reference's semantics; this loop is the pre-screen-and-recheck form (tests/test_drain.py pins it gang by gang against
// tests/drain_ref.py, which states the same rule, including over-subscribed gangs).
// A failing device call leaves the caller's arrays as they were before the cycle: every placement of a cycle is journalled and
// undone, and the group counters are only written after the three patches went through.
// No arithmetic of the hot path happens here: every PreFilter / Filter / quorum answer comes from bs_batch_run.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/bsched.h"

extern "C" {

typedef struct bsh_drain_io {
  bs_ctx* ctx;                 /* nodes, fit, groups and pods loaded; single rank                                    */
  uint32_t lanes;              /* L = 4 + scalar lanes of the context                                                 */
  /* host copies of the loaded state (what the shim marshalled): node requests and group counters are UPDATED in place */
  uint32_t n;
  const int64_t* allocatable;  /* [L][n] */
  int64_t* requested;          /* [L][n] */
  const uint32_t* allocatable_present;
  uint32_t* requested_present;
  const uint8_t* node_flags;
  const uint32_t* fit_bits;    /* [n_classes][ceil(n/32)] */
  uint32_t n_classes;
  uint32_t g;
  const uint32_t* min_member;
  uint32_t* status_scheduled;
  uint32_t* matched;
  uint8_t* group_flags;
  bs_pods_soa pods;            /* the loaded queue */
  uint32_t stages;             /* BS_STAGE_PREFILTER | BS_STAGE_TALLY [| BS_STAGE_FILTER] [| BS_BATCH_HOST_RESULTS]  */
  uint32_t max_cycles;         /* 0 = until nothing is ready */
  /* results */
  uint32_t cap;                /* capacity of the three per-gang arrays                                              */
  uint32_t* admitted_group;    /* [cap] gang released k-th                                                            */
  uint32_t* admitted_pods;     /* [cap] pods released with it (its members placed in that cycle + the pods that were waiting) */
  int64_t* admitted_ns;        /* [cap] time since the drain began when it was released                               */
  int64_t* cycle_ns;           /* [cap] duration of the cycle that decided it (score + read + place + patch)          */
  int32_t* pod_node;           /* [pods.p] node each pod of the ORIGINAL queue was assumed on, -1 = still pending     */
  uint32_t n_admitted, n_cycles, n_stuck, pods_left;
  int64_t total_ns;
} bsh_drain_io;

int bsh_drain(bsh_drain_io* io);

}  // extern "C"

namespace {

inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Placer {
  const bsh_drain_io& io;
  uint32_t L, S, N;
  std::vector<int64_t> bmax;   // per 64-node block: max cpu left, max mem left over eligible nodes (skip blocks that cannot hold the pod)
  explicit Placer(const bsh_drain_io& i) : io(i), L(i.lanes), S(i.lanes - 4), N(i.n), bmax((size_t)2 * ((i.n + 63) / 64)) {
    for (uint32_t b = 0; b < (N + 63) / 64; ++b) refresh(b);
  }
  int64_t left(uint32_t j, uint32_t k) const { return (int64_t)((uint64_t)io.allocatable[(size_t)j * N + k] - (uint64_t)io.requested[(size_t)j * N + k]); }
  void refresh(uint32_t b) {
    int64_t c = INT64_MIN, m = INT64_MIN;
    for (uint32_t k = b * 64; k < std::min(N, b * 64 + 64); ++k) {
      if (io.node_flags[k]) continue;
      c = std::max(c, left(0, k));
      m = std::max(m, left(1, k));
    }
    bmax[2 * b] = c;
    bmax[2 * b + 1] = m;
  }
  bool holds(uint32_t k, const int64_t* req, uint32_t pres) const {
    for (uint32_t j = 0; j < 3; ++j)
      if (req[j] > 0 && req[j] > left(j, k)) return false;
    if (io.requested[(size_t)3 * N + k] + 1 > io.allocatable[(size_t)3 * N + k]) return false;
    for (uint32_t s = 0; s < S; ++s) {
      if (!((pres >> s) & 1u) || req[4 + s] <= 0) continue;
      if (!((io.allocatable_present[k] >> s) & 1u)) return false;
      const int64_t rq = ((io.requested_present[k] >> s) & 1u) ? io.requested[(size_t)(4 + s) * N + k] : 0;
      if (req[4 + s] > io.allocatable[(size_t)(4 + s) * N + k] - rq) return false;
    }
    return true;
  }
  void assume(uint32_t k, const int64_t* req, uint32_t pres, int sign) {
    for (uint32_t j = 0; j < 3; ++j) io.requested[(size_t)j * N + k] += sign * req[j];
    io.requested[(size_t)3 * N + k] += sign;
    for (uint32_t s = 0; s < S; ++s)
      if ((pres >> s) & 1u) {
        if (!((io.requested_present[k] >> s) & 1u)) { io.requested[(size_t)(4 + s) * N + k] = 0; }
        io.requested[(size_t)(4 + s) * N + k] += sign * req[4 + s];
        if (sign > 0) io.requested_present[k] |= 1u << s;
      }
    refresh(k / 64);
  }
};

}  // namespace

int bsh_drain(bsh_drain_io* io) {
  if (!io || !io->ctx || io->lanes < 4 || io->lanes > BS_MAX_LANES) return BS_ERR_INVALID;
  const uint32_t L = io->lanes, N = io->n, G = io->g, P0 = io->pods.p, W = (N + 63) / 64;
  const bool filter = io->stages & BS_STAGE_FILTER;
  const uint32_t fw = (N + 31) / 32;
  bs_ctx* ctx = io->ctx;
  io->n_admitted = io->n_cycles = io->n_stuck = 0;
  for (uint32_t i = 0; i < P0; ++i) io->pod_node[i] = -1;
  // the queue as the device holds it: current position -> pod of the original queue
  std::vector<uint32_t> orig(P0);
  for (uint32_t i = 0; i < P0; ++i) orig[i] = i;
  std::vector<uint8_t> pf_code(std::max<uint32_t>(P0, 1)), fl_code(std::max<uint32_t>(P0, 1)), ready(std::max<uint32_t>(G, 1)), stuck(std::max<uint32_t>(G, 1), 0);
  std::vector<uint32_t> feasible(std::max<uint32_t>(P0, 1)), fl_slot(std::max<uint32_t>(P0, 1)), members, placed_node, rows_n(1);
  struct Placed { uint32_t i, node, old_rp; };
  std::vector<Placed> journal;                             // every placement of the current cycle, in the order it was made
  std::vector<uint64_t> rows;
  Placer pl(*io);
  uint32_t rows_cap = 0;
  const int64_t t_begin = now_ns();
  int rc = BS_OK;
  for (;;) {
    if (io->max_cycles && io->n_cycles >= io->max_cycles) break;
    const int64_t t_cycle = now_ns();
    const uint32_t P = (uint32_t)orig.size();
    if (!P) break;
    // ---- score the whole queue against the frozen state
    if ((rc = bs_batch_run(ctx, io->stages))) return rc;
    bs_batch_out out;
    std::memset(&out, 0, sizeof(out));
    out.pf_code = pf_code.data();
    out.group_ready = ready.data();
    if (filter) {
      uint32_t need = 0;
      if ((rc = bs_filter_rows_count(ctx, &need))) return rc;
      if (need > rows_cap) { rows_cap = need + need / 2 + 64; rows.assign((size_t)W * rows_cap, 0); }
      out.fl_code = fl_code.data();
      out.fl_feasible = feasible.data();
      out.fl_slot = fl_slot.data();
      out.fl_rows = rows.data();
      out.fl_rows_cap = rows_cap;
      out.fl_rows_n = rows_n.data();
    }
    if ((rc = bs_batch_read(ctx, &out))) return rc;
    io->n_cycles++;
    auto passes = [&](uint32_t i) { return BS_PF_IS_PASS(pf_code[i]) && (!filter || feasible[i] > 0); };
    auto filter_ok = [&](uint32_t i, uint32_t k) {
      if (!filter) return true;
      if (fl_code[i] != BS_FL_EVALUATED) return fl_code[i] < 16u;
      return (bool)((rows[(size_t)(k >> 6) * rows_cap + fl_slot[i]] >> (k & 63u)) & 1ull);
    };
    // ---- walk the queue: pods without a PodGroup label that pass (core.go:89-92) are placed as they are met — Permit lets
    // them through at once (:269-272) —, then the first ready gang whose pods can all be placed; everything is committed to the
    // device together.  When no gang is ready the remaining unlabelled pods still go through.
    auto place_one = [&](uint32_t i, uint32_t* node_out) -> bool {
      const uint32_t o = orig[i];
      int64_t req[BS_MAX_LANES];
      for (uint32_t j = 0; j < L; ++j) req[j] = io->pods.req[(size_t)j * P0 + o];
      const uint32_t pres = io->pods.req_present[o], cls = io->pods.cls[o];
      for (uint32_t b = 0; b < W; ++b) {
        if ((req[0] > 0 && pl.bmax[2 * b] < req[0]) || (req[1] > 0 && pl.bmax[2 * b + 1] < req[1])) continue;
        for (uint32_t k = b * 64; k < std::min(N, b * 64 + 64); ++k) {
          if (io->node_flags[k]) continue;
          if (cls >= io->n_classes || !((io->fit_bits[(size_t)cls * fw + (k >> 5)] >> (k & 31u)) & 1u)) continue;
          if (!filter_ok(i, k) || !pl.holds(k, req, pres)) continue;
          journal.push_back({i, k, io->requested_present[k]});
          pl.assume(k, req, pres, +1);
          *node_out = k;
          return true;
        }
      }
      return false;
    };
    auto unplace = [&](uint32_t i, uint32_t node, uint32_t old_rp) {
      const uint32_t o = orig[i];
      int64_t req[BS_MAX_LANES];
      for (uint32_t j = 0; j < L; ++j) req[j] = io->pods.req[(size_t)j * P0 + o];
      pl.assume(node, req, io->pods.req_present[o], -1);
      io->requested_present[node] = old_rp;
    };
    std::vector<uint32_t> gone, gone_node;                  // queue positions leaving in this cycle (ascending), and their nodes
    int32_t gang = -1;
    uint32_t gang_pods = 0, gang_released = 0;   // members placed this cycle; pods released with the gang (members + the ones already waiting)
    journal.clear();
    // undo of everything this cycle placed, should a device call fail
    auto undo_cycle = [&]() {
      for (size_t m = journal.size(); m-- > 0;) unplace(journal[m].i, journal[m].node, journal[m].old_rp);
      journal.clear();
    };
    for (uint32_t i0 = 0; i0 < P && gang < 0; ++i0) {
      const int32_t gi = io->pods.group[orig[i0]];
      if (gi == BS_POD_NOT_GROUPED) {
        uint32_t at;
        if (passes(i0) && place_one(i0, &at)) { gone.push_back(i0); gone_node.push_back(at); }
        continue;
      }
      if (gi < 0 || (uint32_t)gi >= G || !ready[gi] || stuck[gi] || !passes(i0)) continue;
      if (io->group_flags[gi] & BS_GROUP_PHASE_CLOSED) continue;   // batchscheduler.go:258-261: StartBatchSchedule releases nobody in this phase
      members.clear();
      placed_node.clear();
      const size_t mark = journal.size();
      for (uint32_t i = i0; i < P; ++i)
        if (io->pods.group[orig[i]] == gi && passes(i)) members.push_back(i);
      bool ok = true;
      for (uint32_t m = 0; m < members.size() && ok; ++m) {
        uint32_t at;
        if (!place_one(members[m], &at)) { ok = false; break; }
        placed_node.push_back(at);
      }
      if (!ok) {                                             // roll the partial gang back: it holds nothing
        for (size_t m = journal.size(); m-- > mark;) unplace(journal[m].i, journal[m].node, journal[m].old_rp);
        journal.resize(mark);
        stuck[gi] = 1;
        io->n_stuck++;
        continue;
      }
      gang = gi;
      gang_pods = (uint32_t)members.size();
    }
    if (gang < 0 && gone.empty()) break;
    // ---- release: Permit for every member (core.go:290), quorum latch (:305), PostBind (:327), then patch the device.
    // Order of the three patches: node requests first (its own launch), then the group patch — whose launch is deferred — and the
    // queue patch, which takes the group patch along in ITS launch.  The caller's group counters are written after all three
    // succeeded; on a failure the cycle's placements are undone and the arrays are what they were before the cycle.
    bs_group_delta gd{0, 0, 0, 0};
    if (gang >= 0) {
      // unlabelled pods BEHIND the gang's first pod wait for the next cycle; the gang's members are merged in queue order
      std::vector<uint32_t> all(gone.size() + members.size()), alln(all.size());
      size_t a = 0, bq = 0, w = 0;
      while (a < gone.size() || bq < members.size()) {
        const bool take_a = bq == members.size() || (a < gone.size() && gone[a] < members[bq]);
        all[w] = take_a ? gone[a] : members[bq];
        alln[w++] = take_a ? gone_node[a++] : placed_node[bq++];
      }
      gone.swap(all);
      gone_node.swap(alln);
      // StartBatchSchedule allows EVERY entry of MatchedPodNodes — the members placed now and the pods that were already waiting
      // (io->matched) —, deletes each entry (batchscheduler.go:292-333) and PostBind counts each into Status.Scheduled (core.go:327);
      // the phase turns Scheduled at MinMember (core.go:329-330)
      const uint32_t bound = io->matched[gang] + gang_pods, scn = io->status_scheduled[gang] + bound;
      gang_released = bound;
      gd = bs_group_delta{(uint32_t)gang, 0u, scn,
                          (uint32_t)(io->group_flags[gang] | BS_GROUP_SCHEDULED_LATCH | (scn >= io->min_member[gang] ? BS_GROUP_PHASE_CLOSED : 0u))};
    }
    {
      std::vector<uint32_t> touched(gone_node);
      std::sort(touched.begin(), touched.end());
      touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
      std::vector<bs_node_request> nr(touched.size());
      for (size_t t = 0; t < touched.size(); ++t) {
        nr[t].index = touched[t];
        nr[t].requested_present = io->requested_present[touched[t]];
        std::memset(nr[t].requested, 0, sizeof(nr[t].requested));
        for (uint32_t j = 0; j < L; ++j) nr[t].requested[j] = io->requested[(size_t)j * N + touched[t]];
      }
      bs_pods_delta pd;
      std::memset(&pd, 0, sizeof(pd));
      pd.n_remove = (uint32_t)gone.size();
      pd.remove = gone.data();                                // ascending queue positions
      if ((rc = bs_nodes_assume(ctx, nr.data(), (uint32_t)nr.size())) || (gang >= 0 && (rc = bs_groups_apply(ctx, &gd, 1))) || (rc = bs_pods_apply(ctx, &pd))) {
        undo_cycle();                                         // the caller's node requests are what they were before the cycle
        return rc;
      }
      if (gang >= 0) {
        io->matched[gang] = gd.matched;
        io->status_scheduled[gang] = gd.status_scheduled;
        io->group_flags[gang] = (uint8_t)gd.flags;
      }
      for (size_t m = 0; m < gone.size(); ++m) io->pod_node[orig[gone[m]]] = (int32_t)gone_node[m];
      uint32_t w = 0;                                         // the host's view of the queue follows the device's
      size_t m = 0;
      for (uint32_t i = 0; i < P; ++i) {
        if (m < gone.size() && gone[m] == i) { ++m; continue; }
        orig[w++] = orig[i];
      }
      orig.resize(w);
    }
    if (gang >= 0) {
      const int64_t t = now_ns();
      if (io->n_admitted < io->cap) {
        io->admitted_group[io->n_admitted] = (uint32_t)gang;
        io->admitted_pods[io->n_admitted] = gang_released;
        io->admitted_ns[io->n_admitted] = t - t_begin;
        io->cycle_ns[io->n_admitted] = t - t_cycle;
      }
      io->n_admitted++;
    } else {
      break;                                                  // only unlabelled pods were left to place: nothing can become ready any more
    }
  }
  io->pods_left = (uint32_t)orig.size();
  io->total_ns = now_ns() - t_begin;
  return BS_OK;
}
