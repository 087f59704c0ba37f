// bs_kernels.hpp — gfx950 kernels of the batched PreFilter / Filter / Permit path: the shared work loops (node scan,
// Filter evaluation, table build for the single-query entry points) and the GENERAL chain.
//
// A batch (bs_batch_run) takes one of three chains:
//   steady state   bs_fast.hpp    two launches (three dependency levels); every gang has its pod and MinResources, the leader has matched pods
//   positional     bs_epoch.hpp   three launches; first-pod captures / MinResources defaults / leader without matched pods
//   general        this file      what is left: more than sixteen leader runs in one batch, early Filter, BS_NO_FAST / BS_NO_EPOCH
// The steady-state and positional chains use scan_core / scan_loop / filter_item / filter_loop / filter_params_for / tables_local_* from here.
//
// General chain, one stream:
//   k_prepass   per-batch resets; per pod: eligibility (core.go:89-110), first eligible pod / first owner
//               per group; LAST block: findMaxPG (core.go:701-739) when no first-pod capture can occur
//   [k_init, k_epochs_a/b, k_leader_scan   only when groups without a pod exist: capture epochs, findMaxPG per epoch]
//   k_query     per pod: fillOccupiedObj check, branch A/B/C/D of core.go:127-166, request vector
//               (getPreAllocatedResource core.go:774-793 [+ pod request :157-159]) into its request SLOT;
//               class mode: also the Filter parameters of the pod's class into the Filter slots
//   k_tables_local_nofix   singleNodeResource (core.go:634-670) + chunk-local running sums of core.go:602,621 per table in
//               use, per-group maxima for pruning   [k_prepass_tables / k_query_tables: the one known table inside those launches]
//   k_scan_filter   node scan "exists k : prefix_k >= request" (core.go:623), first such k, per scan slot,
//               and computeResourceSatisfied (core.go:514-564) per Filter slot x node, in one launch
//               [k_scan, k_filter: the same two work loops as separate launches when slot = pod or with early Filter]
//   k_reject/k_final   REJECT codes, deny-cache replay in queue order (core.go:105-110,142,163),
//               stale sop.maxFinishedPG propagation, first_k -> node list index, Filter code + slot per pod
//   k_tally     per-pod feasible counts from the slots, per-group admit counts; last block: quorum predicate core.go:303,
//               re-arm for the next batch
//   k_filter_expand   only when a caller asks bs_batch_read for the pods x nodes bitmap: every pod's row from its slot's
//
// Request slots: pods of a gang share a template, so derived requests repeat; bs_pods_load builds request
// classes (k_pod_class_a/b) and the batch evaluates each distinct request once (see BatchDev).
//
// No MFMA anywhere: this is int64 compare/add work (north_star).  Scan: lanes of a wave are request slots,
// node rows are wave-uniform (LDS broadcast), so one 64-bit v_cmp per resource lane decides 64 slot x node
// pairs.  Filter: lanes are nodes while comparing, slots' requests are broadcast from LDS.
#pragma once

#include "bs_common.hpp"

namespace bs {

struct NodesDev {
  uint32_t n, stride;            // lane stride of alloc/req (>= n)
  const int64_t* alloc;          // [L][stride]
  const int64_t* req;            // [L][stride]
  const uint32_t* apres;         // [n]
  const uint32_t* rpres;         // [n]
  const uint8_t* flags;          // [n]
  const uint32_t* fit;           // [C][fit_words]
  uint32_t fit_words, n_classes;
  const uint32_t* kmap;          // [m] row -> node list index (non-skipped nodes, list order)
  uint32_t m;                    // rows
  const int64_t* left4;          // [4][stride] getLeftResource lanes (core.go:460-463)
  const int64_t* lglob;          // [8] min[4], max[4] of left4 over the nodes Filter can evaluate
};

struct GroupsDev {
  uint32_t g;
  const uint32_t* min_member;
  const uint32_t* status_scheduled;
  const uint32_t* matched;
  const uint8_t* flags;
  const uint32_t* cls;
  const int64_t* minres;         // [L][g]
  const uint32_t* mrpres;
  const uint64_t* occupied;
};

struct PodsDev {
  uint32_t p;
  const int32_t* group;
  const int64_t* req;            // [L][p]
  const uint32_t* pres;
  const uint32_t* cls;
  const uint64_t* owner;
  const uint8_t* flags;
};

struct TableDesc { uint32_t cls; float pct; };

// per-pod stage bits (scratch)
constexpr uint8_t ST_ELIG = 1;      // passed core.go:89-110 against the batch-start deny flags
constexpr uint8_t ST_REACH6 = 2;    // (tentatively) reached findMaxPG, core.go:118-123
constexpr uint8_t ST_OWNED = 4;     // evaluated by this rank
constexpr uint8_t ST_QUERY = 8;     // has a scan query

struct BatchDev {
  // per group
  uint32_t* first_elig;     // min pod index that reaches fillOccupiedObj for the group
  uint32_t* first_owner;    // min eligible pod index with OwnerReferences (OccupiedBy == "" at start)
  uint32_t* first_reject;   // min pod index rejected at core.go:140/161 (AddToDenyCache)
  uint32_t* first_pod;      // min pod index of the group (shard ownership)
  uint32_t* cap_epoch;      // epoch at which pgs.Pod becomes non-nil (0 = already set, INF = never)
  // epochs
  uint32_t* epoch;          // [P] captures at indices <= i
  uint32_t* nepochs;        // [4] [0] E + 1; [1] first pod that reaches findMaxPG (early Filter)
  int32_t* leader_epoch;    // [E+1]
  uint8_t* panic_epoch;     // [E+1]
  // per pod
  uint8_t* tcode;           // tentative PreFilter code
  uint8_t* stage;
  int32_t* leader_raw;      // leader findMaxPG returned for this pod (valid iff ST_REACH6)
  uint32_t* first_row;      // [scan slots] min table row satisfying the slot's request (INF none)
  unsigned long long* chunk_rec;    // [64 chunks][kRecStride] k_fast_step_a's whole-step form: the chunks' totals / first key rows as tagged 64-bit words (bs_fast.hpp)
  unsigned long long* first_row64;  // [scan slots] whole-step form, large queues: first_row[] as a 64-bit minimum keyed by ~batch_seq (never reset)
  unsigned long long* scan_rec;     // [256 class slots][kScanRecChunks] whole-step form: tag << 32 | first row of the slot inside the chunk
  unsigned long long* feas_rec;     // [512 Filter slots][kFeasRecGroups] whole-step form: tag << 32 | feasible nodes of the slot in a group of node runs
  int64_t* qreq_s;          // [scan slots][LP] effective request (absent scalar -> INT64_MIN)
  uint32_t* qflags_s;       // [scan slots] bits 0..11 request key present, bits 16..27 "zero/absent" (passes w/o left key)
  uint32_t* qpos;           // [P] pod -> scan slot (valid iff ST_QUERY)
  // tables / tiles
  uint32_t* needed;         // [2C+1] table (class + C*(pct==0.7)) is used by some query of the batch
  uint32_t* qcount;         // [1] scan queries emitted
  uint32_t* ticket;         // [4] last-block tickets
  TableDesc* desc;          // [slots]
  uint32_t* ntables;        // [1]
  int64_t* tables;          // [slots][mcap][LP] running sums, row-major (one s_load per row)
  uint32_t* kp;             // [slots][16] first row at which scalar key s exists in the running sum
  uint64_t* stats;          // [8] counters (only touched when collect_stats)
  unsigned long long* chunk_tot;   // [slots][nchunks][16] chunk totals of the two-level table scan
  uint32_t* chunk_kp;       // [slots][nchunks][16] per chunk: first row at which scalar key s is present
  uint32_t* blk_scratch;    // per-block summaries of the two-level pod scans
  int64_t* gmax;            // [slot][ceil(mcap/64)][LP] per 64-row group: max running sum per resource lane (pruning)
  // filter
  int64_t* fparams;         // [P][8]: R[4] = pod + maxSingle, M[4] = maxSingle  (fixed lanes)
  uint32_t* fflags;         // [P] bit0 scalar_block (case 2 impossible), bit1 leader_block, bits 8.. fl_code
  // Request slots.  Pods of one gang share a template, so derived requests repeat massively; work is done
  // per SLOT and copied to the pods.  With request classes (pclass, built at bs_pods_load) a slot is
  //   scan:    class c (reserve check, core.go:157-166)  or  K + group g (first check, core.go:136-147)
  //   Filter:  class c + K * (pod precedes the first findMaxPG of the batch)
  // and every pod of a slot writes the same values into it (plain stores, no atomics).  When a first-pod
  // capture or a MinResources default can occur in the batch a slot is simply the pod itself.
  const uint32_t* pclass;   // [P] request class of the pod: equal (request lanes, present bits) <=> equal class
  const uint32_t* kclass;   // [1] K = number of classes
  unsigned long long* cls_slots;  // [cls_mask+1] hash slots of the class builder: bit63 | hash31 | pod
  uint32_t cls_mask;
  int32_t* qtab_s;          // [scan slots] table id of the slot's query, -1 = slot unused in this batch
  uint32_t* fu_slot;        // [P] Filter slot of the pod
  int64_t* uparams;         // [filter slots][8] R[4] = pod + maxSingle, M[4] = maxSingle (fixed lanes)
  uint32_t* uflags;         // [filter slots] as fflags; fl_code NOT_RUN = slot unused
  uint32_t* uclaim;         // [filter slots] k_fast_step_a: stamp of the batch whose (one) writer claimed the slot
  uint64_t* fu_bitmap;      // [W][filter slots] rows of the slots
  uint32_t* fu_feas;        // [filter slots] feasible-node counts of the slots
  // node words of the batch (round 6, node_words_block in bs_fast.hpp; null = not built for this batch).  Three tables of [stride] word PAIRS, one
  // pair per 64-node block w — one s_load_dwordx4 of the Filter item:
  //   nodew[(t * stride + w) * 2]       nodes Filter can evaluate (in range, neither nil nor without a Node object: core.go:442-449)
  //   nodew[(t * stride + w) * 2 + 1]   nodes that can NOT hold one member of the leader's gang (left < maxSingle on some fixed lane): the nodes
  //                                     case 3 (core.go:558-563) lets pass.  t = 0: the batch's findMaxPG result, t = 1: the leader carried in,
  //                                     t = 2: a leader whose MinResources names a scalar resource (no node holds a member: getLeftResource has no scalars, Q4)
  //   nodew[6 * stride + 4 t + j]       the maxSingle the pairs of table t < 2 were built from (int64), [6 * stride + 8 + t] bit 0: built, bit 1: scalar MinResources
  uint64_t* nodew;
  uint32_t nodew_stride;
  uint32_t tiles2_min;      // the transposed Filter items take PAIRS of tiles from this many tiles of Filter slots on (filter_loop_t; run_fast: BS_TP_TMIN x ranks)
  // ---- steady-state fast path (bs_fast.hpp): nothing here is reset per batch
  uint32_t* qstamp_s;       // [scan slots] batch stamp of the slot's last writer (slot live iff == prm.stamp)
  const uint32_t* first_pod_s;    // [G] min pod index of the group                               } derived from the pods alone
  const uint32_t* first_np_s;     // [G] min pod index without BS_POD_LAST_PERMITTED              } at bs_pods_load
  const uint32_t* first_owner_s;  // [G] min such pod index that has OwnerReferences              } (k_pod_pairs)
  const unsigned long long* pair_head;  // [G] first (group, request class) pair of the group: (class << 32) | representative pod, low word BS_INF = none
  const uint32_t* ppair;          // [P] pod -> its pair (= index of the pair's representative pod)
  const unsigned long long* pair_next;  // [P] at representatives: the next link of the group's chain, same encoding
  unsigned long long* pair_firstq;// [views][pair_stride] by pair id: (~batch_seq << 32) | first pod of the pair with a scan query
  uint32_t pair_stride;           // id space of the pairs (a pair's id is its representative pod at bs_pods_load, a drawn number after bs_pods_apply)
  unsigned long long* first_reach64;  // [blocks of launch A] (~batch_seq << 32) | first pod of the block that reaches findMaxPG
  uint32_t* fast_reject;    // [G] first rejected pod of the group (only maintained for BS_BATCH_COMMIT)
  const uint32_t* own_start;// [P] sharded contexts only: pods owned before queue position i (see k_owner_starts); owner = own_start[anchor] * nranks / P
  const uint32_t* gcount;   // [G] pods of the group in the resident queue (bs_pods_load / bs_pods_apply keep it): the thread whose add
  unsigned long long* admit64;  // [G] brings (pods seen << 32 | pods admitted) up to gcount closes the group — quorum without a last-block pass
  // BS_BATCH_HOST_RESULTS: mirrors of the results in pinned host memory, written by the last launch (null = off)
  uint8_t* h_pf_code; uint32_t* h_pf_first_k; int32_t* h_pf_leader; uint8_t* h_fl_code; uint32_t* h_fl_feasible; uint32_t* h_fl_slot;
  uint32_t* h_admit; uint8_t* h_ready; uint32_t* h_feas; uint64_t* h_rows; int32_t* h_tag; uint32_t hstride;
  int32_t* h_err;           // pinned host word: a final block of the fused launch gave up waiting for the producers (see fast_final_block)
  uint32_t* epoch_group;    // [E+1] group captured at epoch e (e >= 1)
  // BS_BATCH_FILTER_DENY (bs_fdeny.hpp)
  unsigned long long* fd_event;   // [G] (~key sequence << 32) | first pod of the group whose Filter failed on a node
  uint32_t* fd_in;          // [G] fixed-point re-runs: pods of the group behind this queue position are turned away at the deny check (BS_INF: nobody); null = off
  uint32_t* fd_flag;        // [1] bit 0: a pod somebody else needed was turned away, bit 1: events found != events given
  int32_t* h_fd;            // pinned [2]: the same two bits for the host
  // outputs
  uint8_t* pf_code;
  uint32_t* pf_first_k;
  int32_t* pf_leader;
  uint8_t* fl_code;
  uint32_t* fl_feasible;
  uint64_t* fl_bitmap;      // [W][P]
  uint32_t* admit;          // [G]
  uint8_t* ready;           // [G]
};

struct BatchParams {
  uint32_t L, S, LP, C;        // lanes, scalar lanes, padded row length, fit classes
  uint32_t eph_gate;
  uint32_t rank, nranks;
  int32_t sop_leader0;         // sop.maxFinishedPG carried into the batch (-1 none)
  uint32_t run_filter;
  uint32_t early_filter;       // Filter parameters come from k_fparams_early (Filter overlaps the node scan)
  uint32_t hash_keep;          // hash bits kept in a class-builder slot (0x7FFFFFFF; fewer = forced collisions, tests)
  uint32_t fuse_filter;        // class mode, no early Filter: k_query fills the Filter slots, k_scan_filter evaluates them
  uint32_t use_classes;        // slots are request classes (no capture / MinResources default possible in this batch)
  uint32_t scan_slots_cap, filter_slots_cap;   // entries to reset per batch
  uint32_t collect_stats;
  uint32_t mcap;               // table row capacity
  uint32_t stamp;              // fast path: slot stamp of this batch (never 0)
  uint32_t seq_inv;            // fast path: ~batch sequence number (64-bit atomicMin keys: a newer batch always wins)
  uint32_t commit;             // fast path: keep fast_reject for k_fast_commit
  uint32_t do_tally, do_ready; // fast path: stages of the final launch
  int32_t host_tag;            // BS_BATCH_HOST_RESULTS: completion word the final launch publishes (0 = off)
  uint32_t scan_nsub;          // steady-state scan: waves of a block that share one item's groups (4: latency regime, few tiles; 1: many tiles)
  uint32_t k_host;             // request classes, when the host already knows the count (0: read *kclass — a dependent load in front of the first round trip)
  uint32_t filter_deny;        // BS_BATCH_FILTER_DENY: the chain's last launch leaves tally and completion word to k_fd_apply
  uint32_t fd_iter;            // > 0: a fixed-point re-run (fd_in holds the events of the run before)
  uint32_t first_reach_hint;   // partitioned mode (bs_first_reach_hint): first LOCAL queue index at or behind the whole job's first pod that reaches
                               // findMaxPG (core.go:118); BS_INF = none (the local queue is the whole queue)
};

// BS_BATCH_FILTER_DENY re-runs: the pod stands behind the position at which its group was deny-listed by a failing Filter
__device__ __forceinline__ bool fd_denied(const BatchDev& b, uint32_t g, uint32_t i) { return b.fd_in && b.fd_in[g] < i; }

// ------------------------------------------------------------------------------------------------
// snapshot-derived data (at bs_nodes_load / bs_nodes_apply)
// ------------------------------------------------------------------------------------------------

// kmap: stable compaction of the nodes compareClusterResourceAndRequire does not skip
// (core.go:606-617); left4: getLeftResource lanes (core.go:460-463).  Single block.
// Incremental: nodes below `base0` (a multiple of the block size) are unchanged since the last call and
// contribute `m_before` rows; only [base0, n) is recomputed (node churn: rescan from the first change).
__device__ __forceinline__ int64_t wave_max_i64(int64_t v);
__device__ __forceinline__ int64_t wave_min_i64(int64_t v);

#if BS_EMIT_MAIN
__global__ __launch_bounds__(kScanBlock) void k_nodes_derive(NodesDev nd, uint32_t* kmap, uint32_t* m_out,
                                                             int64_t* left4, int64_t* lglob, uint32_t base0, uint32_t m_before) {
  __shared__ uint32_t lds[16];
  __shared__ int64_t s_mm[16][8];
  int64_t gmin[4] = {INT64_MAX, INT64_MAX, INT64_MAX, INT64_MAX}, gmx[4] = {INT64_MIN, INT64_MIN, INT64_MIN, INT64_MIN};
  // unchanged prefix: its left4 is already resident, fold it into the cluster-wide bounds
  for (uint32_t i = threadIdx.x; i < base0; i += kScanBlock) {
    if (nd.flags[i] & (BS_NODE_NIL | BS_NODE_NO_NODE)) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = left4[(size_t)j * nd.stride + i];
      gmin[j] = v < gmin[j] ? v : gmin[j];
      gmx[j] = v > gmx[j] ? v : gmx[j];
    }
  }
  uint32_t carry = m_before;
  for (uint32_t base = base0; base < nd.n; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    uint32_t keep = 0;
    if (i < nd.n) {
      keep = (nd.flags[i] & BS_NODE_SKIP_MASK) ? 0u : 1u;
      const bool ok = !(nd.flags[i] & (BS_NODE_NIL | BS_NODE_NO_NODE));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t v = wsub(nd.alloc[(size_t)j * nd.stride + i], nd.req[(size_t)j * nd.stride + i]);
        left4[(size_t)j * nd.stride + i] = v;
        if (ok) { gmin[j] = v < gmin[j] ? v : gmin[j]; gmx[j] = v > gmx[j] ? v : gmx[j]; }
      }
    }
    uint32_t total;
    const uint32_t incl = block_incl_scan_add<uint32_t>(keep, lds, total);
    if (keep) kmap[carry + incl - 1] = i;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *m_out = carry;
  // cluster-wide min / max of left per fixed lane (k_filter's lane-subset test)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t mn = wave_min_i64(gmin[j]), mx = wave_max_i64(gmx[j]);
    if (lane_id() == 0) { s_mm[wave_id()][j] = mn; s_mm[wave_id()][4 + j] = mx; }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int64_t r = s_mm[0][threadIdx.x];
    for (int w = 1; w < kScanBlock / 64; ++w) {
      const int64_t x = s_mm[w][threadIdx.x];
      r = threadIdx.x < 4 ? (x < r ? x : r) : (x > r ? x : r);
    }
    lglob[threadIdx.x] = r;
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// k_init: full scratch reset (after a load, or when the previous batch left the per-group minima
// dirty).  In steady state it is not launched: k_tally's last block re-arms them for the next batch.
// ------------------------------------------------------------------------------------------------
#if BS_EMIT_MAIN
__global__ void k_init(GroupsDev gr, BatchDev b) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < gr.g) {
    b.first_elig[t] = BS_INF;
    b.first_owner[t] = BS_INF;
    b.first_reject[t] = BS_INF;
    b.first_pod[t] = BS_INF;
    b.cap_epoch[t] = (gr.flags[t] & BS_GROUP_HAS_POD) ? 0u : BS_INF;
  }
  if (t < 4) b.ticket[t] = 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// k_prepass: per-batch resets + the steps of PreFilter that do not need any other pod
// (core.go:89-110).  When no first-pod capture can happen in the batch (every group already has its
// pod) the LAST block runs findMaxPG for the single epoch instead (k_leader's body): one launch less.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Exact de-duplication of per-pod work.  Pods of one gang share a template, so the derived request
// vectors repeat massively; a request is evaluated once and its row is copied to the duplicates.
// Open-addressing table, one 64-bit slot = valid | 31 hash bits | representative pod.  The first pod to
// claim a slot is the representative; a later pod that meets the same hash bits compares the full key
// (the representative stored it before its CAS: release / acquire at agent scope) and otherwise probes on.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  return x ^ (x >> 33);
}
template <class Same>
__device__ __forceinline__ uint32_t dedupe_insert(unsigned long long* slots, uint32_t mask, uint32_t hash_keep, uint64_t h, uint32_t i,
                                                  Same same_as, bool& winner) {
  const uint32_t tag = (((uint32_t)(h >> 32)) & hash_keep) | 0x80000000u;
  const unsigned long long mine = ((unsigned long long)tag << 32) | i;
  uint32_t sl = (uint32_t)h & mask;
  for (;;) {
    unsigned long long cur = __hip_atomic_load(&slots[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0ull) {
      unsigned long long expected = 0ull;
      if (__hip_atomic_compare_exchange_strong(&slots[sl], &expected, mine, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
        winner = true;
        return i;
      }
      cur = expected;
    }
    if ((uint32_t)(cur >> 32) == tag) {
      const uint32_t rep = (uint32_t)cur;
      if (same_as(rep)) { winner = false; return rep; }
    }
    sl = (sl + 1u) & mask;
  }
}

__device__ __forceinline__ uint64_t bcast64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

// Request classes of the loaded pods (bs_pods_load): equal (request lanes, present bits) <=> equal class.
// k_pod_class_a: every pod looks its request up in the hash table; the first of a kind becomes the representative (rep[i] == i);
// each block leaves its count of representatives.  k_pod_class_ids: a representative's class id = its rank among the
// representatives in QUEUE order (block counts + in-block scan), so class ids grow with the queue position of the class: the
// pods of a contiguous piece of the queue — a rank's share under bs_shard_set, a gang — fall into neighbouring class slots, and
// the tiles of 64 slots the throughput regime's Filter items and scan items work on are either a rank's or empty (ids drawn in
// arrival order — one atomicAdd per class, rounds 1-4 — spread every rank's classes over all tiles: a rank's step cost as much
// as the whole job's).  Classes drawn later by the resident queue (bs_queue.hpp) follow behind in arrival order, as before.
// k_pod_class_b: every pod takes its representative's id.
#if BS_EMIT_MAIN
__global__ __launch_bounds__(256) void k_pod_class_a(PodsDev pods, unsigned long long* slots, uint32_t mask, uint32_t hash_keep, uint32_t L, uint32_t* rep,
                                                     uint32_t* blk_count) {
  __shared__ uint32_t lds[16];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool winner = false;
  if (i < pods.p) {
    const uint32_t pres = pods.pres[i];
    uint64_t h = mix64((uint64_t)pres + 0x9e3779b97f4a7c15ull);
    for (uint32_t j = 0; j < L; ++j) h = mix64(h ^ (uint64_t)pods.req[(size_t)j * pods.p + i]);
    rep[i] = dedupe_insert(slots, mask, hash_keep, h, i, [&](uint32_t o) {
      if (o >= pods.p || pods.pres[o] != pres) return false;
      for (uint32_t j = 0; j < L; ++j)
        if (pods.req[(size_t)j * pods.p + o] != pods.req[(size_t)j * pods.p + i]) return false;
      return true;
    }, winner);
  }
  uint32_t total;
  (void)block_incl_scan_add<uint32_t>(winner ? 1u : 0u, lds, total);
  if (threadIdx.x == 0) blk_count[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_pod_class_ids(uint32_t p, const uint32_t* rep, const uint32_t* blk_count, uint32_t* id, uint32_t* kcount) {
  __shared__ uint32_t lds[16];
  uint32_t part = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += blockDim.x) part += blk_count[j];
  uint32_t prev;
  (void)block_incl_scan_add<uint32_t>(part, lds, prev);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool is_rep = i < p && rep[i] == i;
  uint32_t total;
  const uint32_t incl = block_incl_scan_add<uint32_t>(is_rep ? 1u : 0u, lds, total);
  if (is_rep) id[i] = prev + incl - 1u;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *kcount = prev + total;
}
#endif
#if BS_EMIT_MAIN
__global__ void k_pod_class_b(uint32_t p, const uint32_t* rep, const uint32_t* id, uint32_t* pclass) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p) pclass[i] = id[rep[i]];
}
#endif

__device__ __forceinline__ void leader_block(const GroupsDev& gr, const BatchDev& b, uint32_t e);

constexpr int kPrepassBlock = 512;

// per-thread part of the pre-pass: thread i of nthreads (resets are strided over all of them)
__device__ __forceinline__ void prepass_thread(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, uint32_t no_capture,
                                               uint32_t i, uint32_t nthreads) {
  // resets whose consumers run in later launches
  if (i < gr.g) b.admit[i] = 0;
  if (i < 2 * prm.C + 1) b.needed[i] = 0;
  if (i == 0) {
    b.qcount[0] = 0;
    b.qcount[1] = 0;
    if (no_capture) *b.nepochs = 1;
    b.nepochs[1] = BS_INF;
  }
  if (i < 8 && prm.collect_stats) b.stats[i] = 0;
  {
    // request slots of this batch start out unused
    for (uint32_t k = i; k < prm.scan_slots_cap; k += nthreads) {
      b.qtab_s[k] = -1;
      b.first_row[k] = BS_INF;
    }
    if (prm.run_filter)
      for (uint32_t k = i; k < prm.filter_slots_cap; k += nthreads) {
        b.uflags[k] = (uint32_t)BS_FL_NOT_RUN << 8;
        b.fu_feas[k] = 0;
      }
  }
  if (i >= pods.p) return;
  b.fl_feasible[i] = 0;
  if (no_capture) b.epoch[i] = 0;
  const int32_t gi = pods.group[i];
  uint8_t st = 0;
  if (gi >= 0 && (uint32_t)gi < gr.g) {
    atomicMin(&b.first_pod[gi], i);
    const bool permitted = pods.flags[i] & BS_POD_LAST_PERMITTED;
    const bool denied0 = (gr.flags[gi] & BS_GROUP_DENIED) || fd_denied(b, (uint32_t)gi, i);
    if (!permitted && !denied0) {
      st = ST_ELIG;
      atomicMin(&b.first_elig[gi], i);
      if (pods.owner[i] != 0 && gr.occupied[gi] == 0) atomicMin(&b.first_owner[gi], i);
    }
  }
  b.stage[i] = st;
}

#if BS_EMIT_MAIN
__global__ __launch_bounds__(kPrepassBlock) void k_prepass(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm, uint32_t no_capture,
                                                           uint32_t fused_leader) {
  if (fused_leader && blockIdx.x == gridDim.x - 1) {
    leader_block(gr, b, 0);
    return;
  }
  prepass_thread(pods, gr, b, prm, no_capture, blockIdx.x * kPrepassBlock + threadIdx.x, (gridDim.x - fused_leader) * kPrepassBlock);
}
#endif

// ------------------------------------------------------------------------------------------------
// k_epochs_a / k_epochs_b: epoch[i] = number of first-pod captures (pgs.Pod = pod, core.go:486-488)
// at queue positions <= i.  findMaxPG skips groups without a pod (core.go:709-711), so its candidate
// set — and therefore the leader — can only change at these positions.  Two-level ordered scan:
// per-block counts, then block prefix + in-block scan.  Skipped entirely when every group already
// has its pod (k_init writes epoch = 0).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t capture_flag(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, uint32_t i, int32_t& gi) {
  gi = -1;
  if (i < pods.p && (b.stage[i] & ST_ELIG)) {
    gi = pods.group[i];
    return (b.first_elig[gi] == i && !(gr.flags[gi] & BS_GROUP_HAS_POD)) ? 1u : 0u;
  }
  return 0u;
}
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kScanBlock) void k_epochs_a(PodsDev pods, GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  int32_t gi;
  const uint32_t cap = capture_flag(pods, gr, b, blockIdx.x * kScanBlock + threadIdx.x, gi);
  uint32_t total;
  (void)block_incl_scan_add<uint32_t>(cap, lds, total);
  if (threadIdx.x == 0) b.blk_scratch[blockIdx.x] = total;
}
#endif
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kScanBlock) void k_epochs_b(PodsDev pods, GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  uint32_t part = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kScanBlock) part += b.blk_scratch[j];
  uint32_t prev;
  (void)block_incl_scan_add<uint32_t>(part, lds, prev);
  const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
  int32_t gi;
  const uint32_t cap = capture_flag(pods, gr, b, i, gi);
  uint32_t total;
  const uint32_t incl = block_incl_scan_add<uint32_t>(cap, lds, total);
  if (i < pods.p) b.epoch[i] = prev + incl;
  if (cap) { b.cap_epoch[gi] = prev + incl; b.epoch_group[prev + incl] = (uint32_t)gi; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *b.nepochs = prev + total + 1;
}
#endif

// ------------------------------------------------------------------------------------------------
// k_leader: findMaxPG (core.go:701-739) for epoch blockIdx.x, groups in array order.
//   finished = 0                                   if uint32(MinMember - Scheduled) == 0   (:712-714)
//            = uint32((matched+Scheduled)*1000) / MinMember   otherwise (panics if MinMember==0) (:716-717)
//   strict '>' replaces (:721-724); on '==' replace iff max is nil or
//   (max.Scheduled >= max.MinMember && cand.Scheduled == 0) (:729-731).
// The sequential fold only depends on the candidates tied at the maximum F, in order: the first of
// them is taken unconditionally, and from then on the holder `cur` is replaced by the next tied
// candidate with Scheduled == 0 while `cur` itself is fully scheduled.  That chain is walked with
// block-wide min reductions (it has length <= 2 unless MinMember == 0 groups exist).
// One 64-bit max reduction carries {panic, any candidate, F}.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool leader_candidate(const GroupsDev& gr, const BatchDev& b, uint32_t g, uint32_t e) {
  return !(gr.flags[g] & BS_GROUP_SCHEDULED_LATCH) && b.cap_epoch[g] <= e;
}
__device__ __forceinline__ uint32_t leader_finished(const GroupsDev& gr, uint32_t g, bool& panic) {
  const uint32_t mm = gr.min_member[g], sc = gr.status_scheduled[g];
  if ((uint32_t)(mm - sc) == 0u) return 0u;
  if (mm == 0u) { panic = true; return 0u; }
  return (uint32_t)((uint32_t)(gr.matched[g] + sc) * 1000u) / mm;
}
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long u = __shfl_xor(v, o);
    v = u > v ? u : v;
  }
  __syncthreads();
  if (lane_id() == 0) lds[wave_id()] = v;
  __syncthreads();
  unsigned long long r = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = lds[w] > r ? lds[w] : r;
  return r;
}

__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t u = __shfl_xor(v, o);
    v = u > v ? u : v;
  }
  return v;
}
__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t u = __shfl_xor(v, o);
    v = u < v ? u : v;
  }
  return v;
}

constexpr int kLeaderBlock = 512;
constexpr int kLeaderPerThread = 16;     // groups cached in registers per thread (G <= 8192), else recomputed

__device__ __forceinline__ void leader_block(const GroupsDev& gr, const BatchDev& b, uint32_t e) {
  __shared__ unsigned long long lds64[16];
  __shared__ uint32_t lds[16];
  // per thread: up to kLeaderPerThread groups cached in registers (candidate bit + finished)
  uint32_t fin[kLeaderPerThread];
  uint32_t cand = 0;
  bool panic = false;
  unsigned long long best = 0;                     // max of finished + 1 over candidates, 0 = none
#pragma unroll
  for (int it = 0; it < kLeaderPerThread; ++it) {
    const uint32_t g = threadIdx.x + (uint32_t)it * kLeaderBlock;
    fin[it] = 0;
    if (g < gr.g && leader_candidate(gr, b, g, e)) {
      cand |= 1u << it;
      fin[it] = leader_finished(gr, g, panic);
      const unsigned long long k64 = (unsigned long long)fin[it] + 1ull;
      best = k64 > best ? k64 : best;
    }
  }
  for (uint32_t g = threadIdx.x + kLeaderBlock * kLeaderPerThread; g < gr.g; g += kLeaderBlock) {   // G > 8192
    if (!leader_candidate(gr, b, g, e)) continue;
    const unsigned long long k64 = (unsigned long long)leader_finished(gr, g, panic) + 1ull;
    best = k64 > best ? k64 : best;
  }
  if (panic) best |= 1ull << 63;
  const unsigned long long top = block_max_u64(best, lds64);
  if (top >> 63) {
    if (threadIdx.x == 0) { b.leader_epoch[e] = -1; b.panic_epoch[e] = 1; }
    return;
  }
  if (top == 0) {
    if (threadIdx.x == 0) { b.leader_epoch[e] = -1; b.panic_epoch[e] = 0; }
    return;
  }
  const uint32_t F = (uint32_t)(top - 1ull);
  auto tied = [&](uint32_t g) -> bool {
    if (!leader_candidate(gr, b, g, e)) return false;
    bool p2 = false;
    return leader_finished(gr, g, p2) == F;
  };
  // first candidate tied at F (group indices grow with `it`, so scan downwards and keep the last hit)
  uint32_t first = BS_INF;
  for (uint32_t g = threadIdx.x + kLeaderBlock * kLeaderPerThread; g < gr.g; g += kLeaderBlock)
    if (tied(g)) { first = g; break; }
#pragma unroll
  for (int it = kLeaderPerThread - 1; it >= 0; --it)
    if (((cand >> it) & 1u) && fin[it] == F) first = threadIdx.x + (uint32_t)it * kLeaderBlock;
  uint32_t cur = block_min_u32(first, lds);
  for (;;) {
    if (!(gr.status_scheduled[cur] >= gr.min_member[cur])) break;   // holder not fully scheduled: stays
    uint32_t nxt = BS_INF;
    for (uint32_t g = threadIdx.x; g < gr.g; g += kLeaderBlock) {
      if (g <= cur) continue;
      if (gr.status_scheduled[g] == 0u && tied(g)) { nxt = g; break; }
    }
    nxt = block_min_u32(nxt, lds);
    if (nxt == BS_INF) break;
    cur = nxt;
  }
  if (threadIdx.x == 0) { b.leader_epoch[e] = (int32_t)cur; b.panic_epoch[e] = 0; }
}

#if BS_EMIT_MAIN
__global__ __launch_bounds__(kLeaderBlock) void k_leader(GroupsDev gr, BatchDev b) {
  if (blockIdx.x >= *b.nepochs) return;
  leader_block(gr, b, blockIdx.x);
}
#endif

// ------------------------------------------------------------------------------------------------
// k_query
// ------------------------------------------------------------------------------------------------
struct Res {            // upstream nodeinfo.Resource flattened
  int64_t v[BS_MAX_LANES];
  uint32_t present;
};

// TS >= 0: scalar-lane count known at compile time (arrays stay in registers); TS < 0: runtime S.
template <int TS>
struct Shape {
  uint32_t rtS;
  __device__ __forceinline__ explicit Shape(uint32_t s) : rtS(s) {}
  __device__ __forceinline__ uint32_t S() const { return TS >= 0 ? (uint32_t)TS : rtS; }
  __device__ __forceinline__ uint32_t L() const { return 4u + S(); }
};

template <int TS>
__device__ __forceinline__ void res_zero(Res& r, Shape<TS> sh) {
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) r.v[j] = 0;
  r.present = 0;
}
// Resource.Add(ResourceList): ephemeral-storage only behind the feature gate; scalar keys are created.
template <int TS>
__device__ __forceinline__ void res_add(Res& r, const Res& rl, Shape<TS> sh, uint32_t gate) {
  r.v[0] = wadd(r.v[0], rl.v[0]);
  r.v[1] = wadd(r.v[1], rl.v[1]);
  if (gate) r.v[2] = wadd(r.v[2], rl.v[2]);
  r.v[3] = wadd(r.v[3], rl.v[3]);
#pragma unroll
  for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s)
    if (s < sh.S() && (rl.present & (1u << s))) { r.v[4 + s] = wadd(r.v[4 + s], rl.v[4 + s]); r.present |= 1u << s; }
}
// getPodResourceRequire(pod): lanes handed over by the shim, normalised through Add
template <int TS>
__device__ __forceinline__ void pod_require(const PodsDev& pods, uint32_t i, Shape<TS> sh, uint32_t gate, Res& out) {
  Res raw;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) raw.v[j] = pods.req[(size_t)j * pods.p + i];
  raw.present = pods.pres[i];
  res_zero(out, sh);
  res_add(out, raw, sh, gate);
}
// Spec.MinResources of group g as pod i sees it: the loaded value, or (MinResources == nil at load)
// the request of the first pod that reached fillOccupiedObj (core.go:489-493), or nil.
template <int TS>
__device__ __forceinline__ bool group_minres_at(const GroupsDev& gr, const PodsDev& pods, const BatchDev& b, uint32_t g,
                                                uint32_t i, Shape<TS> sh, uint32_t gate, Res& out) {
  if (gr.flags[g] & BS_GROUP_HAS_MINRES) {
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < sh.L()) out.v[j] = gr.minres[(size_t)j * gr.g + g];
    out.present = gr.mrpres[g];
    return true;
  }
  const uint32_t fe = b.first_elig[g];
  if (fe <= i) { pod_require(pods, fe, sh, gate, out); return true; }
  return false;
}
__device__ __forceinline__ uint32_t group_cls_at(const GroupsDev& gr, const PodsDev& pods, const BatchDev& b, uint32_t g) {
  if (gr.flags[g] & BS_GROUP_HAS_POD) return gr.cls[g];
  return pods.cls[b.first_elig[g]];
}
// getPreAllocatedResource, core.go:774-793 (repeated Add == wrapping multiply)
template <int TS>
__device__ __forceinline__ void pre_allocated(const GroupsDev& gr, uint32_t g, int64_t matched, bool have_mr, const Res& mr,
                                              Shape<TS> sh, uint32_t gate, Res& out) {
  res_zero(out, sh);
  const int64_t mm = (int64_t)gr.min_member[g];
  const int64_t not_finished = matched != 0 ? mm - matched : mm - (int64_t)gr.status_scheduled[g];
  if (not_finished > 0 && have_mr) {
    Res times;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < sh.L()) times.v[j] = wmul(mr.v[j], not_finished);
    times.present = mr.present;
    res_add(out, times, sh, gate);
  }
  if (out.v[BS_LANE_PODS] == 0) out.v[BS_LANE_PODS] = mm + 1;
}

// Pod-axis sharding: which rank evaluates a pod.  Whole groups: the anchor of a grouped pod is its group's first pod in the
// queue, an ungrouped pod is its own anchor.  Balanced by POD COUNT: walk the queue, give every anchor the weight of what hangs
// on it (its group's pods, or 1), and cut the running weight into nranks equal shares — own_start[i] = weight before position i
// (k_owner_starts).  Cutting the queue positions instead (round 2) put nearly every pod of a queue that is not gang-sorted on
// rank 0: the first pods of all gangs sit early in such a queue.
__device__ __forceinline__ uint32_t owner_rank_of(const BatchDev& b, const BatchParams& prm, uint32_t anchor, uint32_t P) {
  if (prm.nranks <= 1u) return 0u;
  return (uint32_t)(((uint64_t)b.own_start[anchor] * prm.nranks) / P);
}
// single block: exclusive prefix over the queue of w(i) = pods of the group whose first pod sits at i | 1 for a pod outside the
// loaded groups | 0 otherwise
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kScanBlock) void k_owner_starts(PodsDev pods, uint32_t G, const uint32_t* first_pod, const uint32_t* gcount, uint32_t* own_start) {
  __shared__ uint32_t lds[kScanBlock / 64];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < pods.p; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    uint32_t w = 0;
    if (i < pods.p) {
      const int32_t gi = pods.group[i];
      if (gi >= 0 && (uint32_t)gi < G) w = first_pod[gi] == i ? gcount[gi] : 0u;
      else w = 1u;
    }
    uint32_t total = 0;
    const uint32_t incl = block_incl_scan_add<uint32_t>(w, lds, total);
    if (i < pods.p) own_start[i] = carry + incl - w;
    carry += total;
    __syncthreads();
  }
}
#endif

template <int TS>
__device__ __forceinline__ void filter_params_for(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm,
                                                  uint32_t i, uint8_t pf, int32_t leader, uint32_t slot, bool write_slot, bool write_pod);

// every lane of the wave calls this (wave-level ballots inside); i >= pods.p is a no-op lane
template <int TS>
__device__ __forceinline__ void query_thread(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, uint32_t i) {
  const Shape<TS> sh(prm.S);
  const bool valid = i < pods.p;
  const uint32_t gate = prm.eph_gate;
  uint8_t code = BS_PF_PASS_NOT_GROUPED, st = 0;
  int32_t leader = -1, table = -1;
  Res q;
  res_zero(q, sh);
  if (valid) {
    st = b.stage[i];
    const int32_t gi = pods.group[i];
    // shard ownership: all pods of a group live on one rank (owner_rank_of)
    uint32_t anchor = i;
    if (gi >= 0 && (uint32_t)gi < gr.g) anchor = b.first_pod[gi];
    if (owner_rank_of(b, prm, anchor, pods.p) == prm.rank) st |= ST_OWNED;

    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                       // core.go:89-92
    else if (pods.flags[i] & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;    // :95-98
    else if (gi < 0 || (uint32_t)gi >= gr.g) code = BS_PF_ERR_PG_NOT_FOUND;              // :100-103
    else if ((gr.flags[gi] & BS_GROUP_DENIED) || fd_denied(b, (uint32_t)gi, i)) code = BS_PF_ERR_DENIED;   // :105-110
    else {
      const uint32_t g = (uint32_t)gi;
      // fillOccupiedObj occupancy rule (core.go:494-511) replayed in queue order
      bool occ_err = false;
      const uint64_t own = pods.owner[i];
      const uint64_t occ0 = gr.occupied[g];
      if (occ0 != 0) occ_err = (own == 0) || (own != occ0);
      else {
        const uint32_t fo = b.first_owner[g];
        if (fo != BS_INF && i > fo) { const uint64_t occ = pods.owner[fo]; occ_err = (own == 0) || (own != occ); }
      }
      if (occ_err) code = BS_PF_ERR_OCCUPIED;                                            // :113-115
      else {
        const uint32_t e = b.epoch[i];
        if (b.panic_epoch[e]) code = BS_PF_PANIC_DIV0;                                   // :716-717
        else {
          st |= ST_REACH6;
          leader = b.leader_epoch[e];                                                    // :118-123
          if (leader < 0) code = BS_PF_PASS_NO_MAX;                                      // :127-130
          else {
            const int64_t matched = (int64_t)gr.matched[leader];                         // :132-135
            Res mr;
            if (matched == 0) {                                                          // :136-147
              const bool have = group_minres_at(gr, pods, b, g, i, sh, gate, mr);
              pre_allocated(gr, g, 0, have, mr, sh, gate, q);
              table = (int32_t)group_cls_at(gr, pods, b, g);                             // pct 1
              code = BS_PF_PASS_FIRST_FITS;                                              // tentative
            } else if (leader == gi) {
              code = BS_PF_PASS_IS_MAX;                                                  // :150-155
            } else {                                                                     // :157-166
              const bool have = group_minres_at(gr, pods, b, (uint32_t)leader, i, sh, gate, mr);
              pre_allocated(gr, (uint32_t)leader, matched, have, mr, sh, gate, q);
              Res cur;
              pod_require(pods, i, sh, gate, cur);
              res_add(q, cur, sh, gate);
              table = (int32_t)(prm.C + group_cls_at(gr, pods, b, (uint32_t)leader));    // pct 0.7
              code = BS_PF_PASS_RESERVE_FITS;                                            // tentative
            }
          }
        }
      }
    }
    if (!(st & ST_OWNED)) table = -1;
    if (table >= 0) st |= ST_QUERY;
    b.tcode[i] = code;
    b.stage[i] = st;
    b.leader_raw[i] = leader;
  }
  {
    // first pod of the queue that reaches findMaxPG (lanes are in queue order)
    const unsigned long long rb = __ballot(valid && (st & ST_REACH6));
    if (rb && lane_id() == __ffsll((long long)rb) - 1) atomicMin(&b.nepochs[1], i);
  }
  // Filter slots (class mode): what Filter needs for a pod is its request class and the leader it sees —
  // the batch's findMaxPG result or, for the pods before the first one that reaches findMaxPG, the leader
  // carried into the batch.  Both are known here, so every pod that may pass fills both slots of its class
  // (equal values from every writer) and Filter can be evaluated beside the node scan.
  if (prm.fuse_filter && valid && (st & ST_OWNED) && BS_PF_IS_PASS(code)) {
    const uint32_t c = b.pclass[i], K = *b.kclass;
    filter_params_for<TS>(pods, gr, b, prm, i, code, b.leader_epoch[0], c, true, false);
    filter_params_for<TS>(pods, gr, b, prm, i, code, prm.sop_leader0, c + K, true, false);
  }
  // The query goes into its request slot.  With request classes every pod of a slot derives the same
  // (table, request, flags), so they all store the same values and the scan sees each distinct query once.
  const bool has_q = valid && table >= 0;
  if (has_q) {
    uint32_t absok = 0;
#pragma unroll
    for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
      if (s < sh.S()) {
        const bool pres = q.present & (1u << s);
        if (!pres || q.v[4 + s] == 0) absok |= 1u << s;       // core.go:688-692
        if (!pres) q.v[4 + s] = INT64_MIN;                    // key not requested: never constrains
      }
    }
    // A first check (core.go:136-147) asks for the GROUP's pre-allocation against the group's class: every pod of the
    // group that gets here derives the same query (the capture and the MinResources default precede findMaxPG), so the
    // slot is the group in every mode: behind the class slots, or behind the per-pod slots.
    uint32_t slot = i;
    if (code == BS_PF_PASS_FIRST_FITS) slot = (prm.use_classes ? *b.kclass : pods.p) + (uint32_t)pods.group[i];
    else if (prm.use_classes) slot = b.pclass[i];
    int64_t* dst = b.qreq_s + (size_t)slot * prm.LP;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < prm.LP) dst[j] = j < sh.L() ? q.v[j] : INT64_MIN;
    b.qflags_s[slot] = q.present | (absok << 16);
    b.qtab_s[slot] = table;
    b.qpos[i] = slot;
    b.needed[table] = 1;
  }
  if (!prm.use_classes) {
    // how many queries sit in per-pod slots and how many in group slots: the scan sizes its shares on the tiles that can be live
    const unsigned long long hp = __ballot(has_q && code != BS_PF_PASS_FIRST_FITS), hg = __ballot(has_q && code == BS_PF_PASS_FIRST_FITS);
    if (lane_id() == 0) {
      if (hp) atomicAdd(&b.qcount[0], (uint32_t)__popcll(hp));
      if (hg) atomicAdd(&b.qcount[1], (uint32_t)__popcll(hg));
    }
  }
  if (prm.collect_stats) {
    const unsigned long long hq = __ballot(has_q);
    if (lane_id() == 0 && hq) atomicAdd((unsigned long long*)&b.stats[2], (unsigned long long)__popcll(hq));
  }
}

template <int TS>
__global__ void k_query(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm) {
  query_thread<TS>(pods, gr, b, prm, blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_tables_local / k_tables_fix: running sums per table, two-level.
// Row k (k-th non-skipped node in list order) holds leftResources after that node (core.go:602,621):
//   left = fit && !taint_err ? int64(float32(alloc)*pct) - requested : 0         (core.go:634-670)
//   scalar lane s contributes only when both allocatable and requested carry the key (:662-668)
// Block (slot, chunk) scans its 256 rows and records the chunk totals; the fix-up pass adds the sum
// of the preceding chunks.  kp[s] = first row at which the running sum owns scalar key s.
// ------------------------------------------------------------------------------------------------
constexpr int kTblChunk = 256;

// table id t: fit class t % C, percent 1 for t < C (core.go:140), 0.7 otherwise (core.go:161);
// `forced` != null: a single explicit descriptor (single-query entry points).
__device__ __forceinline__ TableDesc table_desc(uint32_t t, uint32_t C, const TableDesc* forced) {
  if (forced) return *forced;
  TableDesc d;
  d.cls = t % C;
  d.pct = t < C ? 1.0f : 0.7f;
  return d;
}

template <int TS>
__device__ __forceinline__ void tables_local_block(const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const TableDesc* forced,
                                                   uint32_t slot, uint32_t chunk, uint32_t nchunks) {
  __shared__ unsigned long long s_wtot[BS_MAX_LANES][4];     // per resource lane, per wave: wave total
  __shared__ uint32_t s_kp[BS_MAX_SCALARS];
  if (chunk * kTblChunk >= nd.m) return;
  const uint32_t k = chunk * kTblChunk + threadIdx.x;
  const bool valid = k < nd.m;
  // level 0: everything that does not depend on another load
  const uint32_t need = forced ? 1u : b.needed[slot];
  const uint32_t n = valid ? nd.kmap[k] : 0u;
  if (!need) return;
  const TableDesc d = table_desc(slot, prm.C, forced);
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S(), LP = prm.LP;
  int64_t* T = b.tables + (size_t)slot * prm.mcap * LP;
  // level 1: every field of the node, issued together
  const uint32_t fw = nd.fit[(size_t)d.cls * nd.fit_words + (n >> 5)];
  const uint8_t fl = nd.flags[n];
  const uint32_t ap = nd.apres[n], rp = nd.rpres[n];
  int64_t al[BS_MAX_LANES], rq[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    if (j < L) {
      al[j] = nd.alloc[(size_t)j * nd.stride + n];
      rq[j] = nd.req[(size_t)j * nd.stride + n];
    }
  }
  const bool fit = valid && ((fw >> (n & 31u)) & 1u) && !(fl & BS_NODE_TAINT_ERR);
  const uint32_t pres = fit ? (ap & rp) : 0u;
  if (threadIdx.x < BS_MAX_SCALARS) s_kp[threadIdx.x] = BS_INF;
  // wave-level inclusive scans of every lane, then ONE exchange of the wave totals
  unsigned long long incl[BS_MAX_LANES];
  const int w = wave_id();
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    if (j < L) {
      const bool live = fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
      const unsigned long long left = live ? (unsigned long long)wsub(scale_f32(al[j], d.pct), rq[j]) : 0ull;
      incl[j] = wave_incl_scan_add_u64(left);
      if (lane_id() == 63) s_wtot[j][w] = incl[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    if (j < L) {
      unsigned long long off = 0, tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned long long x = s_wtot[j][i];
        if (i < w) off += x;
        tot += x;
      }
      incl[j] += off;
      if (valid) T[(size_t)k * LP + j] = (int64_t)incl[j];
      if (threadIdx.x == 0) b.chunk_tot[((size_t)slot * nchunks + chunk) * 16 + j] = tot;
    } else if (j < LP && valid) {
      T[(size_t)k * LP + j] = INT64_MAX;
    }
  }
  if (chunk == 0) {                    // rows of the first chunk are already final: their group maxima
    const uint32_t grp = k >> 6, ngroups = (prm.mcap + 63u) >> 6;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const int64_t mx = wave_max_i64(valid ? (int64_t)incl[j] : INT64_MIN);
        if (lane_id() == 0 && (chunk * kTblChunk + (threadIdx.x & ~63u)) < nd.m) b.gmax[((size_t)slot * ngroups + grp) * LP + j] = mx;
      }
    }
  }
  // first row of this chunk at which key s joins the running sum; the fix-up pass (or, for a single
  // chunk, this block) reduces the per-chunk values to kp[s] — no pre-initialised global needed
#pragma unroll
  for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
    if (s < S) {
      const unsigned long long m = __ballot(valid && (pres & (1u << s)));
      if (m && lane_id() == 0) atomicMin(&s_kp[s], chunk * kTblChunk + (uint32_t)(threadIdx.x & ~63u) + (uint32_t)(__ffsll((long long)m) - 1));
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const uint32_t v = threadIdx.x < BS_MAX_SCALARS ? s_kp[threadIdx.x] : BS_INF;
    b.chunk_kp[((size_t)slot * nchunks + chunk) * 16 + threadIdx.x] = v;
    if (nchunks == 1) b.kp[slot * 16 + threadIdx.x] = v;
  }
}
template <int TS>
__global__ __launch_bounds__(kTblChunk) void k_tables_local(NodesDev nd, BatchDev b, BatchParams prm, const TableDesc* forced) {
  tables_local_block<TS>(nd, b, prm, forced, blockIdx.x, blockIdx.y, gridDim.y);
}

// block (slot, fix_index): fixes chunk fix_index + 1 (chunk 0 needs no fix-up); fix_index 0 also reduces kp
__device__ __forceinline__ void tables_fix_block(const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const TableDesc* forced,
                                                 uint32_t slot, uint32_t fix_index, uint32_t nchunks) {
  __shared__ unsigned long long off[BS_MAX_LANES];
  __shared__ unsigned long long part[kTblChunk];
  static_assert(kTblChunk % 16 == 0, "a thread must always land on the same lane of the [chunk][16] arrays");
  if (!forced && !b.needed[slot]) return;
  const uint32_t chunk = fix_index + 1;
  // Both reductions below run over [chunk][16] arrays with all threads: thread t only ever meets lane t % 16,
  // keeps a private partial and 16 threads fold the 16 partials of their lane (no serial walk over the chunks).
  if (fix_index == 0) {                                                // kp[s] = min over the chunks' first rows
    const uint32_t nvalid = min(nchunks, (nd.m + kTblChunk - 1u) / kTblChunk);
    const uint32_t* src = b.chunk_kp + (size_t)slot * nchunks * 16;
    uint32_t v = BS_INF;
    for (uint32_t e = threadIdx.x; e < nvalid * 16u; e += kTblChunk) v = min(v, src[e]);
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 16) {
      uint32_t r = BS_INF;
      for (int mm = 0; mm < kTblChunk / 16; ++mm) r = min(r, (uint32_t)part[threadIdx.x + 16 * mm]);
      b.kp[slot * 16 + threadIdx.x] = r;
    }
    __syncthreads();
  }
  if (chunk * kTblChunk >= nd.m) return;
  const uint32_t L = prm.L, LP = prm.LP;
  {
    const unsigned long long* src = b.chunk_tot + (size_t)slot * nchunks * 16;
    unsigned long long acc = 0;
    for (uint32_t e = threadIdx.x; e < chunk * 16u; e += kTblChunk) acc += src[e];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
      unsigned long long r = 0;
      for (int mm = 0; mm < kTblChunk / 16; ++mm) r += part[threadIdx.x + 16 * mm];
      off[threadIdx.x] = r;
    }
  }
  __syncthreads();
  const uint32_t k = chunk * kTblChunk + threadIdx.x;
  const bool valid = k < nd.m;
  int64_t* row = b.tables + ((size_t)slot * prm.mcap + k) * LP;
  // fix the rows and record, per 64-row group, the max of the running sum per resource lane — lets k_scan skip groups no
  // request can pass
  const uint32_t grp = k >> 6, ngroups = (prm.mcap + 63u) >> 6;
  const bool grp_valid = (chunk * kTblChunk + (threadIdx.x & ~63u)) < nd.m;
  for (uint32_t j = 0; j < L; ++j) {
    int64_t v = INT64_MIN;
    if (valid) {
      v = (int64_t)((unsigned long long)row[j] + off[j]);
      row[j] = v;
    }
    const int64_t mx = wave_max_i64(v);
    if (grp_valid && lane_id() == 0) b.gmax[((size_t)slot * ngroups + grp) * LP + j] = mx;
  }
}
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kTblChunk) void k_tables_fix(NodesDev nd, BatchDev b, BatchParams prm, const TableDesc* forced) {
  tables_fix_block(nd, b, prm, forced, blockIdx.x, blockIdx.y, gridDim.y + 1);
}
#endif

// ------------------------------------------------------------------------------------------------
// Steady state (one known table per batch): the table build rides in the first two launches instead of on a
// side stream.  Its two passes have the dependency shape of the launches they join — the local scans need only
// the node arrays (k_prepass), the fix-up needs the local pass (k_query), the node scan needs the fix-up — so
// no cross-queue event is left on the critical path and the host issues four calls less per batch.
// `bt` = the batch view shifted to the table's slot (slot index 0), `forced` its descriptor.
// ------------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(kPrepassBlock) void k_prepass_tables(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchDev bt, BatchParams prm,
                                                                  const TableDesc* forced, uint32_t nchunks, uint32_t pre_blocks) {
  // block layout: [0, pre_blocks) pre-pass | pre_blocks: findMaxPG | pre_blocks + 1 + c: table chunk c
  if (blockIdx.x < pre_blocks) {
    prepass_thread(pods, gr, b, prm, 1u, blockIdx.x * kPrepassBlock + threadIdx.x, pre_blocks * kPrepassBlock);
  } else if (blockIdx.x == pre_blocks) {
    leader_block(gr, b, 0);
  } else {
    if (threadIdx.x >= kTblChunk) return;          // whole waves leave: the chunk code is written for kTblChunk threads
    tables_local_block<TS>(nd, bt, prm, forced, 0u, blockIdx.x - pre_blocks - 1u, nchunks);
  }
}
template <int TS>
__global__ __launch_bounds__(kTblChunk) void k_query_tables(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchDev bt, BatchParams prm,
                                                            const TableDesc* forced, uint32_t nchunks, uint32_t query_blocks) {
  if (blockIdx.x < query_blocks) query_thread<TS>(pods, gr, b, prm, blockIdx.x * kTblChunk + threadIdx.x);
  else tables_fix_block(nd, bt, prm, forced, 0u, blockIdx.x - query_blocks, nchunks);
}

// ------------------------------------------------------------------------------------------------
// k_scan — the dominant kernel: "exists k : running_sum_k >= request" (core.go:602-631) for every
// query, and the first such k (the reference's early exit, core.go:623-627).
//
// Geometry.  One wave = one tile of 64 request slots x its share of the live 64-row groups of the
// table.  Lane l holds the request lanes of its slot in VGPRs.  Rows are wave-uniform: a group's 64 rows
// are fetched with one vector load per lane into the wave's LDS slice and broadcast back row by row.
//
// Inner step (one row x 64 queries), hand-written because the compiler's form costs ~20 SALU
// instructions per row (mask ANDs, selects, branches) around 5 compares:
//     s_mov_b64   exec, nf            ; lanes still looking for their first row
//     v_cmpx_ge_i64 vcc, row[0], r0   ; EXEC narrows: each compare only keeps lanes that also
//     ...                             ; satisfied the previous ones  (compareResourceAndRequire,
//     v_cmpx_ge_i64 vcc, row[L-1], r  ;  core.go:672-699, is an AND over lanes)
//     v_min_u32   myk, k, myk         ; surviving lanes record k — rows ascend, so the first one sticks
//     s_mov_b64   exec, -1
// = 2 SALU + (L+1) VALU per 64 slot x node evaluations, no branch.  v_cmpx on a lane outside EXEC
// yields 0, so the chain is the AND; one v_cmp_*_i64 decides 64 pairs.
//
// Scalar keys (core.go:686-697).  Before row kp[s] the running sum has no key s: a lane passes iff
// it requests nothing of s (bit precomputed in qflags); from kp[s] on it is an ordinary compare.  k
// is wave-uniform, so a segment is cut at the kp[s] that fall inside it and each piece runs the
// same branch-free loop with the lanes that cannot pass masked out of EXEC (their padded request
// INT64_MIN / 0 makes the absent-key compare trivially true for the others).
//
// Shares of one tile combine through atomicMin on first_row.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) int64_t* crow_t;
__device__ __forceinline__ crow_t as_const_rows(const int64_t* p) { return (crow_t)(uintptr_t)p; }

#define BS_S_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)   /* lgkmcnt(0), vmcnt/expcnt untouched */

#define BS_CX(j) "v_cmpx_ge_i64 vcc, %[a" #j "], %[r" #j "]\n\t"
#define BS_ROW_HEAD "s_mov_b64 exec, %[nf]\n\t"
#define BS_ROW_TAIL "v_min_u32 %[myk], %[k], %[myk]\n\ts_mov_b64 exec, -1"
#define BS_OPS4 [a0] "v"(a[0]), [r0] "v"(r[0]), [a1] "v"(a[1]), [r1] "v"(r[1]), [a2] "v"(a[2]), [r2] "v"(r[2]), [a3] "v"(a[3]), [r3] "v"(r[3])

template <int L>
__device__ __forceinline__ void row_step(unsigned long long nf, uint32_t& myk, uint32_t k, const int64_t (&a)[L], const int64_t (&r)[L]) {
  if constexpr (L == 4) {
    asm volatile(BS_ROW_HEAD BS_CX(0) BS_CX(1) BS_CX(2) BS_CX(3) BS_ROW_TAIL
                 : [myk] "+v"(myk) : [nf] "s"(nf), [k] "s"(k), BS_OPS4 : "vcc");
  } else if constexpr (L == 5) {
    asm volatile(BS_ROW_HEAD BS_CX(0) BS_CX(1) BS_CX(2) BS_CX(3) BS_CX(4) BS_ROW_TAIL
                 : [myk] "+v"(myk) : [nf] "s"(nf), [k] "s"(k), BS_OPS4, [a4] "v"(a[4]), [r4] "v"(r[4]) : "vcc");
  } else if constexpr (L == 6) {
    asm volatile(BS_ROW_HEAD BS_CX(0) BS_CX(1) BS_CX(2) BS_CX(3) BS_CX(4) BS_CX(5) BS_ROW_TAIL
                 : [myk] "+v"(myk) : [nf] "s"(nf), [k] "s"(k), BS_OPS4, [a4] "v"(a[4]), [r4] "v"(r[4]), [a5] "v"(a[5]), [r5] "v"(r[5]) : "vcc");
  } else if constexpr (L == 7) {
    asm volatile(BS_ROW_HEAD BS_CX(0) BS_CX(1) BS_CX(2) BS_CX(3) BS_CX(4) BS_CX(5) BS_CX(6) BS_ROW_TAIL
                 : [myk] "+v"(myk) : [nf] "s"(nf), [k] "s"(k), BS_OPS4, [a4] "v"(a[4]), [r4] "v"(r[4]), [a5] "v"(a[5]), [r5] "v"(r[5]),
                   [a6] "v"(a[6]), [r6] "v"(r[6]) : "vcc");
  } else if constexpr (L == 8) {
    asm volatile(BS_ROW_HEAD BS_CX(0) BS_CX(1) BS_CX(2) BS_CX(3) BS_CX(4) BS_CX(5) BS_CX(6) BS_CX(7) BS_ROW_TAIL
                 : [myk] "+v"(myk) : [nf] "s"(nf), [k] "s"(k), BS_OPS4, [a4] "v"(a[4]), [r4] "v"(r[4]), [a5] "v"(a[5]), [r5] "v"(r[5]),
                   [a6] "v"(a[6]), [r6] "v"(r[6]), [a7] "v"(a[7]), [r7] "v"(r[7]) : "vcc");
  } else {
    // wide rows (S > 4): two chained statements; the surviving-lane mask travels in an SGPR pair
    static_assert(L > 8 && L <= 16, "row width");
    unsigned long long m = nf;
    {
      const int64_t(&a0)[8] = reinterpret_cast<const int64_t(&)[8]>(a[0]);
      const int64_t(&r0)[8] = reinterpret_cast<const int64_t(&)[8]>(r[0]);
      asm volatile("s_mov_b64 exec, %[m]\n\t"
                   "v_cmpx_ge_i64 vcc, %[a0], %[r0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[r1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[r2]\n\t"
                   "v_cmpx_ge_i64 vcc, %[a3], %[r3]\n\tv_cmpx_ge_i64 vcc, %[a4], %[r4]\n\tv_cmpx_ge_i64 vcc, %[a5], %[r5]\n\t"
                   "v_cmpx_ge_i64 vcc, %[a6], %[r6]\n\tv_cmpx_ge_i64 vcc, %[a7], %[r7]\n\t"
                   "s_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1"
                   : [m] "+s"(m)
                   : [a0] "v"(a0[0]), [r0] "v"(r0[0]), [a1] "v"(a0[1]), [r1] "v"(r0[1]), [a2] "v"(a0[2]), [r2] "v"(r0[2]), [a3] "v"(a0[3]),
                     [r3] "v"(r0[3]), [a4] "v"(a0[4]), [r4] "v"(r0[4]), [a5] "v"(a0[5]), [r5] "v"(r0[5]), [a6] "v"(a0[6]), [r6] "v"(r0[6]),
                     [a7] "v"(a0[7]), [r7] "v"(r0[7])
                   : "vcc");
    }
#pragma unroll
    for (int j = 8; j < L; ++j) {
      asm volatile("s_mov_b64 exec, %[m]\n\tv_cmpx_ge_i64 vcc, %[a], %[r]\n\ts_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1"
                   : [m] "+s"(m) : [a] "v"(a[j]), [r] "v"(r[j]) : "vcc");
    }
    asm volatile("s_mov_b64 exec, %[m]\n\tv_min_u32 %[myk], %[k], %[myk]\n\ts_mov_b64 exec, -1"
                 : [myk] "+v"(myk) : [k] "s"(k), [m] "s"(m));
  }
}

// Two query blocks against one row in ONE statement: EXEC is restored once.
#define BS_CXQ(q, j) "v_cmpx_ge_i64 vcc, %[a" #j "], %[r" #q #j "]\n\t"
#define BS_QHEAD(q) "s_mov_b64 exec, %[nf" #q "]\n\t"
#define BS_QTAIL(q) "v_min_u32 %[myk" #q "], %[k], %[myk" #q "]\n\t"
#define BS_QOUT [myk0] "+v"(myk0), [myk1] "+v"(myk1)
#define BS_AOPS4 [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3])
#define BS_ROPS4(q, arr) [r##q##0] "v"(arr[0]), [r##q##1] "v"(arr[1]), [r##q##2] "v"(arr[2]), [r##q##3] "v"(arr[3])
template <int L>
__device__ __forceinline__ void row_step2(unsigned long long nf0, uint32_t& myk0, unsigned long long nf1, uint32_t& myk1, uint32_t k,
                                          const int64_t (&a)[L], const int64_t (&r0)[L], const int64_t (&r1)[L]) {
  if constexpr (L == 4) {
    asm volatile(BS_QHEAD(0) BS_CXQ(0, 0) BS_CXQ(0, 1) BS_CXQ(0, 2) BS_CXQ(0, 3) BS_QTAIL(0)
                 BS_QHEAD(1) BS_CXQ(1, 0) BS_CXQ(1, 1) BS_CXQ(1, 2) BS_CXQ(1, 3) BS_QTAIL(1) "s_mov_b64 exec, -1"
                 : BS_QOUT : [nf0] "s"(nf0), [nf1] "s"(nf1), [k] "s"(k), BS_AOPS4, BS_ROPS4(0, r0), BS_ROPS4(1, r1) : "vcc");
  } else if constexpr (L == 5) {
    asm volatile(BS_QHEAD(0) BS_CXQ(0, 0) BS_CXQ(0, 1) BS_CXQ(0, 2) BS_CXQ(0, 3) BS_CXQ(0, 4) BS_QTAIL(0)
                 BS_QHEAD(1) BS_CXQ(1, 0) BS_CXQ(1, 1) BS_CXQ(1, 2) BS_CXQ(1, 3) BS_CXQ(1, 4) BS_QTAIL(1) "s_mov_b64 exec, -1"
                 : BS_QOUT : [nf0] "s"(nf0), [nf1] "s"(nf1), [k] "s"(k), BS_AOPS4, [a4] "v"(a[4]), BS_ROPS4(0, r0), [r04] "v"(r0[4]), BS_ROPS4(1, r1), [r14] "v"(r1[4]) : "vcc");
  } else if constexpr (L == 6) {
    asm volatile(BS_QHEAD(0) BS_CXQ(0, 0) BS_CXQ(0, 1) BS_CXQ(0, 2) BS_CXQ(0, 3) BS_CXQ(0, 4) BS_CXQ(0, 5) BS_QTAIL(0)
                 BS_QHEAD(1) BS_CXQ(1, 0) BS_CXQ(1, 1) BS_CXQ(1, 2) BS_CXQ(1, 3) BS_CXQ(1, 4) BS_CXQ(1, 5) BS_QTAIL(1) "s_mov_b64 exec, -1"
                 : BS_QOUT : [nf0] "s"(nf0), [nf1] "s"(nf1), [k] "s"(k), BS_AOPS4, [a4] "v"(a[4]), [a5] "v"(a[5]), BS_ROPS4(0, r0), [r04] "v"(r0[4]), [r05] "v"(r0[5]),
                   BS_ROPS4(1, r1), [r14] "v"(r1[4]), [r15] "v"(r1[5]) : "vcc");
  } else {
    row_step<L>(nf0, myk0, k, a, r0);
    row_step<L>(nf1, myk1, k, a, r1);
  }
}
template <int L, int Q>
__device__ __forceinline__ void row_all(const unsigned long long (&nf)[Q], uint32_t (&myk)[Q], uint32_t k, const int64_t (&a)[L], const int64_t (&r)[Q][L]) {
  if constexpr (Q == 2) row_step2<L>(nf[0], myk[0], nf[1], myk[1], k, a, r[0], r[1]);
  else row_step<L>(nf[0], myk[0], k, a, r[0]);
}

// One wave's share of one tile of 64 request slots, for the slots of the tile that use table `slot`
// (`valid` lanes).  The 64-row groups of the table that are LIVE for them (on every fixed lane the group's
// largest running sum reaches the smallest request) are dealt round-robin over the J waves of the tile:
// wave `share` takes the live groups whose rank is = share (mod J).  Every wave recomputes the live set
// itself (one gather of the group maxima per 64 groups), so the deal is balanced wherever the live groups
// are.  A group's 64 rows are fetched with ONE vector load per lane (lane = row) and parked in this wave's
// LDS slice `rows`; the row loop reads them back with uniform-address (broadcast) ds_reads, so no memory
// round trip sits inside the loop.
// LOCAL: the table holds chunk-local running sums (no fix-up pass, and nothing after the local scans in the building
// launch).  A row's final value is local + (sum of the preceding chunks' totals) in wrapping arithmetic.  The scan keeps
// the exclusive prefix of the chunk totals in registers — lane l of the wave owns chunk 64 w + l of the current window w
// (one load + one wave scan per 16 384 rows) — and adds a group's offset while its rows travel to LDS.  A group is
// pruned only when its local sums and the offset both lie inside (-2^62, 2^62) (then nothing wraps and local max + off
// bounds every row of it) and that bound is below the tile's smallest request.  kp[s] = min of the chunks' first key rows.
// What a wave derives from a chunk-local table before it scans it: the exclusive prefix of the chunk totals for window 0
// (lane l <-> chunk l), the running total behind that window, and the first row of every scalar key.  When every item of
// a launch uses the same table (steady state) a wave computes this ONCE, up front, while its first slot loads are in flight.
template <int S>
struct LocalPre {
  unsigned long long offl[4 + S], carry[4 + S];
  uint32_t kp[S > 0 ? S : 1];
  int64_t gm[2][4 + S];          // per-group local maxima of groups lane and 64 + lane (the first 8192 rows): fetched with everything
                                 // else a wave needs before it can look at a row, instead of one round trip later
};
// raw pieces of LocalPre as they come from memory: lane l <-> chunk l (totals, first key rows) and groups l, 64 + l (maxima)
template <int S>
struct LocalRaw {
  unsigned long long v[4 + S];
  uint32_t kv[S > 0 ? S : 1];
  int64_t gm[2][4 + S];
};
template <int S>
__device__ __forceinline__ void local_fetch_chunks(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t slot, LocalRaw<S>& raw) {
  constexpr int L = 4 + S;
  const uint32_t nchunks = (m + kTblChunk - 1u) / kTblChunk, cstride = (prm.mcap + kTblChunk - 1u) / kTblChunk;
  const uint32_t ch = (uint32_t)lane_id();
  const unsigned long long* ct = b.chunk_tot + ((size_t)slot * cstride + ch) * 16;
#pragma unroll
  for (int j = 0; j < L; ++j) raw.v[j] = ch < nchunks ? ct[j] : 0ull;
#pragma unroll
  for (int s = 0; s < S; ++s) raw.kv[s] = ch < nchunks ? b.chunk_kp[((size_t)slot * cstride + ch) * 16 + s] : BS_INF;
}
template <int S>
__device__ __forceinline__ void local_fetch_gmax(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t slot, int w, int64_t (&gm)[4 + S]) {
  constexpr int L = 4 + S;
  constexpr int LP = (S == 0) ? 4 : (S <= 4 ? 8 : 16);
  const uint32_t ngroups = (m + 63u) >> 6, gstride = (prm.mcap + 63u) >> 6;
  const uint32_t g = (uint32_t)w * 64u + (uint32_t)lane_id();
  const int64_t* src = b.gmax + ((size_t)slot * gstride + min(g, ngroups ? ngroups - 1u : 0u)) * LP;
#pragma unroll
  for (int j = 0; j < L; ++j) gm[j] = src[j];
}
// wave scans over what was fetched: chunk offsets, carry, first key rows
template <int S>
__device__ __forceinline__ void local_pre_finish(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t slot, const LocalRaw<S>& raw, LocalPre<S>& pre) {
  constexpr int L = 4 + S;
  const int lane = lane_id();
  const uint32_t nchunks = (m + kTblChunk - 1u) / kTblChunk, cstride = (prm.mcap + kTblChunk - 1u) / kTblChunk;
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int j = 0; j < L; ++j) pre.gm[w][j] = raw.gm[w][j];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const unsigned long long incl = wave_incl_scan_add_u64(raw.v[j]);
    pre.offl[j] = incl - raw.v[j];
    pre.carry[j] = (unsigned long long)readlane63_i64((long long)incl);
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    uint32_t mn = wave_min_u32(raw.kv[s]);
    for (uint32_t w0 = 64u; w0 < nchunks; w0 += 64u) {          // tables beyond 16 384 rows: the remaining chunks' key rows
      const uint32_t c2 = w0 + (uint32_t)lane;
      mn = min(mn, wave_min_u32(c2 < nchunks ? b.chunk_kp[((size_t)slot * cstride + c2) * 16 + s] : BS_INF));
    }
    pre.kp[s] = __builtin_amdgcn_readfirstlane(mn);               // wave-uniform by construction: keep it scalar (it cuts row pieces)
  }
}
template <int S>
__device__ __forceinline__ void local_pre_load(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t slot, LocalPre<S>& pre) {
  LocalRaw<S> raw;
  local_fetch_chunks<S>(b, prm, m, slot, raw);
  local_fetch_gmax<S>(b, prm, m, slot, 0, raw.gm[0]);
  local_fetch_gmax<S>(b, prm, m, slot, 1, raw.gm[1]);
  local_pre_finish<S>(b, prm, m, slot, raw, pre);
}

template <int S, bool LOCAL = false, bool HAVE_PRE = false>
__device__ __forceinline__ void scan_core(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t slot, uint32_t pos, bool valid,
                                          const int64_t (&r)[1][4 + S], uint32_t qf, uint32_t share, uint32_t J, int64_t (*rows)[4 + S],
                                          const LocalPre<S>& given, uint32_t sub = 0, uint32_t nsub = 1) {
  constexpr int LP = (S == 0) ? 4 : (S <= 4 ? 8 : 16);
  constexpr int L = 4 + S;
  constexpr int U = 4;                           // rows per step: their LDS reads are issued together
  const int lane = lane_id();
  const uint32_t ngroups = (m + 63u) >> 6, gstride = (prm.mcap + 63u) >> 6;
  int64_t rmin[L];                               // smallest request of the tile per resource lane (INT64_MIN where some slot does not ask)
#pragma unroll
  for (int j = 0; j < L; ++j) rmin[j] = wave_min_i64_all(valid ? r[0][j] : INT64_MAX);

  uint32_t myk[1] = {BS_INF};
  uint32_t seen = 0;
  unsigned long long nf = __ballot(valid);       // lanes still looking for their first row
  uint32_t kp[S > 0 ? S : 1];
  unsigned long long absok[S > 0 ? S : 1];
  bool loaded = false;
  const int64_t* T = b.tables + (size_t)slot * prm.mcap * LP;
  uint32_t rows_done = 0;
  uint32_t turn = 0;                             // rank of the next live group modulo J
  [[maybe_unused]] uint32_t probe_groups = 0;    // (probe builds: groups this item fetched)
  // chunk-local tables: exclusive prefix of the chunk totals, window by window (forward only)
  const uint32_t nchunks = (m + kTblChunk - 1u) / kTblChunk, cstride = (prm.mcap + kTblChunk - 1u) / kTblChunk;
  unsigned long long offl[L], carry[L];
  uint32_t win = BS_INF;
#pragma unroll
  for (int j = 0; j < L; ++j) { offl[j] = 0; carry[j] = 0; }
  int64_t gmw[2][L];                             // maxima of the first 128 groups, by VALUE (see below)
#pragma unroll
  for (int j = 0; j < L; ++j) gmw[0][j] = gmw[1][j] = 0;
  if constexpr (LOCAL) {
    // everything is copied out of the struct here, with constant indices: a pointer that may name either of two structs and is
    // followed inside the loop keeps both in scratch memory (it did: 336 bytes per lane, and a launch that uses scratch pays for it)
    if constexpr (HAVE_PRE) {                        // steady state: the caller derived it once per wave (one table for every item)
#pragma unroll
      for (int j = 0; j < L; ++j) { gmw[0][j] = given.gm[0][j]; gmw[1][j] = given.gm[1][j]; offl[j] = given.offl[j]; carry[j] = given.carry[j]; }
#pragma unroll
      for (int s = 0; s < S; ++s) kp[s] = __builtin_amdgcn_readfirstlane(given.kp[s]);
    } else {
      LocalPre<S> mine_pre;
      local_pre_load<S>(b, prm, m, slot, mine_pre);
#pragma unroll
      for (int j = 0; j < L; ++j) { gmw[0][j] = mine_pre.gm[0][j]; gmw[1][j] = mine_pre.gm[1][j]; offl[j] = mine_pre.offl[j]; carry[j] = mine_pre.carry[j]; }
#pragma unroll
      for (int s = 0; s < S; ++s) kp[s] = __builtin_amdgcn_readfirstlane(mine_pre.kp[s]);
    }
    win = 0;
  }
  auto ensure_window = [&](uint32_t w) {
    while (win != w) {
      const uint32_t nxt = win + 1u;               // BS_INF + 1 == 0
      const uint32_t ch = nxt * 64u + (uint32_t)lane;
      const unsigned long long* ct = b.chunk_tot + ((size_t)slot * cstride + ch) * 16;
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const unsigned long long v = ch < nchunks ? ct[j] : 0ull;
        const unsigned long long incl = wave_incl_scan_add_u64(v);
        offl[j] = carry[j] + incl - v;
        carry[j] += (unsigned long long)readlane63_i64((long long)incl);
      }
      win = nxt;
    }
  };

  BS_STAMP(2, 2);
  int64_t gm[L];
#pragma unroll
  for (int j = 0; j < L; ++j) gm[j] = gmw[0][j];
  for (uint32_t c0 = 0; c0 < ngroups; c0 += 64u) {
    // live mask of groups c0 .. c0+63 (lane l <-> group c0+l)
    bool dead = true;
    const uint32_t g = c0 + (uint32_t)lane;
    if constexpr (LOCAL) {
      ensure_window(c0 >> 8);                      // groups c0 .. c0+63 = chunks (c0 >> 2) .. +15: inside one window
      unsigned long long og[L];
#pragma unroll
      for (int j = 0; j < L; ++j) og[j] = __shfl(offl[j], (int)((g >> 2) & 63u));
      // local max per lane (INT64_MAX = do not prune) of this window's groups: the first two windows came with the wave's first
      // fetch (kept as VALUES — a runtime choice between the two arrays turns into an indexed load and the struct into scratch)
      if (c0 == 64u) {
#pragma unroll
        for (int j = 0; j < L; ++j) gm[j] = gmw[1][j];
      } else if (c0 >= 128u && g < ngroups) {
        const int64_t* src = b.gmax + ((size_t)slot * gstride + g) * LP;
#pragma unroll
        for (int j = 0; j < L; ++j) gm[j] = src[j];
      }
      if (g < ngroups) {
        constexpr long long kSafe = 1ll << 62;
        dead = false;
#pragma unroll
        for (int j = 0; j < L; ++j) {
          const long long o = (long long)og[j];
          if (gm[j] != INT64_MAX && o > -kSafe && o < kSafe && gm[j] + o < rmin[j]) dead = true;   // |max|, |off| < 2^62: no wrap, exact bound
        }
      }
    } else if (g < ngroups) {
      const int64_t* gm = b.gmax + ((size_t)slot * gstride + g) * LP;
      dead = false;
#pragma unroll
      for (int j = 0; j < L; ++j) dead = dead || gm[j] < rmin[j];
    }
    unsigned long long live = __ballot(!dead);
    while (live) {
      const uint32_t bit = (uint32_t)__ffsll((long long)live) - 1u;
      live &= live - 1ull;
      const bool mine = turn == share;
      turn = turn + 1u == J ? 0u : turn + 1u;
      if (!mine) continue;
      // ---- this group is ours: rows [g0, gend)
      if (!loaded) BS_STAMP(2, 5);
      ++probe_groups;
      const uint32_t g0 = (c0 + bit) << 6;
      const uint32_t gend = min(m, g0 + 64u);
      int64_t mine_row[L];
      {
        const uint32_t row = min(g0 + (uint32_t)lane, gend - 1u);
        const int64_t* src = T + (size_t)row * LP;
#pragma unroll
        for (int j = 0; j < L; ++j) mine_row[j] = src[j];
        if constexpr (LOCAL) {                     // (the window of this group's chunk is the current one)
#pragma unroll
          for (int j = 0; j < L; ++j) mine_row[j] = (int64_t)((unsigned long long)mine_row[j] + bcast64(offl[j], (int)((g0 / kTblChunk) & 63u)));   // (uniform lane: v_readlane, not ds_bpermute)
        }
      }
      if (!loaded || (LOCAL && J > 2u)) {        // first live group of this wave; with many shares per tile, every group: what
        seen = valid ? __hip_atomic_load(&b.first_row[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // another wave found since
      }
      if (!loaded) {
        loaded = true;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if constexpr (!LOCAL) kp[s] = __builtin_amdgcn_readfirstlane(b.kp[slot * 16 + s]);
          absok[s] = __ballot((qf >> (16 + s)) & 1u);
        }
      }
      __builtin_amdgcn_wave_barrier();           // the previous group's reads are behind us (LDS is in order)
#pragma unroll
      for (int j = 0; j < L; ++j) rows[lane][j] = mine_row[j];
      __builtin_amdgcn_wave_barrier();
      BS_STAMP(2, 3);
      // nsub waves share a group: wave `sub` looks at rows [g0 + sub * 64 / nsub, + 64 / nsub) — the row loop is one wave's
      // dependent chain of LDS reads and compares (~50 ns per row), the longest thing a steady-state scan does
      const uint32_t part = 64u / nsub;
      uint32_t a = g0 + sub * part;
      const uint32_t pend = min(gend, a + part);
      // lanes another wave already served with an earlier row need nothing from this group — nor from any later one (this
      // wave walks its groups in increasing row order)
      nf &= __ballot(seen >= a);
      if (nf == 0) { c0 = ngroups; break; }
      unsigned long long want = nf;
      while (a < pend) {
        // piece [a, e): no kp[s] strictly inside
        uint32_t e = pend;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (kp[s] > a && kp[s] < e) e = kp[s];
        // lanes that can pass while key s is absent from the running sum (core.go:688-692)
        unsigned long long el = ~0ull;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (kp[s] > a) el &= absok[s];
        const unsigned long long act[1] = {want & el};
        uint32_t k = a;
        bool open = act[0] != 0;                 // some lane of this piece is still without a row
        while (open && k + U <= e) {
          int64_t A[U][L];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < L; ++j) A[u][j] = rows[k - g0 + u][j];
#pragma unroll
          for (int u = 0; u < U; ++u) row_all<L, 1>(act, myk, k + u, A[u], r);
          k += U;
          open = (act[0] & __ballot(myk[0] == BS_INF)) != 0;
        }
        if (open) {
          for (; k < e; ++k) {                   // < U leftover rows of the piece
            int64_t R1[L];
#pragma unroll
            for (int j = 0; j < L; ++j) R1[j] = rows[k - g0][j];
            row_all<L, 1>(act, myk, k, R1, r);
          }
        }
        rows_done += k - a;
        want &= __ballot(myk[0] == BS_INF);
        if (want == 0) break;
        a = e;
      }
      // found lanes are done for good (this wave walks its groups in increasing row order)
      nf &= __ballot(myk[0] == BS_INF);
      if (nf == 0) { c0 = ngroups; break; }
    }
  }
  BS_STAMP(2, 4);
  BS_COUNT(2, 6, probe_groups);
  if (!loaded) return;
  if (myk[0] != BS_INF) atomicMin(&b.first_row[pos], myk[0]);
  if (prm.collect_stats && lane == 0) {
    const uint32_t nq = (uint32_t)__popcll(__ballot(valid));
    atomicAdd((unsigned long long*)&b.stats[0], (unsigned long long)rows_done);
    atomicAdd((unsigned long long*)&b.stats[1], (unsigned long long)rows_done * nq);
  }
}

// Work loop: item = (tile of 64 request slots, share j of J).  The slots of a tile that use the same
// table are scanned together (steady state: one table for everything); consecutive waves take different
// tiles with the same share.  The grid is fixed; the slot count is read on the device.
// MODE 0: an item's table is whatever its slots ask for (general chain); 1: steady state, one table for every item, an item per wave;
// 2: steady state, an item per BLOCK (its four waves share the first fetch and take a quarter of every group's rows each)
template <int S, bool LOCAL = false, int MODE = 0>
__device__ __forceinline__ void scan_loop(const BatchDev& b, const BatchParams& prm, uint32_t m, uint32_t jcap, uint32_t nslots_fixed,
                                          uint32_t ngroups_g, uint32_t tsplit, uint32_t bx, uint32_t nblocks, int64_t (*rows)[4 + S]) {
  constexpr int LP = (S == 0) ? 4 : (S <= 4 ? 8 : 16);
  constexpr int L = 4 + S;
  // slots: [request classes | groups] or [pods | groups] (the group slots carry the first checks, core.go:136-147)
  const uint32_t nslots = (prm.use_classes ? (prm.k_host ? prm.k_host : __builtin_amdgcn_readfirstlane(*b.kclass)) : nslots_fixed) + ngroups_g;
  const uint32_t ntiles = (nslots + 63u) >> 6;
  if (!ntiles || !m) return;
  // tiles that can hold a query: all of them with request classes; otherwise the per-pod slots only if some pod has a query
  // of its own, the group slots only if some first check (core.go:136-147) was asked
  uint32_t t_lo = 0, t_hi = ntiles;
  if (!prm.use_classes && ngroups_g) {
    const uint32_t np = __builtin_amdgcn_readfirstlane(b.qcount[0]), ng = __builtin_amdgcn_readfirstlane(b.qcount[1]);
    if (np == 0) t_lo = nslots_fixed >> 6;
    if (ng == 0) t_hi = (nslots_fixed + 63u) >> 6;
    if (t_hi <= t_lo) return;
  }
  const uint32_t ntl = t_hi - t_lo;
  // tsplit > 1 (several tables in use): the distinct tables of a tile are dealt over tsplit waves as well
  // steady state (one table for every item, stamped slots): the four waves of a block work on ONE item, a quarter of every
  // group's rows each; elsewhere an item is one wave's
  constexpr bool uni = LOCAL && MODE != 0;
  constexpr bool split = uni && MODE == 2;          // few tiles: a block = one item; many tiles (every pod its own request): a wave = one item,
                                                   // as everywhere else — four waves per item would fetch every group four times
  const uint32_t nsub = split ? 4u : 1u, sub = split ? (uint32_t)__builtin_amdgcn_readfirstlane(wave_id()) : 0u;   // (wave-uniform: row numbers stay scalar)
  const uint32_t wpb = 4u / nsub;                 // items a block works on at a time
  const uint32_t J = max(1u, min(min(jcap, (m + 63u) >> 6), (nblocks * wpb) / (ntl * tsplit)));
  const uint32_t items = ntl * tsplit * J;
  const int lane = lane_id();
  const uint32_t w_first = __builtin_amdgcn_readfirstlane(split ? bx : bx * 4u + (uint32_t)wave_id());
  // table, stamp, request and flags of a tile's slots in ONE round trip (a dead slot's request is loaded for nothing)
  struct SlotLoad { int32_t tab; uint32_t stp, qf; int64_t r[L]; };
  auto load_slots = [&](uint32_t w, SlotLoad& sl) {
    const uint32_t rest = w / ntl, tile = t_lo + (w - rest * ntl);
    const uint32_t ps = min(tile * 64u + (uint32_t)lane, nslots - 1u);
    sl.tab = b.qtab_s[ps];
    sl.stp = prm.stamp ? b.qstamp_s[ps] : 0u;
    const int64_t* src = b.qreq_s + (size_t)ps * LP;
#pragma unroll
    for (int j = 0; j < L; ++j) sl.r[j] = src[j];
    sl.qf = b.qflags_s[ps];
  };
  // the first item's slots are asked for BEFORE the table's offsets / key rows / first pruning bounds are derived (once per
  // wave: loads, then wave scans that wait for them): one round trip for both, not two in a row
  SlotLoad first;
  LocalPre<S> pre = {};
  uint32_t w_live = w_first;                        // MODE 1: the first item of this wave that has work
  if constexpr (LOCAL) {
    if constexpr (uni && !split) {
      // (throughput regime) the wave's first item with a slot in use: two words per lane are looked at before anything else is asked
      // for.  A wave whose items are all idle — the tiles of other ranks' request classes under bs_shard_set: class ids follow the
      // queue (k_pod_class_ids), so 7 tiles of 8 are idle on a rank of 8 — leaves without the first fetch (~60 KB), and a rank can
      // cut its live tiles into as many shares as its part of the job allows (run_fast: share_b).
      while (w_live < items) {
        const uint32_t rest = w_live / ntl, tile = t_lo + (w_live - rest * ntl);
        const uint32_t ps = tile * 64u + (uint32_t)lane;
        const bool live = ps < nslots && b.qtab_s[ps] >= 0 && (!prm.stamp || b.qstamp_s[ps] == prm.stamp);
        if (__ballot(live)) break;
        w_live += nblocks * wpb;
      }
      if (w_live >= items) return;
      load_slots(w_live, first);
      local_pre_load<S>(b, prm, m, 0u, pre);
    } else if constexpr (split) {
      // The four waves of the block work on ONE item and need the SAME first fetch (the tile's slots, the table's chunk
      // totals / key rows, the maxima of its first 128 groups): 64 lanes x 64-byte strides, ~900 cache lines per wave — four
      // waves asking for all of it keep this CU's L1 busy for longer than the memory latency.  Each wave fetches a quarter,
      // leaves it in LDS, everybody reads it back.
      if (w_first < items) {                                    // (block-uniform: w_first = block index)
        __shared__ int64_t sh_req[64][L];
        __shared__ uint32_t sh_meta[3][64];
        __shared__ int64_t sh_gm[2][64][L];
        __shared__ unsigned long long sh_ct[64][L];
        __shared__ uint32_t sh_kv[S > 0 ? S : 1][64];
        const int wv = wave_id();
        if (wv == 0) {
          SlotLoad sl;
          load_slots(w_first, sl);
          sh_meta[0][lane] = (uint32_t)sl.tab; sh_meta[1][lane] = sl.stp; sh_meta[2][lane] = sl.qf;
#pragma unroll
          for (int j = 0; j < L; ++j) sh_req[lane][j] = sl.r[j];
        } else if (wv == 3) {
          LocalRaw<S> raw;
          local_fetch_chunks<S>(b, prm, m, 0u, raw);
#pragma unroll
          for (int j = 0; j < L; ++j) sh_ct[lane][j] = raw.v[j];
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) sh_kv[s2][lane] = raw.kv[s2];
        } else {
          int64_t gm[L];
          local_fetch_gmax<S>(b, prm, m, 0u, wv - 1, gm);
#pragma unroll
          for (int j = 0; j < L; ++j) sh_gm[wv - 1][lane][j] = gm[j];
        }
        __syncthreads();
        first.tab = (int32_t)sh_meta[0][lane]; first.stp = sh_meta[1][lane]; first.qf = sh_meta[2][lane];
        LocalRaw<S> raw;
#pragma unroll
        for (int j = 0; j < L; ++j) { first.r[j] = sh_req[lane][j]; raw.v[j] = sh_ct[lane][j]; raw.gm[0][j] = sh_gm[0][lane][j]; raw.gm[1][j] = sh_gm[1][lane][j]; }
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) raw.kv[s2] = sh_kv[s2][lane];
        local_pre_finish<S>(b, prm, m, 0u, raw, pre);
      }
    } else if (w_first < items) {
      load_slots(w_first, first);
    }
  } else if (w_first < items) {
    load_slots(w_first, first);
  }
  for (uint32_t w = w_live; w < items; w += nblocks * wpb) {
    const uint32_t rest = w / ntl, tile = t_lo + (w - rest * ntl);
    const uint32_t share = rest / tsplit, ts = rest - share * tsplit;
    const uint32_t pos = tile * 64u + (uint32_t)lane;
    SlotLoad cur;
    if (w == w_live) cur = first; else load_slots(w, cur);
    int32_t tab = cur.tab;
    const uint32_t stp = cur.stp;
    int64_t r[1][L];
#pragma unroll
    for (int j = 0; j < L; ++j) r[0][j] = cur.r[j];
    uint32_t qf = cur.qf;
    if (pos >= nslots || stp != prm.stamp) tab = -1;   // fast path: a slot is live iff a pod of THIS batch wrote it (stamp 0: every slot the pre-pass left >= 0)
    unsigned long long todo = __ballot(tab >= 0);
    BS_STAMP(2, 1);
    if (!todo) continue;
    if (tab < 0) {
      qf = 0;
#pragma unroll
      for (int j = 0; j < L; ++j) r[0][j] = INT64_MAX;
    }
    if (prm.collect_stats && share == 0 && ts == 0 && sub == 0 && lane == 0) atomicAdd((unsigned long long*)&b.stats[4], (unsigned long long)__popcll(todo));
    uint32_t turn = 0;
    while (todo) {
      const int32_t t0 = __builtin_amdgcn_readlane(tab, __ffsll((long long)todo) - 1);
      const bool member = tab == t0;
      if (turn == ts) scan_core<S, LOCAL, uni>(b, prm, m, (uint32_t)t0, pos, member, r, qf, share, J, rows, pre, sub, nsub);
      turn = turn + 1u == tsplit ? 0u : turn + 1u;
      todo &= ~__ballot(member);
    }
  }
}
template <int S, bool LOCAL = false>
__global__ __launch_bounds__(256) void k_scan(BatchDev b, BatchParams prm, uint32_t m, uint32_t jcap, uint32_t nslots_fixed, uint32_t ngroups_g,
                                              uint32_t tsplit) {
  __shared__ int64_t s_rows[4][64][4 + S];
  scan_loop<S, LOCAL>(b, prm, m, jcap, nslots_fixed, ngroups_g, tsplit, blockIdx.x, gridDim.x, s_rows[wave_id()]);
}

// ------------------------------------------------------------------------------------------------
// k_reject / k_final_a / k_final_b
// ------------------------------------------------------------------------------------------------
#if BS_EMIT_MAIN
__global__ void k_reject(PodsDev pods, BatchDev b) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  if (!(b.stage[i] & ST_QUERY)) return;
  if (b.first_row[b.qpos[i]] == BS_INF) {
    // compareClusterResourceAndRequire returned false: AddToDenyCache (core.go:142,163)
    b.tcode[i] = (b.tcode[i] == BS_PF_PASS_FIRST_FITS) ? BS_PF_REJECT_FIRST : BS_PF_REJECT_RESERVE;
    atomicMin(&b.first_reject[pods.group[i]], i);
  }
}
#endif

// Filter per-pod parameters (see the Filter section below); defined here because k_final fuses it.
template <int TS>
__device__ __forceinline__ void filter_params_for(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm,
                                                  uint32_t i, uint8_t pf, int32_t leader, uint32_t slot, bool write_slot, bool write_pod) {
  const Shape<TS> sh(prm.S);
  const uint32_t gate = prm.eph_gate;
  uint8_t fl = BS_FL_NOT_RUN;
  uint32_t ff = 0;
  int64_t R[4] = {0, 0, 0, 0}, M[4] = {0, 0, 0, 0};
  if (pf != BS_PF_NOT_OWNED && BS_PF_IS_PASS(pf)) {
    const int32_t gi = pods.group[i];
    if (gi == BS_POD_NOT_GROUPED) fl = BS_FL_PASS_NOT_GROUPED;                 // core.go:171-174
    else if (gi < 0 || (uint32_t)gi >= gr.g) fl = BS_FL_ERR_PG_NOT_FOUND;      // :177-180
    else if (leader < 0) fl = BS_FL_PANIC_NIL_MAX;                             // :525
    else {
      Res mr, ms;
      res_zero(ms, sh);
      const bool have = group_minres_at(gr, pods, b, (uint32_t)leader, i, sh, gate, mr);
      if (have) res_add(ms, mr, sh, gate);                                     // :526-527
      if (leader == gi) fl = BS_FL_PASS_IS_MAX;                                // :531-535
      else if (!have) fl = BS_FL_PASS_NO_MINRES;                               // :542-544
      else {
        fl = BS_FL_EVALUATED;
        Res cur;
        pod_require(pods, i, sh, gate, cur);                                   // :551
        res_add(cur, ms, sh, gate);                                            // :552
#pragma unroll
        for (int j = 0; j < 4; ++j) { R[j] = cur.v[j]; M[j] = ms.v[j]; }
#pragma unroll
        for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
          if (s < sh.S()) {
            if ((cur.present & (1u << s)) && cur.v[4 + s] != 0) ff |= 1u;      // case 2 can never hold
            if ((ms.present & (1u << s)) && ms.v[4 + s] != 0) ff |= 2u;        // node "cannot hold" a leader member
          }
        }
      }
    }
  }
  const uint32_t ffw = ff | ((uint32_t)fl << 8);
  if (write_pod) {
    b.fflags[i] = ffw;
    b.fl_code[i] = fl;
    b.fu_slot[i] = slot;
  }
  if (write_slot && fl == BS_FL_EVALUATED) {
    // every pod of the slot stores the same values (see BatchDev)
    b.uflags[slot] = ffw | (prm.stamp << 16);      // fast path: stamped instead of reset per batch (prm.stamp == 0 otherwise)
    if (prm.stamp) b.fu_feas[slot] = 0;
    int64_t* dst2 = b.uparams + (size_t)slot * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst2[j] = R[j]; dst2[4 + j] = M[j]; }
  }
}

// Did pod j really reach findMaxPG (core.go:118-123)?  Tentatively yes (k_query) and not behind the
// first rejected pod of its group (deny entry, core.go:105-110).  Non-owned pods: tentative value.
__device__ __forceinline__ bool reached_find_max(const PodsDev& pods, const BatchDev& b, uint32_t j) {
  const uint8_t st = b.stage[j];
  if (!(st & ST_REACH6)) return false;
  if ((st & ST_OWNED) && (st & ST_ELIG) && b.first_reject[pods.group[j]] < j) return false;
  return true;
}

// Final codes in queue order.  A pod behind the first rejected pod of its group meets the deny entry
// at core.go:105-110 and never gets further.
// pf_leader = sop.maxFinishedPG after the pod's PreFilter returned: the value findMaxPG produced for
// it, or — if the call returned before core.go:120 — what the latest earlier pod left there (carried in
// from before the batch when there is none): a prefix "last pod that reached findMaxPG".  Inside the
// block it is a wave/block scan; for the pods before the block it is a backward search that almost
// always ends at the block's immediate predecessor.  Fused: the Filter parameters of the pod.
template <int TS>
__global__ __launch_bounds__(256) void k_final(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm) {
  __shared__ uint32_t lds[16];
  const uint32_t base = blockIdx.x * 256u;
  // last pod before this block that reached findMaxPG (as index + 1, 0 = none)
  uint32_t prev = 0;
  for (uint32_t hi = base; hi > 0 && prev == 0;) {
    const uint32_t lo = hi >= 256u ? hi - 256u : 0u;
    const uint32_t j = lo + threadIdx.x;
    uint32_t cand = 0;
    if (j < hi && reached_find_max(pods, b, j)) cand = j + 1u;
    prev = block_max_u32(cand, lds);
    hi = lo;
  }
  const uint32_t i = base + threadIdx.x;
  uint32_t v = 0;
  uint8_t code = 0;
  if (i < pods.p) {
    code = b.tcode[i];
    const uint8_t st = b.stage[i];
    uint32_t fk = BS_K_NOT_SCANNED;
    if (st & ST_OWNED) {
      if ((st & ST_ELIG) && b.first_reject[pods.group[i]] < i) {
        code = BS_PF_ERR_DENIED;
      } else if (st & ST_QUERY) {
        const uint32_t row = b.first_row[b.qpos[i]];
        fk = row == BS_INF ? BS_K_NONE : nd.kmap[row];
      }
    } else {
      code = BS_PF_NOT_OWNED;
    }
    b.pf_code[i] = code;
    b.pf_first_k[i] = fk;
    if (reached_find_max(pods, b, i)) v = i + 1u;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)v, o);
    if (lane_id() >= o) v = max(v, u);
  }
  __syncthreads();
  if (lane_id() == 63) lds[wave_id()] = v;
  __syncthreads();
  uint32_t off = prev;
  for (int w = 0; w < wave_id(); ++w) off = max(off, lds[w]);
  const uint32_t jp1 = max(v, off);
  if (i < pods.p) {
    const int32_t leader = jp1 ? b.leader_raw[jp1 - 1u] : prm.sop_leader0;
    b.pf_leader[i] = leader;
    if (prm.run_filter && !prm.early_filter) {
      // Filter slot: the pod's request class, apart for the pods that still see the leader carried into the batch
      const uint32_t slot = prm.use_classes ? b.pclass[i] + (jp1 ? 0u : *b.kclass) : i;
      filter_params_for<TS>(pods, gr, b, prm, i, code, leader, slot, !prm.fuse_filter, true);
    }
  }
}

// stand-alone Filter parameters (bs_filter_one): pf_code / pf_leader given
#if BS_EMIT_MAIN
__global__ void k_filter_params(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  filter_params_for<-1>(pods, gr, b, prm, i, b.pf_code[i], b.pf_leader[i], i, true, true);
}
#endif

// Filter parameters BEFORE the node scan has run (no first-pod capture possible in this batch, so there
// is one findMaxPG result Lc for the whole batch).  What the scan can still change for a pod is only
// whether it passes PreFilter (REJECT_* / replayed ERR_DENIED), never its Filter inputs:
//   * sop.maxFinishedPG seen by pod i (core.go:524) is Lc once some pod at or before i has reached
//     findMaxPG, and the value carried into the batch before that.  Whether later pods reach findMaxPG
//     depends on rejections, but every pod that does writes the same Lc — and the FIRST pod that
//     tentatively reaches it really does (a replayed deny needs an earlier, reaching, rejected pod).
//   * the pod's own request and the leader's MinResources are batch inputs.
// So Filter runs for every tentatively passing pod concurrently with the scan; k_tally then voids the
// rows of the pods PreFilter turned down.
template <int TS>
__global__ void k_fparams_early(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  const uint8_t pf = (b.stage[i] & ST_OWNED) ? b.tcode[i] : (uint8_t)BS_PF_NOT_OWNED;
  const bool after = i >= b.nepochs[1];
  const int32_t leader = after ? b.leader_epoch[0] : prm.sop_leader0;
  const uint32_t slot = prm.use_classes ? b.pclass[i] + (after ? 0u : *b.kclass) : i;
  filter_params_for<TS>(pods, gr, b, prm, i, pf, leader, slot, true, true);
}

// ------------------------------------------------------------------------------------------------
// Filter: computeResourceSatisfied (core.go:514-564) for every (pod, node).
//   maxSingle = Resource{} + leader.MinResources                       (:524-528)
//   case 1 pod's group is the leader -> pass                            (:531-535)
//   maxSingle == nil -> pass                                            (:542-544)
//   left = getLeftResource(node): alloc - requested on the 4 fixed lanes, no scalar keys (:436-475)
//   case 2 left >= pod + maxSingle -> pass                              (:551-555)
//   case 3 !(left >= maxSingle)   -> pass ; else ErrorResourceNotEnough (:558-563)
// left has no scalar keys, so any non-zero scalar in a request fails compareResourceAndRequire
// (core.go:688-691) for every node: that is one per-pod bit (scalar_block / leader_block).
// ------------------------------------------------------------------------------------------------
// One wave = 64 consecutive pods x a range of 64-node blocks, two blocks at a time.
// Lanes are NODES while comparing: lane n holds left[n] (4 int64) of two node blocks in VGPRs, the
// pod's request R is wave-uniform (one s_load_dwordx8), and the EXEC-chained
//     s_mov_b64 exec, ok ; 4 x v_cmpx_le_i64 vcc, R[j], left[j]
// leaves in EXEC the 64 node-feasibility bits of case 2 for that pod — the compare result IS the
// bitmap word ("ballot") — which v_writelane (lane select in M0, data = exec_lo / exec_hi) drops
// into lane pp.  After 64 pods lane pp owns pod pp's words: lanes are PODS for the outputs, where
// the pod-independent case-3 mask is OR-ed in, non-evaluated pods get their constant word, the
// popcount accumulates the feasible-node count and the store is one coalesced 512 bytes per block.
// Per pod and node block: 3 SALU + 6 VALU, no branch.
// Lane subsets: when, for a whole pod tile, even the cluster-wide smallest `left` of a resource lane
// covers the tile's largest request, that lane's compare is true for every (pod, node) of the tile and is
// left out.  MASK bit j = lane j is compared.  (Typically one or two lanes bind; the others are free.)
#define BS_OPT_0(x) ""
#define BS_OPT_1(x) x
#define BS_OPT(flag, x) BS_OPT_##flag(x)
template <int MASK>
__device__ __forceinline__ void filter_pod2(uint32_t pp, const int64_t (&R)[4], unsigned long long ok0, unsigned long long ok1,
                                            const int64_t (&l0)[4], const int64_t (&l1)[4], uint32_t& vlo0, uint32_t& vhi0,
                                            uint32_t& vlo1, uint32_t& vhi1);
#define BS_DEF_FILTER_POD2(MASK, f0, f1, f2, f3)                                                                          \
  template <>                                                                                                             \
  __device__ __forceinline__ void filter_pod2<MASK>(uint32_t pp, const int64_t (&R)[4], unsigned long long ok0,          \
                                                    unsigned long long ok1, const int64_t (&l0)[4], const int64_t (&l1)[4], \
                                                    uint32_t& vlo0, uint32_t& vhi0, uint32_t& vlo1, uint32_t& vhi1) {       \
    asm volatile("s_mov_b32 m0, %[pp]\n\t"                                                                                \
                 "s_mov_b64 exec, %[ok0]\n\t"                                                                             \
                 BS_OPT(f0, "v_cmpx_le_i64 vcc, %[R0], %[a0]\n\t") BS_OPT(f1, "v_cmpx_le_i64 vcc, %[R1], %[a1]\n\t")       \
                 BS_OPT(f2, "v_cmpx_le_i64 vcc, %[R2], %[a2]\n\t") BS_OPT(f3, "v_cmpx_le_i64 vcc, %[R3], %[a3]\n\t")       \
                 "s_nop 3\n\t"                                                                                            \
                 "v_writelane_b32 %[vlo0], exec_lo, m0\n\t"                                                               \
                 "v_writelane_b32 %[vhi0], exec_hi, m0\n\t"                                                               \
                 "s_mov_b64 exec, %[ok1]\n\t"                                                                             \
                 BS_OPT(f0, "v_cmpx_le_i64 vcc, %[R0], %[b0]\n\t") BS_OPT(f1, "v_cmpx_le_i64 vcc, %[R1], %[b1]\n\t")       \
                 BS_OPT(f2, "v_cmpx_le_i64 vcc, %[R2], %[b2]\n\t") BS_OPT(f3, "v_cmpx_le_i64 vcc, %[R3], %[b3]\n\t")       \
                 "s_nop 3\n\t"                                                                                            \
                 "v_writelane_b32 %[vlo1], exec_lo, m0\n\t"                                                               \
                 "v_writelane_b32 %[vhi1], exec_hi, m0\n\t"                                                               \
                 "s_mov_b64 exec, -1"                                                                                     \
                 : [vlo0] "+v"(vlo0), [vhi0] "+v"(vhi0), [vlo1] "+v"(vlo1), [vhi1] "+v"(vhi1)                             \
                 : [pp] "s"(pp), [ok0] "s"(ok0), [ok1] "s"(ok1), [R0] "v"(R[0]), [R1] "v"(R[1]), [R2] "v"(R[2]),          \
                   [R3] "v"(R[3]), [a0] "v"(l0[0]), [a1] "v"(l0[1]), [a2] "v"(l0[2]), [a3] "v"(l0[3]), [b0] "v"(l1[0]),   \
                   [b1] "v"(l1[1]), [b2] "v"(l1[2]), [b3] "v"(l1[3])                                                      \
                 : "vcc");                                                                                                \
  }
BS_DEF_FILTER_POD2(1, 1, 0, 0, 0)
BS_DEF_FILTER_POD2(2, 0, 1, 0, 0)
BS_DEF_FILTER_POD2(3, 1, 1, 0, 0)
BS_DEF_FILTER_POD2(4, 0, 0, 1, 0)
BS_DEF_FILTER_POD2(5, 1, 0, 1, 0)
BS_DEF_FILTER_POD2(6, 0, 1, 1, 0)
BS_DEF_FILTER_POD2(7, 1, 1, 1, 0)
BS_DEF_FILTER_POD2(8, 0, 0, 0, 1)
BS_DEF_FILTER_POD2(9, 1, 0, 0, 1)
BS_DEF_FILTER_POD2(10, 0, 1, 0, 1)
BS_DEF_FILTER_POD2(11, 1, 1, 0, 1)
BS_DEF_FILTER_POD2(12, 0, 0, 1, 1)
BS_DEF_FILTER_POD2(13, 1, 0, 1, 1)
BS_DEF_FILTER_POD2(14, 0, 1, 1, 1)
BS_DEF_FILTER_POD2(15, 1, 1, 1, 1)

// the 64-pod loop of one step for a given lane subset; requests come from this wave's LDS slice
// PU = pods per step (their LDS reads are issued together): 4 everywhere except in the lean throughput-regime kernel (k_fast_filter),
// where the registers of two more requests are worth less than another resident wave
template <int MASK, int PU = 4>
__device__ __forceinline__ void filter_pod_loop(uint32_t np, const int64_t (*sR)[4], const unsigned long long (&okmask)[2],
                                                const int64_t (&l)[2][4], uint32_t (&vlo)[2], uint32_t (&vhi)[2]) {
  for (uint32_t pp = 0; pp < np; pp += PU) {
    int64_t R[PU][4];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const uint32_t pu = min(pp + (uint32_t)u, np - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) R[u][j] = ((MASK >> j) & 1) ? sR[pu][j] : 0;
    }
#pragma unroll
    for (int u = 0; u < PU; ++u)
      filter_pod2<MASK>(min(pp + (uint32_t)u, np - 1), R[u], okmask[0], okmask[1], l[0], l[1], vlo[0], vhi[0], vlo[1], vhi[1]);
  }
}

// stamp != 0 (fast path): a slot is in use iff the stamp in bits 16.. of its flags word is this batch's.
// DB: the node blocks of step w+NB are loaded during the pod loop of step w (double-buffered, 18 VGPRs); !DB: behind it (the lean
// throughput-regime kernel: other resident waves cover the round trip)
// REGS (k_fast_step_a's whole-step form): the lane's slot — flags word (stamp already resolved), R, M — comes in registers from the caller, nothing of
// the slot arrays is read, and the slot's feasible count is handed back (*cnt_out) instead of being added to fu_feas[] (the caller adds it once the
// block that zeroes the counters is through).
template <int NB, int PU = 4, bool DB = true, bool REGS = false>
__device__ __forceinline__ void filter_item(const NodesDev& nd, const BatchDev& b, uint32_t U, uint32_t ustride, uint32_t ptile,
                                            uint32_t w0, uint32_t w1, uint32_t stamp = 0, uint32_t ff_in = 0, const int64_t* R_in = nullptr,
                                            const int64_t* M_in = nullptr, uint32_t* cnt_out = nullptr) {
  static_assert(NB == 2, "the inner statement handles two node blocks");
  typedef const __attribute__((address_space(4))) uint32_t* cflag_t;
  const int lane = lane_id();
  const uint32_t p0 = ptile * 64u;
  const uint32_t np = min(64u, U - p0);

  // Is the leader's single-member request M the same for every evaluated pod of the tile?  (It is,
  // unless the tile straddles a first-pod capture.)  Then case 3 is one mask per node block.
  const bool mine = (uint32_t)lane < np;
  const uint32_t src = p0 + (uint32_t)lane;                       // lanes are request slots
  // Everything this item needs from memory before it can compare is issued in ONE round trip: the slot's flags word, its
  // request R and leader request M, the cluster-wide bounds, and the first two node blocks (they do not depend on the
  // slots; a tile without a live slot has loaded them for nothing, which costs less than a second trip costs the rest).
  uint32_t myff = (uint32_t)BS_FL_NOT_RUN << 8;
  int64_t M[4] = {0, 0, 0, 0}, myR[4] = {0, 0, 0, 0};
  if constexpr (REGS) {
    if (mine) {
      myff = ff_in;
#pragma unroll
      for (int j = 0; j < 4; ++j) { myR[j] = R_in[j]; M[j] = M_in[j]; }
    }
  } else if (mine) {
    myff = b.uflags[src];
    const int64_t* rs = b.uparams + (size_t)src * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { myR[j] = rs[j]; M[j] = rs[4 + j]; }
  }
  int64_t gl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gl[j] = nd.lglob[j];
  // node blocks are double-buffered: the loads of step w+NB are in flight during the pod loop of step w
  int64_t l[NB][4], ln[NB][4];
  uint8_t nfl[NB], nfln[NB];
  auto load_blocks = [&](uint32_t w, int64_t (&dst)[NB][4], uint8_t (&fl)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const uint32_t n = (w + nb) * 64u + (uint32_t)lane;
      const bool nvalid = (w + nb) < w1 && n < nd.n;
      fl[nb] = 0xFF;                                    // invalid
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[nb][j] = INT64_MIN;
      if (nvalid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[nb][j] = nd.left4[(size_t)j * nd.stride + n];
        fl[nb] = nd.flags[n];
      }
    }
  };
  load_blocks(w0, l, nfl);
  if (!REGS && stamp) myff = (myff >> 16) == stamp ? (myff & 0xFFFFu) : ((uint32_t)BS_FL_NOT_RUN << 8);
  const uint32_t myfl = myff >> 8;
  const bool ev = myfl == BS_FL_EVALUATED;
  if (!ev) {
#pragma unroll
    for (int j = 0; j < 4; ++j) M[j] = 0;
  }
  __shared__ int64_t s_R[4][64][4];               // per wave: the tile's requests (pod + maxSingle, fixed lanes)
  __builtin_amdgcn_wave_barrier();                // the previous item of this wave is done with its slice
  // which resource lanes can decide anything for this tile?  (nd.lglob: cluster-wide min[4] / max[4] of left
  // over the nodes Filter can evaluate)
  uint32_t lane_mask = 0;          // bit j: lane j must be compared
  bool tile_allfail = false;       // some lane fails for every (pod, node): case 2 never holds
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) s_R[wave_id()][lane][j] = myR[j];
    __builtin_amdgcn_wave_barrier();               // same wave writes and reads: LDS is in order, keep the compiler honest
    const bool c2pod = ev && !(myff & 1u);
    const unsigned long long c2mask = __ballot(c2pod);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // lane j is free when even the cluster's smallest left covers every request of the tile (ballots, no
      // 64-bit reductions); it fails everywhere when the largest left is below every request
      if (__ballot(c2pod && !(gl[j] >= myR[j]))) lane_mask |= 1u << j;
      if (c2mask && !__ballot(c2pod && gl[4 + j] >= myR[j])) tile_allfail = true;
    }
  }
  const unsigned long long evmask = __ballot(ev);
  if (!evmask) return;                             // no slot of this tile is in use
  bool uniformM = true;
  int64_t M0[4] = {0, 0, 0, 0};
  uint32_t lb0 = 0;
  if (evmask) {
    const int first = __ffsll((long long)evmask) - 1;
    bool same = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      M0[j] = __shfl(M[j], first);
      same = same && (M[j] == M0[j]);
    }
    lb0 = (uint32_t)__shfl((int)(myff & 2u), first);
    same = same && ((myff & 2u) == lb0);
    uniformM = __ballot(ev && !same) == 0;
  }

  crow_t FP = as_const_rows(b.uparams);
  cflag_t FF = (cflag_t)(uintptr_t)b.uflags;
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; w += NB) {
    if constexpr (DB) {
      if (w + NB < w1) load_blocks(w + NB, ln, nfln);
    }
    unsigned long long okmask[NB], in_range[NB], nlf[NB];
    uint32_t vlo[NB], vhi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const bool nvalid = nfl[nb] != 0xFF;
      const bool node_ok = nvalid && !(nfl[nb] & (BS_NODE_NIL | BS_NODE_NO_NODE));     // core.go:442-449
      in_range[nb] = __ballot(nvalid);
      okmask[nb] = __ballot(node_ok);
      // case 3 for the tile's common leader: nodes that cannot hold one leader member
      unsigned long long lf = 0;
      if (!lb0) lf = __ballot(l[nb][0] >= M0[0]) & __ballot(l[nb][1] >= M0[1]) & __ballot(l[nb][2] >= M0[2]) & __ballot(l[nb][3] >= M0[3]);
      nlf[nb] = okmask[nb] & ~lf;
      vlo[nb] = 0;
      vhi[nb] = 0;
    }
    if (uniformM) {
      // the tile's 64 requests sit in this wave's LDS slice; a uniform-address ds_read_b128 broadcasts one
      // into VGPRs (in-order LDS counter: the compiler pipelines four pods per step) — no global/scalar
      // memory latency inside the pod loop.  Only the lanes that can bind for this tile are compared.
      const int64_t (*sR)[4] = s_R[wave_id()];
      if (tile_allfail) {
        // case 2 impossible for every pod: words stay 0
      } else {
        switch (lane_mask) {
          case 0:                                     // every lane is free: case 2 holds on every evaluable node
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { vlo[nb] = (uint32_t)okmask[nb]; vhi[nb] = (uint32_t)(okmask[nb] >> 32); }
            break;
          case 1: filter_pod_loop<1, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 2: filter_pod_loop<2, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 3: filter_pod_loop<3, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 4: filter_pod_loop<4, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 5: filter_pod_loop<5, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 6: filter_pod_loop<6, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 7: filter_pod_loop<7, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 8: filter_pod_loop<8, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 9: filter_pod_loop<9, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 10: filter_pod_loop<10, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 11: filter_pod_loop<11, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 12: filter_pod_loop<12, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 13: filter_pod_loop<13, PU>(np, sR, okmask, l, vlo, vhi); break;
          case 14: filter_pod_loop<14, PU>(np, sR, okmask, l, vlo, vhi); break;
          default: filter_pod_loop<15, PU>(np, sR, okmask, l, vlo, vhi); break;
        }
      }
      // lanes are pods now: finish the word
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const unsigned long long c2 = ((unsigned long long)vhi[nb] << 32) | vlo[nb];
        unsigned long long word;
        if (ev) word = ((myff & 1u) ? 0ull : c2) | nlf[nb];
        else word = myfl < 16u ? in_range[nb] : 0ull;           // nil before any node lookup / error
        vlo[nb] = (uint32_t)word;
        vhi[nb] = (uint32_t)(word >> 32);
      }
    } else {
      // generic path: per-pod leader request (tile straddles a capture); plain ballots
      for (uint32_t pp = 0; pp < np; ++pp) {
        const uint32_t p = p0 + pp;
        uint32_t ff;
        int64_t fp[8];
        if constexpr (REGS) {                          // the slot of lane pp, broadcast (pp is uniform)
          ff = (uint32_t)__shfl((int)myff, (int)pp);
#pragma unroll
          for (int j = 0; j < 4; ++j) { fp[j] = __shfl(myR[j], (int)pp); fp[4 + j] = __shfl(M[j], (int)pp); }
        } else {
          ff = FF[p];
          if (stamp) ff = (ff >> 16) == stamp ? (ff & 0xFFFFu) : ((uint32_t)BS_FL_NOT_RUN << 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) fp[j] = FP[(size_t)p * 8 + j];
        }
        const uint32_t fl = ff >> 8;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          unsigned long long word;
          if (fl == BS_FL_EVALUATED) {
            unsigned long long c2 = 0, lf = 0;
            if (!(ff & 1u)) c2 = __ballot(l[nb][0] >= fp[0]) & __ballot(l[nb][1] >= fp[1]) & __ballot(l[nb][2] >= fp[2]) & __ballot(l[nb][3] >= fp[3]);
            if (!(ff & 2u)) lf = __ballot(l[nb][0] >= fp[4]) & __ballot(l[nb][1] >= fp[5]) & __ballot(l[nb][2] >= fp[6]) & __ballot(l[nb][3] >= fp[7]);
            word = okmask[nb] & (c2 | ~lf);
          } else {
            word = fl < 16u ? in_range[nb] : 0ull;
          }
          vlo[nb] = writelane_u32((uint32_t)word, pp, vlo[nb]);
          vhi[nb] = writelane_u32((uint32_t)(word >> 32), pp, vhi[nb]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if ((w + nb) < w1 && mine) {
        const unsigned long long word = ((unsigned long long)vhi[nb] << 32) | vlo[nb];
        cnt += (uint32_t)__popcll(word);
        b.fu_bitmap[(size_t)(w + nb) * ustride + p0 + lane] = word;
        if (b.h_rows && p0 + (uint32_t)lane < b.hstride) b.h_rows[(size_t)(w + nb) * b.hstride + p0 + lane] = word;   // latency mode: the row goes home as well
      }
    }
    if constexpr (DB) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        nfl[nb] = nfln[nb];
#pragma unroll
        for (int j = 0; j < 4; ++j) l[nb][j] = ln[nb][j];
      }
    } else {
      if (w + NB < w1) load_blocks(w + NB, l, nfl);
    }
  }
  if constexpr (REGS) *cnt_out = mine ? cnt : 0u;
  else if (mine && cnt) atomicAdd(&b.fu_feas[p0 + lane], cnt);
}

// Work loop over (tile of 64 request slots, run of node blocks).  The slot count is only known on the
// device, so the grid is fixed and every wave derives the split itself: as many node runs as it takes to
// give the whole grid something to do.  ustride = row stride of fu_bitmap (slot capacity).
template <int NB, int PU = 4, bool DB = true>
__device__ __forceinline__ void filter_loop(const PodsDev& pods, const NodesDev& nd, const BatchDev& b, uint32_t target_waves, uint32_t use_classes,
                                            uint32_t ustride, uint32_t collect_stats, uint32_t bx, uint32_t nblocks, uint32_t stamp = 0,
                                            uint32_t slots = 0) {
  const uint32_t U = slots ? slots : (use_classes ? 2u * __builtin_amdgcn_readfirstlane(*b.kclass) : pods.p);     // slots != 0: (view, class) slots of bs_epoch.hpp
  const uint32_t W = (nd.n + 63u) / 64u;
  if (!U || !W) return;
  const uint32_t tiles = (U + 63u) / 64u;
  uint32_t nsplit = max(1u, target_waves / tiles);
  nsplit = min(nsplit, max((W + 1u) / 2u, 1u));
  const uint32_t bpw = max(2u, (((W + nsplit - 1u) / nsplit + 1u) / 2u) * 2u);     // multiple of NB
  const uint32_t nchunk = (W + bpw - 1u) / bpw;
  const uint32_t items = tiles * nchunk;
  for (uint32_t it = __builtin_amdgcn_readfirstlane(bx * 4u + (uint32_t)wave_id()); it < items; it += nblocks * 4u) {
    const uint32_t chunk = it / tiles, tile = it - chunk * tiles;                   // neighbours share the node run
    if (collect_stats && chunk == 0) {
      const uint32_t sl = tile * 64u + (uint32_t)lane_id();
      const uint32_t uf = sl < U ? b.uflags[sl] : 0u;
      const unsigned long long evs = __ballot(sl < U && ((uf >> 8) & 0xFFu) == BS_FL_EVALUATED && (!stamp || (uf >> 16) == stamp));
      if (lane_id() == 0 && evs) atomicAdd((unsigned long long*)&b.stats[3], (unsigned long long)__popcll(evs));
    }
    filter_item<NB, PU, DB>(nd, b, U, ustride, tile, chunk * bpw, min(W, chunk * bpw + bpw), stamp);
  }
}
template <int NB>
__global__ __launch_bounds__(256) void k_filter(PodsDev pods, NodesDev nd, BatchDev b, uint32_t target_waves, uint32_t use_classes,
                                                uint32_t ustride, uint32_t collect_stats) {
  filter_loop<NB>(pods, nd, b, target_waves, use_classes, ustride, collect_stats, blockIdx.x, gridDim.x);
}

// Node scan and Filter evaluation in ONE launch (class mode): both only need what k_query left behind and
// are independent of each other, so the first `scan_blocks` blocks run the scan work loop and the rest the
// Filter work loop — one launch boundary less on the critical path, and the two latency chains overlap.
template <int S, bool LOCAL = false>
__global__ __launch_bounds__(256) void k_scan_filter(PodsDev pods, NodesDev nd, BatchDev b, BatchParams prm, uint32_t m, uint32_t jcap,
                                                     uint32_t nslots_fixed, uint32_t ngroups_g, uint32_t tsplit, uint32_t scan_blocks,
                                                     uint32_t filter_waves, uint32_t ustride) {
  __shared__ int64_t s_rows[4][64][4 + S];
  if (blockIdx.x < scan_blocks)
    scan_loop<S, LOCAL>(b, prm, m, jcap, nslots_fixed, ngroups_g, tsplit, blockIdx.x, scan_blocks, s_rows[wave_id()]);
  else
    filter_loop<2>(pods, nd, b, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x - scan_blocks, gridDim.x - scan_blocks);
}

// Early Filter ran on the tentative PreFilter verdict.  The framework never calls Filter for a pod that
// PreFilter turned down, so such a pod's result is void: code NOT_RUN, no feasible node (its slot row, if it
// has one of its own, is simply never referenced).
#if BS_EMIT_MAIN
__global__ __launch_bounds__(256) void k_void_rows(PodsDev pods, BatchDev b) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pods.p) return;
  const uint8_t pf = b.pf_code[i];
  if (pf != BS_PF_NOT_OWNED && BS_PF_IS_PASS(pf)) return;
  if (((b.fflags[i] >> 8) & 0xFFu) == BS_FL_NOT_RUN) return;   // the early pass already left it out
  b.fl_code[i] = BS_FL_NOT_RUN;
  b.fflags[i] = (uint32_t)BS_FL_NOT_RUN << 8;
}
#endif

// ------------------------------------------------------------------------------------------------
// k_tally / k_ready
// ------------------------------------------------------------------------------------------------
// Per-group admit counts with one atomic per distinct group per wave; the LAST block to finish
// (ticket) then evaluates the quorum predicate of Permit (core.go:303) for every group and re-arms
// the per-group minima for the next batch — no separate launches for either.
constexpr int kTallyBlock = 1024;

// Called by every thread of every participating block (`nblocks` of BLOCK threads, block index bx).
// feasible = nodes on which Filter passes for pod i (any non-zero value when Filter did not run).
template <int BLOCK>
__device__ __forceinline__ void tally_block(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, uint32_t i, uint32_t feasible,
                                            uint32_t do_ready, uint32_t rearm, uint32_t bx, uint32_t nblocks, uint32_t* s_last) {
  bool admit = false;
  uint32_t g = 0;
  if (i < pods.p) {
    const int32_t gi = pods.group[i];
    const uint8_t pf = b.pf_code[i];
    if (gi >= 0 && (uint32_t)gi < gr.g && pf != BS_PF_NOT_OWNED && BS_PF_IS_PASS(pf) && feasible > 0) {
      admit = true;
      g = (uint32_t)gi;
    }
  }
  wave_aggregated_add(b.admit, g, admit);
  // nobody reads the per-group minima any more in this batch: every block re-arms a slice of them
  if (rearm) {
    for (uint32_t gg = bx * BLOCK + threadIdx.x; gg < gr.g; gg += nblocks * BLOCK) {
      b.first_elig[gg] = BS_INF;
      b.first_owner[gg] = BS_INF;
      b.first_reject[gg] = BS_INF;
      b.first_pod[gg] = BS_INF;
      b.cap_epoch[gg] = (gr.flags[gg] & BS_GROUP_HAS_POD) ? 0u : BS_INF;     // what k_init would write (the batch was a what-if: flags unchanged)
    }
  }
  if (!do_ready) return;
  // publish, take a ticket (every wave drains its own atomics before the block-level hand-off)
  // the counters are agent-scope atomics and drained (vmcnt) before the ticket; the last block reads them with agent-scope
  // loads: no release / acquire fence (= an L2 write-back / invalidate per block on the 8-XCD part)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) *s_last = __hip_atomic_fetch_add(&b.ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1 ? 1u : 0u;
  __syncthreads();
  if (!*s_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(&b.ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t gg = threadIdx.x; gg < gr.g; gg += BLOCK) {
    const uint32_t have = gr.matched[gg] + __hip_atomic_load(&b.admit[gg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.ready[gg] = have >= (uint32_t)(gr.min_member[gg] - gr.status_scheduled[gg]) ? 1 : 0;
  }
}

// Nodes on which Filter passes for pod i, from its slot: pods Filter passes without looking at a node (not
// grouped, leader itself, no MinResources: core.go:171-174, :531-535, :542-544) pass on every list entry;
// pods it errors for, or never sees, pass nowhere.
__device__ __forceinline__ uint32_t pod_feasible(const NodesDev& nd, const BatchDev& b, uint32_t i) {
  const uint32_t fl = (b.fflags[i] >> 8) & 0xFFu;
  if (fl == BS_FL_EVALUATED) return b.fu_feas[b.fu_slot[i]];
  return fl < 16u ? nd.n : 0u;
}

// Per-pod feasible counts (from the Filter slots) and, with do_tally, the per-group admit counts / quorum.
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kTallyBlock) void k_tally(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, uint32_t run_filter, uint32_t do_tally,
                                                        uint32_t do_ready, uint32_t rearm) {
  __shared__ uint32_t s_last;
  const uint32_t i = blockIdx.x * kTallyBlock + threadIdx.x;
  uint32_t feasible = 1u;
  if (run_filter && i < pods.p) {
    feasible = pod_feasible(nd, b, i);
    b.fl_feasible[i] = feasible;
  }
  if (do_tally) tally_block<kTallyBlock>(pods, gr, b, i, feasible, do_ready, rearm, blockIdx.x, gridDim.x, &s_last);
}
#endif

// The pods x nodes bitmap, materialised ON REQUEST (bs_batch_read with fl_bitmap set): every pod's row from
// its slot's.  The batch itself never needs it: Filter's answer for (pod, node) is bit `node` of row
// fu_slot[pod] (exported by bs_batch_read as fl_rows / fl_slot), or "every node" / "no node" by fl_code.
// Pure streaming: P x ceil(N/64) words out.
constexpr int kExpandWords = 8;                    // bitmap words per thread (grid.y slices the row)
#if BS_EMIT_MAIN
__global__ __launch_bounds__(256) void k_filter_expand(PodsDev pods, NodesDev nd, BatchDev b, uint32_t words, uint32_t ustride) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pods.p) return;
  const uint32_t fl = (b.fflags[i] >> 8) & 0xFFu;
  const uint32_t w0 = blockIdx.y * kExpandWords;
  unsigned long long v[kExpandWords];
  if (fl == BS_FL_EVALUATED) {
    const uint32_t u = b.fu_slot[i];
#pragma unroll
    for (int k = 0; k < kExpandWords; ++k)       // all loads first: one memory round trip per thread
      v[k] = w0 + k < words ? b.fu_bitmap[(size_t)(w0 + k) * ustride + u] : 0ull;
  } else {
    const bool all = fl < 16u;
    const unsigned long long last = (nd.n & 63u) ? ((1ull << (nd.n & 63u)) - 1ull) : ~0ull;
#pragma unroll
    for (int k = 0; k < kExpandWords; ++k) v[k] = all ? (w0 + k + 1u == words ? last : ~0ull) : 0ull;
  }
#pragma unroll
  for (int k = 0; k < kExpandWords; ++k)
    if (w0 + k < words) b.fl_bitmap[(size_t)(w0 + k) * pods.p + i] = v[k];
}
#endif

// quorum predicate of Permit, core.go:303, with every admitted pod counted as matched (used after
// the cross-rank all-reduce of the admit counters)
#if BS_EMIT_MAIN
__global__ void k_ready(GroupsDev gr, BatchDev b) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= gr.g) return;
  const uint32_t have = gr.matched[g] + b.admit[g];
  b.ready[g] = have >= (uint32_t)(gr.min_member[g] - gr.status_scheduled[g]) ? 1 : 0;
}
#endif

// computeResourceSatisfied for pod 0 of a one-pod view and one node, with the exact case identity
#if BS_EMIT_MAIN
__global__ void k_filter_one(NodesDev nd, BatchDev b, uint32_t node, uint8_t* fn_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t ff = b.fflags[0];
  uint8_t fn = BS_FN_PASS_CASE2;
  if ((ff >> 8) == BS_FL_EVALUATED) {
    if (node >= nd.n || (nd.flags[node] & (BS_NODE_NIL | BS_NODE_NO_NODE))) fn = BS_FN_ERR_SNAPSHOT;   // core.go:545-548
    else {
      bool c2 = !(ff & 1u), lf = !(ff & 2u);
      for (int j = 0; j < 4; ++j) {
        const int64_t l = nd.left4[(size_t)j * nd.stride + node];
        c2 = c2 && l >= b.fparams[j];
        lf = lf && l >= b.fparams[4 + j];
      }
      fn = c2 ? BS_FN_PASS_CASE2 : (!lf ? BS_FN_PASS_CASE3 : BS_FN_ERR_NOT_ENOUGH);                     // :553-563
    }
  }
  *fn_out = fn;
}
#endif

// BS_BATCH_COMMIT: persist what the sequential PreFilter calls would have left in the cache —
// first-pod capture and MinResources default (core.go:486-493), OccupiedBy (:494-500), deny entry
// (:142,:163).  A pod behind its group's first rejection never reaches fillOccupiedObj.
#if BS_EMIT_MAIN
__global__ void k_commit(PodsDev pods, BatchDev b, BatchParams prm, uint8_t* gflags, uint32_t* gcls, int64_t* gminres,
                         uint32_t* gmrpres, uint64_t* gocc, uint32_t G, const uint32_t* gate) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G || (gate && *gate)) return;            // (gate: BS_BATCH_FILTER_DENY's flag word — only the fixed point commits, bs_fdeny.hpp)
  const uint32_t fe = b.first_elig[g], fr = b.first_reject[g];
  uint8_t fl = gflags[g];
  if (fe != BS_INF) {
    if (!(fl & BS_GROUP_HAS_POD)) { fl |= BS_GROUP_HAS_POD; gcls[g] = pods.cls[fe]; }
    if (!(fl & BS_GROUP_HAS_MINRES)) {
      Res r;
      pod_require(pods, fe, Shape<-1>(prm.S), prm.eph_gate, r);
      for (uint32_t j = 0; j < prm.L; ++j) gminres[(size_t)j * G + g] = r.v[j];
      gmrpres[g] = r.present;
      fl |= BS_GROUP_HAS_MINRES;
    }
    if (gocc[g] == 0) {
      const uint32_t fo = b.first_owner[g];
      if (fo != BS_INF && fo <= fr) gocc[g] = pods.owner[fo];
    }
  }
  if (fr != BS_INF) fl |= BS_GROUP_DENIED;
  gflags[g] = fl;
}
#endif

// ------------------------------------------------------------------------------------------------
// single-query helpers (bs_node_left, bs_cluster_total)
// ------------------------------------------------------------------------------------------------
#if BS_EMIT_MAIN
__global__ void k_node_left(NodesDev nd, uint32_t cls, float pct, uint32_t L, int64_t* left, uint32_t* present) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nd.n) return;
  const uint32_t* fitrow = nd.fit + (size_t)cls * nd.fit_words;
  const bool fit = ((fitrow[n >> 5] >> (n & 31u)) & 1u) && !(nd.flags[n] & BS_NODE_TAINT_ERR);
  const uint32_t pres = fit ? (nd.apres[n] & nd.rpres[n]) : 0u;
  for (uint32_t j = 0; j < L; ++j) {
    int64_t v = 0;
    if (fit && (j < 4 || (pres & (1u << (j - 4)))))
      v = wsub(scale_f32(nd.alloc[(size_t)j * nd.stride + n], pct), nd.req[(size_t)j * nd.stride + n]);
    left[(size_t)j * nd.n + n] = v;
  }
  present[n] = pres;
}
#endif

}  // namespace bs
