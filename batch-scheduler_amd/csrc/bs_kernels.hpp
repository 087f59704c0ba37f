// bs_kernels.hpp — gfx950 kernels of the batched PreFilter / Filter / Permit path.
//
// Pipeline of one batch (bs_batch_run), all on one HIP stream:
//   k_init      reset per-batch scratch
//   k_prepass   per pod: eligibility (core.go:89-110), first eligible pod / first owner per group
//   k_epochs    queue-order prefix count of first-pod captures (core.go:486-488) -> epoch per pod
//   k_leader    findMaxPG (core.go:701-739) once per epoch (candidate set grows with captures)
//   k_query     per pod: fillOccupiedObj check, branch A/B/C/D of core.go:127-166, request vector
//               (getPreAllocatedResource core.go:774-793 [+ pod request :157-159]) and scan table id
//   k_plan/k_scatter   bucket queries by (fit class, percent) table, build 64-query tiles
//   k_tables    singleNodeResource (core.go:634-670) + running sums of core.go:602,621 per table
//   k_scan      THE hot kernel: exists k : prefix_k >= request (core.go:623), first such k
//   k_reject/k_final   REJECT codes, deny-cache replay in queue order (core.go:105-110,142,163),
//               stale sop.maxFinishedPG propagation, first_k -> node list index
//   k_filter_params/k_filter   computeResourceSatisfied (core.go:514-564) pods x nodes bitmap
//   k_tally/k_ready    per-group admit counts and the quorum predicate core.go:303
//
// No MFMA anywhere: this is int64 compare/add work (north_star).  Lanes of a wave are pods
// (queries); node rows are wave-uniform and arrive through the scalar cache, so one 64-bit
// v_cmp per resource lane decides 64 pod x node pairs.
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
struct Tile { uint32_t slot, q0, count, pad; };

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
  uint32_t* nepochs;        // [1] E + 1
  int32_t* leader_epoch;    // [E+1]
  uint8_t* panic_epoch;     // [E+1]
  // per pod
  uint8_t* tcode;           // tentative PreFilter code
  uint8_t* stage;
  int32_t* leader_raw;      // leader findMaxPG returned for this pod (valid iff ST_REACH6)
  int32_t* qtable;          // scan table id (class + C * (pct==0.7)), -1 none
  int64_t* qreq;            // [P][LP] effective request (absent scalar -> INT64_MIN)
  uint32_t* qflags;         // bits 0..11 request key present, bits 16..27 "zero/absent" (passes w/o left key)
  uint32_t* first_row;      // [P] min table row satisfying the request (INF none)
  // tables / tiles
  uint32_t* tbl_count;      // [2C]
  uint32_t* tbl_off;        // [2C+1]
  uint32_t* tbl_cursor;     // [2C]
  int32_t* tbl_slot;        // [2C] slot in `tables`, -1 not needed
  TableDesc* desc;          // [slots]
  uint32_t* ntables;        // [1]
  Tile* tiles;
  uint32_t* ntiles;         // [1]
  uint32_t* qlist;          // [P] pod indices grouped by table
  int64_t* tables;          // [slots][mcap][LP] running sums, row-major (one s_load per row)
  uint32_t* kp;             // [slots][16] first row at which scalar key s exists in the running sum
  uint64_t* stats;          // [8] counters (only touched when collect_stats)
  // filter
  int64_t* fparams;         // [P][8]: R[4] = pod + maxSingle, M[4] = maxSingle  (fixed lanes)
  uint32_t* fflags;         // [P] bit0 scalar_block (case 2 impossible), bit1 leader_block, bits 8.. fl_code
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
  uint32_t collect_stats;
  uint32_t mcap;               // table row capacity
  uint32_t seg_len;            // rows per scan segment
};

// ------------------------------------------------------------------------------------------------
// snapshot-derived data (at bs_nodes_load / bs_nodes_apply)
// ------------------------------------------------------------------------------------------------

// kmap: stable compaction of the nodes compareClusterResourceAndRequire does not skip
// (core.go:606-617); left4: getLeftResource lanes (core.go:460-463).  Single block.
__global__ __launch_bounds__(kScanBlock) void k_nodes_derive(NodesDev nd, uint32_t* kmap, uint32_t* m_out,
                                                             int64_t* left4) {
  __shared__ uint32_t lds[16];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nd.n; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    uint32_t keep = 0;
    if (i < nd.n) {
      keep = (nd.flags[i] & BS_NODE_SKIP_MASK) ? 0u : 1u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        left4[(size_t)j * nd.stride + i] = wsub(nd.alloc[(size_t)j * nd.stride + i], nd.req[(size_t)j * nd.stride + i]);
    }
    uint32_t total;
    const uint32_t incl = block_incl_scan_add<uint32_t>(keep, lds, total);
    if (keep) kmap[carry + incl - 1] = i;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *m_out = carry;
}

// ------------------------------------------------------------------------------------------------
// k_init
// ------------------------------------------------------------------------------------------------
__global__ void k_init(GroupsDev gr, BatchDev b, BatchParams prm, uint32_t P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < gr.g) {
    b.first_elig[t] = BS_INF;
    b.first_owner[t] = BS_INF;
    b.first_reject[t] = BS_INF;
    b.first_pod[t] = BS_INF;
    b.cap_epoch[t] = (gr.flags[t] & BS_GROUP_HAS_POD) ? 0u : BS_INF;
    b.admit[t] = 0;
  }
  if (t < 2 * prm.C) {
    b.tbl_count[t] = 0;
    b.tbl_cursor[t] = 0;
  }
  if (t < P) b.first_row[t] = BS_INF;
  if (t < 8 && prm.collect_stats) b.stats[t] = 0;
}

// ------------------------------------------------------------------------------------------------
// k_prepass: steps of PreFilter that do not need any other pod (core.go:89-110)
// ------------------------------------------------------------------------------------------------
__global__ void k_prepass(PodsDev pods, GroupsDev gr, BatchDev b) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  const int32_t gi = pods.group[i];
  uint8_t st = 0;
  if (gi >= 0 && (uint32_t)gi < gr.g) {
    atomicMin(&b.first_pod[gi], i);
    const bool permitted = pods.flags[i] & BS_POD_LAST_PERMITTED;
    const bool denied0 = gr.flags[gi] & BS_GROUP_DENIED;
    if (!permitted && !denied0) {
      st = ST_ELIG;
      atomicMin(&b.first_elig[gi], i);
      if (pods.owner[i] != 0 && gr.occupied[gi] == 0) atomicMin(&b.first_owner[gi], i);
    }
  }
  b.stage[i] = st;
}

// ------------------------------------------------------------------------------------------------
// k_epochs: epoch[i] = number of first-pod captures (pgs.Pod = pod, core.go:486-488) at queue
// positions <= i.  findMaxPG skips groups without a pod (core.go:709-711), so its candidate set —
// and therefore the leader — can only change at these positions.  Single block, ordered chunks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanBlock) void k_epochs(PodsDev pods, GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < pods.p; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    uint32_t cap = 0;
    int32_t gi = -1;
    if (i < pods.p && (b.stage[i] & ST_ELIG)) {
      gi = pods.group[i];
      cap = (b.first_elig[gi] == i && !(gr.flags[gi] & BS_GROUP_HAS_POD)) ? 1u : 0u;
    }
    uint32_t total;
    const uint32_t incl = block_incl_scan_add<uint32_t>(cap, lds, total);
    if (i < pods.p) b.epoch[i] = carry + incl;
    if (cap) b.cap_epoch[gi] = carry + incl;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *b.nepochs = carry + 1;
}

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

__global__ __launch_bounds__(256) void k_leader(GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  const uint32_t e = blockIdx.x;
  if (e >= *b.nepochs) return;
  bool panic = false;
  uint32_t fmax = 0, any = 0;
  for (uint32_t g = threadIdx.x; g < gr.g; g += blockDim.x) {
    if (!leader_candidate(gr, b, g, e)) continue;
    any = 1;
    fmax = max(fmax, leader_finished(gr, g, panic));
  }
  const uint32_t any_panic = block_max_u32(panic ? 1u : 0u, lds);
  const uint32_t any_cand = block_max_u32(any, lds);
  const uint32_t F = block_max_u32(fmax, lds);
  if (any_panic) {
    if (threadIdx.x == 0) { b.leader_epoch[e] = -1; b.panic_epoch[e] = 1; }
    return;
  }
  if (!any_cand) {
    if (threadIdx.x == 0) { b.leader_epoch[e] = -1; b.panic_epoch[e] = 0; }
    return;
  }
  // first candidate tied at F
  uint32_t first = BS_INF;
  for (uint32_t g = threadIdx.x; g < gr.g; g += blockDim.x) {
    bool p2 = false;
    if (leader_candidate(gr, b, g, e) && leader_finished(gr, g, p2) == F) { first = g; break; }
  }
  uint32_t cur = block_min_u32(first, lds);
  for (;;) {
    if (!(gr.status_scheduled[cur] >= gr.min_member[cur])) break;   // holder not fully scheduled: stays
    uint32_t nxt = BS_INF;
    for (uint32_t g = threadIdx.x; g < gr.g; g += blockDim.x) {
      if (g <= cur) continue;
      bool p2 = false;
      if (leader_candidate(gr, b, g, e) && gr.status_scheduled[g] == 0u && leader_finished(gr, g, p2) == F) { nxt = g; break; }
    }
    nxt = block_min_u32(nxt, lds);
    if (nxt == BS_INF) break;
    cur = nxt;
  }
  if (threadIdx.x == 0) { b.leader_epoch[e] = (int32_t)cur; b.panic_epoch[e] = 0; }
}

// ------------------------------------------------------------------------------------------------
// k_query
// ------------------------------------------------------------------------------------------------
struct Res {            // upstream nodeinfo.Resource flattened
  int64_t v[BS_MAX_LANES];
  uint32_t present;
};

__device__ __forceinline__ void res_zero(Res& r, uint32_t L) {
  for (uint32_t j = 0; j < L; ++j) r.v[j] = 0;
  r.present = 0;
}
// Resource.Add(ResourceList): ephemeral-storage only behind the feature gate; scalar keys are created.
__device__ __forceinline__ void res_add(Res& r, const Res& rl, uint32_t S, uint32_t gate) {
  r.v[0] = wadd(r.v[0], rl.v[0]);
  r.v[1] = wadd(r.v[1], rl.v[1]);
  if (gate) r.v[2] = wadd(r.v[2], rl.v[2]);
  r.v[3] = wadd(r.v[3], rl.v[3]);
  for (uint32_t s = 0; s < S; ++s)
    if (rl.present & (1u << s)) { r.v[4 + s] = wadd(r.v[4 + s], rl.v[4 + s]); r.present |= 1u << s; }
}
// getPodResourceRequire(pod): lanes handed over by the shim, normalised through Add
__device__ __forceinline__ void pod_require(const PodsDev& pods, uint32_t i, uint32_t L, uint32_t S, uint32_t gate, Res& out) {
  Res raw;
  for (uint32_t j = 0; j < L; ++j) raw.v[j] = pods.req[(size_t)j * pods.p + i];
  raw.present = pods.pres[i];
  res_zero(out, L);
  res_add(out, raw, S, gate);
}
// Spec.MinResources of group g as pod i sees it: the loaded value, or (MinResources == nil at load)
// the request of the first pod that reached fillOccupiedObj (core.go:489-493), or nil.
__device__ __forceinline__ bool group_minres_at(const GroupsDev& gr, const PodsDev& pods, const BatchDev& b, uint32_t g,
                                                uint32_t i, uint32_t L, uint32_t S, uint32_t gate, Res& out) {
  if (gr.flags[g] & BS_GROUP_HAS_MINRES) {
    for (uint32_t j = 0; j < L; ++j) out.v[j] = gr.minres[(size_t)j * gr.g + g];
    out.present = gr.mrpres[g];
    return true;
  }
  const uint32_t fe = b.first_elig[g];
  if (fe <= i) { pod_require(pods, fe, L, S, gate, out); return true; }
  return false;
}
__device__ __forceinline__ uint32_t group_cls_at(const GroupsDev& gr, const PodsDev& pods, const BatchDev& b, uint32_t g) {
  if (gr.flags[g] & BS_GROUP_HAS_POD) return gr.cls[g];
  return pods.cls[b.first_elig[g]];
}
// getPreAllocatedResource, core.go:774-793 (repeated Add == wrapping multiply)
__device__ __forceinline__ void pre_allocated(const GroupsDev& gr, uint32_t g, int64_t matched, bool have_mr, const Res& mr,
                                              uint32_t L, uint32_t S, uint32_t gate, Res& out) {
  res_zero(out, L);
  const int64_t mm = (int64_t)gr.min_member[g];
  const int64_t not_finished = matched != 0 ? mm - matched : mm - (int64_t)gr.status_scheduled[g];
  if (not_finished > 0 && have_mr) {
    Res times;
    for (uint32_t j = 0; j < L; ++j) times.v[j] = wmul(mr.v[j], not_finished);
    times.present = mr.present;
    res_add(out, times, S, gate);
  }
  if (out.v[BS_LANE_PODS] == 0) out.v[BS_LANE_PODS] = mm + 1;
}

__global__ void k_query(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < pods.p;
  const uint32_t L = prm.L, S = prm.S, gate = prm.eph_gate;
  uint8_t code = BS_PF_PASS_NOT_GROUPED, st = 0;
  int32_t leader = -1, table = -1;
  Res q;
  res_zero(q, L);
  if (valid) {
    st = b.stage[i];
    const int32_t gi = pods.group[i];
    // shard ownership: all pods of a group live on the rank of the group's first pod
    uint32_t anchor = i;
    if (gi >= 0 && (uint32_t)gi < gr.g) anchor = b.first_pod[gi];
    const uint32_t owner_rank = (uint32_t)(((uint64_t)anchor * prm.nranks) / pods.p);
    if (owner_rank == prm.rank) st |= ST_OWNED;

    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                       // core.go:89-92
    else if (pods.flags[i] & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;    // :95-98
    else if (gi < 0 || (uint32_t)gi >= gr.g) code = BS_PF_ERR_PG_NOT_FOUND;              // :100-103
    else if (gr.flags[gi] & BS_GROUP_DENIED) code = BS_PF_ERR_DENIED;                    // :105-110
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
              const bool have = group_minres_at(gr, pods, b, g, i, L, S, gate, mr);
              pre_allocated(gr, g, 0, have, mr, L, S, gate, q);
              table = (int32_t)group_cls_at(gr, pods, b, g);                             // pct 1
              code = BS_PF_PASS_FIRST_FITS;                                              // tentative
            } else if (leader == gi) {
              code = BS_PF_PASS_IS_MAX;                                                  // :150-155
            } else {                                                                     // :157-166
              const bool have = group_minres_at(gr, pods, b, (uint32_t)leader, i, L, S, gate, mr);
              pre_allocated(gr, (uint32_t)leader, matched, have, mr, L, S, gate, q);
              Res cur;
              pod_require(pods, i, L, S, gate, cur);
              res_add(q, cur, S, gate);
              table = (int32_t)(prm.C + group_cls_at(gr, pods, b, (uint32_t)leader));    // pct 0.7
              code = BS_PF_PASS_RESERVE_FITS;                                            // tentative
            }
          }
        }
      }
    }
    if (!(st & ST_OWNED)) table = -1;
    if (table >= 0) {
      st |= ST_QUERY;
      uint32_t absok = 0;
      for (uint32_t s = 0; s < S; ++s) {
        const bool pres = q.present & (1u << s);
        if (!pres || q.v[4 + s] == 0) absok |= 1u << s;       // core.go:688-692
        if (!pres) q.v[4 + s] = INT64_MIN;                    // key not requested: never constrains
      }
      int64_t* dst = b.qreq + (size_t)i * prm.LP;
      for (uint32_t j = 0; j < L; ++j) dst[j] = q.v[j];
      for (uint32_t j = L; j < prm.LP; ++j) dst[j] = INT64_MIN;
      b.qflags[i] = q.present | (absok << 16);
    }
    b.tcode[i] = code;
    b.stage[i] = st;
    b.leader_raw[i] = leader;
    b.qtable[i] = table;
  }
  // one atomic per distinct table per wave
  wave_aggregated_inc(b.tbl_count, (uint32_t)(table < 0 ? 0 : table), valid && table >= 0);
}

// ------------------------------------------------------------------------------------------------
// k_plan: offsets per table, table slots + descriptors, 64-query tiles.  Single block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanBlock) void k_plan(BatchDev b, BatchParams prm) {
  __shared__ uint32_t lds[16];
  __shared__ uint32_t s_carry[3];
  const uint32_t T = 2 * prm.C;
  if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; s_carry[2] = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < T; base += kScanBlock) {
    const uint32_t t = base + threadIdx.x;
    const uint32_t cnt = t < T ? b.tbl_count[t] : 0u;
    const uint32_t need = cnt ? 1u : 0u;
    const uint32_t ntile = (cnt + 63u) / 64u;
    uint32_t tot_c, tot_n, tot_t;
    const uint32_t inc_c = block_incl_scan_add<uint32_t>(cnt, lds, tot_c);
    const uint32_t inc_n = block_incl_scan_add<uint32_t>(need, lds, tot_n);
    const uint32_t inc_t = block_incl_scan_add<uint32_t>(ntile, lds, tot_t);
    const uint32_t c0 = s_carry[0], n0 = s_carry[1], t0 = s_carry[2];
    if (t < T) {
      const uint32_t off = c0 + inc_c - cnt;
      b.tbl_off[t] = off;
      if (need) {
        const uint32_t slot = n0 + inc_n - 1;
        b.tbl_slot[t] = (int32_t)slot;
        TableDesc d;
        d.cls = t % prm.C;
        d.pct = t < prm.C ? 1.0f : 0.7f;        // core.go:140 (percent 1) / :161 (percent 0.7)
        b.desc[slot] = d;
        const uint32_t tile0 = t0 + inc_t - ntile;
        for (uint32_t k = 0; k < ntile; ++k) {
          Tile tl;
          tl.slot = slot;
          tl.q0 = off + 64u * k;
          tl.count = min(64u, cnt - 64u * k);
          tl.pad = 0;
          b.tiles[tile0 + k] = tl;
        }
      } else {
        b.tbl_slot[t] = -1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_carry[0] = c0 + tot_c; s_carry[1] = n0 + tot_n; s_carry[2] = t0 + tot_t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    b.tbl_off[T] = s_carry[0];
    *b.ntables = s_carry[1];
    *b.ntiles = s_carry[2];
  }
}

__global__ void k_scatter(PodsDev pods, BatchDev b) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < pods.p;
  const int32_t table = valid ? b.qtable[i] : -1;
  const uint32_t slot = wave_aggregated_inc(b.tbl_cursor, (uint32_t)(table < 0 ? 0 : table), table >= 0);
  if (table >= 0) b.qlist[b.tbl_off[table] + slot] = i;
}

// ------------------------------------------------------------------------------------------------
// k_tables: one block per table slot.  Row k (k-th non-skipped node in list order) holds the running
// sum leftResources after that node (core.go:602,621):
//   left = fit && !taint_err ? int64(float32(alloc)*pct) - requested : 0         (core.go:634-670)
//   scalar lane s contributes only when both allocatable and requested carry the key (:662-668)
// kp[s] = first row at which the running sum owns scalar key s.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanBlock) void k_tables(NodesDev nd, BatchDev b, BatchParams prm) {
  __shared__ unsigned long long lds64[16];
  __shared__ uint32_t s_kp[BS_MAX_SCALARS];
  const uint32_t slot = blockIdx.x;
  if (slot >= *b.ntables) return;
  const TableDesc d = b.desc[slot];
  const uint32_t L = prm.L, S = prm.S, LP = prm.LP;
  int64_t* T = b.tables + (size_t)slot * prm.mcap * LP;
  if (threadIdx.x < BS_MAX_SCALARS) s_kp[threadIdx.x] = BS_INF;
  __syncthreads();
  unsigned long long carry[BS_MAX_LANES];
  for (uint32_t j = 0; j < L; ++j) carry[j] = 0;
  const uint32_t* fitrow = nd.fit + (size_t)d.cls * nd.fit_words;
  for (uint32_t base = 0; base < nd.m; base += kScanBlock) {
    const uint32_t k = base + threadIdx.x;
    const bool valid = k < nd.m;
    uint32_t n = 0, pres = 0;
    bool fit = false;
    if (valid) {
      n = nd.kmap[k];
      fit = ((fitrow[n >> 5] >> (n & 31u)) & 1u) && !(nd.flags[n] & BS_NODE_TAINT_ERR);
      if (fit) pres = nd.apres[n] & nd.rpres[n];
    }
    for (uint32_t j = 0; j < L; ++j) {
      unsigned long long left = 0;
      const bool lane_live = fit && (j < 4 || (pres & (1u << (j - 4))));
      if (lane_live && !(j == BS_LANE_EPH && !prm.eph_gate))
        left = (unsigned long long)wsub(scale_f32(nd.alloc[(size_t)j * nd.stride + n], d.pct), nd.req[(size_t)j * nd.stride + n]);
      unsigned long long total;
      const unsigned long long incl = block_incl_scan_add<unsigned long long>(left, lds64, total);
      if (valid) T[(size_t)k * LP + j] = (int64_t)(carry[j] + incl);
      carry[j] += total;
    }
    if (valid) {
      for (uint32_t j = L; j < LP; ++j) T[(size_t)k * LP + j] = INT64_MAX;
      for (uint32_t s = 0; s < S; ++s)
        if (pres & (1u << s)) atomicMin(&s_kp[s], k);
    }
    __syncthreads();
  }
  if (threadIdx.x < BS_MAX_SCALARS) b.kp[slot * 16 + threadIdx.x] = s_kp[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// k_scan — the dominant kernel.
// One wave = one tile of <=64 queries (pods) that share a table, x one segment of table rows.
// Lane l holds query l's request lanes in VGPRs; the loop walks rows k (wave-uniform), whose running
// sums arrive as scalar loads; per resource lane ONE v_cmp_ge_i64 (VGPR vs SGPR) decides 64 pod x
// node pairs and lands as a 64-bit lane mask in SGPRs; masks are ANDed on the scalar unit
// (compareResourceAndRequire core.go:672-699).  A lane records the first row whose mask bit is set
// (the reference's early exit, core.go:623-627); the wave leaves when every lane has one.
// Scalar key rule (core.go:686-697): for rows before kp[s] the running sum has no key s, the lane
// passes iff it requests nothing of s (precomputed bit); from kp[s] on it is a plain compare.
// Segments of one tile combine through atomicMin on first_row.
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_scan(BatchDev b, BatchParams prm, uint32_t m) {
  constexpr int LP = (S == 0) ? 4 : (S <= 4 ? 8 : 16);
  constexpr int L = 4 + S;
  const int lane = lane_id();
  const uint32_t tile_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (uint32_t)wave_id());
  if (tile_id >= *b.ntiles) return;
  const uint32_t k0 = blockIdx.y * prm.seg_len;
  if (k0 >= m) return;
  const uint32_t k1 = min(m, k0 + prm.seg_len);
  const Tile tl = b.tiles[tile_id];
  const uint32_t slot = __builtin_amdgcn_readfirstlane(tl.slot);
  const uint32_t q0 = __builtin_amdgcn_readfirstlane(tl.q0);
  const uint32_t cnt = __builtin_amdgcn_readfirstlane(tl.count);

  const bool valid = (uint32_t)lane < cnt;
  const uint32_t pod = valid ? b.qlist[q0 + (uint32_t)lane] : 0u;
  int64_t r[L];
  uint32_t qf = 0;
  if (valid) {
    const int64_t* src = b.qreq + (size_t)pod * LP;
#pragma unroll
    for (int j = 0; j < L; ++j) r[j] = src[j];
    qf = b.qflags[pod];
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) r[j] = INT64_MAX;
  }
  // lanes that already have an earlier row from another segment need nothing from this one
  const uint32_t seen = valid ? __hip_atomic_load(&b.first_row[pod], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  unsigned long long found = __ballot(!valid || seen < k0);
  if (found == ~0ull) return;

  unsigned long long absok[S > 0 ? S : 1];
  uint32_t kp[S > 0 ? S : 1];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    absok[s] = __ballot((qf >> (16 + s)) & 1u);
    kp[s] = __builtin_amdgcn_readfirstlane(b.kp[slot * 16 + s]);
  }

  const int64_t* __restrict__ T = b.tables + (size_t)slot * prm.mcap * LP;
  uint32_t myk = BS_INF;
  uint32_t k = k0;
  for (; k < k1; ++k) {
    const int64_t* __restrict__ row = T + (size_t)k * LP;
    unsigned long long mk = __ballot(row[0] >= r[0]);
    mk &= __ballot(row[1] >= r[1]);
    mk &= __ballot(row[2] >= r[2]);
    mk &= __ballot(row[3] >= r[3]);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const unsigned long long cmp = __ballot(row[4 + s] >= r[4 + s]);
      mk &= (k >= kp[s]) ? cmp : absok[s];
    }
    const unsigned long long fresh = mk & ~found;
    if (fresh) {
      if ((fresh >> lane) & 1ull) myk = k;
      found |= mk;
      if (found == ~0ull) { ++k; break; }
    }
  }
  if (valid && myk != BS_INF) atomicMin(&b.first_row[pod], myk);
  if (prm.collect_stats && lane == 0) {
    atomicAdd((unsigned long long*)&b.stats[0], (unsigned long long)(k - k0));          // rows visited by this wave
    atomicAdd((unsigned long long*)&b.stats[1], (unsigned long long)(k - k0) * cnt);    // pod x node pairs evaluated
  }
}

// ------------------------------------------------------------------------------------------------
// k_reject / k_final
// ------------------------------------------------------------------------------------------------
__global__ void k_reject(PodsDev pods, BatchDev b) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  if (!(b.stage[i] & ST_QUERY)) return;
  if (b.first_row[i] == BS_INF) {
    // compareClusterResourceAndRequire returned false: AddToDenyCache (core.go:142,163)
    b.tcode[i] = (b.tcode[i] == BS_PF_PASS_FIRST_FITS) ? BS_PF_REJECT_FIRST : BS_PF_REJECT_RESERVE;
    atomicMin(&b.first_reject[pods.group[i]], i);
  }
}

// Final codes in queue order.  A pod behind the first rejected pod of its group meets the deny entry
// at core.go:105-110 and never gets further.  pf_leader = sop.maxFinishedPG after the pod's PreFilter
// returned: the value findMaxPG produced for it, or — if the call returned before core.go:120 — what
// the latest earlier pod left there (carried in from before the batch when there is none).
__global__ __launch_bounds__(kScanBlock) void k_final(PodsDev pods, NodesDev nd, BatchDev b, BatchParams prm) {
  __shared__ uint32_t lds[16];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = -1;
  __syncthreads();
  for (uint32_t base = 0; base < pods.p; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    int key = -1;
    uint8_t code = 0, st = 0;
    uint32_t fk = BS_K_NOT_SCANNED;
    if (i < pods.p) {
      code = b.tcode[i];
      st = b.stage[i];
      bool reach6 = st & ST_REACH6;
      if (st & ST_OWNED) {
        if ((st & ST_ELIG) && b.first_reject[pods.group[i]] < i) {
          code = BS_PF_ERR_DENIED;
          reach6 = false;
        } else if (st & ST_QUERY) {
          const uint32_t row = b.first_row[i];
          fk = row == BS_INF ? BS_K_NONE : nd.kmap[row];
        }
      } else {
        code = BS_PF_NOT_OWNED;
      }
      if (reach6) key = (int)i;
    }
    // inclusive prefix max of key (block), then across chunks
    int v = key;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane_id() >= o) v = max(v, u);
    }
    __syncthreads();
    if (lane_id() == 63) lds[wave_id()] = (uint32_t)v;
    __syncthreads();
    int off = s_carry;
    for (int w = 0; w < wave_id(); ++w) off = max(off, (int)lds[w]);
    int blockmax = s_carry;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) blockmax = max(blockmax, (int)lds[w]);
    const int jstar = max(v, off);
    if (i < pods.p) {
      b.pf_code[i] = code;
      b.pf_first_k[i] = fk;
      b.pf_leader[i] = jstar >= 0 ? b.leader_raw[jstar] : prm.sop_leader0;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry = blockmax;
    __syncthreads();
  }
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
__global__ void k_filter_params(PodsDev pods, GroupsDev gr, BatchDev b, BatchParams prm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  const uint32_t L = prm.L, S = prm.S, gate = prm.eph_gate;
  uint8_t fl = BS_FL_NOT_RUN;
  uint32_t ff = 0;
  int64_t R[4] = {0, 0, 0, 0}, M[4] = {0, 0, 0, 0};
  const uint8_t pf = b.pf_code[i];
  if (pf != BS_PF_NOT_OWNED && BS_PF_IS_PASS(pf)) {
    const int32_t gi = pods.group[i];
    const int32_t leader = b.pf_leader[i];
    if (gi == BS_POD_NOT_GROUPED) fl = BS_FL_PASS_NOT_GROUPED;                 // core.go:171-174
    else if (gi < 0 || (uint32_t)gi >= gr.g) fl = BS_FL_ERR_PG_NOT_FOUND;      // :177-180
    else if (leader < 0) fl = BS_FL_PANIC_NIL_MAX;                             // :525
    else {
      Res mr, ms;
      res_zero(ms, L);
      const bool have = group_minres_at(gr, pods, b, (uint32_t)leader, i, L, S, gate, mr);
      if (have) res_add(ms, mr, S, gate);                                      // :526-527
      if (leader == gi) fl = BS_FL_PASS_IS_MAX;                                // :531-535
      else if (!have) fl = BS_FL_PASS_NO_MINRES;                               // :542-544
      else {
        fl = BS_FL_EVALUATED;
        Res cur;
        pod_require(pods, i, L, S, gate, cur);                                 // :551
        res_add(cur, ms, S, gate);                                             // :552
        for (int j = 0; j < 4; ++j) { R[j] = cur.v[j]; M[j] = ms.v[j]; }
        for (uint32_t s = 0; s < S; ++s) {
          if ((cur.present & (1u << s)) && cur.v[4 + s] != 0) ff |= 1u;        // case 2 can never hold
          if ((ms.present & (1u << s)) && ms.v[4 + s] != 0) ff |= 2u;          // node "cannot hold" a leader member
        }
      }
    }
  }
  int64_t* dst = b.fparams + (size_t)i * 8;
  for (int j = 0; j < 4; ++j) { dst[j] = R[j]; dst[4 + j] = M[j]; }
  b.fflags[i] = ff | ((uint32_t)fl << 8);
  b.fl_code[i] = fl;
}

// One wave = 64 consecutive pods x a range of 64-node blocks.  Lanes are NODES while comparing (a
// v_cmp_ge_i64 against the pod's wave-uniform request yields the 64 node-feasibility bits of that pod
// directly as an SGPR pair — the "ballot is the bitmap word"), and lanes are PODS for the outputs
// (v_writelane collects pod pp's word into lane pp; popcount accumulates its feasible-node count).
__global__ __launch_bounds__(256) void k_filter(PodsDev pods, NodesDev nd, BatchDev b, uint32_t blocks_per_wave,
                                                uint32_t want_bitmap) {
  const int lane = lane_id();
  const uint32_t ptile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (uint32_t)wave_id());
  const uint32_t p0 = ptile * 64u;
  if (p0 >= pods.p) return;
  const uint32_t W = (nd.n + 63u) / 64u;
  const uint32_t w0 = blockIdx.y * blocks_per_wave;
  const uint32_t w1 = min(W, w0 + blocks_per_wave);
  const uint32_t np = min(64u, pods.p - p0);
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; ++w) {
    const uint32_t n = w * 64u + (uint32_t)lane;
    const bool nvalid = n < nd.n;
    int64_t l0 = INT64_MIN, l1 = INT64_MIN, l2 = INT64_MIN, l3 = INT64_MIN;
    bool node_ok = false;
    if (nvalid) {
      l0 = nd.left4[n];
      l1 = nd.left4[(size_t)nd.stride + n];
      l2 = nd.left4[(size_t)2 * nd.stride + n];
      l3 = nd.left4[(size_t)3 * nd.stride + n];
      node_ok = !(nd.flags[n] & (BS_NODE_NIL | BS_NODE_NO_NODE));              // core.go:442-449
    }
    const unsigned long long in_range = __ballot(nvalid);
    const unsigned long long okmask = __ballot(node_ok);
    uint32_t word_lo = 0, word_hi = 0;
    for (uint32_t pp = 0; pp < np; ++pp) {
      const uint32_t p = p0 + pp;                                              // wave-uniform
      const uint32_t ff = b.fflags[p];
      const uint32_t fl = ff >> 8;
      unsigned long long word;
      if (fl == BS_FL_EVALUATED) {
        const int64_t* __restrict__ prm = b.fparams + (size_t)p * 8;
        unsigned long long c2 = 0, lf = 0;
        if (!(ff & 1u)) c2 = __ballot(l0 >= prm[0]) & __ballot(l1 >= prm[1]) & __ballot(l2 >= prm[2]) & __ballot(l3 >= prm[3]);
        if (!(ff & 2u)) lf = __ballot(l0 >= prm[4]) & __ballot(l1 >= prm[5]) & __ballot(l2 >= prm[6]) & __ballot(l3 >= prm[7]);
        word = okmask & (c2 | ~lf);
      } else if (fl < 16u) {
        word = in_range;              // returned nil before looking at the node
      } else {
        word = 0;
      }
      word_lo = writelane_u32((uint32_t)word, pp, word_lo);
      word_hi = writelane_u32((uint32_t)(word >> 32), pp, word_hi);
    }
    if ((uint32_t)lane < np) {
      const unsigned long long mine = ((unsigned long long)word_hi << 32) | word_lo;
      cnt += (uint32_t)__popcll(mine);
      if (want_bitmap) b.fl_bitmap[(size_t)w * pods.p + p0 + lane] = mine;
    }
  }
  if ((uint32_t)lane < np && cnt) atomicAdd(&b.fl_feasible[p0 + lane], cnt);
}

// ------------------------------------------------------------------------------------------------
// k_tally / k_ready
// ------------------------------------------------------------------------------------------------
__global__ void k_tally(PodsDev pods, GroupsDev gr, BatchDev b, uint32_t run_filter) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool admit = false;
  uint32_t g = 0;
  if (i < pods.p) {
    const int32_t gi = pods.group[i];
    const uint8_t pf = b.pf_code[i];
    if (gi >= 0 && (uint32_t)gi < gr.g && pf != BS_PF_NOT_OWNED && BS_PF_IS_PASS(pf) &&
        (!run_filter || b.fl_feasible[i] > 0)) {
      admit = true;
      g = (uint32_t)gi;
    }
  }
  wave_aggregated_inc(b.admit, g, admit);
}

// quorum predicate of Permit, core.go:303, with every admitted pod counted as matched
__global__ void k_ready(GroupsDev gr, BatchDev b) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= gr.g) return;
  const uint32_t have = gr.matched[g] + b.admit[g];
  b.ready[g] = have >= (uint32_t)(gr.min_member[g] - gr.status_scheduled[g]) ? 1 : 0;
}

// computeResourceSatisfied for pod 0 of a one-pod view and one node, with the exact case identity
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

// BS_BATCH_COMMIT: persist what the sequential PreFilter calls would have left in the cache —
// first-pod capture and MinResources default (core.go:486-493), OccupiedBy (:494-500), deny entry
// (:142,:163).  A pod behind its group's first rejection never reaches fillOccupiedObj.
__global__ void k_commit(PodsDev pods, BatchDev b, BatchParams prm, uint8_t* gflags, uint32_t* gcls, int64_t* gminres,
                         uint32_t* gmrpres, uint64_t* gocc, uint32_t G) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t fe = b.first_elig[g], fr = b.first_reject[g];
  uint8_t fl = gflags[g];
  if (fe != BS_INF) {
    if (!(fl & BS_GROUP_HAS_POD)) { fl |= BS_GROUP_HAS_POD; gcls[g] = pods.cls[fe]; }
    if (!(fl & BS_GROUP_HAS_MINRES)) {
      Res r;
      pod_require(pods, fe, prm.L, prm.S, prm.eph_gate, r);
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

// ------------------------------------------------------------------------------------------------
// single-query helpers (bs_node_left, bs_cluster_total)
// ------------------------------------------------------------------------------------------------
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

__global__ void k_scale_probe(const int64_t* a, const float* pct, int64_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = scale_f32(a[i], pct[i]);
}

}  // namespace bs
