// bs_epoch.hpp — batches in which the queue position matters, in THREE launches.
//
// The steady-state chain (bs_fast.hpp) needs "nothing a pod derives depends on where it stands in the queue".
// That is false while gangs are still being seen for the first time: the first pod of a group that reaches
// fillOccupiedObj is captured (pgs.Pod = pod, core.go:486-488), which puts the group into findMaxPG's candidate
// set (:709-711) and may default its MinResources (:489-493); and while the leader has no matched pod every
// group asks its own first check (:136-147).  The general chain of bs_kernels.hpp replays that with a pre-pass,
// a two-level epoch scan, findMaxPG per epoch, one slot per pod, k_reject, k_final, k_tally: eleven launches.
//
// What this chain does instead:
//   * everything positional that depends on groups + pods only is derived ONCE, when the later of the two is
//     loaded (analysis launches below): first eligible pod / first owner per group, the capture epochs, findMaxPG
//     per epoch (k_leader_scan), and from those the leader RUNS — maximal stretches of epochs with one leader.
//     A batch has few runs (a cold start has one: the first gang seen leads until something is permitted).
//   * a pod's derived vectors depend on (its request class, the leader it sees, whether that leader's MinResources
//     are visible yet) -> scan and Filter slots are (view, class) with view = 2 x run + visible; one extra run
//     stands for the leader carried into the batch (sop.maxFinishedPG before the first findMaxPG).
//   * a first check (core.go:136-147) asks for the GROUP's pre-allocation against the group's class: one slot per
//     group, and the groups are laid out in class order (counting sort at analysis time), so a tile of 64 group
//     slots scans one table, or two at a class boundary, with every lane busy.
//   * slots are stamped, minima are keyed by the inverted batch number: nothing is reset per batch.
//   * the deny replay (core.go:105-110) asks "is an earlier pod of my group rejected": the group's first check
//     failed (first asking pod), or a (pair, view) of the group has a rejected slot (first asking pod of it).
//   * sop.maxFinishedPG after a pod's PreFilter (the stale shared field Filter reads, core.go:121,:525) is the
//     findMaxPG result of the last pod at or before it that really got there: a block-local prefix maximum, and a
//     backward search over earlier blocks only when the block's first pod did not get there itself.
//
// Several ranks: ownership as everywhere (all pods of a group on the rank of the group's first pod), decisions of owned pods
// only, the quorum pass after the collective (k_ready) — a non-owned pod's "reached findMaxPG" stays the tentative one.
//
//   launch A  k_epoch_query_tables   per pod: decisions that need no scan, scan / Filter slots | chunk-local running
//                                    sums of every table the loaded state can ask for (known at analysis time)
//   launch B  k_epoch_scan_filter    node scan per live scan slot | computeResourceSatisfied per Filter slot x node
//   launch C  k_epoch_final          REJECT / deny replay / stale leader, Filter code + slot + feasible count per
//                                    pod, per-group admit counts and the quorum predicate core.go:303 (tally_tail)
//   (BS_BATCH_COMMIT: + k_epoch_reject_groups + k_commit, then the analysis is redone for the committed state)
#pragma once

#include "bs_fast.hpp"

namespace bs {

constexpr uint32_t kMaxRuns = 16;           // leader runs a batch may have on this chain (= the run_leader array; more: general chain).  Round 3 stopped at 4;
                                            // the slot spaces ((view, class) scan slots 2 R K, Filter slots (R + 1) K) are checked against their capacities per batch
constexpr uint32_t kEpochHistCap = 8192;    // fit classes + 1 the counting sort holds in LDS

struct EpochDev {
  uint32_t* run_of_epoch;         // [E + 1] run the epoch belongs to
  int32_t* run_leader;            // [16] leader of run r
  uint32_t* gslot;                // [G] position of the group in class order (its first-check slot is GB + gslot[g])
  unsigned long long* gfirstq;    // [G] (~batch_seq << 32) | first pod of the group that asked its first check
  uint32_t R, K, GB;              // runs, request classes, first group slot (host-known after the analysis)
  uint32_t has_first, has_reserve;
};

// ------------------------------------------------------------------------------------------------
// analysis (at load time; again after anything changed the groups or the pods)
// ------------------------------------------------------------------------------------------------
// per group: the minima of the pod load gated by the group's flags — what k_prepass derives with per-pod atomics
__global__ void k_epoch_groups(GroupsDev gr, BatchDev b) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= gr.g) return;
  const uint8_t fl = gr.flags[g];
  // (BS_BATCH_FILTER_DENY re-runs: a group deny-listed in front of its first eligible pod never gets one; an owner behind the
  // position never writes OccupiedBy)
  const uint32_t fdp = b.fd_in ? b.fd_in[g] : BS_INF;
  const bool denied = (fl & BS_GROUP_DENIED) || fdp < b.first_np_s[g];
  b.first_pod[g] = b.first_pod_s[g];
  b.first_elig[g] = denied ? BS_INF : b.first_np_s[g];
  b.first_owner[g] = (denied || gr.occupied[g] != 0 || fdp < b.first_owner_s[g]) ? BS_INF : b.first_owner_s[g];
  b.first_reject[g] = BS_INF;
  b.cap_epoch[g] = (fl & BS_GROUP_HAS_POD) ? 0u : BS_INF;
}

__device__ __forceinline__ uint32_t capture_flag2(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, uint32_t i, int32_t& gi) {
  gi = -1;
  if (i >= pods.p) return 0u;
  const int32_t g = pods.group[i];
  if (g < 0 || (uint32_t)g >= gr.g) return 0u;
  gi = g;
  return (b.first_elig[g] == i && !(gr.flags[g] & BS_GROUP_HAS_POD)) ? 1u : 0u;      // first_elig covers LAST_PERMITTED and the deny flag
}
__global__ __launch_bounds__(kScanBlock) void k_epochs2_a(PodsDev pods, GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  int32_t gi;
  const uint32_t cap = capture_flag2(pods, gr, b, blockIdx.x * kScanBlock + threadIdx.x, gi);
  uint32_t total;
  (void)block_incl_scan_add<uint32_t>(cap, lds, total);
  if (threadIdx.x == 0) b.blk_scratch[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void k_epochs2_b(PodsDev pods, GroupsDev gr, BatchDev b) {
  __shared__ uint32_t lds[16];
  uint32_t part = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kScanBlock) part += b.blk_scratch[j];
  uint32_t prev;
  (void)block_incl_scan_add<uint32_t>(part, lds, prev);
  const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
  int32_t gi;
  const uint32_t cap = capture_flag2(pods, gr, b, i, gi);
  uint32_t total;
  const uint32_t incl = block_incl_scan_add<uint32_t>(cap, lds, total);
  if (i < pods.p) b.epoch[i] = prev + incl;
  if (cap) { b.cap_epoch[gi] = prev + incl; b.epoch_group[prev + incl] = (uint32_t)gi; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *b.nepochs = prev + total + 1;
}

// Leader runs, the tables the loaded state can ask for, the class order of the groups.  One block.
//   info[0] runs  info[1] bit0: some run's leader has no matched pod (first checks), bit1: some run's leader has
//   (reservation checks), bit2: this chain cannot take the batch  info[3] tag
__global__ __launch_bounds__(kLeaderBlock) void k_epoch_views(PodsDev pods, GroupsDev gr, BatchDev b, EpochDev ep, uint32_t C, int32_t tag, int32_t* info) {
  __shared__ uint32_t s_hist[kEpochHistCap];
  __shared__ uint32_t lds[16];
  __shared__ uint32_t s_carry, s_flags, s_base;
  const uint32_t E1 = *b.nepochs;
  if (threadIdx.x == 0) { s_carry = 0; s_flags = 0; }
  for (uint32_t t = threadIdx.x; t < 2 * C + 1; t += kLeaderBlock) b.needed[t] = 0;
  __syncthreads();
  // ---- runs: epoch e starts one when its leader differs from the previous epoch's
  for (uint32_t base = 0; base < E1; base += kLeaderBlock) {
    const uint32_t e = base + threadIdx.x;
    uint32_t start = 0;
    int32_t l = -1;
    if (e < E1) {
      l = b.leader_epoch[e];
      start = (e == 0 || l != b.leader_epoch[e - 1]) ? 1u : 0u;
    }
    uint32_t total;
    const uint32_t incl = block_incl_scan_add<uint32_t>(start, lds, total);
    const uint32_t r = s_carry + incl - 1u;
    if (e < E1) {
      ep.run_of_epoch[e] = r;
      if (start && r < 16u) ep.run_leader[r] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  const uint32_t R = s_carry;
  auto cls_of = [&](uint32_t g) -> uint32_t {       // group_cls_at: the loaded class, or the capturing pod's
    if (gr.flags[g] & BS_GROUP_HAS_POD) return gr.cls[g];
    const uint32_t fe = b.first_elig[g];
    return fe != BS_INF ? pods.cls[fe] : BS_INF;
  };
  if (threadIdx.x < min(R, 16u)) {
    const int32_t l = ep.run_leader[threadIdx.x];
    if (l >= 0) {
      if (gr.matched[l] == 0) atomicOr(&s_flags, 1u);
      else {
        atomicOr(&s_flags, 2u);
        const uint32_t cl = cls_of((uint32_t)l);
        if (cl < C) b.needed[C + cl] = 1;           // core.go:161: percent 0.7 against the leader's class
        else atomicOr(&s_flags, 4u);
      }
    }
  }
  if (threadIdx.x == 0 && (R > kMaxRuns || C + 1u > kEpochHistCap)) atomicOr(&s_flags, 4u);
  __syncthreads();
  const bool has_first = s_flags & 1u;
  // ---- groups in class order (a group no pod can ask for goes behind the last class)
  const uint32_t nb = min(C, kEpochHistCap - 1u) + 1u;
  for (uint32_t k = threadIdx.x; k < nb; k += kLeaderBlock) s_hist[k] = 0;
  __syncthreads();
  auto key_of = [&](uint32_t g) -> uint32_t {
    if (b.first_elig[g] == BS_INF) return nb - 1u;
    const uint32_t cl = cls_of(g);
    return cl < nb - 1u ? cl : nb - 1u;
  };
  for (uint32_t g = threadIdx.x; g < gr.g; g += kLeaderBlock) {
    const uint32_t k = key_of(g);
    ep.gslot[g] = atomicAdd(&s_hist[k], 1u);
    if (has_first && k < nb - 1u && k < C) b.needed[k] = 1;     // core.go:140: percent 1 against the group's class
  }
  __syncthreads();
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += kLeaderBlock) {    // exclusive prefix of the class counts
    const uint32_t k = base + threadIdx.x;
    const uint32_t v = k < nb ? s_hist[k] : 0u;
    uint32_t total;
    const uint32_t incl = block_incl_scan_add<uint32_t>(v, lds, total);
    if (k < nb) s_hist[k] = s_base + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_base += total;
    __syncthreads();
  }
  for (uint32_t g = threadIdx.x; g < gr.g; g += kLeaderBlock) ep.gslot[g] += s_hist[key_of(g)];
  if (threadIdx.x == 0) {
    info[0] = (int32_t)R;
    info[1] = (int32_t)s_flags;
    __hip_atomic_store(&info[3], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------
// launch A, pod part (core.go:88-167 up to the node scan)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool minres_visible(const GroupsDev& gr, const BatchDev& b, uint32_t g, uint32_t i) {
  return (gr.flags[g] & BS_GROUP_HAS_MINRES) || b.first_elig[g] <= i;      // group_minres_at's predicate
}

template <int TS>
__device__ __forceinline__ void epoch_query_thread(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, const EpochDev& ep,
                                                   uint32_t i, uint32_t nthreads) {
  const Shape<TS> sh(prm.S);
  arm_tally(gr, b, prm, i, nthreads);                                  // consumed by launch C
  if (i < 8 && prm.collect_stats) b.stats[i] = 0;
  const bool valid = i < pods.p;
  const uint32_t gate = prm.eph_gate;
  uint8_t code = BS_PF_PASS_NOT_GROUPED, st = 0;
  bool has_q = false, first_q = false, grouped = false;
  int32_t leader = -1, table = -1, gi = BS_POD_NOT_GROUPED;
  uint32_t run = 0, slot = 0, view = 0;
  Res q;
  res_zero(q, sh);
  if (valid) {
    gi = pods.group[i];
    grouped = gi >= 0 && (uint32_t)gi < gr.g;
    // shard ownership: all pods of a group live on one rank (owner_rank_of)
    const uint32_t anchor = grouped ? b.first_pod[gi] : i;
    if (owner_rank_of(b, prm, anchor, pods.p) == prm.rank) st |= ST_OWNED;
    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                         // core.go:89-92
    else if (pods.flags[i] & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;      // :95-98
    else if (!grouped) code = BS_PF_ERR_PG_NOT_FOUND;                                      // :100-103
    else if ((gr.flags[gi] & BS_GROUP_DENIED) || fd_denied(b, (uint32_t)gi, i)) code = BS_PF_ERR_DENIED;   // :105-110
    else {
      st |= ST_ELIG;
      const uint32_t g = (uint32_t)gi;
      bool occ_err = false;                                                                // :494-511 in queue order
      const uint64_t own = pods.owner[i];
      const uint64_t occ0 = gr.occupied[g];
      if (occ0 != 0) occ_err = (own == 0) || (own != occ0);
      else {
        const uint32_t fo = b.first_owner[g];
        if (fo != BS_INF && i > fo) { const uint64_t occ = pods.owner[fo]; occ_err = (own == 0) || (own != occ); }
      }
      const uint32_t e = b.epoch[i];
      if (occ_err) code = BS_PF_ERR_OCCUPIED;                                              // :113-115
      else if (b.panic_epoch[e]) code = BS_PF_PANIC_DIV0;                                  // :716-717
      else {
        st |= ST_REACH6;                                                                   // :118-123
        leader = b.leader_epoch[e];
        run = ep.run_of_epoch[e];
        if (leader < 0) code = BS_PF_PASS_NO_MAX;                                          // :127-130
        else {
          const int64_t matched = (int64_t)gr.matched[leader];                             // :132-135
          Res mr;
          if (matched == 0) {                                                              // :136-147
            const bool have = group_minres_at(gr, pods, b, g, i, sh, gate, mr);
            pre_allocated(gr, g, 0, have, mr, sh, gate, q);
            table = (int32_t)group_cls_at(gr, pods, b, g);                                 // percent 1
            code = BS_PF_PASS_FIRST_FITS;                                                  // tentative
            has_q = first_q = true;
            slot = ep.GB + ep.gslot[g];
          } else if (leader == gi) {
            code = BS_PF_PASS_IS_MAX;                                                      // :150-155
          } else {                                                                         // :157-166
            const bool have = group_minres_at(gr, pods, b, (uint32_t)leader, i, sh, gate, mr);
            pre_allocated(gr, (uint32_t)leader, matched, have, mr, sh, gate, q);
            Res cur;
            pod_require(pods, i, sh, gate, cur);
            res_add(q, cur, sh, gate);
            table = (int32_t)(prm.C + group_cls_at(gr, pods, b, (uint32_t)leader));        // percent 0.7
            code = BS_PF_PASS_RESERVE_FITS;                                                // tentative
            has_q = true;
            view = 2u * run + (have ? 1u : 0u);
            slot = view * ep.K + b.pclass[i];
          }
        }
      }
    }
    if (!(st & ST_OWNED)) has_q = first_q = false;                    // another rank evaluates this pod
    if (has_q) st |= ST_QUERY;
    b.tcode[i] = code;
    b.stage[i] = st;
  }
  // Filter slots: run x K + class — the leader of the run with its MinResources visible (without them Filter passes
  // without looking at a node, core.go:542-544); run R = the leader carried into the batch.  A pod that reaches findMaxPG sees its own epoch's leader; a pod that passes without
  // getting there (LAST_PERMITTED) sees what the latest earlier pod left: any run up to its own, or the carried-in one.
  if (prm.run_filter) {
    const uint32_t c = valid ? b.pclass[i] : 0u;
    const bool may = valid && (st & ST_OWNED) && BS_PF_IS_PASS(code) && grouped;
    const bool own = may && (st & ST_REACH6) && leader >= 0 && leader != gi && minres_visible(gr, b, (uint32_t)leader, i);
    const uint32_t fslot = run * ep.K + c;
    if (wave_elect_by_key(fslot, own)) filter_params_for<TS>(pods, gr, b, prm, i, code, leader, fslot, true, false);
    if (may && !(st & ST_REACH6)) {
      const uint32_t rmax = min(ep.run_of_epoch[b.epoch[i]], ep.R - 1u);
      for (uint32_t r2 = 0; r2 <= rmax; ++r2) {
        const int32_t l2 = ep.run_leader[r2];
        if (l2 >= 0 && l2 != gi) filter_params_for<TS>(pods, gr, b, prm, i, code, l2, r2 * ep.K + c, true, false);
      }
      const int32_t l0 = prm.sop_leader0;
      if (l0 >= 0 && (uint32_t)l0 < gr.g && l0 != gi) filter_params_for<TS>(pods, gr, b, prm, i, code, l0, ep.R * ep.K + c, true, false);
    }
  }
  const bool fill = wave_elect_by_key(slot, has_q);     // one writer per (wave, slot): every asker derives the same contents
  if (has_q) {
    b.qpos[i] = slot;
    const unsigned long long key = ((unsigned long long)prm.seq_inv << 32) | i;
    if (first_q) atomicMin(&ep.gfirstq[gi], key);
    else atomicMin(&b.pair_firstq[(size_t)view * b.pair_stride + b.ppair[i]], key);
  }
  if (fill) {
    uint32_t absok = 0;
#pragma unroll
    for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
      if (s < sh.S()) {
        const bool pres = q.present & (1u << s);
        if (!pres || q.v[4 + s] == 0) absok |= 1u << s;       // core.go:688-692
        if (!pres) q.v[4 + s] = INT64_MIN;                    // key not requested: never constrains
      }
    }
    int64_t* dst = b.qreq_s + (size_t)slot * prm.LP;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < prm.LP) dst[j] = j < sh.L() ? q.v[j] : INT64_MIN;
    b.qflags_s[slot] = q.present | (absok << 16);
    b.qtab_s[slot] = table;
    b.first_row[slot] = BS_INF;                               // every writer stores the same; launch B takes minima
    b.qstamp_s[slot] = prm.stamp;
  }
  if (prm.collect_stats) {
    const unsigned long long hq = __ballot(has_q);
    if (lane_id() == 0 && hq) atomicAdd((unsigned long long*)&b.stats[2], (unsigned long long)__popcll(hq));
  }
}

// Chunk `chunk` (256 rows) of the tables [t0, t1) that the loaded state can ask for.  What a row needs from its node
// (allocatable, requested, key presence, flags: 2 L + 3 loads) is the same for every table — only the fit bit of the
// table's class and its percent differ — so one block builds the chunk for a whole group of tables behind ONE load chain
// (two tables per block: more made the block's own chain the longest thing in the launch — measured with 8).  Per table it is
// tables_local_fast's body: wave scans (DPP), one exchange of wave totals, rows, chunk total, per-group local max,
// first key rows.  LDS exchange buffers alternate between consecutive tables: one barrier pair per table, no more.
constexpr uint32_t kTableGroup = 2;

template <int TS>
__device__ __forceinline__ void tables_local_multi(const NodesDev& nd, const BatchDev& b, const BatchParams& prm, uint32_t t0, uint32_t t1, uint32_t chunk,
                                                   uint32_t cstride, uint32_t gstride) {
  __shared__ unsigned long long s_wtot[2][BS_MAX_LANES][4];
  __shared__ uint32_t s_kp[2][BS_MAX_SCALARS];
  __shared__ __attribute__((aligned(16))) int64_t s_rowbuf[kTblChunk * 16];      // the chunk's rows, row-major, as they will lie in the table
  if (chunk * kTblChunk >= nd.m) return;
  const uint32_t k = chunk * kTblChunk + threadIdx.x;
  const bool valid = k < nd.m;
  const uint32_t n = valid ? nd.kmap[k] : 0u;
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S(), LP = prm.LP;
  // every load that does not depend on the table, then the fit words of the group's tables, issued together
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
  uint32_t fw[kTableGroup], need[kTableGroup];
#pragma unroll
  for (uint32_t u = 0; u < kTableGroup; ++u) {
    const uint32_t t = t0 + u;
    need[u] = t < t1 ? b.needed[t] : 0u;
    fw[u] = t < t1 ? nd.fit[(size_t)(t % prm.C) * nd.fit_words + (n >> 5)] : 0u;
  }
  const int w = wave_id();
  uint32_t par = 0;
#pragma unroll 1
  for (uint32_t u = 0; u < kTableGroup; ++u) {
    uint32_t fwu = 0, needu = 0;
#pragma unroll
    for (uint32_t x = 0; x < kTableGroup; ++x)        // (constant indexing keeps fw[] / need[] in registers)
      if (x == u) { fwu = fw[x]; needu = need[x]; }
    if (!needu) continue;                              // block-uniform
    const uint32_t slot = t0 + u;
    const TableDesc d = table_desc(slot, prm.C, nullptr);
    int64_t* T = b.tables + (size_t)slot * prm.mcap * LP;
    const bool fit = valid && ((fwu >> (n & 31u)) & 1u) && !(fl & BS_NODE_TAINT_ERR);
    const uint32_t pres = fit ? (ap & rp) : 0u;
    if (threadIdx.x < BS_MAX_SCALARS) s_kp[par][threadIdx.x] = BS_INF;
    unsigned long long incl[BS_MAX_LANES];
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const bool live = fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
        const unsigned long long left = live ? (unsigned long long)wsub(scale_f32(al[j], d.pct), rq[j]) : 0ull;
        incl[j] = wave_incl_scan_add_u64(left);
        if (lane_id() == 63) s_wtot[par][j][w] = incl[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        unsigned long long off = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned long long x = s_wtot[par][j][i];
          if (i < w) off += x;
          tot += x;
        }
        incl[j] += off;
        s_rowbuf[threadIdx.x * LP + j] = (int64_t)incl[j];
        if (threadIdx.x == 0) b.chunk_tot[((size_t)slot * cstride + chunk) * 16 + j] = tot;
      } else if (j < LP) {
        s_rowbuf[threadIdx.x * LP + j] = INT64_MAX;
      }
    }
    __syncthreads();
    {
      // A chunk's rows are one contiguous piece of the table: copy it out 16 bytes per lane, consecutive lanes consecutive
      // addresses (a row per lane, 8 bytes at a time, is 8 store instructions that each touch 64 different cache lines:
      // with 64 tables in flight the write path, not the arithmetic, was what this launch waited for).
      const uint32_t rows = min((uint32_t)kTblChunk, nd.m - chunk * kTblChunk);
      const uint32_t units = rows * LP / 2u;
      const int4* src = reinterpret_cast<const int4*>(s_rowbuf);
      int4* dst = reinterpret_cast<int4*>(T + (size_t)chunk * kTblChunk * LP);
      for (uint32_t x = threadIdx.x; x < units; x += kTblChunk) dst[x] = src[x];
    }
    {
      const uint32_t grp = k >> 6;
      const bool grp_valid = (chunk * kTblChunk + (threadIdx.x & ~63u)) < nd.m;
      constexpr int64_t kSafe = (int64_t)1 << 62;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const int64_t x = (int64_t)incl[j];
          const long long mx = wave_max_i64_lane63(valid ? x : INT64_MIN);
          const bool risky = __ballot(valid && (x >= kSafe || x <= -kSafe)) != 0ull;      // some local sum could wrap with an offset on top
          if (lane_id() == 63 && grp_valid) b.gmax[((size_t)slot * gstride + grp) * LP + j] = risky ? INT64_MAX : mx;
        }
      }
    }
#pragma unroll
    for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
      if (s2 < S) {
        const unsigned long long mk = __ballot(valid && (pres & (1u << s2)));
        if (mk && lane_id() == 0) atomicMin(&s_kp[par][s2], chunk * kTblChunk + (uint32_t)(threadIdx.x & ~63u) + (uint32_t)(__ffsll((long long)mk) - 1));
      }
    }
    __syncthreads();
    if (threadIdx.x < 16) b.chunk_kp[((size_t)slot * cstride + chunk) * 16 + threadIdx.x] = threadIdx.x < BS_MAX_SCALARS ? s_kp[par][threadIdx.x] : BS_INF;
    par ^= 1u;
  }
}

// block layout: [0, query_blocks) pods | query_blocks + tg * nchunks + c: chunk c of the tables [tg * kTableGroup, + kTableGroup)
template <int TS>
__global__ __launch_bounds__(kTblChunk) void k_epoch_query_tables(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm, EpochDev ep,
                                                                  uint32_t nchunks, uint32_t cstride, uint32_t gstride, uint32_t query_blocks, uint32_t ntab) {
  if (blockIdx.x < query_blocks) {
    epoch_query_thread<TS>(pods, gr, b, prm, ep, blockIdx.x * kTblChunk + threadIdx.x, query_blocks * kTblChunk);
    return;
  }
  const uint32_t idx = blockIdx.x - query_blocks;
  const uint32_t tg = idx / nchunks, chunk = idx - tg * nchunks;
  tables_local_multi<TS>(nd, b, prm, tg * kTableGroup, min(ntab, tg * kTableGroup + kTableGroup), chunk, cstride, gstride);
}

// ------------------------------------------------------------------------------------------------
// launch B: node scan over the live (view, class) slots and group slots | Filter over the Filter slots
// ------------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void epoch_scan_loop(const BatchDev& b, const BatchParams& prm, const EpochDev& ep, uint32_t m, uint32_t jcap, uint32_t ngroups_g,
                                                uint32_t bx, uint32_t nblocks, int64_t (*rows)[4 + S]) {
  constexpr int LP = (S == 0) ? 4 : (S <= 4 ? 8 : 16);
  constexpr int L = 4 + S;
  if (!m) return;
  const uint32_t tiles_v = ep.has_reserve ? (2u * ep.R * ep.K + 63u) >> 6 : 0u;     // (view, class) slots [0, 2 R K)
  const uint32_t tiles_g = ep.has_first ? (ngroups_g + 63u) >> 6 : 0u;              // group slots [GB, GB + G), GB a multiple of 64
  const uint32_t ntl = tiles_v + tiles_g;
  if (!ntl) return;
  const uint32_t nslots = ep.GB + ngroups_g;
  // item = (tile, table turn, share): a tile of group slots straddles a class boundary more often than not, and its two
  // tables are two independent scans — turn 0 takes the tile's first table, turn 1 the rest
  const uint32_t J = max(1u, min(min(jcap, (m + 63u) >> 6), (nblocks * 4u) / (ntl * 2u)));
  const uint32_t items = ntl * 2u * J;
  const int lane = lane_id();
  for (uint32_t w = __builtin_amdgcn_readfirstlane(bx * 4u + (uint32_t)wave_id()); w < items; w += nblocks * 4u) {
    const uint32_t rest = w / ntl, t = w - rest * ntl;
    const uint32_t share = rest >> 1, ts = rest & 1u;
    const uint32_t tile = t < tiles_v ? t : (ep.GB >> 6) + (t - tiles_v);
    const uint32_t pos = tile * 64u + (uint32_t)lane;
    // everything a slot holds in ONE round trip (stamp, table, request, flags), then the liveness decision
    const uint32_t ps = min(pos, nslots - 1u);
    const uint32_t stp = b.qstamp_s[ps];
    const int32_t tb = b.qtab_s[ps];
    int64_t r[1][L];
    {
      const int64_t* src = b.qreq_s + (size_t)ps * LP;
#pragma unroll
      for (int j = 0; j < L; ++j) r[0][j] = src[j];
    }
    uint32_t qf = b.qflags_s[ps];
    const int32_t tab = (pos < nslots && stp == prm.stamp) ? tb : -1;               // live iff a pod of THIS batch wrote it
    unsigned long long todo = __ballot(tab >= 0);
    if (!todo) continue;
    if (tab < 0) {
      qf = 0;
#pragma unroll
      for (int j = 0; j < L; ++j) r[0][j] = INT64_MAX;
    }
    if (prm.collect_stats && share == 0 && ts == 0 && lane == 0) atomicAdd((unsigned long long*)&b.stats[4], (unsigned long long)__popcll(todo));
    uint32_t turn = 0;
    while (todo) {
      const int32_t t0 = __builtin_amdgcn_readlane(tab, __ffsll((long long)todo) - 1);
      const bool member = tab == t0;
      if (turn == ts) scan_core<S, true, false>(b, prm, m, (uint32_t)t0, pos, member, r, qf, share, J, rows, LocalPre<S>{});   // (no shared pre-fetch: every tile has its own table)
      turn = 1u;
      todo &= ~__ballot(member);
      if (ts == 0) break;                            // (the first table was this wave's)
    }
  }
}

template <int S>
__global__ __launch_bounds__(256) void k_epoch_scan_filter(PodsDev pods, NodesDev nd, BatchDev b, BatchParams prm, EpochDev ep, uint32_t m, uint32_t jcap,
                                                           uint32_t ngroups_g, uint32_t scan_blocks, uint32_t filter_waves, uint32_t ustride,
                                                           uint32_t filter_slots) {
  __shared__ int64_t s_rows[4][64][4 + S];
  if (blockIdx.x < scan_blocks)
    epoch_scan_loop<S>(b, prm, ep, m, jcap, ngroups_g, blockIdx.x, scan_blocks, s_rows[wave_id()]);
  else
    filter_loop<2>(pods, nd, b, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x - scan_blocks, gridDim.x - scan_blocks, prm.stamp, filter_slots);
}

// ------------------------------------------------------------------------------------------------
// launch C
// ------------------------------------------------------------------------------------------------
// First pod of group g that the node scan turned down in this batch (AddToDenyCache, core.go:142,163), BS_INF = none.
__device__ __forceinline__ uint32_t epoch_first_reject(const BatchDev& b, const BatchParams& prm, const EpochDev& ep, uint32_t P, uint32_t g) {
  uint32_t fr = BS_INF;
  if (ep.has_first) {
    const unsigned long long fq = ep.gfirstq[g];                         // } one round trip
    const uint32_t gs = ep.gslot[g];                                     // }
    const uint32_t row = b.first_row[ep.GB + gs];
    if ((uint32_t)(fq >> 32) == prm.seq_inv && row == BS_INF) fr = (uint32_t)fq;    // a pod asked the group's first check and it failed
  }
  if (ep.has_reserve) {
    const uint32_t nv = 2u * ep.R;
    for (unsigned long long link = b.pair_head[g]; (uint32_t)link != BS_INF;) {
      const uint32_t r = (uint32_t)link, cls = (uint32_t)(link >> 32);
      link = b.pair_next[r];
      for (uint32_t v = 0; v < nv; ++v) {
        const unsigned long long pq = b.pair_firstq[(size_t)v * b.pair_stride + r];
        const uint32_t row = b.first_row[v * ep.K + cls];
        if ((uint32_t)(pq >> 32) != prm.seq_inv) continue;               // no pod of the pair asked with this view in this batch
        if (row == BS_INF) fr = min(fr, (uint32_t)pq);
      }
    }
  }
  return fr;
}

// BS_BATCH_COMMIT: what k_commit needs besides the analysis — the group's first rejected pod (its deny entry, core.go:142,163)
__global__ void k_epoch_reject_groups(GroupsDev gr, BatchDev b, BatchParams prm, EpochDev ep, uint32_t P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= gr.g) return;
  b.first_reject[g] = b.first_elig[g] != BS_INF ? epoch_first_reject(b, prm, ep, P, g) : BS_INF;
}

__global__ __launch_bounds__(256) void k_epoch_final(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm, EpochDev ep) {
  __shared__ uint32_t lds[16];
  __shared__ uint32_t s_need_prev;
  const uint32_t base = blockIdx.x * 256u;
  const uint32_t i = base + threadIdx.x;
  uint8_t code = 0;
  int32_t gi = BS_POD_NOT_GROUPED;
  bool reached = false;
  if (i < pods.p) {
    code = b.tcode[i];
    const uint8_t st = b.stage[i];
    gi = pods.group[i];
    uint32_t fk = BS_K_NOT_SCANNED;
    bool denied = false;
    if ((st & ST_OWNED) && (st & ST_ELIG)) {
      const uint32_t fr = epoch_first_reject(b, prm, ep, pods.p, (uint32_t)gi);
      denied = fr < i;
    }
    if (!(st & ST_OWNED)) code = BS_PF_NOT_OWNED;                                          // (its reach status stays the tentative one)
    else if (denied) code = BS_PF_ERR_DENIED;                                              // core.go:105-110 replayed
    else if (st & ST_QUERY) {
      const uint32_t row = b.first_row[b.qpos[i]];
      if (row == BS_INF) { code = code == BS_PF_PASS_FIRST_FITS ? BS_PF_REJECT_FIRST : BS_PF_REJECT_RESERVE; fk = BS_K_NONE; }   // :140-146, :161-165
      else fk = nd.kmap[row];
    }
    reached = (st & ST_REACH6) && !denied;
    b.pf_code[i] = code;
    b.pf_first_k[i] = fk;
    if (prm.host_tag) { b.h_pf_code[i] = code; b.h_pf_first_k[i] = fk; }
  }
  // last pod at or before i that really reached findMaxPG (as index + 1, 0 = none)
  uint32_t v = reached ? i + 1u : 0u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)v, o);
    if (lane_id() >= o) v = max(v, u);
  }
  if (threadIdx.x == 0) s_need_prev = (!reached && base > 0) ? 1u : 0u;
  __syncthreads();
  if (lane_id() == 63) lds[wave_id()] = v;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wave_id(); ++w) off = max(off, lds[w]);
  uint32_t prev = 0;
  if (s_need_prev) {                      // the block's first pod did not get there: look back (nearly always one block)
    for (uint32_t hi = base; hi > 0 && prev == 0;) {
      const uint32_t lo = hi >= 256u ? hi - 256u : 0u;
      const uint32_t j = lo + threadIdx.x;
      uint32_t cand = 0;
      if (j < hi) {
        const uint8_t sj = b.stage[j];
        if (sj & ST_REACH6) {
          bool den = false;
          if ((sj & ST_OWNED) && (sj & ST_ELIG)) den = epoch_first_reject(b, prm, ep, pods.p, (uint32_t)pods.group[j]) < j;
          if (!den) cand = j + 1u;
        }
      }
      prev = block_max_u32(cand, lds);
      hi = lo;
    }
  }
  const uint32_t jp1 = max(max(v, off), prev);
  bool admit = false;
  if (i < pods.p) {
    int32_t leader = prm.sop_leader0;
    uint32_t rr = ep.R;
    if (jp1) {
      const uint32_t e2 = b.epoch[jp1 - 1u];
      leader = b.leader_epoch[e2];
      rr = ep.run_of_epoch[e2];
    }
    b.pf_leader[i] = leader;
    const bool pass = code != BS_PF_NOT_OWNED && BS_PF_IS_PASS(code);
    uint32_t feasible = 1u, slot = 0;
    uint8_t fl = BS_FL_NOT_RUN;
    if (prm.run_filter) {
      if (pass) {
        if (gi == BS_POD_NOT_GROUPED) fl = BS_FL_PASS_NOT_GROUPED;                         // core.go:171-174
        else if (gi < 0 || (uint32_t)gi >= gr.g) fl = BS_FL_ERR_PG_NOT_FOUND;              // :177-180
        else if (leader < 0) fl = BS_FL_PANIC_NIL_MAX;                                     // :525
        else if (leader == gi) fl = BS_FL_PASS_IS_MAX;                                     // :531-535
        else if (!minres_visible(gr, b, (uint32_t)leader, i)) fl = BS_FL_PASS_NO_MINRES;   // :542-544
        else { fl = BS_FL_EVALUATED; slot = rr * ep.K + b.pclass[i]; }
      }
      feasible = fl == BS_FL_EVALUATED ? b.fu_feas[slot] : (fl < 16u ? nd.n : 0u);
      b.fu_slot[i] = slot;
      b.fl_feasible[i] = feasible;
    } else {
      b.fl_feasible[i] = 0;
    }
    b.fl_code[i] = fl;
    b.fflags[i] = (uint32_t)fl << 8;
    if (prm.host_tag) { b.h_pf_leader[i] = leader; b.h_fl_code[i] = fl; b.h_fl_feasible[i] = prm.run_filter ? feasible : 0u; b.h_fl_slot[i] = slot; }
    if (gi >= 0 && (uint32_t)gi < gr.g && pass && feasible > 0) admit = true;
    // BS_BATCH_FILTER_DENY: Filter fails on some node -> the group's first such pod (k_fd_apply takes it from here, bs_fdeny.hpp)
    if (prm.filter_deny && fl == BS_FL_EVALUATED && feasible < nd.n) atomicMin(&b.fd_event[gi], ((unsigned long long)prm.seq_inv << 32) | i);
  }
  if (prm.host_tag && prm.run_filter) {             // per-row feasible counts of the (run, class) slots in use
    const uint32_t U = min((ep.R + 1u) * ep.K, b.hstride);
    for (uint32_t k = i; k < U; k += gridDim.x * 256u) b.h_feas[k] = b.fu_feas[k];
  }
  const bool grouped = i < pods.p && gi >= 0 && (uint32_t)gi < gr.g;
  if (!prm.filter_deny) tally_tail(gr, b, prm, grouped, grouped ? (uint32_t)gi : 0u, admit, gridDim.x);       // (else: k_fd_apply, bs_fdeny.hpp)
}

}  // namespace bs
