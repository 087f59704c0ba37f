// bs_fast.hpp — the steady-state batch in TWO launches (three dependency levels; the last two share a launch).
//
// Steady state = every group already has its pod and its MinResources (nothing a pod derives can depend on
// its queue position: request classes stand for the pods) and the leader findMaxPG returns has matched pods
// (every scan query of the batch is a reservation check against ONE table: core.go:157-166).  That is the
// state a scheduler lives in once each gang has been seen; everything else takes the general chain of
// bs_kernels.hpp.
//
// What makes three dependency levels (two launches) enough:
//   * everything that depends on the pods alone is derived when the pods are loaded (bs_pods_load,
//     k_pod_pairs): request classes, per-group first pod / first non-permitted pod / first owner, and the
//     (group, request class) pairs of each group.  The per-batch pre-pass with its grid-wide minima is gone.
//   * findMaxPG depends on the group state alone: it runs when the groups are loaded / patched
//     (bs_groups_load, bs_groups_apply) and leaves leader, panic flag and the steady table's descriptor on
//     the device.
//   * nothing is reset per batch: slots carry the stamp of the batch that wrote them, the "first pod that
//     ..." minima are 64-bit atomicMin keys with the inverted batch number in the high word (a later batch
//     always wins).
//   * the running-sum table stays CHUNK-LOCAL: the scan adds a chunk's offset while the rows travel to LDS
//     (scan_core<S, true>), so the fix-up pass and its launch dependency disappear.  The scan derives the offsets
//     (exclusive prefix of the chunk totals, in registers), the key rows and the pruning bounds itself: nothing
//     follows the local scans in the building launch.
//   * a rejection (core.go:161-165) is a property of the request class, so the deny replay (core.go:105-110)
//     of a pod is "is there an earlier pod of my group whose class was rejected": a walk over the group's
//     (group, class) pairs — no grid-wide first_reject minimum, k_reject is gone.
//   * Filter's answer per (pod, node) is a bit of the pod's slot row; the pods x nodes bitmap is not
//     materialised (k_filter_expand runs only when a caller asks for it).
//
//   launch A    k_fast_query_tables        per pod: decisions that need no scan, its scan query and Filter parameters into
//                                          the class slots | chunk-local running sums of the table
//   launch B+C  k_fast_scan_filter_final   producer blocks: node scan per scan slot | computeResourceSatisfied per Filter
//                                          slot x node; final blocks (same launch, handed the producers' count): REJECT / deny
//                                          replay / stale leader, Filter code + slot + feasible count per pod, per-group admit
//                                          counts and the quorum predicate core.go:303 (the lane that completes a group)
#pragma once

#include "bs_kernels.hpp"
#include "bs_filter_t.hpp"

#ifdef BS_NT_TABLES
#define BS_TBL_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define BS_TBL_STORE(p, v) (*(p) = (v))
#endif
namespace bs {

template <bool INL = false>
__device__ __forceinline__ void arm_tally(const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, uint32_t i, uint32_t nthreads);

// ------------------------------------------------------------------------------------------------
// bs_pods_load: what the batch needs from the pods alone.
//   gstat[0][g] first pod of group g (shard ownership)           gstat[1][g] first pod without LAST_PERMITTED
//   gstat[2][g] first such pod with OwnerReferences              gstat[3][g] head of the group's pair chain
// A pair = (group, request class); its id is the index of its representative pod.
// ------------------------------------------------------------------------------------------------
#if BS_EMIT_MAIN
__global__ void k_pods_prep(unsigned long long* tables, uint32_t ntab, uint32_t* gstat, uint32_t ngstat, uint32_t* kcount, uint32_t* gcount, uint32_t G) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (uint32_t i = t; i < ntab; i += nt) tables[i] = 0ull;
  for (uint32_t i = t; i < G; i += nt) gcount[i] = 0;                // pods per group: counted by k_pod_pairs
  for (uint32_t i = t; i < ngstat; i += nt) gstat[i] = BS_INF;     // [3][G] minima + [G] 64-bit chain heads = 5 G words, all ones
  if (t == 0 && kcount) *kcount = 0;
}
#endif

// Second half of the class builder (every pod takes its representative's dense id) fused with the per-group
// minima and the pair table.  `hinfo` = pinned host memory: K is handed to the host without a copy or an event
// (value, then the tag with system-scope release; the host only looks when it needs the row count).
#if BS_EMIT_MAIN
__global__ void k_pod_pairs(PodsDev pods, uint32_t G, const uint32_t* rep, const uint32_t* id, uint32_t* pclass, unsigned long long* slots, uint32_t mask,
                            uint32_t hash_keep, uint32_t* gstat, uint32_t* ppair, unsigned long long* pair_next, const uint32_t* kcount, int32_t tag,
                            int32_t* hinfo, uint32_t* gcount) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && hinfo) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&hinfo[4]), ((unsigned long long)(uint32_t)tag << 32) | *kcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // K + its tag in one store
  }
  {
    const int32_t gq = i < pods.p ? pods.group[i] : -1;
    const bool in = gq >= 0 && (uint32_t)gq < G;
    wave_aggregated_add(gcount, in ? (uint32_t)gq : 0u, in);          // (every lane of the wave gets here)
  }
  if (i >= pods.p) return;
  const uint32_t c = id[rep[i]];
  if (pclass) pclass[i] = c;
  const int32_t gi = pods.group[i];
  if (gi < 0 || (uint32_t)gi >= G) { ppair[i] = BS_INF; return; }
  atomicMin(&gstat[gi], i);
  if (!(pods.flags[i] & BS_POD_LAST_PERMITTED)) {
    atomicMin(&gstat[(size_t)G + gi], i);
    if (pods.owner[i] != 0) atomicMin(&gstat[(size_t)2 * G + gi], i);
  }
  const uint64_t h = mix64(((uint64_t)(uint32_t)gi << 32) | c);
  bool winner;
  const uint32_t r = dedupe_insert(slots, mask, hash_keep, h, i, [&](uint32_t o) { return o < pods.p && pods.group[o] == gi && id[rep[o]] == c; }, winner);
  ppair[i] = r;
  // chain links carry the class of the pair they point to: (class << 32) | representative — the walker can ask for the
  // class slot's scan result in the same round trip as the pair's own fields
  if (winner) {
    unsigned long long* head = reinterpret_cast<unsigned long long*>(gstat + (((size_t)3 * G + 1) & ~(size_t)1)) + gi;   // 8-byte aligned behind the minima
    pair_next[i] = atomicExch(head, ((unsigned long long)c << 32) | i);
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// bs_groups_load / bs_groups_apply: findMaxPG for the loaded state + what the host wants to know about it
// (read back asynchronously; bs_batch_run waits for it only if it has not arrived yet).
//   info[0] leader (-1 none)  info[1] panic  info[2] steady table id (-1: none)  info[3] sequence tag
// ------------------------------------------------------------------------------------------------
struct GroupDelta { uint32_t index, matched, status_scheduled, flags; };

#if BS_EMIT_MAIN
__global__ void k_groups_apply(const GroupDelta* d, uint32_t n, uint32_t* matched, uint32_t* status_scheduled, uint8_t* flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const GroupDelta x = d[t];
  matched[x.index] = x.matched;
  status_scheduled[x.index] = x.status_scheduled;
  flags[x.index] = (uint8_t)x.flags;
}
#endif

constexpr int kInlineDeltas = 48;                  // group deltas that ride in the kernel arguments (no H2D, no staging)
struct DeltaPack { uint32_t n; GroupDelta d[kInlineDeltas]; };

// [apply up to kInlineDeltas group deltas] -> findMaxPG -> the steady table's descriptor -> info to the host.
// `info` is pinned host memory the kernel writes directly; the tag goes last with system-scope release.
// One round trip for everything the fold of core.go:701-739 needs (flags, MinMember, Status.Scheduled, matched and the fit
// class of every group a thread owns: all loads issued together), ONE 64-bit block maximum of (progress + 1) << 32 | ~index
// (largest progress, FIRST group at it), and the thread that owns the winner already holds what the descriptor and the host
// want to know — no second trip.  Only when the winner is fully scheduled (the tie rule :729-731 may hand over) or G is
// beyond what the registers hold does it fall back to the general fold (leader_block).
__device__ __forceinline__ void leader_publish(const GroupsDev& gr, const BatchDev& b, uint32_t C, int32_t tag, int32_t* info, int32_t l, int32_t pn,
                                               uint32_t l_matched, uint8_t l_flags, uint32_t l_cls) {
  int32_t steady = -1;
  if (!pn && l >= 0 && C && l_matched > 0 && (l_flags & BS_GROUP_HAS_POD) && l_cls < C) {
    steady = (int32_t)(C + l_cls);
    TableDesc d;
    d.cls = l_cls;
    d.pct = 0.7f;                                   // core.go:161
    b.desc[steady] = d;
  }
  b.leader_epoch[0] = l;
  b.panic_epoch[0] = pn ? 1 : 0;
  // to the host: three words, drained, then the tag — all system-scope stores that go straight out (posted PCIe writes to one
  // destination arrive in order).  A release fence at system scope would also write this XCD's L2 back first (~1-2 us) for
  // device-side words only the next launch reads.
  __hip_atomic_store(&info[0], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&info[1], pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&info[2], steady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(&info[3], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (a block of kLeaderBlock threads; k_leader_info is this and nothing else, k_pods_apply runs it in one extra block when a
// group patch and a queue patch arrive in the same cycle)
__device__ __forceinline__ void leader_info_block(const GroupsDev& gr, const BatchDev& b, uint32_t C, int32_t tag, int32_t* info, const DeltaPack& dp,
                                                  uint32_t* matched, uint32_t* status_scheduled, uint8_t* flags) {
  __shared__ unsigned long long s_w[kLeaderBlock / 64];
  __shared__ uint32_t s_slow;
  if (threadIdx.x < dp.n) {
    const GroupDelta x = dp.d[threadIdx.x];
    matched[x.index] = x.matched;
    status_scheduled[x.index] = x.status_scheduled;
    flags[x.index] = (uint8_t)x.flags;
  }
  if (threadIdx.x == 0) s_slow = 0;
  __syncthreads();                                     // (the stores are performed: every thread may load a patched group)
  if (!gr.g) {
    if (threadIdx.x == 0) leader_publish(gr, b, C, tag, info, -1, 0, 0u, 0, 0u);
    return;
  }
  if (gr.g <= (uint32_t)kLeaderBlock * kLeaderPerThread) {
    uint32_t f_[kLeaderPerThread], mm_[kLeaderPerThread], sc_[kLeaderPerThread], ma_[kLeaderPerThread], cl_[kLeaderPerThread];
#pragma unroll
    for (int it = 0; it < kLeaderPerThread; ++it) {
      const uint32_t g = threadIdx.x + (uint32_t)it * kLeaderBlock;
      f_[it] = mm_[it] = sc_[it] = ma_[it] = cl_[it] = 0;
      if (g < gr.g) { f_[it] = gr.flags[g]; mm_[it] = gr.min_member[g]; sc_[it] = gr.status_scheduled[g]; ma_[it] = gr.matched[g]; cl_[it] = gr.cls[g]; }
    }
    unsigned long long best = 0;
    bool panic = false;
#pragma unroll
    for (int it = 0; it < kLeaderPerThread; ++it) {
      const uint32_t g = threadIdx.x + (uint32_t)it * kLeaderBlock;
      // candidates of the fold: groups with their pod that have not been let through (core.go:705-711); epoch 0 = the loaded state
      if (g >= gr.g || (f_[it] & BS_GROUP_SCHEDULED_LATCH) || !(f_[it] & BS_GROUP_HAS_POD)) continue;
      uint32_t fin = 0;
      if ((uint32_t)(mm_[it] - sc_[it]) != 0u) {                                        // :713-717
        if (mm_[it] == 0u) panic = true;
        else fin = (uint32_t)((uint32_t)(ma_[it] + sc_[it]) * 1000u) / mm_[it];
      }
      const unsigned long long k64 = (((unsigned long long)fin + 1ull) << 32) | (unsigned long long)(0xFFFFFFFFu - g);
      best = k64 > best ? k64 : best;
    }
    if (panic) best = ~0ull;
    const unsigned long long top = block_max_u64(best, s_w);
    if (top == ~0ull || top == 0ull) {
      if (threadIdx.x == 0) leader_publish(gr, b, C, tag, info, -1, top ? 1 : 0, 0u, 0, 0u);
      return;
    }
    const uint32_t first = 0xFFFFFFFFu - (uint32_t)top;
    const int own = (first % kLeaderBlock) == threadIdx.x ? (int)(first / kLeaderBlock) : -1;
    if (own >= 0) {
      uint32_t mm = 0, sc = 0, ma = 0, cl = 0, fl = 0;
#pragma unroll
      for (int it = 0; it < kLeaderPerThread; ++it)
        if (it == own) { mm = mm_[it]; sc = sc_[it]; ma = ma_[it]; cl = cl_[it]; fl = f_[it]; }
      if (sc >= mm) s_slow = 1;                                                         // the tie rule may hand over: general fold
      else leader_publish(gr, b, C, tag, info, (int32_t)first, 0, ma, (uint8_t)fl, cl);
    }
    __syncthreads();
    if (!s_slow) return;
  }
  // cap_epoch mirrors HAS_POD for epoch 0 (k_init / the positional analysis keep it so)
  leader_block(gr, b, 0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t l = b.leader_epoch[0], pn = b.panic_epoch[0];
    leader_publish(gr, b, C, tag, info, l, pn, l >= 0 ? gr.matched[l] : 0u, l >= 0 ? gr.flags[l] : (uint8_t)0, l >= 0 ? gr.cls[l] : 0u);
  }
}

#if BS_EMIT_MAIN
__global__ __launch_bounds__(kLeaderBlock) void k_leader_info(GroupsDev gr, BatchDev b, uint32_t C, int32_t tag, int32_t* info, DeltaPack dp,
                                                              uint32_t* matched, uint32_t* status_scheduled, uint8_t* flags) {
  leader_info_block(gr, b, C, tag, info, dp, matched, status_scheduled, flags);
}
#endif

// ------------------------------------------------------------------------------------------------
// launch A, table part: chunk-local running sums (core.go:602,621 restarted at every 256-row chunk), chunk
// totals, per 64-row group max of the local sums, per chunk first row of every scalar key.
// ------------------------------------------------------------------------------------------------
template <int TS>
__device__ __forceinline__ void tables_local_fast(const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const TableDesc* forced, uint32_t chunk) {
  __shared__ unsigned long long s_wtot[BS_MAX_LANES][4];
  __shared__ uint32_t s_kp[BS_MAX_SCALARS];
  const uint32_t k = chunk * kTblChunk + threadIdx.x;
  const bool valid = k < nd.m;
  const uint32_t n = valid ? nd.kmap[k] : 0u;
  const TableDesc d = *forced;
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S(), LP = prm.LP;
  int64_t* T = b.tables;
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
  BS_STAMP(1, 1);
  if (threadIdx.x < BS_MAX_SCALARS) s_kp[threadIdx.x] = BS_INF;
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
      if (valid) BS_TBL_STORE(&T[(size_t)k * LP + j], (int64_t)incl[j]);
      if (threadIdx.x == 0) b.chunk_tot[(size_t)chunk * 16 + j] = tot;
    } else if (j < LP && valid) {
      BS_TBL_STORE(&T[(size_t)k * LP + j], (int64_t)INT64_MAX);
    }
  }
  // per 64-row group: max of the local sums per resource lane; the scan bounds the group's FINAL sums with max + chunk
  // offset.  That bound is exact when nothing can wrap: a group whose local sums leave (-2^62, 2^62) is marked
  // "cannot be pruned" (INT64_MAX) here, and the scan does not prune behind an offset outside that range either.
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
        if (lane_id() == 63 && grp_valid) b.gmax[(size_t)grp * LP + j] = risky ? INT64_MAX : mx;
      }
    }
  }
#pragma unroll
  for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
    if (s < S) {
      const unsigned long long mk = __ballot(valid && (pres & (1u << s)));
      if (mk && lane_id() == 0) atomicMin(&s_kp[s], chunk * kTblChunk + (uint32_t)(threadIdx.x & ~63u) + (uint32_t)(__ffsll((long long)mk) - 1));
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) b.chunk_kp[(size_t)chunk * 16 + threadIdx.x] = threadIdx.x < BS_MAX_SCALARS ? s_kp[threadIdx.x] : BS_INF;
  // nothing else: offsets, key rows and pruning bounds are derived by the scan itself (scan_core<S, true>) from the chunk
  // totals, the per-chunk key rows and the per-group max / min — no ticket, no tail, no grid-wide dependency in this launch
}

// ------------------------------------------------------------------------------------------------
// launch A, pod part (core.go:88-167 up to the node scan).  Differences to query_thread: no capture epochs
// (one findMaxPG result for the batch, from the group load), the per-group minima come from the pod load,
// every scan query is a reservation check -> slot = request class, slots are stamped, not reset.
// ------------------------------------------------------------------------------------------------
// What a steady-state batch needs from a leader group (the batch's, or the one carried into it): MinResources, MinMember,
// Status.Scheduled, matched.  Wave-uniform addresses; loaded by every thread at the very top, together with the pod's own
// fields, so that nothing the pod derives later waits for another round trip.
struct LeaderPre { Res mr; bool have_mr; int64_t min_member, status_scheduled, matched; };
// stores a consumer block of the SAME launch reads after a ticket (k_fast_step_a): write-through, agent scope (sc1) — a plain store
// stays in this XCD's L2 and another XCD's reader would see the previous batch's value (MI355X_MICROARCH.md, inter-workgroup visibility)
template <bool PUB, typename T>
__device__ __forceinline__ void st_pub(T* p, T v) {
  if constexpr (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// Host mirrors (pinned memory the host reads when the completion word arrives, latency mode).  BS_HOME_WT=1 (an experiment, off): stored write-through at
// system scope, so that a block's "all my mirrors are out" would be s_waitcnt vmcnt(0) instead of final_tail's system-scope release fence.  Measured
// (profiles/r06_home_wt_ab.txt): the resident cycle's wait goes from 31 to 95 us at cfg3 — system-scope stores cross PCIe lane by lane (60 000 small writes
// instead of a few thousand combined ones).  What the fence's ~7 us are is the mirrors' own way home: 340 KB (per-pod arrays + Filter rows) at PCIe rate.
#ifndef BS_HOME_WT
#define BS_HOME_WT 0
#endif
template <typename T>
__device__ __forceinline__ void st_home(T* p, T v) {
#if BS_HOME_WT
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *p = v;
#endif
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int TS>
__device__ __forceinline__ void leader_pre_load(const GroupsDev& gr, int32_t leader, Shape<TS> sh, LeaderPre& o) {
  const uint32_t l = leader >= 0 ? (uint32_t)leader : 0u;           // (clamped: the values are only used when leader >= 0)
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) o.mr.v[j] = j < sh.L() ? gr.minres[(size_t)j * gr.g + l] : 0;
  o.mr.present = gr.mrpres[l];
  o.have_mr = (gr.flags[l] & BS_GROUP_HAS_MINRES) != 0;             // steady state: every group has it (bs_batch_run's chain choice)
  o.min_member = (int64_t)gr.min_member[l];
  o.status_scheduled = (int64_t)gr.status_scheduled[l];
  o.matched = (int64_t)gr.matched[l];
}
// getPreAllocatedResource (core.go:774-793) from the preloaded leader
template <int TS>
__device__ __forceinline__ void pre_allocated_from(const LeaderPre& lp, Shape<TS> sh, uint32_t gate, Res& out) {
  res_zero(out, sh);
  const int64_t mm = lp.min_member;
  const int64_t not_finished = lp.matched != 0 ? mm - lp.matched : mm - lp.status_scheduled;
  if (not_finished > 0 && lp.have_mr) {
    Res times;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < sh.L()) times.v[j] = wmul(lp.mr.v[j], not_finished);
    times.present = lp.mr.present;
    res_add(out, times, sh, gate);
  }
  if (out.v[BS_LANE_PODS] == 0) out.v[BS_LANE_PODS] = mm + 1;
}
// the Filter slot of (class of `cur`, leader `lp`): computeResourceSatisfied's R = pod + maxSingle, M = maxSingle (core.go:526-552); ff bit 0: case 2 can
// never hold, bit 1: every node "cannot hold" a leader member
template <int TS>
__device__ __forceinline__ uint32_t filter_slot_values(const Res& cur_in, const LeaderPre& lp, Shape<TS> sh, uint32_t gate, int64_t (&R)[4], int64_t (&M)[4]) {
  Res ms, cur = cur_in;
  res_zero(ms, sh);
  res_add(ms, lp.mr, sh, gate);                                      // :526-527
  res_add(cur, ms, sh, gate);                                        // :551-552
  uint32_t ff = 0;
#pragma unroll
  for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
    if (s < sh.S()) {
      if ((cur.present & (1u << s)) && cur.v[4 + s] != 0) ff |= 1u;  // case 2 can never hold
      if ((ms.present & (1u << s)) && ms.v[4 + s] != 0) ff |= 2u;    // node "cannot hold" a leader member
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { R[j] = cur.v[j]; M[j] = ms.v[j]; }
  return ff;
}
template <int TS, bool PUB = false>
__device__ __forceinline__ void filter_slot_from(const BatchDev& b, const BatchParams& prm, const Res& cur_in, const LeaderPre& lp, Shape<TS> sh, uint32_t gate,
                                                 uint32_t slot, bool zero_feas = true) {
  if (!lp.have_mr) return;                                           // PASS_NO_MINRES (core.go:542-544): no slot to evaluate
  int64_t R[4], M[4];
  const uint32_t ff = filter_slot_values(cur_in, lp, sh, gate, R, M);
  // zero_feas = false: the whole-step launch of a small queue — its pod blocks STORE every slot's count at their end and wait for nobody, so a zero
  // written here by a block that happened to run late would be the value that stays (k_fd_apply and bs_batch_read read fu_feas[] behind the launch)
  if (zero_feas) st_pub<PUB>(&b.fu_feas[slot], 0u);
  int64_t* dst2 = b.uparams + (size_t)slot * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) { st_pub<PUB>(&dst2[j], R[j]); st_pub<PUB>(&dst2[4 + j], M[j]); }
  st_pub<PUB>(&b.uflags[slot], ff | ((uint32_t)BS_FL_EVALUATED << 8) | (prm.stamp << 16));      // (the stamp goes last)
}
// the scan query of a class against the batch's leader (core.go:157-159): pre-allocation + the class's request; returns the slot's flags word
// (present | absok << 16), scalar keys nobody asks for become INT64_MIN (never constrain)
template <int TS>
__device__ __forceinline__ uint32_t class_scan_query(const Res& cur, const LeaderPre& lp0, Shape<TS> sh, uint32_t gate, Res& q) {
  pre_allocated_from(lp0, sh, gate, q);
  res_add(q, cur, sh, gate);
  uint32_t absok = 0;
#pragma unroll
  for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
    if (s2 < sh.S()) {
      const bool pres = q.present & (1u << s2);
      if (!pres || q.v[4 + s2] == 0) absok |= 1u << s2;       // core.go:688-692
      if (!pres) q.v[4 + s2] = INT64_MIN;
    }
  }
  return q.present | (absok << 16);
}

// Three rounds of loads, issued as early as their addresses are known, then arithmetic, then stores:
//   round 1   the pod's own fields (group, flags, owner, request, class, pair) | the batch's leader and panic flag
//   round 2   the pod's group (flags, OccupiedBy, first owner, first pod) | both leaders' resources (uniform)
//   round 3   the owner of the group's first owning pod (only where OccupiedBy is still empty)
template <int TS, bool PUB = false, bool SLOTS = true, bool INL = false>
__device__ __forceinline__ void fast_query_thread(const PodsDev& pods, const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, uint32_t i,
                                                  uint32_t nthreads) {
  const Shape<TS> sh(prm.S);
  const bool valid = i < pods.p;
  const uint32_t gate = prm.eph_gate;
  const uint32_t ii = valid ? i : 0u;                                 // (clamped loads: invalid lanes read pod 0 and store nothing)
  // ---- round 1
  const int32_t leader0 = b.leader_epoch[0];
  const uint8_t panic0 = b.panic_epoch[0];
  const uint32_t K = prm.run_filter ? (prm.k_host ? prm.k_host : *b.kclass) : 0u;
  const int32_t gi = pods.p ? pods.group[ii] : BS_POD_NOT_GROUPED;
  const uint8_t pfl = pods.p ? pods.flags[ii] : (uint8_t)0;
  const uint64_t own = pods.p ? pods.owner[ii] : 0ull;
  const uint32_t pc = pods.p ? b.pclass[ii] : 0u;
  const uint32_t pp = pods.p ? b.ppair[ii] : BS_INF;
  Res cur;
  if (pods.p) pod_require(pods, ii, sh, gate, cur); else res_zero(cur, sh);
  arm_tally<INL>(gr, b, prm, i, nthreads);                             // consumed by launch C (INL: by the second half of this launch)
  if (i < 8 && prm.collect_stats) b.stats[i] = 0;
  // ---- round 2
  const bool grouped = valid && gi >= 0 && (uint32_t)gi < gr.g;
  const uint32_t g = grouped ? (uint32_t)gi : 0u;
  const uint8_t gfl = gr.g ? gr.flags[g] : (uint8_t)0;
  const uint64_t occ0 = gr.g ? gr.occupied[g] : 0ull;
  const uint32_t fo = gr.g ? b.first_owner_s[g] : BS_INF;
  const uint32_t anchor = grouped ? b.first_pod_s[g] : i;
  LeaderPre lp0{}, lp1{};
  const bool two = prm.run_filter && prm.sop_leader0 >= 0 && prm.sop_leader0 != leader0;
  if (gr.g) {
    leader_pre_load(gr, leader0, sh, lp0);
    if (two) leader_pre_load(gr, prm.sop_leader0, sh, lp1); else lp1 = lp0;
  }
  // ---- round 3
  const bool need_fo = grouped && occ0 == 0 && fo != BS_INF && i > fo;
  const uint64_t occ_fo = need_fo ? pods.owner[fo] : 0ull;
  const uint32_t owner_rank = valid ? owner_rank_of(b, prm, anchor, pods.p) : 0u;     // (a load only on sharded contexts)

  uint8_t code = BS_PF_PASS_NOT_GROUPED, st = 0;
  bool has_q = false;
  Res q;
  res_zero(q, sh);
  if (valid) {
    // shard ownership: all pods of a group live on one rank (owner_rank_of)
    if (owner_rank == prm.rank) st |= ST_OWNED;
    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                         // core.go:89-92
    else if (pfl & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;                // :95-98
    else if (!grouped) code = BS_PF_ERR_PG_NOT_FOUND;                                      // :100-103
    else if ((gfl & BS_GROUP_DENIED) || fd_denied(b, (uint32_t)gi, i)) code = BS_PF_ERR_DENIED;   // :105-110
    else {
      st |= ST_ELIG;
      bool occ_err = false;                                                                // :494-511 in queue order
      if (occ0 != 0) occ_err = (own == 0) || (own != occ0);
      else if (need_fo) occ_err = (own == 0) || (own != occ_fo);     // the group is not denied: every non-permitted pod of it is eligible
      if (occ_err) code = BS_PF_ERR_OCCUPIED;                                              // :113-115
      else if (panic0) code = BS_PF_PANIC_DIV0;                                            // :716-717
      else {
        st |= ST_REACH6;                                                                   // :118-123
        if (leader0 < 0) code = BS_PF_PASS_NO_MAX;                                         // :127-130
        else if (leader0 == gi) code = BS_PF_PASS_IS_MAX;                                  // :150-155 (leader.matched > 0 on this path)
        else {                                                                             // :157-166
          pre_allocated_from(lp0, sh, gate, q);
          res_add(q, cur, sh, gate);
          code = BS_PF_PASS_RESERVE_FITS;                                                  // tentative
          has_q = (st & ST_OWNED) != 0;
        }
      }
    }
    if (has_q) st |= ST_QUERY;
    b.tcode[i] = code;
    b.stage[i] = st;
  }
  BS_STAMP(1, 1);
  {
    // First pod of the queue that reaches findMaxPG (it really does: a replayed deny needs an earlier, reaching, rejected
    // pod).  Blocks are in queue order, so it is the first reaching pod of the first block that has one: every block
    // leaves its own candidate in its own word (a single shared minimum was ~800 atomics on one address at 50k pods).
    __shared__ uint32_t s_reach;
    if (threadIdx.x == 0) s_reach = BS_INF;
    __syncthreads();
    const unsigned long long rb = __ballot(valid && (st & ST_REACH6));
    if (rb && lane_id() == __ffsll((long long)rb) - 1) atomicMin(&s_reach, i);
    __syncthreads();
    if (threadIdx.x == 0) st_pub<INL>(&b.first_reach64[blockIdx.x], ((unsigned long long)prm.seq_inv << 32) | s_reach);
  }
  BS_STAMP(1, 2);
  // Filter slots: class c with the batch's leader, class c + K with the leader carried into the batch.  Every pod of a
  // class that may pass and is not in the leader's own group derives the same slot contents: one lane per (wave, class)
  // is elected to fill it (tens of thousands of identical stores to a few hundred cache lines were a measurable part
  // of this launch; electing ONE writer per batch with an atomic swap of the stamp was worse: a hot-spot of returning atomics).
  const uint32_t qslot = has_q ? pc : 0u;
  // (SLOTS == false, k_fast_step_a's class-slot form: the slots of EVERY class are written by class_slots_block of the same launch, from the
  // class directory — nothing here may store into them: the scan blocks of that launch are already taking minima in first_row[])
  bool fill = SLOTS && wave_elect_by_key(qslot, has_q);         // one writer per (wave, class): ~10x fewer identical stores, no atomics
  bool w1 = false, w2 = false;
  if (SLOTS && prm.run_filter) {
    const bool may = valid && (st & ST_OWNED) && BS_PF_IS_PASS(code) && grouped;
    w1 = wave_elect_by_key(pc, may && leader0 >= 0 && leader0 != gi);
    w2 = wave_elect_by_key(pc, may && prm.sop_leader0 >= 0 && prm.sop_leader0 != gi);
  }
  if constexpr (PUB) {
    // k_fast_step_a: the slots are read by other blocks of THIS launch, so they go out write-through (st_pub) — one fabric write per
    // store, and ~2 400 (wave, class) writers x 18 stores kept the pod blocks draining for 30 us (profiles/r05_stamps_step_a.txt; a
    // look at the stamp first does not help: every wave looks before anybody's stamp has landed).  One writer per slot and batch: the
    // elected lanes CLAIM their slots with a swap of the stamp word, all three swaps in flight together; whoever finds the batch's
    // stamp already there leaves the slot to the claimant (whose block finishes its stores before it takes the ticket the readers
    // wait for).
    uint32_t o1 = 0, o2 = 0, o3 = 0;
    // (claim words of their own for the Filter slots: a swap and a write-through store of DIFFERENT values to one word were seen to
    // land in either order — the readers found the claim, not the flags; the scan slot's stamp word takes the same value both times)
    if (w1) o1 = __hip_atomic_exchange(&b.uclaim[pc], prm.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w2) o2 = __hip_atomic_exchange(&b.uclaim[pc + K], prm.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fill) o3 = __hip_atomic_exchange(&b.qstamp_s[qslot], prm.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w1 && o1 == prm.stamp) w1 = false;
    if (w2 && o2 == prm.stamp) w2 = false;
    if (fill && o3 == prm.stamp) fill = false;
  }
  if (w1) filter_slot_from<TS, PUB>(b, prm, cur, lp0, sh, gate, pc);
  if (w2) filter_slot_from<TS, PUB>(b, prm, cur, lp1, sh, gate, pc + K);
  BS_STAMP(1, 3);
  if (has_q) {
    b.qpos[i] = qslot;
    atomicMin(&b.pair_firstq[pp], ((unsigned long long)prm.seq_inv << 32) | i);
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
    const uint32_t slot = qslot;
    int64_t* dst = b.qreq_s + (size_t)slot * prm.LP;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < prm.LP) st_pub<PUB>(&dst[j], j < sh.L() ? q.v[j] : (int64_t)INT64_MIN);
    st_pub<PUB>(&b.qflags_s[slot], q.present | (absok << 16));
    st_pub<PUB>(&b.qtab_s[slot], (int32_t)0);
    st_pub<PUB>(&b.first_row[slot], BS_INF);                  // every writer stores the same; launch B takes minima
    st_pub<PUB>(&b.qstamp_s[slot], prm.stamp);
  }
  if (prm.collect_stats) {
    const unsigned long long hq = __ballot(has_q);
    if (lane_id() == 0 && hq) atomicAdd((unsigned long long*)&b.stats[2], (unsigned long long)__popcll(hq));
  }
}

// Node words (round 6): what computeResourceSatisfied (core.go:514-564) needs of a NODE that no pod has a say in, once per batch instead of once
// per (tile of 64 request slots, 64-node block) in the Filter items of the throughput regime (filter_item_t, bs_filter_t.hpp).  Thread = node,
// wave = 64-node block: the nodes Filter can evaluate (getLeftResource finds them: core.go:442-449) and, for each of the batch's two leaders
// (findMaxPG's result, and sop.maxFinishedPG as it was carried into the batch), the nodes whose left covers one member of the leader's gang
// (maxSingle = Resource{} + MinResources, core.go:526-527) — case 3 lets exactly the OTHER nodes pass (core.go:558-563).  The item then fetches two
// words per block through the scalar cache where it loaded 33 bytes per node into the lanes, double-buffered, and took six ballots (PMC of round 5:
// 3.6 VALU per node and wave against the 2.1 of the node loop, profiles/r05_distinct4_pmc_summary.txt).  The maxSingle the words were built from is
// left beside them: an item whose slots carry another one (never in this chain today) keeps the old path.
template <int TS>
__device__ __forceinline__ void node_words_block(const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchParams& prm, uint32_t blk) {
  const Shape<TS> sh(prm.S);
  const uint32_t gate = prm.eph_gate;
  const uint32_t n = blk * kTblChunk + threadIdx.x, w = n >> 6, stride = b.nodew_stride;
  const int32_t leaders[2] = {b.leader_epoch[0], prm.sop_leader0};
  int64_t l[4] = {INT64_MIN, INT64_MIN, INT64_MIN, INT64_MIN};
  uint8_t fl = 0xFF;
  if (n < nd.n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = nd.left4[(size_t)j * nd.stride + n];
    fl = nd.flags[n];
  }
  const unsigned long long ok = __ballot(fl != 0xFF && !(fl & (BS_NODE_NIL | BS_NODE_NO_NODE)));
  if (lane_id() == 0) {                                                   // table 2: a leader no node can hold a member of (scalar MinResources)
    b.nodew[((size_t)2 * stride + w) * 2] = ok;
    b.nodew[((size_t)2 * stride + w) * 2 + 1] = ~0ull;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    LeaderPre lp{};
    const bool have = gr.g != 0 && leaders[s] >= 0;
    if (have) leader_pre_load(gr, leaders[s], sh, lp);
    Res ms;
    res_zero(ms, sh);
    if (have && lp.have_mr) res_add(ms, lp.mr, sh, gate);               // core.go:526-527
    uint32_t ff = (have && lp.have_mr) ? 1u : 0u;
#pragma unroll
    for (uint32_t q = 0; q < BS_MAX_SCALARS; ++q)
      if (q < sh.S() && (ms.present & (1u << q)) && ms.v[4 + q] != 0) ff |= 2u;
    const unsigned long long holds = __ballot(l[0] >= ms.v[0] && l[1] >= ms.v[1] && l[2] >= ms.v[2] && l[3] >= ms.v[3]);
    if (lane_id() == 0) {
      b.nodew[((size_t)s * stride + w) * 2] = ok;
      b.nodew[((size_t)s * stride + w) * 2 + 1] = ~holds;
    }
    if (blk == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) b.nodew[(size_t)6 * stride + 4 * s + j] = (uint64_t)ms.v[j];
      b.nodew[(size_t)6 * stride + 8 + s] = ff;
    }
  }
}

// block layout: [0, query_blocks) pods | query_blocks + c: table chunk c | behind the chunks: node words (only launched when launch B takes the
// transposed Filter item)
template <int TS>
__global__ __launch_bounds__(kTblChunk) void k_fast_query_tables(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchDev bt, BatchParams prm,
                                                                 const TableDesc* forced, uint32_t nchunks, uint32_t query_blocks) {
  BS_STAMP(1, 0);
  if (blockIdx.x < query_blocks) fast_query_thread<TS>(pods, gr, b, prm, blockIdx.x * kTblChunk + threadIdx.x, query_blocks * kTblChunk);
  else if (blockIdx.x < query_blocks + nchunks) tables_local_fast<TS>(nd, bt, prm, forced, blockIdx.x - query_blocks);
  else node_words_block<TS>(gr, nd, b, prm, blockIdx.x - query_blocks - nchunks);
  BS_STAMP(1, 7);
}

// ------------------------------------------------------------------------------------------------
// Round 5: launch A and the scan / Filter roles of launch B as ONE launch — k_fast_step_a — followed by k_fast_final.
//
// The three-level step (tables -> scan -> final) spends most of its 21 us on what sits BETWEEN the levels (stamps,
// profiles/r03_stamps_step.txt): launch A's drain, the boundary, the scan blocks' first fetches of what A wrote (slots, chunk totals,
// group maxima, then the table rows of a live group), 944 KB of table rows that go out through L2 and come back.  Here the block that
// BUILDS chunk c of the table keeps its 256 rows in registers (thread = row) and scans them itself:
//   pod blocks    [0, qb)                    fast_query_thread, slot stores write-through (st_pub), then ticket[8] += 1
//   table blocks  qb + c * SS + q            chunk c (built by each of its SS blocks: a few us of arithmetic on L2-resident node lanes),
//                                            publishes the chunk's totals / first key rows (q == 0), ticket[9] += 1; waits for BOTH
//                                            tickets; chunk offset = exclusive prefix of the totals (<= 64 chunks: one lane each);
//                                            then share q of the class slots against its rows: lanes are ROWS, a slot's request is the
//                                            uniform operand (LDS copy of the block's slots), first row per slot -> atomicMin(first_row)
//   filter blocks the rest                   wait for ticket[8], agent acquire, filter_loop<2> as in launch B
// No table row, group maximum or chunk key row is written at all.  A slot's first row is the minimum over the chunks of the first row
// inside each chunk: the same answer as the legacy scan's early exit.  Tickets are never reset: the host passes the counter values
// this launch starts from (wrap-safe differences).  Every block of the grid must be resident at once (the waiting blocks spin):
// run_fast asks the occupancy API and takes the legacy chain otherwise; spins are bounded and raise the context's error word.
// ------------------------------------------------------------------------------------------------
// lanes whose row covers the request on every resource lane: compareResourceAndRequire (core.go:672-699) as an EXEC chain, row and
// request both per-lane operands (rows in the lanes, the slot's request broadcast from LDS)
template <int NL>
__device__ __forceinline__ unsigned long long rows_cover(unsigned long long m, const int64_t (&a)[BS_MAX_LANES], const int64_t (&r)[BS_MAX_LANES]) {
  asm volatile("s_mov_b64 exec, %[m]\n\tv_cmpx_ge_i64 vcc, %[a0], %[r0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[r1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[r2]\n\t"
               "v_cmpx_ge_i64 vcc, %[a3], %[r3]\n\ts_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1"
               : [m] "+s"(m)
               : [a0] "v"(a[0]), [r0] "v"(r[0]), [a1] "v"(a[1]), [r1] "v"(r[1]), [a2] "v"(a[2]), [r2] "v"(r[2]), [a3] "v"(a[3]), [r3] "v"(r[3])
               : "vcc");
#pragma unroll
  for (int j = 4; j < NL; ++j)
    asm volatile("s_mov_b64 exec, %[m]\n\tv_cmpx_ge_i64 vcc, %[a], %[r]\n\ts_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1" : [m] "+s"(m) : [a] "v"(a[j]), [r] "v"(r[j]) : "vcc");
  return m;
}
#ifndef BS_SCAN_BALLOT
#define BS_SCAN_BALLOT 1
#endif
// In-launch waits are bounded: kSpinBound looks (each an agent-scope load, about a microsecond) — a quarter of a second where a wait that goes well takes
// tens of microseconds; then the context's error word is raised, the batch is void (BS_ERR_RETRY) and the context goes back to separate launches.
constexpr uint32_t kSpinBound = 1u << 18;
// Experiment builds only (-DBS_TEST_CHAOS, profiles/r06_late_class_slots_race.txt): every block of the kernels that hand data over inside a launch starts 0 .. ~50 us
// late, a different amount per block and launch — what a busy GPU can do to block order.  Results must not change; only the time may.
__device__ __forceinline__ void test_chaos_delay() {
#ifdef BS_TEST_CHAOS
  uint32_t h = (blockIdx.x + 1u) * 2654435761u ^ (uint32_t)wall_clock64() * 40503u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  for (uint32_t s = 0; s < (h & 15u); ++s) __builtin_amdgcn_s_sleep(127);
#endif
}
constexpr uint32_t kStepSlotsMax = 256;        // class slots the one-launch form handles (the latency regime: K <= 256)
__device__ __forceinline__ bool step_wait(const uint32_t* word, uint32_t base, uint32_t need, int32_t* h_err) {
  uint32_t spins = 0;
  while ((uint32_t)(ld_agent(word) - base) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinBound) { if (h_err) __hip_atomic_store(h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
  }
  return true;
}
// Where the tickets of k_fast_step_a live (u32 indices into BatchDev::ticket; every word other blocks poll has a 64-byte line of its own: a poll and an
// add to one line queue behind each other at the memory side, and with one line for everything the pod blocks' polling of the last counter slowed the
// table blocks' wait for the first by 1.3 us, profiles/r06_stamps_whole_step.txt).  The two counters with hundreds of adds per launch are spread over
// kTkWays lines: same-address agent-scope adds complete one after the other (~50 ns each: 200 producers = 10 us on one word).  kTkDone: the scan / Filter blocks of a large queue (more than
// kGatherDirectBlocks pod blocks), counted in after their atomics have drained; a small queue's results are tagged words (scan_rec / feas_rec), no counter.
constexpr uint32_t kScanRecChunks = 64;  // BatchDev::scan_rec[chunk][slot]: tag << 32 | the slot's first row inside the chunk (BS_INF: none) — one writer per word
constexpr uint32_t kFeasRecGroups = 32;  // BatchDev::feas_rec[group of 4 node runs][slot]: tag << 32 | feasible nodes of the slot in those runs — one writer per word
#ifndef BS_GATHER_DIRECT
#define BS_GATHER_DIRECT 8
#endif
constexpr uint32_t kGatherDirectBlocks = BS_GATHER_DIRECT;   // up to this many pod blocks poll the result words directly (fast_final_block<true>)
constexpr uint32_t kRecStride = 32;      // 64-bit words per chunk record (BatchDev::chunk_rec): [2 j], [2 j + 1] = tag << 32 | low / high half of lane j's total; [16 + s] = tag << 32 | first row of key s
constexpr uint32_t kTkSlots = 32, kTkTab = 48, kTkP1 = 64, kTkDone = 64 + 16 * 16, kTkWays = 16, kTkWords = 64 + 2 * 16 * 16;
__device__ __forceinline__ void spread_add(uint32_t* words, uint32_t who) {
  (void)__hip_atomic_fetch_add(&words[16u * (who % kTkWays)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the whole of wave 0 calls this: lane l < kTkWays polls way l, the sum is the counter
__device__ __forceinline__ bool spread_wait(const uint32_t* words, uint32_t base, uint32_t need, int32_t* h_err) {
  uint32_t spins = 0;
  for (;;) {
    uint32_t v = lane_id() < (int)kTkWays ? ld_agent(&words[16u * (uint32_t)lane_id()]) : 0u;
#pragma unroll
    for (int o = 1; o < (int)kTkWays; o <<= 1) v += (uint32_t)__shfl_xor((int)v, o);
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    if ((uint32_t)(v - base) >= need) return true;
    __builtin_amdgcn_s_sleep(4);
    if (++spins > kSpinBound) { if (h_err && lane_id() == 0) __hip_atomic_store(h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
  }
}
// REGS (the whole step in one launch, BS_STEP_A=3): the block derives its share's scan queries ITSELF from the class directory and the leader
// (class_scan_query: what class_slots_block publishes) while it waits for the chunk totals — it waits for no slot ticket and fetches no slot; the first
// row of every slot of the share inside this chunk goes to scan_rec[chunk][slot] as a tagged word.
template <int TS, bool REGS = false>
__device__ __forceinline__ void table_scan_block(const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const TableDesc* forced, uint32_t chunk, uint32_t nchunks,
                                                 uint32_t share, uint32_t nshares, uint32_t pod_blocks, uint32_t tk_pods0, uint32_t tk_tab0, const GroupsDev& gr,
                                                 const int64_t* ckeys, const uint32_t* cpres, uint32_t kcap, bool direct = false, uint32_t forced_cls = 0) {
  __shared__ unsigned long long s_wtot[BS_MAX_LANES][4];
  __shared__ uint32_t s_kp[BS_MAX_SCALARS];
  __shared__ unsigned long long s_off[BS_MAX_LANES];
  __shared__ uint32_t s_kpg[BS_MAX_SCALARS];
  __shared__ int64_t s_req[kStepSlotsMax][BS_MAX_LANES];
  __shared__ uint32_t s_qf[kStepSlotsMax];
  __shared__ uint32_t s_first[kStepSlotsMax];
  BS_STAMP(2, 0);
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S(), LP = prm.LP;
  const uint32_t k = chunk * kTblChunk + threadIdx.x;
  // REGS (whole-step form): the table's rows are the NODES in list order, a node compareClusterResourceAndRequire skips (core.go:606-617) is a row that adds
  // nothing and can never be the first covering row — the same running sums and the same first node as over the compacted rows (kmap), without the
  // dependent trip through kmap here and without the one back from row to node in the pod blocks: a "row" of this form IS the node's list index
  const bool in_range = REGS ? k < nd.n : k < nd.m;
  const uint32_t n = REGS ? (in_range ? k : 0u) : (in_range ? nd.kmap[k] : 0u);
  // (REGS: the steady table's descriptor is a function of its slot — fit class = slot - C, 70 % (leader_publish, core.go:161) — and the host knows the
  // slot: it comes in the kernel arguments, not through a load the fit word would have to wait for)
  const TableDesc d = REGS ? TableDesc{forced_cls, 0.7f} : *forced;
  const uint32_t fw = nd.fit[(size_t)d.cls * nd.fit_words + (n >> 5)];
  const uint8_t fl = nd.flags[n];
  const bool valid = in_range && (!REGS || !(fl & BS_NODE_SKIP_MASK));
  const uint32_t ap = nd.apres[n], rp = nd.rpres[n];
  int64_t al[BS_MAX_LANES], rq[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    if (j < L) {
      al[j] = nd.alloc[(size_t)j * nd.stride + n];
      rq[j] = nd.req[(size_t)j * nd.stride + n];
    }
  }
  // REGS: the share's class keys and the leader, asked for together with the node lanes
  const uint32_t K = prm.k_host ? prm.k_host : *b.kclass;
  const uint32_t per = (K + nshares - 1u) / nshares, s_lo = share * per, s_hi = min(K, s_lo + per);
  Res raw;
  int32_t leader0 = -1;
  uint8_t panic0 = 0;
  LeaderPre lp0{};
  if constexpr (REGS) {
    const uint32_t cc = s_lo + threadIdx.x < s_hi ? s_lo + threadIdx.x : 0u;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < L) raw.v[j] = ckeys[(size_t)j * kcap + cc];
    raw.present = cpres[cc];
    leader0 = b.leader_epoch[0];
    panic0 = b.panic_epoch[0];
    if (gr.g) leader_pre_load(gr, leader0, sh, lp0);
  }
  const bool fit = valid && ((fw >> (n & 31u)) & 1u) && !(fl & BS_NODE_TAINT_ERR);
  const uint32_t pres = fit ? (ap & rp) : 0u;
  if (threadIdx.x < BS_MAX_SCALARS) s_kp[threadIdx.x] = BS_INF;
  for (uint32_t t = threadIdx.x; t < kStepSlotsMax; t += kTblChunk) s_first[t] = BS_INF;
  unsigned long long incl[BS_MAX_LANES];
  const int w = wave_id();
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    incl[j] = 0;
    if (j < L) {
      const bool live = fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
      const unsigned long long left = live ? (unsigned long long)wsub(scale_f32(al[j], d.pct), rq[j]) : 0ull;
      incl[j] = wave_incl_scan_add_u64(left);
      if (lane_id() == 63) s_wtot[j][w] = incl[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
    if (s2 < S) {
      const unsigned long long mk = __ballot(valid && (pres & (1u << s2)));
      if (mk && lane_id() == 0) atomicMin(&s_kp[s2], chunk * kTblChunk + (uint32_t)(threadIdx.x & ~63u) + (uint32_t)(__ffsll((long long)mk) - 1));
    }
  }
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
      if (threadIdx.x == 0 && share == 0) {
        if constexpr (REGS) {
          // tagged record (see kRecStride): each half of the total travels in a 64-bit word of its own beside the batch's tag — a reader that finds
          // the tag in both words has both halves of THIS batch's total, whatever order the words landed in: no drain, no ticket, no second fetch
          const unsigned long long tag = (unsigned long long)prm.seq_inv << 32;
          st_pub<true>(&b.chunk_rec[(size_t)chunk * kRecStride + 2u * j], tag | (uint32_t)tot);
          st_pub<true>(&b.chunk_rec[(size_t)chunk * kRecStride + 2u * j + 1u], tag | (uint32_t)(tot >> 32));
        } else {
          st_pub<true>(&b.chunk_tot[(size_t)chunk * 16 + j], tot);
        }
      }
    }
  }
  __syncthreads();
  __shared__ uint32_t s_ok;
  if constexpr (REGS) {
    if (share == 0 && threadIdx.x < S) st_pub<true>(&b.chunk_rec[(size_t)chunk * kRecStride + 16u + threadIdx.x], ((unsigned long long)prm.seq_inv << 32) | s_kp[threadIdx.x]);
    BS_STAMP(2, 1);
  } else {
    if (share == 0 && threadIdx.x < 16) st_pub<true>(&b.chunk_kp[(size_t)chunk * 16 + threadIdx.x], threadIdx.x < BS_MAX_SCALARS ? s_kp[threadIdx.x] : BS_INF);
    // ---- publish (share 0 of every chunk), then wait for every chunk's totals and for every pod block's slots
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BS_STAMP(2, 1);
    if (threadIdx.x == 0 && share == 0) (void)__hip_atomic_fetch_add(&b.ticket[kTkTab], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (REGS) {
    // the share's scan queries -> LDS (one slot per thread), in the shadow of the other chunks' totals
    const uint32_t t = s_lo + threadIdx.x;
    if (t < s_hi) {
      uint32_t qf = 0x80000000u;
      Res q;
      res_zero(q, sh);
      if (leader0 >= 0 && !panic0) {                                                     // core.go:157-166 for every class that could ask
        Res cur;
        res_zero(cur, sh);
        res_add(cur, raw, sh, prm.eph_gate);
        qf = class_scan_query(cur, lp0, sh, prm.eph_gate, q);
      }
      s_qf[threadIdx.x] = qf;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) s_req[threadIdx.x][j] = q.v[j];
    }
  }
  if constexpr (!REGS) {
    if (threadIdx.x == 0) s_ok = (step_wait(&b.ticket[kTkTab], tk_tab0, nchunks, b.h_err) && step_wait(&b.ticket[kTkSlots], tk_pods0, pod_blocks, b.h_err)) ? 1u : 0u;
    __syncthreads();
    BS_STAMP(2, 2);
    if (!s_ok) return;
  }
  // ---- this block's slots -> LDS; the chunk's offset and the table's first key rows (wave 0: lane <-> chunk)
  {
    // one round trip for everything: wave 0's lanes take the chunk totals / key rows (lane <-> chunk), every thread a slot of the share
    const uint32_t ch = (uint32_t)lane_id();
    unsigned long long cv[BS_MAX_LANES];
    uint32_t ckp[BS_MAX_SCALARS];
    if constexpr (REGS) {
      // wave 0 polls the records themselves: lane <-> chunk, every word of the record in one round trip, again until every chunk's words carry the tag
      if (w == 0) {
        const uint32_t tagw = prm.seq_inv;
        const unsigned long long* rec = b.chunk_rec + (size_t)(ch < nchunks ? ch : 0u) * kRecStride;
        bool ok = false;
        for (uint32_t spins = 0;; ++spins) {
          unsigned long long lo[BS_MAX_LANES], hi[BS_MAX_LANES], kw[BS_MAX_SCALARS];
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
            if (j < L) { lo[j] = ld_agent64(&rec[2u * j]); hi[j] = ld_agent64(&rec[2u * j + 1u]); }
#pragma unroll
          for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2)
            if (s2 < S) kw[s2] = ld_agent64(&rec[16u + s2]);
          bool mine_ok = true;
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
            cv[j] = 0ull;
            if (j < L) {
              mine_ok = mine_ok && (uint32_t)(lo[j] >> 32) == tagw && (uint32_t)(hi[j] >> 32) == tagw;
              cv[j] = (hi[j] << 32) | (uint32_t)lo[j];
            }
          }
#pragma unroll
          for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
            ckp[s2] = BS_INF;
            if (s2 < S) { mine_ok = mine_ok && (uint32_t)(kw[s2] >> 32) == tagw; ckp[s2] = (uint32_t)kw[s2]; }
          }
          if (__ballot(!(mine_ok || ch >= nchunks)) == 0ull) { ok = true; break; }
          if (spins > kSpinBound) { if (b.h_err && ch == 0) __hip_atomic_store(b.h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          __builtin_amdgcn_s_sleep(2);
        }
        if (ch >= nchunks) {
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) cv[j] = 0ull;
#pragma unroll
          for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) ckp[s2] = BS_INF;
        }
        if (ch == 0) s_ok = ok ? 1u : 0u;
      }
      BS_STAMP(2, 2);
    } else if (w == 0) {
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) cv[j] = (j < L && ch < nchunks) ? ld_agent64(&b.chunk_tot[(size_t)ch * 16 + j]) : 0ull;
#pragma unroll
      for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) ckp[s2] = (s2 < S && ch < nchunks) ? ld_agent(&b.chunk_kp[(size_t)ch * 16 + s2]) : BS_INF;
    }
    const uint32_t t = s_lo + threadIdx.x;
    uint32_t stp = 0, qf = 0x80000000u;
    int32_t tab = -1;
    unsigned long long rv[BS_MAX_LANES];
    if (!REGS && t < s_hi) {
      stp = ld_agent(&b.qstamp_s[t]);
      tab = (int32_t)ld_agent(reinterpret_cast<const uint32_t*>(&b.qtab_s[t]));
      qf = ld_agent(&b.qflags_s[t]);
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) rv[j] = j < L ? ld_agent64(&b.qreq_s[(size_t)t * LP + j]) : 0ull;
    }
    if (w == 0) {
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const unsigned long long incl_c = wave_incl_scan_add_u64(cv[j]);
          const unsigned long long mine = bcast64(incl_c - cv[j], (int)(chunk & 63u));
          if (ch == 0) s_off[j] = mine;
        }
      }
#pragma unroll
      for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
        if (s2 < S) {
          const uint32_t mn = wave_min_u32(ckp[s2]);
          if (ch == 0) s_kpg[s2] = mn;
        }
      }
    }
    if (!REGS && t < s_hi) {                                 // (per <= 256: one slot per thread)
      if (stp != prm.stamp || tab != 0) qf = 0x80000000u;    // not written by a pod of THIS batch: no query
      s_qf[threadIdx.x] = qf;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) s_req[threadIdx.x][j] = (int64_t)rv[j];
    }
  }
  __syncthreads();
  if constexpr (REGS) {
    if (!s_ok) return;
  }
  int64_t fin[BS_MAX_LANES];
  uint32_t keyrow = 0;                                         // bit s: key s is in the running sum at this row (core.go:686-697)
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    fin[j] = 0;
    if (j < L) fin[j] = (int64_t)(incl[j] + s_off[j]);
  }
#pragma unroll
  for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2)
    if (s2 < S && k >= s_kpg[s2]) keyrow |= 1u << s2;
  BS_STAMP(2, 3);
  // ---- the scan: lanes are rows, the slot's request is the uniform operand
  const uint32_t nmine = s_hi > s_lo ? s_hi - s_lo : 0u;
  const uint32_t smask = S ? ((1u << S) - 1u) : 0u;
  const uint32_t row0 = chunk * kTblChunk + ((uint32_t)w << 6);
  for (uint32_t t0 = 0; t0 < nmine; t0 += 4u) {            // four slots per step: their LDS reads travel together
    uint32_t qf[4];
    int64_t r[4][BS_MAX_LANES];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t t = min(t0 + u, nmine - 1u);
      qf[u] = t0 + u < nmine ? s_qf[t] : 0x80000000u;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) r[u][j] = j < L ? s_req[t][j] : 0;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      // a scalar key that is not in the running sum yet at this row: only a slot that asks for nothing of it can pass (core.go:686-697)
      const uint32_t nab = ~(qf[u] >> 16) & smask;
      unsigned long long m = (qf[u] & 0x80000000u) ? 0ull : __ballot(valid && (nab & ~keyrow) == 0u);
      static_assert(TS >= 0 && TS <= 4, "k_fast_step_a is instantiated for 0..4 scalar lanes (run_fast: step_a_possible)");
#if BS_SCAN_BALLOT
      // one v_cmp_ge_i64 -> SGPR mask per resource lane, ANDed: no EXEC write, the four slots' compares are independent of each other
#pragma unroll
      for (uint32_t j = 0; j < 4u + (uint32_t)TS; ++j) m &= __ballot(fin[j] >= r[u][j]);
#else
      m = rows_cover<4 + TS>(m, fin, r[u]);
#endif
      if (m && lane_id() == 0) atomicMin(&s_first[t0 + u], row0 + (uint32_t)(__ffsll((long long)m) - 1));
    }
  }
  __syncthreads();
  BS_STAMP(2, 4);
  if constexpr (REGS) {
    // a small queue: every (slot of the share, this chunk) word is written, found or not — the pod blocks know a slot is complete when all its chunks'
    // words carry the batch's tag: no counter to add to, no drain in front of it, and their poll IS their fetch
    if (direct) {
      for (uint32_t t = threadIdx.x; t < nmine; t += kTblChunk)
        st_pub<true>(&b.scan_rec[(size_t)chunk * kStepSlotsMax + s_lo + t], ((unsigned long long)prm.seq_inv << 32) | s_first[t]);
    } else {
      // a large queue (see kGatherDirectBlocks): one 64-bit minimum per slot, keyed by ~batch_seq (never reset), behind the kTkDone counter
      for (uint32_t t = threadIdx.x; t < nmine; t += kTblChunk)
        if (s_first[t] != BS_INF) atomicMin(&b.first_row64[s_lo + t], ((unsigned long long)prm.seq_inv << 32) | s_first[t]);
    }
    BS_STAMP(2, 7);
    return;
  }
  for (uint32_t t = threadIdx.x; t < nmine; t += kTblChunk) {
    if (s_first[t] == BS_INF) continue;
    atomicMin(&b.first_row[s_lo + t], s_first[t]);
  }
  BS_STAMP(2, 7);
}

// The Filter role of the whole-step launch: filter_loop<2>'s split into (tile of 64 slots, run of node blocks) items, with the tile's slots derived on the
// spot — lane = slot: class c = slot mod K, leader = the batch's (slots [0, K)) or the one carried in (slots [K, 2K)); what class_slots_block writes into
// uflags[] / uparams[] for that slot, computed from the class directory and the leader's MinResources (filter_slot_values), never read back.  A block takes
// FOUR node runs of ONE tile (a wave each); a small queue's block leaves the slots' feasible counts over those runs in feas_rec[group][slot] as tagged
// words (nothing is added to, nothing has to be zeroed first, nobody is waited for), a large queue's adds them to fu_feas[].
__device__ __forceinline__ void step_filter_split(uint32_t K, uint32_t W, uint32_t target_waves, uint32_t& tiles, uint32_t& bpw, uint32_t& nchunk) {
  tiles = (2u * K + 63u) / 64u;
  uint32_t nsplit = max(1u, target_waves / max(tiles, 1u));
  nsplit = min(nsplit, max((W + 1u) / 2u, 1u));
  bpw = max(2u, (((W + nsplit - 1u) / nsplit + 1u) / 2u) * 2u);
  nchunk = (W + bpw - 1u) / bpw;
  if ((nchunk + 3u) / 4u > kFeasRecGroups) {                       // (never with <= 64 table chunks; keeps the record within its row whatever the caller passes)
    bpw = ((W + 4u * kFeasRecGroups - 1u) / (4u * kFeasRecGroups) + 1u) / 2u * 2u;
    nchunk = (W + bpw - 1u) / bpw;
  }
}
template <int TS>
__device__ __forceinline__ void step_filter_block(const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const int64_t* ckeys,
                                                  const uint32_t* cpres, uint32_t kcap, uint32_t target_waves, uint32_t ustride, uint32_t bx, uint32_t nblocks,
                                                  bool direct, uint32_t tk_slots0, uint32_t slot_producers) {
  __shared__ uint32_t s_cnt[4][64];
  BS_STAMP_AT(4, 0, bx);
  const Shape<TS> sh(prm.S);
  const uint32_t gate = prm.eph_gate, K = prm.k_host ? prm.k_host : (uint32_t)__builtin_amdgcn_readfirstlane((int)*b.kclass), U = 2u * K;
  const uint32_t W = (nd.n + 63u) / 64u;
  if (!U || !W) return;
  uint32_t tiles, bpw, nchunk;
  step_filter_split(K, W, target_waves, tiles, bpw, nchunk);
  const uint32_t groups = (nchunk + 3u) / 4u;
  const int32_t leader0 = b.leader_epoch[0];
  LeaderPre lp0{}, lp1{};
  const bool two = prm.sop_leader0 >= 0 && prm.sop_leader0 != leader0;
  if (gr.g) {
    leader_pre_load(gr, leader0, sh, lp0);
    if (two) leader_pre_load(gr, prm.sop_leader0, sh, lp1); else lp1 = lp0;
  }
  bool waited = false;
  for (uint32_t bi = bx; bi < tiles * groups; bi += nblocks) {
    const uint32_t tile = bi % tiles, group = bi / tiles, chunk = group * 4u + (uint32_t)wave_id();
    const uint32_t src = tile * 64u + (uint32_t)lane_id();
    const bool second = src >= K;
    const uint32_t c = src < U ? (second ? src - K : src) : 0u;
    uint32_t cnt = 0;
    if (chunk < nchunk) {
      Res raw;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < sh.L()) raw.v[j] = ckeys[(size_t)j * kcap + c];
      raw.present = cpres[c];
      Res cur;
      res_zero(cur, sh);
      res_add(cur, raw, sh, gate);
      LeaderPre lp = lp0;
      if (second) lp = lp1;
      const bool on = src < U && (second ? prm.sop_leader0 >= 0 : leader0 >= 0) && lp.have_mr;      // class_slots_block / filter_slot_from: who gets a slot
      int64_t R[4], M[4];
      const uint32_t ff = filter_slot_values(cur, lp, sh, gate, R, M);
      filter_item<2, 4, true, true>(nd, b, U, ustride, tile, chunk * bpw, min(W, chunk * bpw + bpw), 0u,
                                    on ? (ff | ((uint32_t)BS_FL_EVALUATED << 8)) : ((uint32_t)BS_FL_NOT_RUN << 8), R, M, &cnt);
    }
    s_cnt[wave_id()][lane_id()] = cnt;
    BS_STAMP_AT(4, 1, bx);
    // latency mode, small queue: the rows this wave sent home must be out before the tagged word says the slot is complete (nobody drains for us here)
    if (direct && b.h_rows) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BS_STAMP_AT(4, 2, bx);
    if (wave_id() == 0 && src < U) {
      const uint32_t sum = s_cnt[0][lane_id()] + s_cnt[1][lane_id()] + s_cnt[2][lane_id()] + s_cnt[3][lane_id()];
      if (direct) {
        st_pub<true>(&b.feas_rec[(size_t)group * (2u * kStepSlotsMax) + src], ((unsigned long long)prm.seq_inv << 32) | sum);
      } else {
        // a large queue: the counts are added to fu_feas[], which the class-slot block zeroed (long through by now: one look)
        if (!waited) { (void)step_wait(&b.ticket[kTkSlots], tk_slots0, slot_producers, b.h_err); waited = true; }
        if (sum) atomicAdd(&b.fu_feas[src], sum);
      }
    }
    __syncthreads();
  }
  BS_STAMP_AT(4, 7, bx);
}

// Round 6, the class-slot form of the one-launch step (BS_STEP_A=2).  What made round 5's form slow was WHO publishes the slots: every pod block
// (40 blocks x 4 waves x a few classes each, write-through, a claim swap per slot) in front of the ticket every table / Filter block waits for.  But a
// class slot's content does not depend on any pod of the batch: the scan query of class c is the leader's pre-allocation + the class's request
// (core.go:157-159), its Filter parameters the class's request + the leader's MinResources (core.go:526-552, for the batch's leader and for the one
// carried in) — class request = the class directory's key (ckeys / cpres, kept by bs_pods_load / bs_pods_apply), leaders = leader_epoch[0] and
// prm.sop_leader0.  One thread per class id writes all three slots write-through; the ticket has as many producers as there are such blocks (one
// per 256 classes), the pod blocks publish nothing and gate nobody (fast_query_thread<TS, false, false>), and a slot nobody asks about costs a scan /
// Filter lane, not an answer.
template <int TS>
__device__ __forceinline__ void class_slots_block(const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, const int64_t* ckeys, const uint32_t* cpres,
                                                  uint32_t kcap, uint32_t blk, bool zero_feas) {
  const Shape<TS> sh(prm.S);
#ifdef BS_TEST_OLD_ZERO            // experiment builds only (tools/r06_flaky2.sh): the behaviour before the fix described at filter_slot_from
  zero_feas = true;
#endif
  const uint32_t gate = prm.eph_gate, K = prm.k_host ? prm.k_host : *b.kclass;        // (K not on the host yet: the grid was sized for a bound, see run_fast)
  const uint32_t c = blk * kTblChunk + threadIdx.x;
  Res raw;                                                                                 // (the class's key: asked for before the leader chain, which is two dependent trips)
  const uint32_t cc = c < K ? c : 0u;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) raw.v[j] = ckeys[(size_t)j * kcap + cc];
  raw.present = cpres[cc];
  const int32_t leader0 = b.leader_epoch[0];
  const uint8_t panic0 = b.panic_epoch[0];
  LeaderPre lp0{}, lp1{};
  const bool two = prm.run_filter && prm.sop_leader0 >= 0 && prm.sop_leader0 != leader0;
  if (gr.g) {
    leader_pre_load(gr, leader0, sh, lp0);
    if (two) leader_pre_load(gr, prm.sop_leader0, sh, lp1); else lp1 = lp0;
  }
  if (c < K) {
    Res cur;                                                                               // pod_require (core.go:761-772) on the key: the raw lanes through Resource.Add's rule
    res_zero(cur, sh);
    res_add(cur, raw, sh, gate);
    if (leader0 >= 0 && !panic0) {                                                       // core.go:157-166 for every class that could ask
      Res q;
      const uint32_t qfw = class_scan_query(cur, lp0, sh, gate, q);
      int64_t* dst = b.qreq_s + (size_t)c * prm.LP;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < prm.LP) st_pub<true>(&dst[j], j < sh.L() ? q.v[j] : (int64_t)INT64_MIN);
      st_pub<true>(&b.qflags_s[c], qfw);
      st_pub<true>(&b.qtab_s[c], (int32_t)0);
      st_pub<true>(&b.first_row[c], BS_INF);
      st_pub<true>(&b.qstamp_s[c], prm.stamp);
    }
    if (prm.run_filter) {
      if (leader0 >= 0) filter_slot_from<TS, true>(b, prm, cur, lp0, sh, gate, c, zero_feas);
      if (prm.sop_leader0 >= 0) filter_slot_from<TS, true>(b, prm, cur, lp1, sh, gate, c + K, zero_feas);
    }
  }
  BS_STAMP(1, 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  BS_STAMP(1, 5);
  if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&b.ticket[kTkSlots], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ------------------------------------------------------------------------------------------------
// The common ends of a steady-state / positional batch (final blocks of k_fast_scan_filter_final, k_fast_final, k_epoch_final;
// 256 threads per block).
//   arm_tally    (launch A) zero the per-group counters of this batch; groups without a pod in the queue get their quorum
//                answer right away (nobody will come by to close them)
//   tally_tail   (launch C) per-group admit counts and the Permit quorum (core.go:303).  Single context: every wave adds
//                (pods seen << 32 | pods admitted) to its groups' 64-bit counters with RETURNING atomics, all in flight
//                together; the lane whose add brings "pods seen" up to the group's pod count closes the group — admit count,
//                ready bit, host mirrors.  No ticket, no last block that walks all groups after everybody else is done.
//                Sharded / external reduction: fire-and-forget adds into admit[] (the collective and k_ready follow).
//   final_tail   latency mode only: the LAST block to get here publishes the completion word the host polls.
// ------------------------------------------------------------------------------------------------
template <bool INL>
__device__ __forceinline__ void arm_tally(const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, uint32_t i, uint32_t nthreads) {
  // INL (k_fast_step_a, whole step in one launch): the counters are added to and closed by blocks of THIS launch — the zeros go out write-through
  // (a plain store parked in this XCD's L2 would be written back over the closing lane's value at the end of the launch)
  if (i == 0) __hip_atomic_store(&b.ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // hand-over counter of the fused scan / final launch
  for (uint32_t g = i; g < gr.g; g += nthreads) {
    st_pub<INL>(&b.admit[g], 0u);
    if (prm.do_ready) {
      st_pub<INL>(&b.admit64[g], 0ull);
      if (b.gcount[g] == 0u) {
        const uint8_t rd = gr.matched[g] >= (uint32_t)(gr.min_member[g] - gr.status_scheduled[g]) ? 1 : 0;
        b.ready[g] = rd;
        if (prm.host_tag) { st_home(&b.h_admit[g], 0u); st_home(&b.h_ready[g], rd); }
      }
    }
  }
}

// WT: every mirror the calling kernel's blocks wrote went out through st_home (fast_final_block); otherwise the fence
template <bool WT = false>
__device__ __forceinline__ void final_tail(const BatchDev& b, const BatchParams& prm, uint32_t nblocks) {
  __shared__ uint32_t s_last;
  if (!prm.host_tag) return;
  // Every block makes its writes (host mirrors included) VISIBLE AT SYSTEM SCOPE before it takes its ticket; the last one sends the
  // completion word behind them with a system-scope release store — the host polls it (bs_batch_read / bs_batch_map) instead of
  // waiting on the stream.  A bare s_waitcnt vmcnt(0) is not enough here (round 3 had that): the mirrors are posted writes that
  // leave through eight XCDs' separate paths, the counter only says they left the CU, and the completion word of the last block
  // could overtake another XCD's mirrors — a host that copied the results out at once (bs_batch_read) saw the previous cycle's
  // values in a few of them, one run in three (tests/test_gpu_speculate.py::test_latency_mode_results_are_complete_when_the_word_arrives).  The release fence costs the latency mode about a microsecond.
  // (BS_HOME_WT=1, the experiment above: mirrors written through at system scope, "out" = s_waitcnt vmcnt(0), the word a relaxed system-scope store.)
  if constexpr (WT && BS_HOME_WT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&b.ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&b.ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (WT && BS_HOME_WT) __hip_atomic_store(b.h_tag, prm.host_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(b.h_tag, prm.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// grouped: the pod names a group of the loaded state (g valid); admit: it passes PreFilter and, if Filter ran, has a feasible node
template <bool WT = false>
__device__ __forceinline__ void tally_tail(const GroupsDev& gr, const BatchDev& b, const BatchParams& prm, bool grouped, uint32_t g, bool admit,
                                           uint32_t nblocks) {
  if (prm.do_tally && prm.do_ready) {
    // what closing a group needs, fetched while the adds are in flight (the lane's own group: the leader of a key is one of its lanes)
    uint32_t want = 0, ma = 0, mm = 0, sc = 0;
    if (grouped) { want = b.gcount[g]; ma = gr.matched[g]; mm = gr.min_member[g]; sc = gr.status_scheduled[g]; }
    // who adds what: one lane per distinct group of the wave (the first kElectRounds groups; lanes left over add for themselves)
    unsigned long long mine = 0;
    bool asked = false;
    unsigned long long todo = __ballot(grouped);
    const unsigned long long adm = __ballot(grouped && admit);
    for (int round = 0; todo && round < kElectRounds; ++round) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)g, leader);
      const unsigned long long same = __ballot(grouped && g == k0) & todo;
      if (lane_id() == leader) {
        mine = ((unsigned long long)__popcll(same) << 32) | (unsigned long long)__popcll(same & adm);
        asked = true;
      }
      todo &= ~same;
    }
    if (todo & (1ull << lane_id())) {
      mine = (1ull << 32) | (admit ? 1ull : 0ull);
      asked = true;
    }
    // ONE returning atomic instruction for all of them (a returning atomic per round would wait for the previous round's result)
    unsigned long long got = 0;
    if (asked) got = __hip_atomic_fetch_add(&b.admit64[g], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (asked) {
      const unsigned long long tot = got + mine;
      if ((uint32_t)(tot >> 32) == want) {           // every pod of the group has been counted: this lane closes it
        const uint32_t ad = (uint32_t)tot;
        const uint8_t rd = (ma + ad) >= (uint32_t)(mm - sc) ? 1 : 0;
        b.admit[g] = ad;
        b.ready[g] = rd;
        if (prm.host_tag) { st_home(&b.h_admit[g], ad); st_home(&b.h_ready[g], rd); }
      }
    }
  } else if (prm.do_tally) {
    wave_aggregated_add(b.admit, g, grouped && admit);
  }
  final_tail<WT>(b, prm, nblocks);
}

// ------------------------------------------------------------------------------------------------
// launch C: final codes in queue order, per pod independent.
//   deny replay   pod i is behind a rejected pod of its group iff some (group, class) pair of the group has a
//                 rejected class slot and its first querying pod precedes i (core.go:142,163 -> :105-110)
//   stale leader  sop.maxFinishedPG after the pod's PreFilter = the batch's findMaxPG result from the first
//                 pod that reaches findMaxPG on, the value carried into the batch before it (core.go:121)
//   Filter        code, slot and feasible-node count of the pod from its class slot
//   Permit        per-group admit counts and the quorum predicate core.go:303 (tally_tail)
// ------------------------------------------------------------------------------------------------
// what launch B left behind, read by a block of the SAME launch: performed at the coherence point (its writers used agent-scope
// atomics), not looked up in this XCD's L2

// block `bx` of `nblocks` final blocks; producers > 0: they run in this very launch (k_fast_scan_filter_final) and count themselves
// into ticket[1] when their results are out — everything that does not depend on them is fetched first
// INL (k_fast_step_a, the whole step in one launch): the block is a POD block of the same launch that has finished its own first half; what other pod
// blocks left for it (first-reach words, the pairs' first querying pods, the armed counters) is read after their ticket (ticket[10], base tk_p1), with
// agent-scope loads.  The scan / Filter blocks leave their results as tagged words, one writer each (scan_rec[chunk][slot], feas_rec[group][slot]): the
// block gathers every class's first row (minimum over the `producers` = table chunks) and every Filter slot's feasible count (sum over the groups of
// node runs; `aux` = the Filter split's target waves) into LDS, again until every word carries the batch's tag — the poll is the fetch, no counter.
template <bool INL = false>
__device__ __forceinline__ void fast_final_block(const PodsDev& pods, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchParams& prm,
                                                 uint32_t query_blocks, uint32_t bx, uint32_t nblocks, uint32_t producers, uint32_t tk_p1 = 0, uint32_t aux = 0,
                                                 uint32_t tk_done = 0, uint32_t hint_blocks = 0) {
  __shared__ uint32_t s_first_reach;
  __shared__ uint32_t s_rowk[INL ? kStepSlotsMax : 1u], s_feask[INL ? 2u * kStepSlotsMax : 1u];
  BS_STAMP(3, 0);
  if constexpr (INL) {
    if (threadIdx.x < 64) (void)spread_wait(&b.ticket[kTkP1], tk_p1, query_blocks, b.h_err);
    __syncthreads();
  }
  const uint32_t i = bx * 256u + threadIdx.x;
  const bool valid = i < pods.p;
  // ---- round trip 1: the pod's own fields, and the block's look at the first reaching pod
  uint8_t code0 = 0, st0 = 0;
  int32_t gi0 = BS_POD_NOT_GROUPED;
  uint32_t qpos0 = 0, pclass0 = 0, pair0 = BS_INF;
  if (valid) { code0 = b.tcode[i]; st0 = b.stage[i]; gi0 = pods.group[i]; qpos0 = b.qpos[i]; pclass0 = b.pclass[i]; pair0 = b.ppair[i]; }
  const uint32_t K = prm.k_host ? prm.k_host : *b.kclass;
  const int32_t leader_now = b.leader_epoch[0];
  // first pod that reaches findMaxPG = the candidate of the first block of launch A that has one (64 blocks per look)
  if (threadIdx.x < 64) {
    uint32_t found = BS_INF;
    for (uint32_t b0 = 0; b0 < query_blocks && found == BS_INF; b0 += 64u) {
      const uint32_t bk = b0 + threadIdx.x;
      uint32_t v = BS_INF;
      if (bk < query_blocks) {
        const unsigned long long w = INL ? ld_agent64(&b.first_reach64[bk]) : b.first_reach64[bk];
        if ((uint32_t)(w >> 32) == prm.seq_inv) v = (uint32_t)w;
      }
      const unsigned long long any = __ballot(v != BS_INF);
      if (any) found = (uint32_t)__shfl((int)v, __ffsll((long long)any) - 1);
    }
    if (threadIdx.x == 0) s_first_reach = found;
  }
  // ---- round trip 2a (still nothing of launch B): the head of the group's pair chain, the pod's OWN pair's first querying pod
  //   and next link (a group with ONE request class — nearly every gang — needs nothing else for the deny replay: the pair's
  //   class slot is the pod's own)
  const bool owned = valid && (st0 & ST_OWNED);
  const bool walk = owned && (st0 & ST_ELIG);
  const bool grouped = valid && gi0 >= 0 && (uint32_t)gi0 < gr.g;
  uint32_t row_q = BS_INF, row_c = BS_INF, feas0 = 0, feas1 = 0;
  unsigned long long head = ~0ull, own_fq = ~0ull, own_next = ~0ull;
  if (walk) {
    head = b.pair_head[gi0];
    if (pair0 != BS_INF) { own_fq = INL ? ld_agent64(&b.pair_firstq[pair0]) : b.pair_firstq[pair0]; own_next = b.pair_next[pair0]; }
  }
  if (producers) {                                   // the scan / Filter blocks of this launch have to be through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (our own loads first: they overlap the producers, not the wait)
    // What crosses this hand-over: first_row[] and fu_feas[] — written by the producers with agent-scope atomics (performed at the
    // coherence point, never parked in an XCD's L2), drained (s_waitcnt vmcnt(0)) before their ticket add, and read here with
    // agent-scope loads: no fence is needed for them, and a release / acquire pair would cost every block an L2 write-back /
    // invalidate.  The producers' PLAIN stores (Filter rows, host mirrors) are read by nobody in this launch.
    // Forward progress: the host only takes the fused launch when the WHOLE grid is resident at once (run_fast), so a producer
    // can never be waiting for a slot a spinning final block holds; the spin is bounded all the same (CU masking, a profiler
    // serialising blocks): on time-out the block raises the context's error word and goes on — the batch is then refused by
    // bs_batch_sync / read / map instead of hanging the GPU.
    if constexpr (INL) {
      const uint32_t tagw = prm.seq_inv, U = prm.run_filter ? 2u * K : 0u;
      uint32_t groups = 0;
      if (prm.run_filter) {
        uint32_t tiles, bpw, nchunk;
        step_filter_split(K, (nd.n + 63u) / 64u, aux, tiles, bpw, nchunk);
        groups = (nchunk + 3u) / 4u;
      }
      const unsigned long long none = ((unsigned long long)tagw << 32) | BS_INF, zero = (unsigned long long)tagw << 32;
      BS_STAMP(3, 4);
      if (query_blocks > kGatherDirectBlocks) {
        // a large queue: the producers' counter, then every pod fetches its own words (below)
        if (threadIdx.x < 64) (void)spread_wait(&b.ticket[kTkDone], tk_done, hint_blocks, b.h_err);
      } else
      // thread = slot: the class's words of every chunk (minimum), the Filter slots' words of every group (sum); eight loads in flight at a time.  (A
      // version that spread the words evenly over the block's threads and reduced them with LDS atomics measured slower at both BASELINE sizes.)
      for (uint32_t spins = 0;; ++spins) {
        uint32_t andt = tagw, ort = tagw, row = BS_INF, fs[2] = {0u, 0u};
        if (threadIdx.x < K) {
          const unsigned long long* rec = b.scan_rec + threadIdx.x;             // [chunk][slot]: a wave's lanes read neighbouring words
          for (uint32_t c0 = 0; c0 < producers; c0 += 8u) {
            unsigned long long w[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) w[u] = c0 + u < producers ? ld_agent64(&rec[(size_t)(c0 + u) * kStepSlotsMax]) : none;
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) { andt &= (uint32_t)(w[u] >> 32); ort |= (uint32_t)(w[u] >> 32); row = min(row, (uint32_t)w[u]); }
          }
        }
#pragma unroll
        for (uint32_t h = 0; h < 2u; ++h) {
          const uint32_t slot = threadIdx.x + 256u * h;
          if (slot < U) {
            const unsigned long long* rec = b.feas_rec + slot;                  // [group][slot]
            for (uint32_t g0 = 0; g0 < groups; g0 += 8u) {
              unsigned long long w[8];
#pragma unroll
              for (uint32_t u = 0; u < 8u; ++u) w[u] = g0 + u < groups ? ld_agent64(&rec[(size_t)(g0 + u) * (2u * kStepSlotsMax)]) : zero;
#pragma unroll
              for (uint32_t u = 0; u < 8u; ++u) { andt &= (uint32_t)(w[u] >> 32); ort |= (uint32_t)(w[u] >> 32); fs[h] += (uint32_t)w[u]; }
            }
          }
        }
        const bool timed_out = spins > kSpinBound / 2u;
        if (__syncthreads_and((andt == tagw && ort == tagw) || timed_out)) {
          if (threadIdx.x < K) s_rowk[threadIdx.x] = row;
#pragma unroll
          for (uint32_t h = 0; h < 2u; ++h)
            if (threadIdx.x + 256u * h < U) s_feask[threadIdx.x + 256u * h] = fs[h];
          if (timed_out && threadIdx.x == 0 && b.h_err) __hip_atomic_store(b.h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          BS_COUNT(3, 5, spins + 1u);                    // (probe build: gather rounds, shown as microseconds)
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    } else if (threadIdx.x == 0) {
      uint32_t spins = 0;
      while (ld_agent(&b.ticket[1]) < producers) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinBound) { if (b.h_err) __hip_atomic_store(b.h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      }
    }
  }
  __syncthreads();
  BS_STAMP(3, 1);
  // ---- round trip 2b: the pod's scan slot's result, both Filter slots' feasible counts (INL: they are in LDS already)
  const bool gathered = INL && query_blocks <= kGatherDirectBlocks;
  auto keyed_row = [&](uint32_t slot) -> uint32_t {                 // INL: from LDS (gathered), or first_row64[], a 64-bit minimum keyed by ~batch_seq
    if (gathered) return s_rowk[INL ? slot : 0u];
    const unsigned long long w = ld_agent64(&b.first_row64[slot]);
    return (uint32_t)(w >> 32) == prm.seq_inv ? (uint32_t)w : BS_INF;
  };
  if (owned && (st0 & ST_QUERY)) row_q = INL ? keyed_row(qpos0) : ld_agent(&b.first_row[qpos0]);
  if (walk) row_c = INL ? keyed_row(pclass0) : ld_agent(&b.first_row[pclass0]);
  if (valid && prm.run_filter && grouped) {
    if (gathered) { feas0 = s_feask[INL ? pclass0 : 0u]; feas1 = s_feask[INL ? pclass0 + K : 0u]; }
    else { feas0 = ld_agent(&b.fu_feas[pclass0]); feas1 = ld_agent(&b.fu_feas[pclass0 + K]); }
  }
  bool admit = false;
  if (valid) {
    uint8_t code = code0;
    const uint8_t st = st0;
    const int32_t gi = gi0;
    const uint32_t first_reach = min(s_first_reach, prm.first_reach_hint);     // (the hint never lies behind the local first reaching pod)
    uint32_t fk = BS_K_NOT_SCANNED;
    if (st & ST_OWNED) {
      bool denied = false;
      if (st & ST_ELIG) {
        uint32_t fr = BS_INF;
        if ((uint32_t)head == pair0 && (uint32_t)own_next == BS_INF) {
          // one pair in the chain, and it is this pod's: rejected class <=> its slot found no row
          if ((uint32_t)(own_fq >> 32) == prm.seq_inv && row_c == BS_INF) fr = (uint32_t)own_fq;
        } else {
          for (unsigned long long link = head; (uint32_t)link != BS_INF;) {
            const uint32_t r = (uint32_t)link, cls = (uint32_t)(link >> 32);
            const unsigned long long pq = INL ? ld_agent64(&b.pair_firstq[r]) : b.pair_firstq[r];   // } one round trip: the pair's first querying pod,
            const uint32_t row = INL ? keyed_row(cls) : ld_agent(&b.first_row[cls]);   // } its class slot's scan result,
            link = b.pair_next[r];                                       // } the next link
            if ((uint32_t)(pq >> 32) != prm.seq_inv) continue;           // no pod of the pair had a query in this batch
            const uint32_t fq = (uint32_t)pq;
            if (fq < fr && row == BS_INF) fr = fq;                       // the pair's class was rejected
          }
        }
        denied = fr < i;
        if (prm.commit && fr == i) b.fast_reject[gi] = fr;             // AddToDenyCache, kept for k_fast_commit
      }
      if (denied) code = BS_PF_ERR_DENIED;
      else if (st & ST_QUERY) {
        if (row_q == BS_INF) { code = BS_PF_REJECT_RESERVE; fk = BS_K_NONE; }                // core.go:161-165
        else fk = INL ? row_q : nd.kmap[row_q];          // (INL: the whole-step form's rows are node list indices already)
      }
    } else {
      code = BS_PF_NOT_OWNED;
    }
    b.pf_code[i] = code;
    b.pf_first_k[i] = fk;
    const bool reached = i >= first_reach;
    const int32_t leader = reached ? leader_now : prm.sop_leader0;
    b.pf_leader[i] = leader;
    if (prm.host_tag) { st_home(&b.h_pf_code[i], code); st_home(&b.h_pf_first_k[i], fk); st_home(&b.h_pf_leader[i], leader); }
    const bool pass = code != BS_PF_NOT_OWNED && BS_PF_IS_PASS(code);
    uint32_t feasible = 1u, slot = 0;
    uint8_t fl = BS_FL_NOT_RUN;
    if (prm.run_filter) {
      if (pass) {
        if (gi == BS_POD_NOT_GROUPED) fl = BS_FL_PASS_NOT_GROUPED;                         // core.go:171-174
        else if (gi < 0 || (uint32_t)gi >= gr.g) fl = BS_FL_ERR_PG_NOT_FOUND;              // :177-180
        else if (leader < 0) fl = BS_FL_PANIC_NIL_MAX;                                     // :525
        else if (leader == gi) fl = BS_FL_PASS_IS_MAX;                                     // :531-535
        else { fl = BS_FL_EVALUATED; slot = pclass0 + (reached ? 0u : K); }                // every group has MinResources here
      }
      feasible = fl == BS_FL_EVALUATED ? (reached ? feas0 : feas1) : (fl < 16u ? nd.n : 0u);
      b.fu_slot[i] = slot;
      b.fl_feasible[i] = feasible;
    } else {
      b.fl_feasible[i] = 0;
    }
    b.fl_code[i] = fl;
    b.fflags[i] = (uint32_t)fl << 8;
    if (prm.host_tag) { st_home(&b.h_fl_code[i], fl); st_home(&b.h_fl_feasible[i], prm.run_filter ? feasible : 0u); st_home(&b.h_fl_slot[i], slot); }
    if (gi >= 0 && (uint32_t)gi < gr.g && pass && feasible > 0) admit = true;
    // BS_BATCH_FILTER_DENY: Filter fails on some node -> the group's first such pod (k_fd_apply takes it from here, bs_fdeny.hpp)
    if (prm.filter_deny && fl == BS_FL_EVALUATED && feasible < nd.n) atomicMin(&b.fd_event[gi], ((unsigned long long)prm.seq_inv << 32) | i);
  }
  if (gathered) {                                    // the slots' counts, for whoever reads fu_feas[] after the launch (bs_fdeny.hpp, bs_batch_read)
    if (prm.run_filter)
      for (uint32_t k = i; k < 2u * K; k += nblocks * 256u) b.fu_feas[k] = s_feask[INL ? k : 0u];
  }
  if (prm.host_tag && prm.run_filter) {             // per-row feasible counts of the slots in use
    const uint32_t U = min(2u * K, b.hstride);
    for (uint32_t k = i; k < U; k += nblocks * 256u) st_home(&b.h_feas[k], gathered ? s_feask[INL ? k : 0u] : ld_agent(&b.fu_feas[k]));
  }
  BS_STAMP(3, 2);
  if (!prm.filter_deny) tally_tail<true>(gr, b, prm, grouped, grouped ? (uint32_t)gi0 : 0u, admit, nblocks);      // (else: k_fd_apply, bs_fdeny.hpp)
  BS_STAMP(3, 7);
}

// The same two levels as separate launches, for batches with MANY tiles of class slots (thousands of distinct requests: the
// throughput regime).  There a scan item is one wave's, nothing is gained by overlapping the final blocks' first fetch, and the
// fused kernel's register footprint (every role's maximum) halves the waves a SIMD can hold.
template <int S>
__global__ __launch_bounds__(256) void k_fast_scan_filter(PodsDev pods, NodesDev nd, BatchDev bt, BatchParams prm, uint32_t m, uint32_t jcap,
                                                          uint32_t scan_blocks, uint32_t filter_waves, uint32_t ustride) {
  __shared__ int64_t s_rows[4][64][4 + S];
  if (blockIdx.x < scan_blocks)
    scan_loop<S, true, 1>(bt, prm, m, jcap, 0u, 0u, 1u, blockIdx.x, scan_blocks, s_rows[wave_id()]);
  else
    filter_loop<2>(pods, nd, bt, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x - scan_blocks, gridDim.x - scan_blocks, prm.stamp, 2u * prm.k_host);
}
// ... and the two roles of that launch as launches of their own (BS_TP_FILTER=1..4, see run_fast): one kernel's register footprint is
// the maximum over its roles — the scan's 137 VGPRs (S = 1) hold the Filter loop, the role that does the work when thousands of
// requests are distinct, at three waves per SIMD.  On its own the Filter loop needs 109 VGPRs as written for the latency regime
// (four requests per step, node blocks double-buffered: 4 waves), 93 with two requests per step (5), 75 without the double buffer (6),
// 72 when the compiler is asked for seven waves (k_fast_filter_w7); none of them touches scratch (tools/kernel_resources.py).
template <int S>
__global__ __launch_bounds__(256) void k_fast_scan(BatchDev bt, BatchParams prm, uint32_t m, uint32_t jcap) {
  __shared__ int64_t s_rows[4][64][4 + S];
  scan_loop<S, true, 1>(bt, prm, m, jcap, 0u, 0u, 1u, blockIdx.x, gridDim.x, s_rows[wave_id()]);
}
template <int PU, bool DB>
__global__ __launch_bounds__(256) void k_fast_filter(PodsDev pods, NodesDev nd, BatchDev bt, BatchParams prm, uint32_t filter_waves, uint32_t ustride) {
  filter_loop<2, PU, DB>(pods, nd, bt, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x, gridDim.x, prm.stamp, 2u * prm.k_host);
}
#if BS_EMIT_FAST
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7))) void k_fast_filter_w7(PodsDev pods, NodesDev nd, BatchDev bt, BatchParams prm, uint32_t filter_waves,
                                                                                                uint32_t ustride) {
  filter_loop<2, 2, false>(pods, nd, bt, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x, gridDim.x, prm.stamp, 2u * prm.k_host);
}
#endif
// BS_TP_FILTER=5: the transposed item (bs_filter_t.hpp): lanes are request slots, nodes come through the scalar cache
#if BS_EMIT_FAST
__global__ __launch_bounds__(256) void k_fast_filter_t(NodesDev nd, BatchDev bt, BatchParams prm, uint32_t filter_waves, uint32_t ustride, uint32_t by_tile) {
  filter_loop_t<false>(nd, bt, filter_waves, ustride, prm.collect_stats, blockIdx.x, gridDim.x, prm.stamp, 2u * prm.k_host, by_tile);   // (no look-ahead: 64 VGPRs, eight waves)
}
#endif
// BS_TP_FILTER=6 / 7: both roles in ONE launch again (the scan's dependent-load chains and the Filter loop's compares overlap), the
// Filter role taken by the transposed item; 7: the Filter blocks carry the LOW block indices (dispatched first)
// Compiled for FOUR waves per SIMD where that costs next to nothing (round 6): the kernel's footprint is the scan role's (112 VGPRs at S = 0, 136 at
// S = 1, more beyond), and the Filter role of the throughput regime is bound by the scalar loads its waves keep in flight (bs_filter_t.hpp) — a
// fourth wave per SIMD is a third more of them.  At S = 1 the limit of 128 VGPRs spills nine dwords to scratch (36 bytes: the ONE listed exception of
// tests/test_kernel_resources.py; cfg4 all-distinct 140 -> 137 / 211 -> 199 / 325 -> 324 us at one / two / four compared lanes, cfg3 34.5 -> 35.0 /
// 38.2 -> 37.4 / 50.5 -> 46.1, gpurun_out r06_g_W4 -> profiles/r06_waves4_ab.txt); from S = 2 on it would spill 116+ bytes: left at three.
template <int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S <= 1 ? 4 : 1))) void k_fast_scan_filter_t(NodesDev nd, BatchDev bt, BatchParams prm, uint32_t m, uint32_t jcap, uint32_t scan_blocks,
                                                            uint32_t filter_waves, uint32_t ustride, uint32_t form) {
  __shared__ int64_t s_rows[4][64][4 + S];
  const uint32_t filter_first = form & 1u, by_tile = form & 2u;      // (bit 0: BS_TP_FILTER=7; bit 1: the Filter items' order, filter_loop_t)
  const uint32_t filter_blocks = gridDim.x - scan_blocks;
  const bool is_scan = filter_first ? blockIdx.x >= filter_blocks : blockIdx.x < scan_blocks;
  BS_STAMP(2, 0);                                   // (probe builds: the first 128 blocks = scan blocks, or Filter blocks with BS_TP_FILTER=7)
  if (is_scan)
    scan_loop<S, true, 1>(bt, prm, m, jcap, 0u, 0u, 1u, filter_first ? blockIdx.x - filter_blocks : blockIdx.x, scan_blocks, s_rows[wave_id()]);
  else
    filter_loop_t<true>(nd, bt, filter_waves, ustride, prm.collect_stats, filter_first ? blockIdx.x : blockIdx.x - scan_blocks, filter_blocks, prm.stamp, 2u * prm.k_host, by_tile);
  BS_STAMP(2, 7);
}
#if BS_EMIT_MAIN
__global__ __launch_bounds__(256) void k_fast_final(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm, uint32_t query_blocks) {
  fast_final_block(pods, gr, nd, b, prm, query_blocks, blockIdx.x, gridDim.x, 0u);
}
#endif

// ------------------------------------------------------------------------------------------------
// launches B and C as ONE launch: [0, scan_blocks) node scan | [.., + filter_blocks) Filter evaluation | the rest: final
// blocks.  A final block fetches what it needs from launch A while the producers run, then waits for their count (bounded
// spin, see fast_final_block).  One launch boundary (~1.5 us) and the final blocks' first round trips (~2 us) come off the
// step's critical path.  Taken only when the WHOLE grid is resident at once (run_fast asks the occupancy API): HIP promises
// no dispatch order, so "producers start first" is never relied on; BS_NO_FUSE_FINAL=1 (or a grid beyond residency) runs
// the two levels as k_fast_scan_filter + k_fast_final.
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_fast_scan_filter_final(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchDev bt, BatchParams prm, uint32_t m,
                                                                uint32_t jcap, uint32_t scan_blocks, uint32_t filter_blocks, uint32_t filter_waves,
                                                                uint32_t ustride, uint32_t query_blocks) {
  __shared__ int64_t s_rows[4][64][4 + S];
  const uint32_t producers = scan_blocks + filter_blocks;
  test_chaos_delay();
#ifdef BS_TEST_LATE_ROLE_B           // experiment builds only (see BS_TEST_LATE_ROLE at k_fast_step_a): 0 scan blocks, 1 Filter blocks, 2 final blocks start ~70 us late
  {
    const int role = blockIdx.x < scan_blocks ? 0 : blockIdx.x < producers ? 1 : 2;
    if (role == BS_TEST_LATE_ROLE_B) for (int spin = 0; spin < 20; ++spin) __builtin_amdgcn_s_sleep(127);
  }
#endif
  if (blockIdx.x < producers) {
    BS_STAMP(2, 0);
    if (blockIdx.x < scan_blocks)
      scan_loop<S, true, 2>(bt, prm, m, jcap, 0u, 0u, 1u, blockIdx.x, scan_blocks, s_rows[wave_id()]);
    else
      filter_loop<2>(pods, nd, bt, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x - scan_blocks, filter_blocks, prm.stamp, 2u * prm.k_host);
    // results out (atomics performed, row stores drained), then count this block in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&b.ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    BS_STAMP(2, 7);
    return;
  }
  fast_final_block(pods, gr, nd, b, prm, query_blocks, blockIdx.x - producers, gridDim.x - producers, producers);
}

// block layout: [0, qb) pods | (class-slot form: pb class-slot blocks) | then c * nshares + q: chunk c, slot share q | the rest: Filter
// WHOLE (BS_STEP_A=3, needs the class-slot form): the whole step in this launch.
//   pod blocks    first half as before (publishing nothing but their first-reach word and the armed counters, write-through), ticket kTkP1; then the
//                 block goes on to its pods' final verdicts (fast_final_block<true>) once every pod block's first half and every table / Filter block
//                 (ticket kTkDone) are through
//   table blocks  derive their share's scan queries themselves (table_scan_block<TS, true>): they wait for the chunk totals only
//   Filter blocks derive their tile's slots themselves (class directory + leaders -> filter_slot_values), run the item on registers, and only their
//                 closing add to fu_feas[] waits for the class-slot block (which still publishes every slot for whoever reads them after the launch,
//                 and zeroes those counters — for a large queue only: a small queue's pod blocks store the counts themselves, see filter_slot_from)
template <int TS, bool WHOLE>
__global__ __launch_bounds__(kTblChunk) void k_fast_step_a(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchDev bt, BatchParams prm, const TableDesc* forced,
                                                           uint32_t nchunks, uint32_t query_blocks, uint32_t nshares, uint32_t filter_blocks, uint32_t filter_waves,
                                                           uint32_t ustride, uint32_t tk_pods0, uint32_t tk_tab0, uint32_t param_blocks, const int64_t* ckeys,
                                                           const uint32_t* cpres, uint32_t kcap, uint32_t tk_p1, uint32_t tk_done, uint32_t forced_cls) {
  BS_STAMP(1, 0);
  test_chaos_delay();
  const uint32_t tb = nchunks * nshares;
#ifdef BS_TEST_LATE_ROLE             // experiment builds only (tools/r06_flaky2.sh, profiles/r06_late_class_slots_race.txt): the blocks of ONE role — 0 pod blocks, 1 class-slot
  {                                  // block, 2 table blocks, 3 Filter blocks — start ~70 us late, as a busy GPU could make them: every hand-over has to hold, only the time may change
    const int role = blockIdx.x < query_blocks ? 0 : blockIdx.x < query_blocks + param_blocks ? 1 : blockIdx.x < query_blocks + param_blocks + tb ? 2 : 3;
    if (role == BS_TEST_LATE_ROLE) for (int spin = 0; spin < 20; ++spin) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const uint32_t producers = param_blocks ? param_blocks : query_blocks;      // blocks the slots' ticket waits for
  // WHOLE: how the scan / Filter results reach the pod blocks.  A few pod blocks (<= kGatherDirectBlocks) poll the result words themselves (tagged, one
  // writer each: no counter, no drain, the poll is the fetch — cfg2: 13.1 -> 12.0 us per step); with forty of them polling the same 40 KB at the memory
  // side everybody slowed down, and gathering after a hint counter measured no better than the compact form (cfg3: +0.5 us): a large queue's producers
  // reduce with atomics (first_row64[], fu_feas[]), drain, count themselves into kTkDone, and every pod fetches its own three words after that
  const bool direct = query_blocks <= kGatherDirectBlocks;
  if (blockIdx.x < query_blocks) {
    if constexpr (WHOLE) {
      fast_query_thread<TS, false, false, true>(pods, gr, b, prm, blockIdx.x * kTblChunk + threadIdx.x, query_blocks * kTblChunk);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first-reach word, armed counters, the pairs' minima: out before the ticket
      __syncthreads();
      if (threadIdx.x == 0) spread_add(&b.ticket[kTkP1], blockIdx.x);
      fast_final_block<true>(pods, gr, nd, b, prm, query_blocks, blockIdx.x, query_blocks, nchunks, tk_p1, filter_waves, tk_done, tb + filter_blocks);
    } else if (param_blocks) {
      fast_query_thread<TS, false, false>(pods, gr, b, prm, blockIdx.x * kTblChunk + threadIdx.x, query_blocks * kTblChunk);
    } else {
      fast_query_thread<TS, true>(pods, gr, b, prm, blockIdx.x * kTblChunk + threadIdx.x, query_blocks * kTblChunk);
      BS_STAMP(1, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // slots, Filter parameters, first-reach word: out before the ticket
      __syncthreads();
      BS_STAMP(1, 5);
      if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&b.ticket[kTkSlots], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (blockIdx.x < query_blocks + param_blocks) {
    class_slots_block<TS>(gr, b, prm, ckeys, cpres, kcap, blockIdx.x - query_blocks, !(WHOLE && direct));
  } else if (blockIdx.x < query_blocks + param_blocks + tb) {
    const uint32_t x = blockIdx.x - query_blocks - param_blocks;
    table_scan_block<TS, WHOLE>(nd, bt, prm, forced, x / nshares, nchunks, x % nshares, nshares, producers, tk_pods0, tk_tab0, gr, ckeys, cpres, kcap, direct, forced_cls);
    if constexpr (WHOLE) {
      if (!direct) {                                           // minima performed, then count this block in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) spread_add(&b.ticket[kTkDone], blockIdx.x);
      }
    }
  } else if constexpr (WHOLE) {
    step_filter_block<TS>(gr, nd, bt, prm, ckeys, cpres, kcap, filter_waves, ustride, blockIdx.x - query_blocks - param_blocks - tb, filter_blocks, direct, tk_pods0, producers);
    if (!direct) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) spread_add(&b.ticket[kTkDone], blockIdx.x);
    }
  } else {
    __shared__ uint32_t s_go;
    if (threadIdx.x == 0) {
      s_go = step_wait(&b.ticket[kTkSlots], tk_pods0, producers, b.h_err) ? 1u : 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the Filter loop reads the slots with plain loads (one lane's acquire + the barrier)
    }
    __syncthreads();
    if (!s_go) return;
    filter_loop<2>(pods, nd, bt, filter_waves, 1u, ustride, prm.collect_stats, blockIdx.x - query_blocks - param_blocks - tb, filter_blocks, prm.stamp, 2u * prm.k_host);
  }
  BS_STAMP(1, 7);
}


// ------------------------------------------------------------------------------------------------
// General chain, tables without the fix-up pass: block (table, chunk) of every table some query of the batch uses builds
// its chunk-local running sums; k_scan<S, true> derives the offsets and adds them while the rows travel to LDS.
// ------------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(kTblChunk) void k_tables_local_nofix(NodesDev nd, BatchDev b, BatchParams prm, uint32_t nchunks, uint32_t cstride,
                                                                  uint32_t gstride) {
  const uint32_t slot = blockIdx.x;
  if (!b.needed[slot]) return;
  BatchDev bt = b;
  bt.tables = b.tables + (size_t)slot * prm.mcap * prm.LP;
  bt.kp = b.kp + (size_t)slot * 16;
  bt.chunk_tot = b.chunk_tot + (size_t)slot * cstride * 16;
  bt.chunk_kp = b.chunk_kp + (size_t)slot * cstride * 16;
  bt.gmax = b.gmax + (size_t)slot * gstride * prm.LP;
  const TableDesc d = table_desc(slot, prm.C, nullptr);
  tables_local_fast<TS>(nd, bt, prm, &d, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// findMaxPG for every capture epoch in ONE block (replaces one block per epoch).
// The candidate set only grows with the epoch (epoch e adds the group captured there), and the fold of
// core.go:701-739 over a candidate set is: F = largest progress, holder = FIRST candidate (group order) at F — unless
// that one is fully scheduled (Status.Scheduled >= MinMember), where the tie rule :729-731 may hand over to a later
// candidate.  (F, first index) is a prefix maximum over the epochs; the rare handed-over epochs are folded exactly
// (leader_block) afterwards.  A uint32 divide by zero (:716-717) at some epoch panics every later epoch too.
// key = (F + 1) << 31 | (0x7FFFFFFF - group): larger F wins, then the smaller group index; 0 = no candidate.
// ------------------------------------------------------------------------------------------------
#if BS_EMIT_MAIN
__global__ __launch_bounds__(kLeaderBlock) void k_leader_scan(GroupsDev gr, BatchDev b) {
  __shared__ unsigned long long s_w[kLeaderBlock / 64];
  __shared__ uint32_t s_p[kLeaderBlock / 64];
  __shared__ unsigned long long s_carry;
  __shared__ uint32_t s_pcarry, s_nexact, s_exact[64];
  const uint32_t E1 = *b.nepochs;                 // epochs 0 .. E1-1
  auto key_of = [&](uint32_t g, bool& panic) -> unsigned long long {
    if (gr.flags[g] & BS_GROUP_SCHEDULED_LATCH) return 0ull;
    const uint32_t f = leader_finished(gr, g, panic);
    return (((unsigned long long)f + 1ull) << 31) | (unsigned long long)(0x7FFFFFFFu - g);
  };
  // epoch 0: the groups that already have their pod
  unsigned long long k0 = 0;
  bool p0 = false;
  for (uint32_t g = threadIdx.x; g < gr.g; g += kLeaderBlock)
    if (b.cap_epoch[g] == 0u) { const unsigned long long k = key_of(g, p0); k0 = k > k0 ? k : k0; }
  {
    const unsigned long long m = block_max_u64(k0, s_w);
    const unsigned long long anyp = __ballot(p0);
    if (lane_id() == 0) s_p[wave_id()] = anyp ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t pp = 0;
      for (int w = 0; w < kLeaderBlock / 64; ++w) pp |= s_p[w];
      s_carry = m;
      s_pcarry = pp;
      s_nexact = 0;
    }
    __syncthreads();
  }
  for (uint32_t base = 0; base < E1; base += kLeaderBlock) {
    const uint32_t e = base + threadIdx.x;
    unsigned long long k = 0;
    bool pn = false;
    if (e >= 1 && e < E1) k = key_of(b.epoch_group[e], pn);
    uint32_t pv = pn ? 1u : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {             // inclusive prefix max / or inside the wave
      const unsigned long long u = __shfl_up(k, o);
      const uint32_t q = (uint32_t)__shfl_up((int)pv, o);
      if (lane_id() >= o) { k = u > k ? u : k; pv |= q; }
    }
    __syncthreads();
    if (lane_id() == 63) { s_w[wave_id()] = k; s_p[wave_id()] = pv; }
    __syncthreads();
    unsigned long long off = s_carry;
    uint32_t poff = s_pcarry;
    for (int w = 0; w < wave_id(); ++w) { off = s_w[w] > off ? s_w[w] : off; poff |= s_p[w]; }
    k = off > k ? off : k;
    pv |= poff;
    if (e < E1) {
      int32_t leader = -1;
      if (!pv && k) {
        const uint32_t g = 0x7FFFFFFFu - (uint32_t)(k & 0x7FFFFFFFull);
        leader = (int32_t)g;
        if (gr.status_scheduled[g] >= gr.min_member[g]) {          // the tie rule may hand over: fold this epoch exactly
          const uint32_t at = atomicAdd(&s_nexact, 1u);
          if (at < 64) s_exact[at] = e;
        }
      }
      b.leader_epoch[e] = leader;
      b.panic_epoch[e] = pv ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == kLeaderBlock - 1) { s_carry = k; s_pcarry = pv; }
    __syncthreads();
  }
  const uint32_t nex = s_nexact;
  if (nex > 64) {                                   // pathological: every epoch exactly
    for (uint32_t e = 0; e < E1; ++e) { __syncthreads(); if (!b.panic_epoch[e]) leader_block(gr, b, e); }
  } else {
    for (uint32_t x = 0; x < nex; ++x) { __syncthreads(); leader_block(gr, b, s_exact[x]); }
  }
}
#endif

// BS_BATCH_COMMIT on the fast path: every group has its pod and MinResources already, so what sequential
// PreFilter calls would leave behind is OccupiedBy (core.go:494-500) and the deny entries (:142,:163).
// gate: BS_BATCH_FILTER_DENY's flag word — a run that is not the fixed point (bs_fdeny.hpp) commits nothing
#if BS_EMIT_MAIN
__global__ void k_fast_commit(PodsDev pods, BatchDev b, uint8_t* gflags, uint64_t* gocc, uint32_t G, const uint32_t* gate) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G || (gate && *gate)) return;
  uint8_t fl = gflags[g];
  const uint32_t fe = ((fl & BS_GROUP_DENIED) || (b.fd_in && b.fd_in[g] < b.first_np_s[g])) ? BS_INF : b.first_np_s[g];
  const uint32_t fr = b.fast_reject[g];
  if (fe != BS_INF && gocc[g] == 0) {
    const uint32_t fo = b.first_owner_s[g];
    if (fo != BS_INF && fo <= fr) gocc[g] = pods.owner[fo];
  }
  if (fr != BS_INF) fl |= BS_GROUP_DENIED;
  gflags[g] = fl;
}
#endif

}  // namespace bs
