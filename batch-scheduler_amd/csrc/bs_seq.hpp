// bs_seq.hpp — the reference's scheduling cycle POD BY POD, resident on the device (bs_seq_run).
//
// What upstream's scheduleOne does with the plugin, for every pending pod in queue order:
//   PreFilter  core.go:88-167    deny / permitted entries, fillOccupiedObj (:477-512), findMaxPG (:701-739), the node scan
//                                compareClusterResourceAndRequire (:595-632) against the CURRENT node requests
//   [Filter    core.go:170-191, :514-564 on every node, when the stage is on]
//   node choice + assume         first fit in list order (the rule host/bs_drain.cpp states; the CPU replay the tests use restates it),
//                                requested += request
//   Permit     core.go:268-309   matched + 1 (:290), quorum (:303), latch (:305)
//   release    batchscheduler.go:254-344 + PostBind core.go:327: the waiting pods of the gang bind, Status.Scheduled += k
// Every pod's decision depends on what the pods before it did to the nodes and the group counters, so the pass is ONE
// persistent workgroup (1024 threads, 16 waves) that walks the queue; the O(nodes) and O(groups) parts of a step are
// data-parallel inside it:
//   * findMaxPG is a block maximum over 64-bit keys (progress + 1) << 32 | (inverted index << 1) | "fully scheduled" kept in
//     LDS (one key changes per Permit / capture / release); the tie rule of :729-731 is walked exactly only when the
//     winner is fully scheduled.
//   * the node scan and the first-fit choice are ONE pass over the nodes: wave w owns a contiguous range of the list,
//     sums singleNodeResource over it (phase 1), the 16 totals are exchanged through LDS, and phase 2 forms the running
//     sums of core.go:621 with DPP wave scans and finds the first row that covers the request (:623) — EXEC-free ballots.
//     int64(float32(allocatable) * percent) (:656-659,667) does not depend on the requests: both percents are derived
//     once at the start of the pass (allocatable never changes during a pass).
//   * the control flow of a pod (a few dozen scalar decisions) runs redundantly in every wave from wave-uniform loads:
//     no broadcast step, and the block only meets at the barriers the reductions need anyway.
// Mutable state (node requests, group counters / flags / MinResources / OccupiedBy) is read with vector loads only
// (relaxed atomics at workgroup scope: never through the scalar cache, which this kernel's own stores do not update) and
// written by one thread; all waves of a workgroup share the CU's L1, so __syncthreads() orders them.
#pragma once

#include "bs_kernels.hpp"

namespace bs {

constexpr int kSeqBlock = 1024;
constexpr int kSeqWaves = kSeqBlock / 64;
constexpr uint32_t kSeqKeysLds = 8192;     // groups whose findMaxPG keys fit the LDS window (64 KB)

struct SeqDev {
  // resident state the pass mutates
  int64_t* nreq;                 // [L][stride] node requests
  uint32_t* rpres;               // [n]
  uint32_t* g_matched; uint32_t* g_sc; uint8_t* g_flags; uint32_t* g_cls; int64_t* g_minres; uint32_t* g_mrpres; uint64_t* g_occ;
  // scratch
  int64_t* sc07; int64_t* sc10;  // [L][stride] int64(float32(allocatable) * 0.7 | 1.0)
  unsigned long long* keys;      // [G] findMaxPG keys when G > kSeqKeysLds
  unsigned long long* wait_rec;  // [P] waiting pod: (next waiting pod of its gang + 1) << 32 | node it was assumed on
  uint32_t* head;                // [G] last waiting pod of the gang + 1, 0 = none
  uint32_t* nwait;               // [G]
  uint32_t* slot_of;             // [G] release record of a gang that is through
  unsigned long long* t_first;   // [G] clock when the gang's first pod entered PreFilter, ~0 = not yet
  // results
  uint8_t* pf_code; int32_t* pod_node; uint32_t* pf_first_k; int32_t* pf_leader;
  uint32_t* released_group; uint32_t* released_pods; unsigned long long* first_tick; unsigned long long* ready_tick;
  uint32_t cap;
  unsigned long long* info;      // [0] gangs released [1] clock ticks of the pass [2] node passes [3] node scans [4] sop leader at the end + 1
};

struct SeqParams {
  uint32_t S, eph_gate, run_filter, C;
  int32_t sop_leader0;           // sop.maxFinishedPG carried into the pass (-1 none)
  uint32_t keys_in_lds;
};

// ---- wave-uniform loads of state this kernel itself writes: vector loads, value moved to SGPRs ---------------------
__device__ __forceinline__ uint32_t seq_ld32(const uint32_t* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ uint32_t seq_ld8(const uint8_t* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ uint64_t seq_ld64(const uint64_t* p) {
  const uint64_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int64_t seq_ldi64(const int64_t* p) { return (int64_t)seq_ld64(reinterpret_cast<const uint64_t*>(p)); }
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  const uint32_t lo = uni32((uint32_t)v), hi = uni32((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// findMaxPG key of one group (core.go:705-717): 0 = not a candidate, ~0 = the uint32 division by zero of :716-717
__device__ __forceinline__ unsigned long long seq_key(uint32_t g, uint32_t flags, uint32_t mm, uint32_t sc, uint32_t matched) {
  if ((flags & BS_GROUP_SCHEDULED_LATCH) || !(flags & BS_GROUP_HAS_POD)) return 0ull;                 // :706-711
  uint32_t fin = 0;
  if ((uint32_t)(mm - sc) != 0u) {                                                                    // :712-714
    if (mm == 0u) return ~0ull;
    fin = (uint32_t)((uint32_t)(matched + sc) * 1000u) / mm;                                          // :716-717
  }
  return (((unsigned long long)fin + 1ull) << 32) | ((unsigned long long)(0x7FFFFFFFu - g) << 1) | (sc >= mm ? 1ull : 0ull);
}

__device__ __forceinline__ unsigned long long wave_max_u64_all(unsigned long long v) {
  // signed DPP maximum on the sign-flipped value; result in every lane
  const long long s = (long long)(v ^ 0x8000000000000000ull);
  const long long m = readlane63_i64(wave_max_i64_lane63(s));
  return (unsigned long long)m ^ 0x8000000000000000ull;
}

struct SeqShared {
  unsigned long long kmax[kSeqWaves];
  uint32_t red[kSeqWaves];
  unsigned long long tot[kSeqWaves][BS_MAX_LANES];
  uint32_t wpres[kSeqWaves];
  uint32_t fk[kSeqWaves];
  uint32_t pick[kSeqWaves];
};

// What one node pass is asked: the PreFilter scan (table = fit class + percent, request) and / or the first-fit choice.
struct SeqQuery {
  bool scan, pick;
  uint32_t tcls; bool pct07;
  Res R;                         // scan request (Resource.Add-normalised)
  uint32_t pcls;                 // pick: the pod's own fit class
  int64_t preq[BS_MAX_LANES];    // pick: raw request lanes of the pod
  uint32_t ppres;
  // Filter (computeResourceSatisfied) of the pod, when the stage is on
  uint32_t fl;                   // BS_FL_*
  uint32_t ff;                   // bit0: case 2 impossible, bit1: the leader's member cannot be "held" (scalar key)
  int64_t FR[4], FM[4];
};

// One pass over the node list.  Returns (wave-uniform, same in every wave) first_k of the scan (BS_K_NONE) and the first
// node that takes the pod (BS_INF).
template <int TS>
__device__ __forceinline__ void seq_node_pass(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const SeqQuery& q,
                                              uint32_t& first_k, uint32_t& at) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  const int lane = lane_id(), w = wave_id();
  const uint32_t N = nd.n;
  const uint32_t cw = ((((N + 63u) >> 6) + kSeqWaves - 1u) / kSeqWaves) << 6;     // nodes per wave, a multiple of 64
  const uint32_t lo = min(N, (uint32_t)w * cw), hi = min(N, lo + cw);
  const int64_t* scp = q.pct07 ? sq.sc07 : sq.sc10;
  const uint32_t* fit_scan = nd.fit + (size_t)q.tcls * nd.fit_words;
  const bool pick_cls_ok = q.pcls < nd.n_classes;
  const uint32_t* fit_pick = nd.fit + (size_t)(pick_cls_ok ? q.pcls : 0u) * nd.fit_words;
  const bool fl_all = q.fl < 16u && q.fl != BS_FL_EVALUATED;       // Filter passes on every node
  const bool fl_none = q.fl >= 16u;                                // ... on none (ERR_PG_NOT_FOUND, the nil-leader panic)

  // ---- phase 1: the wave's total of singleNodeResource over its range; first node of the range that takes the pod
  unsigned long long acc[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) acc[j] = 0;
  uint32_t por = 0, mypick = BS_INF;
  const bool want_pick = q.pick && pick_cls_ok && !fl_none;
  for (uint32_t base = lo; base < hi; base += 64u) {
    const uint32_t n = base + (uint32_t)lane;
    const bool valid = n < hi;
    const uint32_t nn = valid ? n : lo;
    const uint32_t fl = nd.flags[nn];
    const uint32_t rp = __hip_atomic_load(&sq.rpres[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t ap = nd.apres[nn];
    int64_t rq[BS_MAX_LANES];
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < L) rq[j] = __hip_atomic_load(&sq.nreq[(size_t)j * nd.stride + nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (q.scan) {
      const bool fit = valid && !(fl & BS_NODE_SKIP_MASK) && !(fl & BS_NODE_TAINT_ERR) && ((fit_scan[nn >> 5] >> (nn & 31u)) & 1u);
      const uint32_t pres = fit ? (ap & rp) : 0u;                                                   // core.go:662-668
      por |= pres;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const bool live = fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
          if (live) acc[j] += (unsigned long long)wsub(scp[(size_t)j * nd.stride + nn], rq[j]);      // :656-659,667
        }
      }
    }
    if (want_pick && mypick == BS_INF) {
      bool ok = valid && fl == 0u && ((fit_pick[nn >> 5] >> (nn & 31u)) & 1u);
      int64_t al[BS_MAX_LANES];
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) al[j] = nd.alloc[(size_t)j * nd.stride + nn];
      if (!fl_all) {                                           // computeResourceSatisfied on this node (core.go:545-563)
        bool c2 = !(q.ff & 1u), c3h = !(q.ff & 2u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t left = wsub(al[j], rq[j]);                                                   // getLeftResource :460-463
          c2 = c2 && left >= q.FR[j];
          c3h = c3h && left >= q.FM[j];
        }
        ok = ok && (c2 || !c3h);                                                                     // case 2 | case 3
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) ok = ok && !(q.preq[j] > 0 && q.preq[j] > wsub(al[j], rq[j]));
      ok = ok && !(wadd(rq[3], 1) > al[3]);
#pragma unroll
      for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
        if (s < S && ((q.ppres >> s) & 1u) && q.preq[4 + s] > 0) {
          const int64_t r0 = ((rp >> s) & 1u) ? rq[4 + s] : 0;
          ok = ok && ((ap >> s) & 1u) && !(q.preq[4 + s] > wsub(al[4 + s], r0));
        }
      }
      const unsigned long long m = __ballot(ok);
      if (m) mypick = base + (uint32_t)(__ffsll((long long)m) - 1);
    }
  }
  if (q.scan) {
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const unsigned long long t = wave_incl_scan_add_u64(acc[j]);
        if (lane == 63) sh_.tot[w][j] = t;
      }
    }
    uint32_t wp = 0;
#pragma unroll
    for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s)
      if (s < S && __ballot((por >> s) & 1u)) wp |= 1u << s;
    if (lane == 0) sh_.wpres[w] = wp;
  }
  if (lane == 0) sh_.pick[w] = mypick;
  __syncthreads();

  // ---- phase 2: running sums inside the range on top of the ranges before it; first row that covers the request
  uint32_t myfk = BS_INF;
  if (q.scan) {
    unsigned long long carry[BS_MAX_LANES];
    uint32_t pcarry = 0;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) carry[j] = 0;
    for (int ww = 0; ww < w; ++ww) {
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) carry[j] += sh_.tot[ww][j];
      pcarry |= sh_.wpres[ww];
    }
    for (uint32_t base = lo; base < hi && myfk == BS_INF; base += 64u) {
      const uint32_t n = base + (uint32_t)lane;
      const bool valid = n < hi;
      const uint32_t nn = valid ? n : lo;
      const uint32_t fl = nd.flags[nn];
      const uint32_t rp = __hip_atomic_load(&sq.rpres[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t ap = nd.apres[nn];
      const bool row = valid && !(fl & BS_NODE_SKIP_MASK);                                            // core.go:606-617
      const bool fit = row && !(fl & BS_NODE_TAINT_ERR) && ((fit_scan[nn >> 5] >> (nn & 31u)) & 1u);
      const uint32_t pres = fit ? (ap & rp) : 0u;
      bool ok = row;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const bool live = fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
          unsigned long long left = 0;
          if (live) left = (unsigned long long)wsub(scp[(size_t)j * nd.stride + nn],
                                                    __hip_atomic_load(&sq.nreq[(size_t)j * nd.stride + nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          const unsigned long long incl = wave_incl_scan_add_u64(left) + carry[j];
          carry[j] = (unsigned long long)readlane63_i64((long long)incl);
          if (j < 4) {
            ok = ok && (int64_t)incl >= q.R.v[j];                                                      // core.go:673-685
          } else {
            const uint32_t s = j - 4;
            const unsigned long long km = __ballot((pres >> s) & 1u);
            const bool have = ((pcarry >> s) & 1u) || (km & ((2ull << lane) - 1ull));                 // key exists in the running sum
            if ((q.R.present >> s) & 1u) ok = ok && (have ? !(q.R.v[j] > (int64_t)incl) : q.R.v[j] == 0);   // :686-697
            if (km) pcarry |= 1u << s;
          }
        }
      }
      const unsigned long long m = __ballot(ok);
      if (m) myfk = base + (uint32_t)(__ffsll((long long)m) - 1);
    }
  }
  if (lane == 0) sh_.fk[w] = myfk;
  __syncthreads();
  uint32_t fk = BS_INF, pk = BS_INF;
#pragma unroll
  for (int ww = 0; ww < kSeqWaves; ++ww) {
    fk = min(fk, sh_.fk[ww]);
    pk = min(pk, sh_.pick[ww]);
  }
  first_k = uni32(fk);
  at = uni32(pk);
}

// findMaxPG (core.go:701-739) over the keys.  Wave-uniform result: leader (-1 none), panic.
__device__ __forceinline__ void seq_find_max(const GroupsDev& gr, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const unsigned long long* lkeys,
                                             int32_t& leader, bool& panic) {
  const uint32_t G = gr.g;
  auto key_at = [&](uint32_t g) -> unsigned long long {
    return prm.keys_in_lds ? lkeys[g] : __hip_atomic_load(&sq.keys[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  unsigned long long best = 0;
  for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
    const unsigned long long k = key_at(g);
    best = k > best ? k : best;
  }
  best = wave_max_u64_all(best);
  if (lane_id() == 0) sh_.kmax[wave_id()] = best;
  __syncthreads();
  unsigned long long top = 0;
#pragma unroll
  for (int ww = 0; ww < kSeqWaves; ++ww) top = sh_.kmax[ww] > top ? sh_.kmax[ww] : top;
  top = uni64(top);
  panic = top == ~0ull;
  leader = -1;
  if (panic || top == 0ull) return;
  const uint32_t F1 = (uint32_t)(top >> 32);
  uint32_t cur = 0x7FFFFFFFu - ((uint32_t)top >> 1);
  bool full = top & 1ull;
  while (full) {                                             // the tie rule :729-731 may hand over (rare: exact walk)
    uint32_t nxt = BS_INF;
    for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
      if (g <= cur || g >= nxt) continue;
      const unsigned long long k = key_at(g);
      if (k == 0ull || (uint32_t)(k >> 32) != F1) continue;
      if (__hip_atomic_load(&sq.g_sc[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) nxt = g;
    }
    nxt = wave_min_u32(nxt);
    __syncthreads();
    if (lane_id() == 0) sh_.red[wave_id()] = nxt;
    __syncthreads();
    uint32_t r = BS_INF;
#pragma unroll
    for (int ww = 0; ww < kSeqWaves; ++ww) r = min(r, sh_.red[ww]);
    r = uni32(r);
    if (r == BS_INF) break;
    cur = r;
    full = uni64(key_at(cur)) & 1ull;
  }
  leader = (int32_t)cur;
}

template <int TS>
__global__ __launch_bounds__(kSeqBlock, 1) void k_seq_pass(PodsDev pods, GroupsDev gr, NodesDev nd, SeqDev sq, SeqParams prm) {
  extern __shared__ unsigned long long s_keys[];             // [G] when prm.keys_in_lds
  __shared__ SeqShared sh_;
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  const uint32_t P = pods.p, G = gr.g, N = nd.n;
  const uint32_t gate = prm.eph_gate;
  const bool t0 = threadIdx.x == 0;

  // ---- prologue: keys, scaled allocatables, per-gang bookkeeping
  for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
    const unsigned long long k = seq_key(g, gr.flags[g], gr.min_member[g], gr.status_scheduled[g], gr.matched[g]);
    if (prm.keys_in_lds) s_keys[g] = k; else sq.keys[g] = k;
    sq.head[g] = 0;
    sq.nwait[g] = 0;
    sq.slot_of[g] = BS_INF;
    sq.t_first[g] = ~0ull;
  }
  for (uint32_t i = threadIdx.x; i < P; i += kSeqBlock) sq.pod_node[i] = -1;
  for (uint32_t n = threadIdx.x; n < N; n += kSeqBlock) {
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const int64_t a = nd.alloc[(size_t)j * nd.stride + n];
        sq.sc07[(size_t)j * nd.stride + n] = scale_f32(a, 0.7f);
        sq.sc10[(size_t)j * nd.stride + n] = scale_f32(a, 1.0f);
      }
    }
  }
  int32_t sop_leader = prm.sop_leader0;                      // sop.maxFinishedPG / maxPGStatus (core.go:58-59), stale between calls
  uint32_t n_released = 0;
  unsigned long long n_pass = 0, n_scan = 0;
  const unsigned long long clk0 = (unsigned long long)wall_clock64();

  for (uint32_t i = 0; i < P; ++i) {
    __syncthreads();                                         // what the previous pod wrote (global state, keys) is in place
    const int32_t gi = pods.group[i];
    const uint32_t pflags = pods.flags[i];
    const bool grouped = gi >= 0 && (uint32_t)gi < G;
    if (grouped && t0 && sq.t_first[gi] == ~0ull) sq.t_first[gi] = (unsigned long long)wall_clock64() - clk0;
    uint32_t code;
    uint32_t fk = BS_K_NOT_SCANNED;
    SeqQuery q;
    q.scan = false;
    q.pick = false;
    q.tcls = 0;
    q.pct07 = false;
    bool deny = false;
    // READ PHASE: from here to the end of the node pass every wave loads the same mutable state (nothing is written except
    // inside the bracketed capture step below); thread 0 writes in the WRITE PHASE behind the node pass, where no other
    // wave reads mutable state any more.
    uint32_t gflags = 0, gmatched = 0, gsc = 0;              // of the pod's own group; gflags as this PreFilter call leaves them
    if (grouped) {
      gflags = seq_ld8(&sq.g_flags[gi]);
      gmatched = seq_ld32(&sq.g_matched[gi]);
      gsc = seq_ld32(&sq.g_sc[gi]);
    }

    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                                    // core.go:89-92
    else if (pflags & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;                      // :95-98
    else if (!grouped) code = BS_PF_ERR_PG_NOT_FOUND;                                               // :100-103
    else {
      if (gflags & BS_GROUP_DENIED) code = BS_PF_ERR_DENIED;                                        // :105-110
      else {
        // fillOccupiedObj, core.go:477-512
        const uint32_t mm = gr.min_member[gi];
        const uint32_t sc = gsc;
        uint32_t nf = gflags;
        uint32_t gcls;
        Res own_mr;                                          // Spec.MinResources of the pod's group after :489-493
        if (!(gflags & BS_GROUP_HAS_POD)) {                  // :486-488
          nf |= BS_GROUP_HAS_POD;
          gcls = pods.cls[i];
        } else gcls = seq_ld32(&sq.g_cls[gi]);
        if (!(gflags & BS_GROUP_HAS_MINRES)) {               // :489-493
          nf |= BS_GROUP_HAS_MINRES;
          pod_require(pods, i, sh, gate, own_mr);
        } else {
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
            if (j < L) own_mr.v[j] = seq_ldi64(&sq.g_minres[(size_t)j * G + gi]);
          own_mr.present = seq_ld32(&sq.g_mrpres[gi]);
        }
        const uint64_t occ = seq_ld64(&sq.g_occ[gi]), refs = pods.owner[i];
        bool occupied = false;
        const bool take_owner = occ == 0 && refs != 0;                                               // :494-501
        if (occ != 0 && (refs == 0 || refs != occ)) occupied = true;                                 // :503-510
        if (nf != gflags || take_owner) {
          __syncthreads();                                   // every wave has read the state this step rewrites
          if (t0) {
            if (!(gflags & BS_GROUP_HAS_POD)) sq.g_cls[gi] = gcls;
            if (!(gflags & BS_GROUP_HAS_MINRES)) {
#pragma unroll
              for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                if (j < L) sq.g_minres[(size_t)j * G + gi] = own_mr.v[j];
              sq.g_mrpres[gi] = own_mr.present;
            }
            if (take_owner) sq.g_occ[gi] = refs;
            if (nf != gflags) {
              sq.g_flags[gi] = (uint8_t)nf;
              const unsigned long long k = seq_key((uint32_t)gi, nf, mm, sc, gmatched);
              if (prm.keys_in_lds) s_keys[gi] = k; else sq.keys[gi] = k;
            }
          }
          gflags = nf;
          __syncthreads();                                   // the capture is a candidate of this very findMaxPG
        }
        if (occupied) code = BS_PF_ERR_OCCUPIED;                                                     // :113-115
        else {
          int32_t leader;
          bool panic;
          seq_find_max(gr, sq, prm, sh_, s_keys, leader, panic);                                     // :118-123
          if (panic) code = BS_PF_PANIC_DIV0;
          else {
            sop_leader = leader;                                                                     // :121-122
            if (leader < 0) code = BS_PF_PASS_NO_MAX;                                                // :127-130
            else {
              const uint32_t lmatched = leader == gi ? gmatched : seq_ld32(&sq.g_matched[leader]);  // :132-135
              if (lmatched == 0) {                                                                   // :136-147
                // getPreAllocatedResource(pgs, 0), core.go:774-793
                res_zero(q.R, sh);
                const int64_t nfin = (int64_t)mm - (int64_t)sc;
                if (nfin > 0) {
                  Res times;
#pragma unroll
                  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                    if (j < L) times.v[j] = wmul(own_mr.v[j], nfin);
                  times.present = own_mr.present;
                  res_add(q.R, times, sh, gate);
                }
                if (q.R.v[BS_LANE_PODS] == 0) q.R.v[BS_LANE_PODS] = (int64_t)mm + 1;
                q.scan = true;
                q.tcls = gcls;
                q.pct07 = false;
                code = BS_PF_PASS_FIRST_FITS;
              } else if (leader == gi) code = BS_PF_PASS_IS_MAX;                                     // :150-155
              else {                                                                                 // :157-166
                const uint32_t lmm = gr.min_member[leader];
                const uint32_t lfl = seq_ld8(&sq.g_flags[leader]);
                Res lmr;
#pragma unroll
                for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                  if (j < L) lmr.v[j] = seq_ldi64(&sq.g_minres[(size_t)j * G + leader]);
                lmr.present = seq_ld32(&sq.g_mrpres[leader]);
                res_zero(q.R, sh);
                const int64_t nfin = (int64_t)lmm - (int64_t)lmatched;                               // matched != 0: :778-779
                if (nfin > 0 && (lfl & BS_GROUP_HAS_MINRES)) {
                  Res times;
#pragma unroll
                  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                    if (j < L) times.v[j] = wmul(lmr.v[j], nfin);
                  times.present = lmr.present;
                  res_add(q.R, times, sh, gate);
                }
                if (q.R.v[BS_LANE_PODS] == 0) q.R.v[BS_LANE_PODS] = (int64_t)lmm + 1;
                Res cur;
                pod_require(pods, i, sh, gate, cur);                                                 // :158
                res_add(q.R, cur, sh, gate);                                                         // :159
                q.scan = true;
                q.tcls = seq_ld32(&sq.g_cls[leader]);
                q.pct07 = true;
                code = BS_PF_PASS_RESERVE_FITS;
              }
            }
          }
        }
      }
    }

    // ---- the node pass: scan of this PreFilter call and (speculatively, in the same sweep) the node the pod would take
    uint32_t at = BS_INF;
    if (BS_PF_IS_PASS(code)) {
      q.pick = true;
      q.pcls = pods.cls[i];
      q.ppres = pods.pres[i];
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) q.preq[j] = j < L ? pods.req[(size_t)j * P + i] : 0;
      q.fl = BS_FL_PASS_NOT_GROUPED;
      q.ff = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { q.FR[j] = 0; q.FM[j] = 0; }
      if (prm.run_filter) {                                  // Filter's per-pod half, core.go:170-180, :524-544
        if (gi == BS_POD_NOT_GROUPED) q.fl = BS_FL_PASS_NOT_GROUPED;
        else if (!grouped) q.fl = BS_FL_ERR_PG_NOT_FOUND;
        else if (sop_leader < 0) q.fl = BS_FL_PANIC_NIL_MAX;
        else {
          const uint32_t lfl = seq_ld8(&sq.g_flags[sop_leader]);
          const bool have = lfl & BS_GROUP_HAS_MINRES;
          if (sop_leader == gi) q.fl = BS_FL_PASS_IS_MAX;
          else if (!have) q.fl = BS_FL_PASS_NO_MINRES;
          else {
            Res mr, ms, cur;
#pragma unroll
            for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
              if (j < L) mr.v[j] = seq_ldi64(&sq.g_minres[(size_t)j * G + sop_leader]);
            mr.present = seq_ld32(&sq.g_mrpres[sop_leader]);
            res_zero(ms, sh);
            res_add(ms, mr, sh, gate);                                                               // :526-527
            pod_require(pods, i, sh, gate, cur);                                                     // :551
            res_add(cur, ms, sh, gate);                                                              // :552
            q.fl = BS_FL_EVALUATED;
#pragma unroll
            for (int j = 0; j < 4; ++j) { q.FR[j] = cur.v[j]; q.FM[j] = ms.v[j]; }
#pragma unroll
            for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
              if (s < S) {
                if ((cur.present & (1u << s)) && cur.v[4 + s] != 0) q.ff |= 1u;
                if ((ms.present & (1u << s)) && ms.v[4 + s] != 0) q.ff |= 2u;
              }
            }
          }
        }
      }
      uint32_t first_k;
      seq_node_pass<TS>(nd, sq, prm, sh_, q, first_k, at);
      n_pass++;
      if (q.scan) {
        n_scan++;
        fk = first_k == BS_INF ? BS_K_NONE : first_k;
        if (first_k == BS_INF) {                             // compareClusterResourceAndRequire false: AddToDenyCache (:142,:163)
          code = code == BS_PF_PASS_FIRST_FITS ? BS_PF_REJECT_FIRST : BS_PF_REJECT_RESERVE;
          deny = true;
          at = BS_INF;
        }
      }
    }
    if (t0) {
      sq.pf_code[i] = (uint8_t)code;
      if (sq.pf_first_k) sq.pf_first_k[i] = fk;
      if (sq.pf_leader) sq.pf_leader[i] = sop_leader;
      if (deny) sq.g_flags[gi] = (uint8_t)(gflags | BS_GROUP_DENIED);
    }
    if (at == BS_INF) continue;                              // rejected, or no node takes the pod: it holds nothing

    // ---- assume (NodeInfo.AddPod): the thread that owns nothing in particular does it — one thread, a handful of words
    if (t0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) sq.nreq[(size_t)j * nd.stride + at] = wadd(sq.nreq[(size_t)j * nd.stride + at], q.preq[j]);
      sq.nreq[(size_t)3 * nd.stride + at] = wadd(sq.nreq[(size_t)3 * nd.stride + at], 1);
      uint32_t rp = sq.rpres[at];
#pragma unroll
      for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
        if (s < S && ((q.ppres >> s) & 1u)) {
          const int64_t old = ((rp >> s) & 1u) ? sq.nreq[(size_t)(4 + s) * nd.stride + at] : 0;
          sq.nreq[(size_t)(4 + s) * nd.stride + at] = wadd(old, q.preq[4 + s]);
          rp |= 1u << s;
        }
      }
      sq.rpres[at] = rp;
    }
    if (!grouped) {                                          // core.go:269-272: Permit lets it through at once
      if (t0) sq.pod_node[i] = (int32_t)at;
      continue;
    }
    // ---- Permit, core.go:268-309 (WRITE PHASE: thread 0 only; the inputs were loaded before the node pass)
    const uint32_t mm = gr.min_member[gi];
    const uint32_t sc0 = gsc;
    const uint32_t m1 = gmatched + 1u;                                                               // :290
    const bool ready = m1 >= (uint32_t)(mm - sc0);                                                   // :303
    if (t0) {
      sq.g_matched[gi] = m1;
      const uint32_t prev = sq.head[gi];
      sq.wait_rec[i] = ((unsigned long long)prev << 32) | at;
      uint32_t nf = gflags, scn = sc0;
      if (!ready) {
        sq.head[gi] = i + 1u;
        sq.nwait[gi] = sq.nwait[gi] + 1u;
      } else {
        const bool first_time = !(gflags & BS_GROUP_SCHEDULED_LATCH);
        nf |= BS_GROUP_SCHEDULED_LATCH;                                                              // :305
        const uint32_t k = sq.nwait[gi] + 1u;
        sq.pod_node[i] = (int32_t)at;
        for (uint32_t wv = prev; wv != 0u;) {                // the waiting pods of the gang bind (batchscheduler.go:254-344)
          const unsigned long long rec = sq.wait_rec[wv - 1u];
          sq.pod_node[wv - 1u] = (int32_t)(uint32_t)rec;
          wv = (uint32_t)(rec >> 32);
        }
        sq.head[gi] = 0;
        sq.nwait[gi] = 0;
        scn = sc0 + k;                                                                               // PostBind, core.go:327
        sq.g_sc[gi] = scn;
        sq.g_flags[gi] = (uint8_t)nf;
        if (first_time) {
          if (n_released < sq.cap) {
            sq.released_group[n_released] = (uint32_t)gi;
            sq.released_pods[n_released] = k;
            sq.first_tick[n_released] = sq.t_first[gi];
            sq.ready_tick[n_released] = (unsigned long long)wall_clock64() - clk0;
            sq.slot_of[gi] = n_released;
          }
        } else if (sq.slot_of[gi] != BS_INF) {
          sq.released_pods[sq.slot_of[gi]] += k;             // a late member of a gang that is already through
        }
      }
      const unsigned long long key = seq_key((uint32_t)gi, nf, mm, scn, m1);
      if (prm.keys_in_lds) s_keys[gi] = key; else sq.keys[gi] = key;
    }
    if (ready && !(gflags & BS_GROUP_SCHEDULED_LATCH)) n_released++;
  }
  __syncthreads();
  if (t0) {
    sq.info[0] = n_released;
    sq.info[1] = (unsigned long long)wall_clock64() - clk0;
    sq.info[2] = n_pass;
    sq.info[3] = n_scan;
    sq.info[4] = (unsigned long long)(uint32_t)(sop_leader + 1);
  }
}

}  // namespace bs
