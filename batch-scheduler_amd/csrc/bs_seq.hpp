// bs_seq.hpp — the reference's scheduling cycle POD BY POD, resident on the device (bs_seq_run).
//
// What upstream's scheduleOne does with the plugin, for every pending pod in queue order:
//   PreFilter  core.go:88-167    deny / permitted entries, fillOccupiedObj (:477-512), findMaxPG (:701-739), the node scan
//                                compareClusterResourceAndRequire (:595-632) against the CURRENT node requests
//   [Filter    core.go:170-191, :514-564 on every node, when the stage is on]
//   node choice + assume         first fit in list order (the rule host/bs_drain.cpp states; the CPU replay the tests use
//                                restates it), requested += request
//   Permit     core.go:268-309   matched + 1 (:290), quorum (:303), latch (:305)
//   release    batchscheduler.go:254-344 + PostBind core.go:327: the waiting pods of the gang bind, Status.Scheduled += k
// Every pod's decision depends on what the pods before it did to the nodes and the group counters, so the pass is ONE
// persistent workgroup (512 threads, 8 waves) that walks the queue; the O(nodes) and O(groups) parts of a step are
// data-parallel inside it:
//   * findMaxPG is a block maximum over 64-bit keys (progress + 1) << 32 | (inverted index << 1) | "fully scheduled" kept in
//     LDS; one key changes per Permit / capture / release, and the fold is only repeated after such a change.  The tie rule
//     of :729-731 is walked exactly only when the winner is fully scheduled.
//   * singleNodeResource (:634-670) is kept as two resident arrays left07 / left10 = int64(float32(allocatable) * percent) -
//     requested (allocatable never changes during a pass; an assume step patches one node), so the node scan reads ONE
//     int64 per resource lane and node.  The scan goes over the list in ROUNDS of 16 tiles of 64 nodes (one tile per wave):
//     DPP wave scans give the running sums inside a tile (v_add_co_u32_dpp / v_addc_co_u32_dpp: the shift rides on the add),
//     the 16 tile totals are exchanged through LDS (one barrier per round, LDS-only fences so that the next round's loads
//     stay in flight across it), and the pass stops at the round of the first row that covers the request — the reference's
//     early exit (:623-627).
//   * the first-fit choice looks only at tiles whose per-tile bound (max free cpu / memory over schedulable nodes, kept in
//     LDS, tightened whenever a tile was looked at in vain) can hold the pod; the lane that owns the chosen node performs the
//     assume step from the registers it already holds.
//   * the control flow of a pod (a few dozen scalar decisions) runs redundantly in every wave from wave-uniform loads:
//     no broadcast step, and the block only meets at the barriers the reductions need anyway.
// Mutable state read at wave-uniform addresses (group counters / flags / MinResources / OccupiedBy) is read with vector loads
// only (relaxed atomics at workgroup scope: never through the scalar cache, which this kernel's own stores do not update) and
// written by one thread; all waves of a workgroup share the CU's L1, so a barrier orders them.  READ PHASE / WRITE PHASE
// discipline: between the barrier at the top of a pod and the end of its node scan nothing is written (except inside the
// barrier-bracketed capture step); behind it only thread 0 (and the lane that owns the chosen node) write, and nobody else
// reads mutable global state until the next pod's top barrier.
#pragma once

#include "bs_kernels.hpp"

namespace bs {

#ifndef BS_SEQ_BLOCK
#define BS_SEQ_BLOCK 512
#endif
// Threads of the one workgroup (a multiple of 64, at most 1024).  Measured (tools/seq_bench.py --probe, profiles/r04_*): the pass is
// instruction-issue bound on ONE CU — every wave repeats the pod's control flow, and rocprofv3 counts ~11 000 wave-instructions per
// pod at 16 waves — so fewer waves win until the parallel parts (tile checks, table builds) run short of them: 16 waves 98 ms,
// 8 waves 48 ms, 4 waves 47 ms for cfg3/tail, 8 ahead of 4 on cfg4.
constexpr int kSeqBlock = BS_SEQ_BLOCK;
constexpr int kSeqWaves = kSeqBlock / 64;
constexpr uint32_t kSeqKeysLds = 8192;     // groups whose findMaxPG keys fit the LDS window (64 KB)
constexpr uint32_t kSeqPruneTiles = 1024;  // 64-node tiles whose first-fit bounds fit the LDS window (65 536 nodes)
constexpr uint32_t kSeqWaitList = 512;     // waiting pods of the current gang kept in LDS
constexpr uint32_t kSeqHasRecord = 0x80000000u;   // in nwait[g]: the gang has been released in this pass (it has a release record)
constexpr uint32_t kSeqCursorBits = 9;     // first-fit cursors in LDS: 512 direct-mapped entries (12 bytes each)
constexpr uint32_t kSeqCursors = 1u << kSeqCursorBits;
#ifndef BS_SEQ_RESULT_WAVE
#define BS_SEQ_RESULT_WAVE 1
#endif
constexpr uint32_t kSeqResultThread = kSeqWaves > BS_SEQ_RESULT_WAVE ? 64u * BS_SEQ_RESULT_WAVE : 0u;   // the thread that writes a pod's PreFilter results
#ifndef BS_SEQ_POD_WIN
#define BS_SEQ_POD_WIN 64
#endif
constexpr uint32_t kSeqPodWin = BS_SEQ_POD_WIN;        // pods whose (immutable) input fields are staged in LDS ahead of their turn
#ifndef BS_SEQ_CACHE_MAX
#define BS_SEQ_CACHE_MAX 4
#endif
constexpr uint32_t kSeqCacheSlots = BS_SEQ_CACHE_MAX;     // table summaries (fit class x percent) kept in LDS at most

struct SeqDev {
  // resident state the pass mutates
  int64_t* nreq;                 // [L][stride] node requests
  uint32_t* rpres;               // [n]
  uint32_t* g_matched; uint32_t* g_sc; uint8_t* g_flags; uint32_t* g_cls; int64_t* g_minres; uint32_t* g_mrpres; uint64_t* g_occ;
  // scratch
  int64_t* left07; int64_t* left10;  // [L][stride] int64(float32(allocatable) * 0.7 | 1.0) - requested (core.go:656-659,667)
  uint32_t* nmeta;               // [n] bit 0: a row of the scan (not skipped, core.go:606-617), bit 1: taint error (:639-641), bits 4..: scalar keys of singleNodeResource (:662-668)
  unsigned long long* keys;      // [G] findMaxPG keys when G > kSeqKeysLds
  unsigned long long* wait_rec;  // [P] waiting pod: (next waiting pod of its gang + 1) << 32 | node it was assumed on
  uint32_t* head;                // [G] last waiting pod of the gang + 1, 0 = none
  uint32_t* nwait;               // [G]
  uint32_t* slot_of;             // [G] release record of a gang that is through
  unsigned long long* t_first;   // [G] clock when the gang's first pod entered PreFilter, ~0 = not yet
  const uint32_t* pclass;        // [P] request class of the pod (equal request lanes + present bits <=> equal class; derived at the pod load)
  // results
  uint8_t* pf_code; int32_t* pod_node; uint32_t* pf_first_k; int32_t* pf_leader;
  uint8_t* last_permitted;       // [P] prm.filter_deny: 1 = the pass left a lastPermittedPod entry for the pod (core.go:188)
  uint32_t* released_group; uint32_t* released_pods; unsigned long long* first_tick; unsigned long long* ready_tick;
  uint32_t cap;
  unsigned long long* info;      // [0] gangs released [1] clock ticks of the pass [2] first-fit searches [3] node scans [4] sop leader at the end + 1
                                 // [5] scan rounds executed [6] tiles the first-fit searches looked at [7] findMaxPG folds
};

// Probe build only (-DBS_SEQ_PROBE, tools/seq_bench.py --probe; never the shipped library): shader-clock cycles per phase of a
// pod, summed over the pass, in info[8..15]: control (state in registers / group loads), capture, findMaxPG fold, scan,
// first fit + assume, Permit / release, the top barrier.
#ifdef BS_SEQ_PROBE
#define BS_SEQ_T(k) do { const unsigned long long _t = (unsigned long long)__builtin_readcyclecounter(); ph[k] += _t - tl; tl = _t; } while (0)
#else
#define BS_SEQ_T(k) ((void)0)
#endif

// ... and a finer split (thread 0's clock, kept in LDS so that it costs the kernel no registers): info[32 + k]
#ifdef BS_SEQ_PROBE
#define BS_SEQ_P(k) do { if (threadIdx.x == 0) { const unsigned long long _t = (unsigned long long)__builtin_readcyclecounter(); sh_.ph2[k] += _t - sh_.tl2; sh_.tl2 = _t; } } while (0)
#else
#define BS_SEQ_P(k) ((void)0)
#endif

#ifdef BS_SEQ_PROBE
__device__ unsigned long long g_seq_scan_ph[8];
#define BS_SCAN_T(k) do { if (threadIdx.x == 0) { const unsigned long long _t = (unsigned long long)__builtin_readcyclecounter(); g_seq_scan_ph[k] += _t - stl; stl = _t; } } while (0)
#else
#define BS_SCAN_T(k) ((void)0)
#endif

struct SeqParams {
  uint32_t S, eph_gate, run_filter, C;
  int32_t sop_leader0;           // sop.maxFinishedPG carried into the pass (-1 none)
  uint32_t keys_in_lds, prune;
  uint32_t cache_slots;          // table summaries kept in LDS (0: every scan walks the node list in rounds)
  uint32_t cache_off;            // byte offset of the summary area in dynamic LDS (behind the key window)
  uint32_t use_cursor;           // first-fit cursors per request class (SeqShared::cur_kn, see k_seq_pass's node choice)
  uint32_t filter_deny;          // BS_BATCH_FILTER_DENY: Filter's TTL writes (core.go:183-188) happen inside the pass, every node offered
};

// ---- wave-uniform loads of state this kernel itself writes: vector loads, value moved to SGPRs ---------------------
__device__ __forceinline__ uint32_t seq_raw32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t seq_raw8(const uint8_t* p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint64_t seq_raw64(const void* p) { return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  const uint32_t lo = uni32((uint32_t)v), hi = uni32((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

#ifdef BS_SEQ_UNSAFE_BARRIERS   // timing experiment only: every barrier LDS-only
#define BS_SEQ_FULL_BARRIER() lds_barrier()
#else
#define BS_SEQ_FULL_BARRIER() __syncthreads()
#endif
// a barrier that orders LDS only: the global loads a wave has in flight (the next round's tile) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 64-bit add with the DPP shift riding on the add itself (two instructions per step instead of two moves + two adds).
// Lanes without a source, and rows masked out, are not written.  The leading s_nop covers the "VALU write -> DPP read"
// hazard of whatever produced the operands (the assembler does not see into inline asm).
#define BS_DPP_ADD64(lo, hi, CTRL) \
  asm volatile("s_nop 1\n\tv_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL : "+v"(lo), "+v"(hi) : : "vcc")
__device__ __forceinline__ unsigned long long seq_wave_scan64(unsigned long long v) {        // inclusive, all 64 lanes active
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  BS_DPP_ADD64(lo, hi, "row_shr:1 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:2 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:4 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:8 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  return ((unsigned long long)hi << 32) | lo;
}
// The same scan for the L resource lanes of a tile at once, STEP-major: the six DPP steps of one lane are a dependent chain
// (each ~10 cycles of issue + hazard), the L chains are independent — interleaved, one chain's wait is the others' issue
// time (the hazard "VALU write -> DPP read" is covered by the 2 L - 1 instructions between two steps of one lane).
#define BS_DPP_ADD64_NN(lo, hi, CTRL) \
  asm volatile("v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL : "+v"(lo), "+v"(hi) : : "vcc")
#define BS_DPP_STEP_ALL(CTRL)                                  \
  _Pragma("unroll") for (uint32_t j = 0; j < BS_MAX_LANES; ++j) \
    if (j < L) BS_DPP_ADD64_NN(lo[j], hi[j], CTRL)
__device__ __forceinline__ void seq_wave_scan64_lanes(unsigned long long (&x)[BS_MAX_LANES], uint32_t L) {   // L >= 4
  uint32_t lo[BS_MAX_LANES], hi[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) { lo[j] = (uint32_t)x[j]; hi[j] = (uint32_t)(x[j] >> 32); }
  asm volatile("s_nop 1" ::: "memory");
  BS_DPP_STEP_ALL("row_shr:1 row_mask:0xf bank_mask:0xf");
  BS_DPP_STEP_ALL("row_shr:2 row_mask:0xf bank_mask:0xf");
  BS_DPP_STEP_ALL("row_shr:4 row_mask:0xf bank_mask:0xf");
  BS_DPP_STEP_ALL("row_shr:8 row_mask:0xf bank_mask:0xf");
  BS_DPP_STEP_ALL("row_bcast:15 row_mask:0xa bank_mask:0xf");
  BS_DPP_STEP_ALL("row_bcast:31 row_mask:0xc bank_mask:0xf");
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) x[j] = ((unsigned long long)hi[j] << 32) | lo[j];
}
__device__ __forceinline__ unsigned long long seq_row_scan64(unsigned long long v) {         // inclusive inside each row of 16 lanes
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  BS_DPP_ADD64(lo, hi, "row_shr:1 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:2 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:4 row_mask:0xf bank_mask:0xf");
  BS_DPP_ADD64(lo, hi, "row_shr:8 row_mask:0xf bank_mask:0xf");
  return ((unsigned long long)hi << 32) | lo;
}
// minimum over lanes 0..15 (one value per wave of the block), wave-uniform result
__device__ __forceinline__ uint32_t seq_row_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)BS_INF, (int)v, 0x111, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)BS_INF, (int)v, 0x112, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)BS_INF, (int)v, 0x114, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)BS_INF, (int)v, 0x118, 0xF, 0xF, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}

// A node holds a request for cpu AND memory only if min(free cpu << 20, free memory) >= min(cpu << 20, memory): one number per
// node (1000 millicores ~ 1 GiB) whose tile maximum rules out the tiles in which the free cpu and the free memory sit on
// DIFFERENT nodes — the per-lane maxima alone let those through for ever.
__device__ __forceinline__ long long seq_shl20_sat(long long v) {
  return v > (1ll << 42) ? INT64_MAX : (v < -(1ll << 42) ? INT64_MIN : v * (1ll << 20));
}
__device__ __forceinline__ long long seq_joint(long long cpu, long long mem) {
  const long long c = seq_shl20_sat(cpu);
  return c < mem ? c : mem;
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
  uint32_t fdw[2][kSeqWaves];                            // Filter's TTL writes: per wave, bit 0 = a node's Filter failed, bit 1 = a node's Filter passed (parity per use)
  unsigned long long tot[2][BS_MAX_LANES][kSeqWaves];   // per round parity: tile totals, lane-major so that lanes 0..15 read one row
  uint32_t wpres[2][kSeqWaves];
  uint32_t fk[2][kSeqWaves];
  unsigned long long cmask[kSeqWaves];                   // first fit: candidate tiles of a chunk of 1024
  uint32_t pick[2][kSeqWaves];
  long long pmax[3][kSeqPruneTiles];                     // per tile, over schedulable nodes: max free cpu, max free memory, max of min(free cpu << 20, free memory)
  uint32_t tight[kSeqPruneTiles / 32];                   // first fit: the tile's bounds are exact (set by the wave that tightened them, cleared by an assume step in the tile)
  uint32_t hit_w[2];                                     // first fit / scan: lowest TILE with a hit so far (nobody looks behind it); searches alternate
                                                         // between the two words, so that a word is re-armed a whole search (a barrier) before its next use
  uint32_t wl_pod[kSeqWaitList], wl_node[kSeqWaitList];  // waiting pods of the CURRENT gang (released in parallel; the chain in global memory is the fallback)
  uint32_t asm_ap[2][kSeqWaves], asm_rp[2][kSeqWaves], asm_fit[2][kSeqWaves];   // first fit: keys / fit bits of each wave's node (see SeqAssumed)
  // first-fit cursors, direct-mapped by (request class, fit class): (request class + 1) << 32 | node the class's last search ended at
  // (nodes: nothing fits any more), and the fit class that search ran under.  In LDS, not in global memory: thread 0 writes a cursor
  // behind the search, the next pod's top barrier (LDS-only) orders it for every wave — all waves read the SAME cursor, which the
  // barriers inside the search rely on.  An entry that was evicted only costs the next search of that class its head start.
  unsigned long long cur_kn[kSeqCursors];
  uint32_t cur_fc[kSeqCursors];
  // the next kSeqPodWin pods' input fields (nothing the pass writes): one bulk fetch per window instead of half a dozen dependent
  // trips to memory per pod — group, flags, fit class, present bits, request class, owner hash, the group's MinMember, request lanes
  int64_t pw_req[BS_MAX_LANES][kSeqPodWin];
  uint64_t pw_owner[kSeqPodWin];
  int32_t pw_group[kSeqPodWin];
  uint32_t pw_flags[kSeqPodWin], pw_cls[kSeqPodWin], pw_pres[kSeqPodWin], pw_pclass[kSeqPodWin], pw_mm[kSeqPodWin];
#ifdef BS_SEQ_PROBE
  unsigned long long ph2[32], tl2;
#endif
};

// ---------------------------------------------------------------------------------------------------------------------
// compareClusterResourceAndRequire (core.go:595-632) against the resident left arrays.  Returns the node index of the
// first row whose running sum covers R (first_k), BS_INF if none; wave-uniform, identical in every wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int TS>
__device__ __forceinline__ uint32_t seq_scan(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, uint32_t tcls, bool pct07,
                                             const Res& R, unsigned long long& rounds_done) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  const int lane = lane_id(), w = (int)uni32((uint32_t)wave_id());
  const uint32_t N = nd.n;
  if (!N) return BS_INF;
  const uint32_t ntiles = (N + 63u) >> 6, rounds = (ntiles + kSeqWaves - 1u) / kSeqWaves;
  const int64_t* lf = pct07 ? sq.left07 : sq.left10;
  const uint32_t* fitrow = nd.fit + (size_t)tcls * nd.fit_words;
  BS_SEQ_FULL_BARRIER();                                            // the assume steps of earlier pods (left arrays, meta words) have landed
  struct Tile { int64_t v[BS_MAX_LANES]; uint32_t meta, fitw; };
  auto load_tile = [&](uint32_t r, Tile& t) {
    const uint32_t n = ((r * kSeqWaves + (uint32_t)w) << 6) + (uint32_t)lane;
    const uint32_t nn = n < N ? n : N - 1u;
    t.meta = n < N ? sq.nmeta[nn] : 0u;
    t.fitw = fitrow[nn >> 5];
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < L) t.v[j] = lf[(size_t)j * nd.stride + nn];
  };
  Tile cur, nxt;
  load_tile(0, cur);
  nxt = cur;
  unsigned long long carry[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) carry[j] = 0;
  uint32_t pcarry = 0, found = BS_INF;
  uint32_t r = 0;
#ifdef BS_SEQ_PROBE
  unsigned long long stl = (unsigned long long)__builtin_readcyclecounter();
#endif
  for (; r < rounds; ++r) {
    if (r + 1 < rounds) load_tile(r + 1, nxt);
    BS_SCAN_T(0);
    const uint32_t b = r & 1u;
    const uint32_t n = ((r * kSeqWaves + (uint32_t)w) << 6) + (uint32_t)lane;
    const bool row = n < N && (cur.meta & 1u);                                                        // core.go:606-617
    const bool fit = row && !(cur.meta & 2u) && ((cur.fitw >> (n & 31u)) & 1u);                        // :639-645
    const uint32_t pres = fit ? (cur.meta >> 4) : 0u;                                                 // :662-668
    unsigned long long x[BS_MAX_LANES];
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      const bool live = j < L && fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
      x[j] = live ? (unsigned long long)cur.v[j] : 0ull;
    }
    BS_SCAN_T(1);
    seq_wave_scan64_lanes(x, L);                                                                      // running sums inside the tile (:621)
    BS_SCAN_T(2);
    if (lane == 63) {
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) sh_.tot[b][j][w] = x[j];
    }
    unsigned long long km[BS_MAX_SCALARS];
    uint32_t wp = 0;
#pragma unroll
    for (uint32_t s = 0; s < BS_MAX_SCALARS; ++s) {
      km[s] = 0;
      if (s < S) {
        km[s] = __ballot((pres >> s) & 1u);
        if (km[s]) wp |= 1u << s;
      }
    }
    if (lane == 0) sh_.wpres[b][w] = wp;
    BS_SCAN_T(3);
    lds_barrier();
    BS_SCAN_T(4);
    // a hit of the PREVIOUS round is visible now: the reference's loop ended there (early exit, :623-627)
    if (r > 0) {
      const uint32_t prev = seq_row_min_u32(lane < kSeqWaves ? sh_.fk[b ^ 1u][lane] : BS_INF);
      if (prev != BS_INF) { found = prev; break; }
    }
    BS_SCAN_T(5);
    // offsets: row r of the wave (16 lanes) holds the tile totals of resource lane 4 q + r — ONE row scan serves four lanes
    bool ok = row;
    const uint32_t pw = lane < kSeqWaves ? sh_.wpres[b][lane] : 0u;
    uint32_t pbefore = pcarry, pround = 0;
    unsigned long long inc4[BS_MAX_LANES / 4];
#pragma unroll
    for (uint32_t qd = 0; qd < BS_MAX_LANES / 4; ++qd) {
      inc4[qd] = 0;
      if (qd * 4u < L) {
        const uint32_t jr = qd * 4u + ((uint32_t)lane >> 4), wi = (uint32_t)lane & 15u;
        inc4[qd] = seq_row_scan64((jr < L && wi < (uint32_t)kSeqWaves) ? sh_.tot[b][jr][wi] : 0ull);
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const unsigned long long inc = inc4[j >> 2];
        const int rl = (int)((j & 3u) << 4);
        const unsigned long long off = carry[j] + (w ? readlane_u64(inc, rl + w - 1) : 0ull);
        carry[j] += readlane_u64(inc, rl + kSeqWaves - 1);
        const int64_t sum = (int64_t)(x[j] + off);
        if (j < 4) {
          ok = ok && sum >= R.v[j];                                                                    // :673-685
        } else {
          const uint32_t s = j - 4;
          const unsigned long long bal = __ballot(lane < kSeqWaves && ((pw >> s) & 1u));
          if (bal & ((1ull << w) - 1ull)) pbefore |= 1u << s;
          if (bal) pround |= 1u << s;
          const bool have = ((pbefore >> s) & 1u) || (km[s] & ((2ull << lane) - 1ull));              // the key exists in the running sum
          if ((R.present >> s) & 1u) ok = ok && (have ? !(R.v[j] > sum) : R.v[j] == 0);              // :686-697
        }
      }
    }
    pcarry |= pround;
    BS_SCAN_T(6);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) sh_.fk[b][w] = m ? ((r * kSeqWaves + (uint32_t)w) << 6) + (uint32_t)(__ffsll((long long)m) - 1) : BS_INF;
    cur = nxt;
    BS_SCAN_T(7);
  }
  rounds_done += r < rounds ? r + 1 : rounds;
  if (found == BS_INF) {                                      // the last round's hits
    lds_barrier();
    found = seq_row_min_u32(lane < kSeqWaves ? sh_.fk[(rounds - 1u) & 1u][lane] : BS_INF);
  }
  // (the next writer of tot / wpres / fk is the next scan: the first-fit search's barriers or the next pod's top barrier lie between)
  return found;
}

// ---------------------------------------------------------------------------------------------------------------------
// Table summaries.  For a (fit class, percent) table the scan only has to look INSIDE a tile of 64 nodes when the tile can
// hold the first covering row: with the running sum in front of the tile (OFFSET: the exclusive prefix of the tile totals)
// and the largest running sum inside it per resource lane (MAX LOCAL PREFIX over the rows the reference compares at), a tile
// is a candidate iff offset + max >= request on every lane.  Offsets, totals, maxima and the scalar keys in front of / inside
// a tile are kept in LDS for up to four tables, one thread per tile.  They are MAINTAINED, not recomputed:
//   * an assume step moves the chosen node's left values by the pod's request (the same delta in every table the node
//     counts in): every thread subtracts it from the offsets of the tiles behind the node's — one LDS subtract per lane;
//   * maxima are left alone (requests only grow in a pass, so an old maximum is still an upper bound: more candidates, never
//     fewer) and are TIGHTENED to the exact value by the wave that looked into a candidate tile in vain;
//   * the rare events that would break a bound (a negative request, a scalar key the node's requests did not have) drop the
//     table; it is summarised afresh at its next use.
// A scan is then: candidate test (one thread per tile, LDS only) -> every wave looks into ITS OWN candidate tiles in list order
// until it has a hit (DPP scan of the tile's 64 rows on top of the tile's offset) -> one barrier -> the smallest hit wins.
// A rejected request (no candidate) never touches the node list.  Sums wrap (Go semantics): a tile whose local sums leave
// (-2^62, 2^62), or an offset outside it, is never pruned.
// ---------------------------------------------------------------------------------------------------------------------
struct SeqCache {
  uint32_t T, K;                 // tiles (<= kSeqBlock: one thread per tile), slots
  unsigned long long* tt;        // [K][L][T] tile totals
  unsigned long long* off;       // [K][L][T] running sum in front of the tile
  long long* mp;                 // [K][L][T] max local prefix per lane (INT64_MIN: no row in the tile, INT64_MAX: not prunable)
  uint32_t* pr;                  // [K][T] scalar keys the tile's nodes bring into the running sum
  uint32_t* pb;                  // [K][T] scalar keys present in front of the tile
};
template <int TS>
__device__ __forceinline__ SeqCache seq_cache_view(unsigned char* base, uint32_t T, uint32_t K, Shape<TS> sh) {
  SeqCache c;
  const size_t L = sh.L();
  c.T = T; c.K = K;
  c.tt = reinterpret_cast<unsigned long long*>(base);
  c.off = c.tt + (size_t)K * L * T;
  c.mp = reinterpret_cast<long long*>(c.off + (size_t)K * L * T);
  c.pr = reinterpret_cast<uint32_t*>(c.mp + (size_t)K * L * T);
  c.pb = c.pr + (size_t)K * T;
  return c;
}

// One tile (64 nodes, one wave): live values -> running sums inside the tile.  x[j] = inclusive local sums, row / pres per lane.
template <int TS>
__device__ __forceinline__ void seq_tile_local(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, const int64_t* lf, const uint32_t* fitrow,
                                               uint32_t tile, unsigned long long (&x)[BS_MAX_LANES], bool& row, uint32_t& pres) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), N = nd.n;
  const uint32_t n = (tile << 6) + (uint32_t)lane_id();
  const uint32_t nn = n < N ? n : N - 1u;
  const uint32_t meta = n < N ? sq.nmeta[nn] : 0u, fitw = fitrow[nn >> 5];
  int64_t v[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < L) v[j] = lf[(size_t)j * nd.stride + nn];
  row = n < N && (meta & 1u);                                                                        // core.go:606-617
  const bool fit = row && !(meta & 2u) && ((fitw >> (n & 31u)) & 1u);                                 // :639-645
  pres = fit ? (meta >> 4) : 0u;                                                                     // :662-668
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    const bool live = j < L && fit && (j < 4 || (pres & (1u << (j - 4)))) && !(j == BS_LANE_EPH && !prm.eph_gate);
    x[j] = live ? (unsigned long long)v[j] : 0ull;
  }
  seq_wave_scan64_lanes(x, L);
}

// exact maxima of a tile's local sums -> the slot's mp (and, when `totals`, its totals and keys): the calling wave holds x / row / pres
template <int TS>
__device__ __forceinline__ void seq_cache_put(const SeqParams& prm, const SeqCache& ch, uint32_t slot, uint32_t tile, const unsigned long long (&x)[BS_MAX_LANES],
                                              bool row, uint32_t pres, bool totals) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  constexpr long long kSafe = 1ll << 62;
  const bool anyrow = __ballot(row) != 0ull;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
    if (j < L) {
      const long long xi = (long long)x[j];
      const bool risky = __ballot(xi >= kSafe || xi <= -kSafe) != 0ull;
      const long long mx = readlane63_i64(wave_max_i64_lane63(row ? xi : INT64_MIN));
      if (lane_id() == 63) {
        if (totals) ch.tt[((size_t)slot * L + j) * ch.T + tile] = x[j];
        ch.mp[((size_t)slot * L + j) * ch.T + tile] = !anyrow ? INT64_MIN : (risky ? INT64_MAX : mx);
      }
    }
  }
  if (totals) {
    uint32_t wp = 0;
#pragma unroll
    for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2)
      if (s2 < S && __ballot((pres >> s2) & 1u)) wp |= 1u << s2;
    if (lane_id() == 0) ch.pr[(size_t)slot * ch.T + tile] = wp;
  }
}

// Summarise a table from scratch into slot `slot`: every tile's totals / maxima / keys, then the offsets and the keys in
// front of every tile (a two-level prefix over the tile totals: one thread per tile).
template <int TS>
__device__ __forceinline__ void seq_cache_build(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const SeqCache& ch, uint32_t slot,
                                                uint32_t tcls, bool pct07) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  const int lane = lane_id(), w = (int)uni32((uint32_t)wave_id());
  const uint32_t T = ch.T;
  const int64_t* lf = pct07 ? sq.left07 : sq.left10;
  const uint32_t* fitrow = nd.fit + (size_t)tcls * nd.fit_words;
  for (uint32_t tile = (uint32_t)w; tile < T; tile += kSeqWaves) {
    unsigned long long x[BS_MAX_LANES];
    bool row;
    uint32_t pres;
    seq_tile_local<TS>(nd, sq, prm, lf, fitrow, tile, x, row, pres);
    seq_cache_put<TS>(prm, ch, slot, tile, x, row, pres, true);
  }
  lds_barrier();
  // offsets and keys in front of every tile: chunks of one tile per thread, the running sums carried from chunk to chunk
  unsigned long long carry[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) carry[j] = 0;
  uint32_t pcarry = 0;
  for (uint32_t chunk = 0; chunk < T; chunk += kSeqBlock) {
    const uint32_t t = chunk + threadIdx.x;
    unsigned long long v[BS_MAX_LANES], inc[BS_MAX_LANES];
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) { v[j] = (j < L && t < T) ? ch.tt[((size_t)slot * L + j) * T + t] : 0ull; inc[j] = v[j]; }
    const uint32_t mypr = t < T ? ch.pr[(size_t)slot * T + t] : 0u;
    seq_wave_scan64_lanes(inc, L);
    if (lane == 63) {
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
        if (j < L) sh_.tot[0][j][w] = inc[j];
    }
    uint32_t wp = 0;
#pragma unroll
    for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2)
      if (s2 < S && __ballot((mypr >> s2) & 1u)) wp |= 1u << s2;
    if (lane == 0) sh_.wpres[0][w] = wp;
    lds_barrier();
    unsigned long long inc4[BS_MAX_LANES / 4];
#pragma unroll
    for (uint32_t qd = 0; qd < BS_MAX_LANES / 4; ++qd) {
      inc4[qd] = 0;
      if (qd * 4u < L) {
        const uint32_t jr = qd * 4u + ((uint32_t)lane >> 4), wi = (uint32_t)lane & 15u;
        inc4[qd] = seq_row_scan64((jr < L && wi < (uint32_t)kSeqWaves) ? sh_.tot[0][jr][wi] : 0ull);
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const int rl = (int)((j & 3u) << 4);
        const unsigned long long woff = w ? readlane_u64(inc4[j >> 2], rl + w - 1) : 0ull;
        if (t < T) ch.off[((size_t)slot * L + j) * T + t] = carry[j] + inc[j] - v[j] + woff;
        carry[j] += readlane_u64(inc4[j >> 2], rl + kSeqWaves - 1);
      }
    }
    const uint32_t pw = (uint32_t)lane < (uint32_t)kSeqWaves ? sh_.wpres[0][lane] : 0u;
    uint32_t pbt = pcarry, pround = 0;
#pragma unroll
    for (uint32_t s2 = 0; s2 < BS_MAX_SCALARS; ++s2) {
      if (s2 < S) {
        const unsigned long long bal = __ballot((uint32_t)lane < (uint32_t)kSeqWaves && ((pw >> s2) & 1u));
        const unsigned long long km = __ballot((mypr >> s2) & 1u);
        if ((bal & ((1ull << w) - 1ull)) || (km & ((1ull << lane) - 1ull))) pbt |= 1u << s2;
        if (bal) pround |= 1u << s2;
      }
    }
    if (t < T) ch.pb[(size_t)slot * T + t] = pbt;
    pcarry |= pround;
    lds_barrier();
  }
}

// compareClusterResourceAndRequire through the summaries of slot `slot`.  first_k or BS_INF; wave-uniform, same in every wave.
template <int TS>
__device__ __forceinline__ uint32_t seq_scan_cached(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const SeqCache& ch, uint32_t slot,
                                                    uint32_t tcls, bool pct07, const Res& R, uint32_t& hit_par, unsigned long long& rounds_done) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L();
  const int lane = lane_id(), w = (int)uni32((uint32_t)wave_id());
  const uint32_t T = ch.T;                                       // tiles; thread tid tests tiles tid, tid + block, ...
  const uint32_t hp = hit_par;
  hit_par ^= 1u;
  constexpr long long kSafe = 1ll << 62;
  const int64_t* lf = pct07 ? sq.left07 : sq.left10;
  const uint32_t* fitrow = nd.fit + (size_t)tcls * nd.fit_words;
  uint32_t mine = BS_INF;
  for (uint32_t chunk = 0; chunk < T && mine == BS_INF; chunk += kSeqBlock) {
    // ---- candidate test, LDS only
    const uint32_t t = chunk + threadIdx.x;
    bool cand = t < T;
    if (t < T) {
      const uint32_t mypr = ch.pr[(size_t)slot * T + t], mypb = ch.pb[(size_t)slot * T + t];
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const long long off = (long long)ch.off[((size_t)slot * L + j) * T + t];
          const long long m = ch.mp[((size_t)slot * L + j) * T + t];
          const bool open = m == INT64_MAX || off >= kSafe || off <= -kSafe;         // not prunable on this lane
          const bool reach = m != INT64_MIN && (open || off + m >= R.v[j]);
          if (j < 4) cand = cand && reach;
          else if ((R.present >> (j - 4)) & 1u) {
            if ((mypb >> (j - 4)) & 1u) cand = cand && reach;                         // the key is in the running sum in front of the tile
            else if (!((mypr >> (j - 4)) & 1u)) cand = cand && R.v[j] == 0 && m != INT64_MIN;   // ... at no row of the tile
            else cand = cand && m != INT64_MIN;                                       // ... appears inside the tile: look
          } else cand = cand && m != INT64_MIN;
        }
      }
    }
    unsigned long long cm = __ballot(cand);
    BS_SEQ_P(5);
    // ---- this wave's candidate tiles of the chunk, in list order, until one holds a covering row
    while (cm && mine == BS_INF) {
      const uint32_t tile = chunk + ((uint32_t)w << 6) + (uint32_t)(__ffsll((long long)cm) - 1);
      cm &= cm - 1ull;
      if (uni32(__hip_atomic_load(&sh_.hit_w[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < tile) { cm = 0; chunk = T; break; }   // a covering row in front of this tile exists
      rounds_done++;
      unsigned long long x[BS_MAX_LANES];
      bool row;
      uint32_t pres;
      seq_tile_local<TS>(nd, sq, prm, lf, fitrow, tile, x, row, pres);
      const uint32_t pbt = ch.pb[(size_t)slot * T + tile];
      bool ok = row;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          const int64_t sum = (int64_t)(x[j] + ch.off[((size_t)slot * L + j) * T + tile]);
          if (j < 4) ok = ok && sum >= R.v[j];                                                         // core.go:673-685
          else {
            const uint32_t s2 = j - 4;
            const unsigned long long km = __ballot((pres >> s2) & 1u);
            const bool have = ((pbt >> s2) & 1u) || (km & ((2ull << lane) - 1ull));
            if ((R.present >> s2) & 1u) ok = ok && (have ? !(R.v[j] > sum) : R.v[j] == 0);            // :686-697
          }
        }
      }
      const unsigned long long m = __ballot(ok);
      if (m) {
        mine = (tile << 6) + (uint32_t)(__ffsll((long long)m) - 1);
        if (lane == 0) __hip_atomic_fetch_min(&sh_.hit_w[hp], tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else seq_cache_put<TS>(prm, ch, slot, tile, x, row, pres, false);      // looked at in vain: the tile's maxima become exact
    }
  }
  BS_SEQ_P(6);
  if (lane == 0) sh_.fk[0][w] = mine;
  lds_barrier();
  if (threadIdx.x == 0) sh_.hit_w[hp] = BS_INF;               // (re-armed behind the barrier; the NEXT search uses the other word)
  const uint32_t fk_all = seq_row_min_u32(lane < kSeqWaves ? sh_.fk[0][lane] : BS_INF);
  BS_SEQ_P(7);
  return fk_all;
}

// ---------------------------------------------------------------------------------------------------------------------
// first fit in list order + the assume step (upstream's node choice / cache.AssumePod, restated; see bs_drain.cpp).
// Returns the node (BS_INF none), wave-uniform, identical in every wave.  Every wave looks into ITS OWN candidate tiles
// (per-tile bound on free cpu / memory, LDS) in list order until it has a node; the smallest one wins; the lane that owns
// it rewrites the node's request lanes, left values and meta word, and what the other waves need to follow the change in the
// table summaries (the node's keys and fit bits) has been published before the barrier that decides.
// ---------------------------------------------------------------------------------------------------------------------
struct SeqPick {
  uint32_t pcls;                 // the pod's own fit class
  int64_t preq[BS_MAX_LANES];    // raw request lanes of the pod
  uint32_t ppres;
  uint32_t fl;                   // Filter (computeResourceSatisfied) of the pod, BS_FL_* (PASS_NOT_GROUPED when the stage is off)
  uint32_t ff;                   // bit0: case 2 impossible, bit1: the leader's member cannot be "held" (scalar key)
  int64_t FR[4], FM[4];
};
struct SeqAssumed { uint32_t ap, rp, fitbits; };     // of the chosen node: allocatable keys, request keys BEFORE the step, bit c = fits the class of slot c

template <int TS>
__device__ __forceinline__ uint32_t seq_pick(const NodesDev& nd, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const SeqPick& q,
                                             const uint32_t (&slot_key)[kSeqCacheSlots], bool drained, uint32_t& hit_par, SeqAssumed& out,
                                             unsigned long long& tiles_looked, uint32_t start_tile) {
  const Shape<TS> sh(prm.S);
  const uint32_t L = sh.L(), S = sh.S();
  const int lane = lane_id(), w = (int)uni32((uint32_t)wave_id());
  const uint32_t N = nd.n;
  out.ap = out.rp = out.fitbits = 0;
  if (!N || q.pcls >= nd.n_classes || q.fl >= 16u) return BS_INF;          // (ERR_PG_NOT_FOUND / the nil-leader panic: Filter fails everywhere)
  const bool fl_all = q.fl != BS_FL_EVALUATED;                              // Filter passes on every node
  const uint32_t ntiles = (N + 63u) >> 6;
  const uint32_t* fitrow = nd.fit + (size_t)q.pcls * nd.fit_words;
  uint32_t found = BS_INF;
  const uint32_t pb = 0, hp = hit_par;
  hit_par ^= 1u;
  if (!drained) __syncthreads();                                            // the assume steps of earlier pods have landed before a tile is read
  BS_SEQ_P(10);
  uint32_t mine = BS_INF;
  int64_t al[BS_MAX_LANES], rq[BS_MAX_LANES], l07[BS_MAX_LANES], l10[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) { al[j] = 0; rq[j] = 0; l07[j] = 0; l10[j] = 0; }
  uint32_t ap = 0, rp = 0, fbits = 0, meta = 0;
  // start_tile: the first-fit cursor of the pod's request class — an identical request, with the same fit class, last ended its search
  // there; while requests only ADD to the nodes nothing in front of it can have started to fit (k_seq_pass keeps that invariant)
  for (uint32_t chunk = (start_tile / kSeqBlock) * kSeqBlock; chunk < ntiles && mine == BS_INF; chunk += kSeqBlock) {
    const uint32_t t = chunk + threadIdx.x;
    bool cand = t < ntiles && t >= start_tile;
    if (cand && prm.prune) {
      cand = !(q.preq[0] > 0 && sh_.pmax[0][t] < q.preq[0]) && !(q.preq[1] > 0 && sh_.pmax[1][t] < q.preq[1]);
      if (q.preq[0] > 0 && q.preq[1] > 0) cand = cand && !(sh_.pmax[2][t] < seq_joint(q.preq[0], q.preq[1]));
    }
    unsigned long long cm = __ballot(cand);
    while (cm && mine == BS_INF) {
      const uint32_t tile = chunk + ((uint32_t)w << 6) + (uint32_t)(__ffsll((long long)cm) - 1);
      cm &= cm - 1ull;
      if (uni32(__hip_atomic_load(&sh_.hit_w[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < tile) { cm = 0; chunk = ntiles; break; }   // a node in front of this tile takes the pod
      tiles_looked++;
      const uint32_t n = (tile << 6) + (uint32_t)lane;
      const bool valid = n < N;
      const uint32_t nn = valid ? n : N - 1u;
      const uint32_t fl = nd.flags[nn];
      ap = nd.apres[nn];
      rp = sq.rpres[nn];
      const uint32_t fw = fitrow[nn >> 5];
      uint32_t fws[kSeqCacheSlots];
#pragma unroll
      for (uint32_t c = 0; c < kSeqCacheSlots; ++c)
        fws[c] = (c < prm.cache_slots && slot_key[c] != BS_INF) ? nd.fit[(size_t)(slot_key[c] >> 1) * nd.fit_words + (nn >> 5)] : 0u;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          al[j] = nd.alloc[(size_t)j * nd.stride + nn];
          rq[j] = sq.nreq[(size_t)j * nd.stride + nn];
          l07[j] = sq.left07[(size_t)j * nd.stride + nn];       // (for the assume step: no second round trip behind the decision)
          l10[j] = sq.left10[(size_t)j * nd.stride + nn];
        }
      }
      meta = sq.nmeta[nn];
      fbits = 0;
#pragma unroll
      for (uint32_t c = 0; c < kSeqCacheSlots; ++c) fbits |= ((fws[c] >> (nn & 31u)) & 1u) << c;
      const bool sched = valid && fl == 0u;
      bool ok = sched && ((fw >> (nn & 31u)) & 1u);
      const int64_t f0 = wsub(al[0], rq[0]), f1 = wsub(al[1], rq[1]);
      if (!fl_all) {                                         // computeResourceSatisfied on this node (core.go:545-563)
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
      if (m) {
        mine = (tile << 6) + (uint32_t)(__ffsll((long long)m) - 1);
        if (lane == 0) __hip_atomic_fetch_min(&sh_.hit_w[hp], tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (prm.prune && !((sh_.tight[tile >> 5] >> (tile & 31u)) & 1u)) {   // looked at in vain: tighten the tile's bounds to what is really there
        const long long m0 = readlane63_i64(wave_max_i64_lane63(sched ? (long long)f0 : INT64_MIN));
        const long long m1 = readlane63_i64(wave_max_i64_lane63(sched ? (long long)f1 : INT64_MIN));
        const long long m2 = readlane63_i64(wave_max_i64_lane63(sched ? seq_joint(f0, f1) : INT64_MIN));
        if (lane == 0) { sh_.pmax[0][tile] = m0; sh_.pmax[1][tile] = m1; sh_.pmax[2][tile] = m2; atomicOr(&sh_.tight[tile >> 5], 1u << (tile & 31u)); }
      }
    }
  }
  BS_SEQ_P(11);
  {
    const bool owner = mine != BS_INF && (uint32_t)lane == (mine & 63u);
    if (owner) { sh_.pick[pb][w] = mine; sh_.asm_ap[pb][w] = ap; sh_.asm_rp[pb][w] = rp; sh_.asm_fit[pb][w] = fbits; }
    if (mine == BS_INF && lane == 0) sh_.pick[pb][w] = BS_INF;
    lds_barrier();
    const uint32_t mypick = lane < kSeqWaves ? sh_.pick[pb][lane] : BS_INF;
    found = seq_row_min_u32(mypick);
    if (found != BS_INF) {
      const unsigned long long wm = __ballot(mypick == found);               // the wave that holds the chosen node
      const uint32_t ww = (uint32_t)(__ffsll((long long)wm) - 1);
      out.ap = sh_.asm_ap[pb][ww]; out.rp = sh_.asm_rp[pb][ww]; out.fitbits = sh_.asm_fit[pb][ww];
    }
    if (threadIdx.x == 0) sh_.hit_w[hp] = BS_INF;             // (re-armed behind the barrier; the NEXT search uses the other word)
    BS_SEQ_P(12);
    if (found != BS_INF && owner && mine == found) {
      // ---- assume (NodeInfo.AddPod): requested += request, pods lane + 1; the left arrays and the meta word follow
      const uint32_t at = found;
      uint32_t nrp = rp;
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
        if (j < L) {
          int64_t nr = rq[j];
          bool touched = false;
          if (j < 3) { nr = wadd(rq[j], q.preq[j]); touched = true; }
          else if (j == 3) { nr = wadd(rq[j], 1); touched = true; }
          else if ((q.ppres >> (j - 4)) & 1u) {
            nr = wadd(((rp >> (j - 4)) & 1u) ? rq[j] : 0, q.preq[j]);
            nrp |= 1u << (j - 4);
            touched = true;
          }
          if (touched) {
            sq.nreq[(size_t)j * nd.stride + at] = nr;
            if (j < 4 || ((rp >> (j - 4)) & 1u)) {             // left = scaled allocatable - requested: it moves by what requested moves by
              const int64_t dlt = wsub(nr, rq[j]);
              sq.left07[(size_t)j * nd.stride + at] = wsub(l07[j], dlt);
              sq.left10[(size_t)j * nd.stride + at] = wsub(l10[j], dlt);
            } else {                                           // a scalar key the node's requests did not have: the lane starts to exist
              sq.left07[(size_t)j * nd.stride + at] = wsub(scale_f32(al[j], 0.7f), nr);
              sq.left10[(size_t)j * nd.stride + at] = wsub(scale_f32(al[j], 1.0f), nr);
            }
          }
        }
      }
      if (nrp != rp) {
        sq.rpres[at] = nrp;
        sq.nmeta[at] = (meta & 0xFu) | ((ap & nrp) << 4);
      }
      if (prm.prune) {                                       // a negative request frees capacity: the bounds must stay upper bounds
        const int64_t n0 = wsub(al[0], wadd(rq[0], q.preq[0])), n1 = wsub(al[1], wadd(rq[1], q.preq[1]));
        if (n0 > sh_.pmax[0][at >> 6]) sh_.pmax[0][at >> 6] = n0;
        if (n1 > sh_.pmax[1][at >> 6]) sh_.pmax[1][at >> 6] = n1;
        if (seq_joint(n0, n1) > sh_.pmax[2][at >> 6]) sh_.pmax[2][at >> 6] = seq_joint(n0, n1);
        atomicAnd(&sh_.tight[at >> 11], ~(1u << ((at >> 6) & 31u)));     // the node that held a maximum may be the one that just filled up
      }
    }
  }
  BS_SEQ_P(13);
  return found;
}

// findMaxPG (core.go:701-739) over the keys.  Wave-uniform result: leader (-1 none), panic.
__device__ __forceinline__ void seq_find_max(const GroupsDev& gr, const SeqDev& sq, const SeqParams& prm, SeqShared& sh_, const unsigned long long* lkeys,
                                             int32_t& leader, bool& panic, unsigned long long& top_out) {
  const uint32_t G = gr.g;
  top_out = 0;
  auto key_at = [&](uint32_t g) -> unsigned long long {
    return prm.keys_in_lds ? lkeys[g] : __hip_atomic_load(&sq.keys[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (!prm.keys_in_lds) BS_SEQ_FULL_BARRIER();                      // keys in global memory: thread 0's stores of earlier pods have landed
  unsigned long long best = 0;
  for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
    const unsigned long long k = key_at(g);
    best = k > best ? k : best;
  }
  best = wave_max_u64_all(best);
  if (lane_id() == 0) sh_.kmax[wave_id()] = best;
  BS_SEQ_FULL_BARRIER();
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
  if (!full) top_out = top;                                  // (a fully scheduled winner may hand over: its key is no bound for the others)
  while (full) {                                             // the tie rule :729-731 may hand over (rare: exact walk)
    uint32_t nxt = BS_INF;
    for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
      if (g <= cur || g >= nxt) continue;
      const unsigned long long k = key_at(g);
      if (k == 0ull || (uint32_t)(k >> 32) != F1) continue;
      if (__hip_atomic_load(&sq.g_sc[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) nxt = g;
    }
    nxt = wave_min_u32(nxt);
    BS_SEQ_FULL_BARRIER();
    if (lane_id() == 0) sh_.red[wave_id()] = nxt;
    BS_SEQ_FULL_BARRIER();
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

// The input fields of pods i0 .. i0 + kSeqPodWin - 1 into LDS (the caller brackets it with barriers).  MinMember of the pod's group
// rides along (one more dependent trip per WINDOW, not per pod); nothing here is written by the pass.
template <int TS>
__device__ __forceinline__ void seq_stage_pods(const PodsDev& pods, const GroupsDev& gr, const SeqDev& sq, SeqShared& sh_, uint32_t i0, Shape<TS> sh) {
  const uint32_t P = pods.p, L = sh.L();
  for (uint32_t e = threadIdx.x; e < kSeqPodWin * L; e += kSeqBlock) {
    const uint32_t j = e / kSeqPodWin, w = e % kSeqPodWin, p = min(i0 + w, P - 1u);
    sh_.pw_req[j][w] = pods.req[(size_t)j * P + p];
  }
  if (threadIdx.x < kSeqPodWin) {
    const uint32_t w = threadIdx.x, p = min(i0 + w, P - 1u);
    const int32_t g = pods.group[p];
    sh_.pw_group[w] = g;
    sh_.pw_flags[w] = pods.flags[p];
    sh_.pw_cls[w] = pods.cls[p];
    sh_.pw_pres[w] = pods.pres[p];
    sh_.pw_owner[w] = pods.owner[p];
    sh_.pw_pclass[w] = sq.pclass ? sq.pclass[p] : 0u;
    sh_.pw_mm[w] = (g >= 0 && (uint32_t)g < gr.g) ? gr.min_member[g] : 0u;
  }
}
// getPodResourceRequire(pod) from the staged fields (pod_require's rule: the lanes normalised through Add)
template <int TS>
__device__ __forceinline__ void seq_pod_require(const SeqShared& sh_, uint32_t w, Shape<TS> sh, uint32_t gate, Res& out) {
  Res raw;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) raw.v[j] = (int64_t)uni64((uint64_t)sh_.pw_req[j][w]);
  raw.present = uni32(sh_.pw_pres[w]);
  res_zero(out, sh);
  res_add(out, raw, sh, gate);
}

// Everything the pass reads from one group, loaded in ONE round trip.  The state of the pod's own group and of the leader
// then stays in registers (wave-uniform, every wave applies the same updates): consecutive pods of a gang, and pods that
// reserve for the same leader, read nothing from global memory.
struct SeqGroup {
  uint32_t flags, matched, sc, cls, head, nwait;
  uint32_t mm;                   // Spec.MinMember (never written by the pass; fetched in the same trip)
  uint32_t seen;                 // a pod of the gang has entered PreFilter in this pass (t_first is set); a word, not a bool: no padding to copy
  uint64_t occ;
  Res mr;
};
template <int TS>
__device__ __forceinline__ void seq_group_load(const SeqDev& sq, const GroupsDev& gr, uint32_t G, uint32_t g, Shape<TS> sh, SeqGroup& o) {
  const uint32_t gmm = gr.min_member[g];
  const uint32_t f = seq_raw8(&sq.g_flags[g]), m = seq_raw32(&sq.g_matched[g]), s = seq_raw32(&sq.g_sc[g]), c = seq_raw32(&sq.g_cls[g]),
                 p = seq_raw32(&sq.g_mrpres[g]), h = seq_raw32(&sq.head[g]), nw = seq_raw32(&sq.nwait[g]);
  const uint64_t oc = seq_raw64(&sq.g_occ[g]), tf = seq_raw64(&sq.t_first[g]);
  uint64_t mr[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) mr[j] = seq_raw64(&sq.g_minres[(size_t)j * G + g]);
  o.flags = uni32(f); o.matched = uni32(m); o.sc = uni32(s); o.cls = uni32(c); o.head = uni32(h); o.nwait = uni32(nw);
  o.mm = uni32(gmm);
  o.occ = uni64(oc);
  o.seen = uni64(tf) != ~0ull;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < sh.L()) o.mr.v[j] = (int64_t)uni64(mr[j]);
  o.mr.present = uni32(p);
}
template <int TS>
__device__ __forceinline__ void seq_group_zero(SeqGroup& o, Shape<TS> sh) {
  o.flags = o.matched = o.sc = o.cls = o.head = o.nwait = o.mm = 0;
  o.seen = false;
  o.occ = 0;
  res_zero(o.mr, sh);
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

  // ---- prologue: keys, left arrays, meta words, first-fit bounds, per-gang bookkeeping
  for (uint32_t g = threadIdx.x; g < G; g += kSeqBlock) {
    const unsigned long long k = seq_key(g, gr.flags[g], gr.min_member[g], gr.status_scheduled[g], gr.matched[g]);
    if (prm.keys_in_lds) s_keys[g] = k; else sq.keys[g] = k;
    sq.head[g] = 0;
    sq.nwait[g] = 0;
    sq.slot_of[g] = BS_INF;
    sq.t_first[g] = ~0ull;
  }
  for (uint32_t i = threadIdx.x; i < P; i += kSeqBlock) { sq.pod_node[i] = -1; if (prm.filter_deny) sq.last_permitted[i] = 0; }
  if (t0) { sh_.hit_w[0] = BS_INF; sh_.hit_w[1] = BS_INF; }
  if (threadIdx.x < kSeqPruneTiles / 32) sh_.tight[threadIdx.x] = 0;
  for (uint32_t base = 0; base < N; base += kSeqBlock) {
    const uint32_t n = base + threadIdx.x;
    const bool valid = n < N;
    const uint32_t nn = valid ? n : N - 1u;
    const uint32_t fl = nd.flags[nn], ap = nd.apres[nn], rp = sq.rpres[nn];
    long long f0 = INT64_MIN, f1 = INT64_MIN;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
      if (j < L) {
        const int64_t a = nd.alloc[(size_t)j * nd.stride + nn], q = sq.nreq[(size_t)j * nd.stride + nn];
        if (valid) {
          sq.left07[(size_t)j * nd.stride + n] = wsub(scale_f32(a, 0.7f), q);
          sq.left10[(size_t)j * nd.stride + n] = wsub(scale_f32(a, 1.0f), q);
        }
        if (valid && fl == 0u) {
          if (j == 0) f0 = wsub(a, q);
          if (j == 1) f1 = wsub(a, q);
        }
      }
    }
    if (valid) sq.nmeta[n] = ((fl & BS_NODE_SKIP_MASK) ? 0u : 1u) | ((fl & BS_NODE_TAINT_ERR) ? 2u : 0u) | ((ap & rp) << 4);
    if (prm.prune) {
      const long long m0 = readlane63_i64(wave_max_i64_lane63(f0)), m1 = readlane63_i64(wave_max_i64_lane63(f1));
      const long long m2 = readlane63_i64(wave_max_i64_lane63(f0 == INT64_MIN ? INT64_MIN : seq_joint(f0, f1)));
      if (lane_id() == 0 && n < N) { sh_.pmax[0][n >> 6] = m0; sh_.pmax[1][n >> 6] = m1; sh_.pmax[2][n >> 6] = m2; }
    }
  }
  // table summaries (see seq_scan_cached): which table sits in which slot, how many of its tiles wait for a refresh
  const uint32_t ntiles = (N + 63u) >> 6;
  const SeqCache ch = seq_cache_view(reinterpret_cast<unsigned char*>(s_keys) + prm.cache_off, ntiles, prm.cache_slots, sh);
  uint32_t slot_key[kSeqCacheSlots], slot_age[kSeqCacheSlots], age_ctr = 0;
#pragma unroll
  for (uint32_t c = 0; c < kSeqCacheSlots; ++c) { slot_key[c] = BS_INF; slot_age[c] = 0; }
  for (uint32_t e = threadIdx.x; e < kSeqCursors; e += kSeqBlock) { sh_.cur_kn[e] = 0ull; sh_.cur_fc[e] = 0u; }   // (ordered by the barrier in front of the loop)
  uint32_t hit_par = 0;                                      // which of the two early-stop words the next search uses
  bool mono = true;                                          // no assumed request has freed capacity so far: the first-fit cursors hold
  bool stores_pending = true;                                // an assume step (or the prologue) stored node state nobody has waited for yet
  int32_t sop_leader = prm.sop_leader0;                      // sop.maxFinishedPG / maxPGStatus (core.go:58-59), stale between calls
  uint32_t n_released = 0;
  unsigned long long n_pick = 0, n_scan = 0, n_rounds = 0, n_tiles = 0, n_folds = 0, n_builds = 0;
  // findMaxPG's answer is kept until a key changes (capture, Permit, release)
  // ... and a change of ONE key only matters when it beats the winner's (fold_top = the winner's key, 0 = repeat the fold on any change)
  bool fold_valid = false, fold_panic = false;
  int32_t fold_leader = -1;
  unsigned long long fold_top = 0;
  // register-resident group state: the pod's own group (own_of) and the leader's (ldr_of); -1 = nothing cached
  SeqGroup own, ldr;
  seq_group_zero(own, sh);
  seq_group_zero(ldr, sh);
  int32_t own_of = -1, ldr_of = -1;
  uint32_t wl_cnt = 0;                                       // waiting pods of group own_of recorded in the LDS list
  bool wl_ok = true;                                         // ... and the list holds ALL of them (nothing waited before the group became current)
  const unsigned long long clk0 = (unsigned long long)wall_clock64();
  BS_SEQ_FULL_BARRIER();
#ifdef BS_SEQ_PROBE
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = (unsigned long long)__builtin_readcyclecounter();
  if (t0) { for (int k = 0; k < 32; ++k) sh_.ph2[k] = 0; sh_.tl2 = tl; }
#endif

  for (uint32_t i = 0; i < P; ++i) {
    const uint32_t pw = i % kSeqPodWin;
    if (pw == 0u) {                                          // the next window of pod fields (every wave is through with the last one)
      if (i) lds_barrier();
      seq_stage_pods(pods, gr, sq, sh_, i, sh);
    }
    lds_barrier();                                           // keys / bounds / wait list as the previous pod left them
    BS_SEQ_T(6);
    BS_SEQ_P(0);
    const int32_t gi = (int32_t)uni32((uint32_t)sh_.pw_group[pw]);
    const uint32_t pflags = uni32(sh_.pw_flags[pw]);
    const uint32_t pod_mm = uni32(sh_.pw_mm[pw]);            // MinMember of the pod's group (0 if it has none)
    const bool grouped = gi >= 0 && (uint32_t)gi < G;
    uint32_t code;
    uint32_t fk = BS_K_NOT_SCANNED;
    bool scan = false, pct07 = false, deny = false;
    uint32_t tcls = 0;
    Res R;
    res_zero(R, sh);
    // READ PHASE.  A group that is not in registers costs one round trip (behind a full barrier: thread 0's stores of
    // earlier pods have landed); the leader the previous call left behind is fetched in the same trip.
    {
      const bool miss_own = grouped && gi != own_of;
      const bool miss_ldr = sop_leader >= 0 && sop_leader != gi && sop_leader != ldr_of;
      if (miss_own || miss_ldr) {
        BS_SEQ_FULL_BARRIER();
        if (miss_own) {
          if (gi == ldr_of) { const uint32_t h = seq_raw32(&sq.head[gi]), nw = seq_raw32(&sq.nwait[gi]); const uint64_t tf = seq_raw64(&sq.t_first[gi]);
                              own = ldr; own.head = uni32(h); own.nwait = uni32(nw); own.seen = uni64(tf) != ~0ull; }
          else seq_group_load(sq, gr, G, (uint32_t)gi, sh, own);
          own_of = gi;
          wl_cnt = 0;
          wl_ok = (own.nwait & ~kSeqHasRecord) == 0;
        }
        if (miss_ldr) { seq_group_load(sq, gr, G, (uint32_t)sop_leader, sh, ldr); ldr_of = sop_leader; }
      }
    }
    if (grouped && !own.seen) {
      if (t0) sq.t_first[gi] = (unsigned long long)wall_clock64() - clk0;
      own.seen = true;
    }
    BS_SEQ_T(0);
    BS_SEQ_P(1);
    uint32_t gflags = grouped ? own.flags : 0u;              // of the pod's own group, as this PreFilter call leaves them
    const uint32_t gmatched = grouped ? own.matched : 0u, gsc = grouped ? own.sc : 0u;

    if (gi == BS_POD_NOT_GROUPED) code = BS_PF_PASS_NOT_GROUPED;                                    // core.go:89-92
    else if (pflags & BS_POD_LAST_PERMITTED) code = BS_PF_PASS_LAST_PERMITTED;                      // :95-98
    else if (!grouped) code = BS_PF_ERR_PG_NOT_FOUND;                                               // :100-103
    else if (gflags & BS_GROUP_DENIED) code = BS_PF_ERR_DENIED;                                     // :105-110
    else {
      // fillOccupiedObj, core.go:477-512
      const uint32_t mm = pod_mm;
      uint32_t nf = gflags;
      if (!(gflags & BS_GROUP_HAS_POD)) { nf |= BS_GROUP_HAS_POD; own.cls = uni32(sh_.pw_cls[pw]); }           // :486-488
      if (!(gflags & BS_GROUP_HAS_MINRES)) { nf |= BS_GROUP_HAS_MINRES; seq_pod_require(sh_, pw, sh, gate, own.mr); }   // :489-493
      const uint64_t occ = own.occ, refs = uni64(sh_.pw_owner[pw]);
      const bool take_owner = occ == 0 && refs != 0;                                                 // :494-501
      const bool occupied = occ != 0 && (refs == 0 || refs != occ);                                  // :503-510
      if (nf != gflags || take_owner) {
        if (t0) {                                            // (nobody reads these words from memory while the group is current)
          if (!(gflags & BS_GROUP_HAS_POD)) sq.g_cls[gi] = own.cls;
          if (!(gflags & BS_GROUP_HAS_MINRES)) {
#pragma unroll
            for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
              if (j < L) sq.g_minres[(size_t)j * G + gi] = own.mr.v[j];
            sq.g_mrpres[gi] = own.mr.present;
          }
          if (take_owner) sq.g_occ[gi] = refs;
          if (nf != gflags) sq.g_flags[gi] = (uint8_t)nf;
        }
        if (take_owner) own.occ = refs;
        if (nf != gflags) {
          const unsigned long long k = seq_key((uint32_t)gi, nf, mm, gsc, gmatched);
          if (t0) { if (prm.keys_in_lds) s_keys[gi] = k; else sq.keys[gi] = k; }
          if (fold_top == 0ull || k > fold_top) fold_valid = false;   // the capture is a candidate of this very findMaxPG: it only matters if it wins
          gflags = nf;
          own.flags = nf;
          if (prm.keys_in_lds) lds_barrier(); else BS_SEQ_FULL_BARRIER();
        }
      }
      BS_SEQ_T(1);
      if (occupied) code = BS_PF_ERR_OCCUPIED;                                                       // :113-115
      else {
        if (!fold_valid) {
          seq_find_max(gr, sq, prm, sh_, s_keys, fold_leader, fold_panic, fold_top);                 // :118-123
          fold_valid = true;
          n_folds++;
        }
        BS_SEQ_T(2);
        const int32_t leader = fold_leader;
        if (fold_panic) code = BS_PF_PANIC_DIV0;
        else {
          sop_leader = leader;                                                                       // :121-122
          if (leader < 0) code = BS_PF_PASS_NO_MAX;                                                  // :127-130
          else {
            if (leader != gi && leader != ldr_of) {          // (the leader changed: one more round trip)
              BS_SEQ_FULL_BARRIER();
              seq_group_load(sq, gr, G, (uint32_t)leader, sh, ldr);
              ldr_of = leader;
            }
            const uint32_t lmatched = leader == gi ? gmatched : ldr.matched;                         // :132-135
            if (lmatched == 0) {                                                                     // :136-147
              // getPreAllocatedResource(pgs, 0), core.go:774-793
              const int64_t nfin = (int64_t)mm - (int64_t)gsc;
              if (nfin > 0) {
                Res times;
#pragma unroll
                for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                  if (j < L) times.v[j] = wmul(own.mr.v[j], nfin);
                times.present = own.mr.present;
                res_add(R, times, sh, gate);
              }
              if (R.v[BS_LANE_PODS] == 0) R.v[BS_LANE_PODS] = (int64_t)mm + 1;
              scan = true;
              tcls = own.cls;
              pct07 = false;
              code = BS_PF_PASS_FIRST_FITS;
            } else if (leader == gi) code = BS_PF_PASS_IS_MAX;                                       // :150-155
            else {                                                                                   // :157-166
              const uint32_t lmm = ldr.mm;
              const int64_t nfin = (int64_t)lmm - (int64_t)lmatched;                                 // matched != 0: :778-779
              if (nfin > 0 && (ldr.flags & BS_GROUP_HAS_MINRES)) {
                Res times;
#pragma unroll
                for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                  if (j < L) times.v[j] = wmul(ldr.mr.v[j], nfin);
                times.present = ldr.mr.present;
                res_add(R, times, sh, gate);
              }
              if (R.v[BS_LANE_PODS] == 0) R.v[BS_LANE_PODS] = (int64_t)lmm + 1;
              Res cur;
              seq_pod_require(sh_, pw, sh, gate, cur);                                               // :158
              res_add(R, cur, sh, gate);                                                             // :159
              scan = true;
              tcls = ldr.cls;
              pct07 = true;
              code = BS_PF_PASS_RESERVE_FITS;
            }
          }
        }
      }
    }

    // ---- Filter's per-pod half (core.go:170-180, :524-544) with sop.maxPGStatus as this PreFilter call left it
    SeqPick q;
    q.fl = BS_FL_PASS_NOT_GROUPED;
    q.ff = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { q.FR[j] = 0; q.FM[j] = 0; }
    if (BS_PF_IS_PASS(code) && prm.run_filter) {
      if (gi == BS_POD_NOT_GROUPED) q.fl = BS_FL_PASS_NOT_GROUPED;
      else if (!grouped) q.fl = BS_FL_ERR_PG_NOT_FOUND;
      else if (sop_leader < 0) q.fl = BS_FL_PANIC_NIL_MAX;
      else if (sop_leader == gi) q.fl = BS_FL_PASS_IS_MAX;
      else {
        // (sop_leader is in ldr: fetched at the top, or by the fold branch above)
        if (!(ldr.flags & BS_GROUP_HAS_MINRES)) q.fl = BS_FL_PASS_NO_MINRES;
        else {
          Res ms, cur;
          res_zero(ms, sh);
          res_add(ms, ldr.mr, sh, gate);                                                             // :526-527
          seq_pod_require(sh_, pw, sh, gate, cur);                                                   // :551
          res_add(cur, ms, sh, gate);                                                                // :552
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

    // ---- the node scan of this PreFilter call
    BS_SEQ_T(0);
    BS_SEQ_P(2);
    bool drained = false;                                    // this pod has already waited for the stores of earlier assume steps
    if (scan) {
      uint32_t first_k;
      if (stores_pending) { __syncthreads(); stores_pending = false; }
      BS_SEQ_P(3);
      drained = true;
      if (prm.cache_slots) {
        const uint32_t key = (tcls << 1) | (pct07 ? 1u : 0u);
        uint32_t sel = BS_INF;
#pragma unroll
        for (uint32_t c = 0; c < kSeqCacheSlots; ++c)
          if (c < prm.cache_slots && slot_key[c] == key) sel = c;
        if (sel == BS_INF) {                                 // not cached: the least recently used slot takes the table (summarised from scratch)
          uint32_t best_age = BS_INF;
#pragma unroll
          for (uint32_t c = 0; c < kSeqCacheSlots; ++c) {
            if (c < prm.cache_slots) {
              const uint32_t a = slot_key[c] == BS_INF ? 0u : slot_age[c] + 1u;
              if (a < best_age) { best_age = a; sel = c; }
            }
          }
          seq_cache_build<TS>(nd, sq, prm, sh_, ch, sel, tcls, pct07);
#pragma unroll
          for (uint32_t c = 0; c < kSeqCacheSlots; ++c)
            if (c == sel) slot_key[c] = key;
          n_builds++;
        }
        age_ctr++;
#pragma unroll
        for (uint32_t c = 0; c < kSeqCacheSlots; ++c)
          if (c == sel) slot_age[c] = age_ctr;
        BS_SEQ_P(4);
        first_k = seq_scan_cached<TS>(nd, sq, prm, sh_, ch, sel, tcls, pct07, R, hit_par, n_rounds);
      } else {
        first_k = seq_scan<TS>(nd, sq, prm, sh_, tcls, pct07, R, n_rounds);
      }
      n_scan++;
      fk = first_k == BS_INF ? BS_K_NONE : first_k;
      if (first_k == BS_INF) {                               // compareClusterResourceAndRequire false: AddToDenyCache (:142,:163)
        code = code == BS_PF_PASS_FIRST_FITS ? BS_PF_REJECT_FIRST : BS_PF_REJECT_RESERVE;
        deny = true;
      }
    }
    // ---- node choice + assume (WRITE PHASE from here on)
    BS_SEQ_T(3);
    BS_SEQ_P(8);
    uint32_t at = BS_INF;
    if (BS_PF_IS_PASS(code) && (grouped || gi == BS_POD_NOT_GROUPED)) {   // (a labelled pod of an unknown group, here on its lastPermittedPod entry:
      //                                                                   Permit answers "can not found pod group", core.go:275-278, and the framework forgets it)
      q.pcls = uni32(sh_.pw_cls[pw]);
      q.ppres = uni32(sh_.pw_pres[pw]);
#pragma unroll
      for (uint32_t j = 0; j < BS_MAX_LANES; ++j) q.preq[j] = j < L ? (int64_t)uni64((uint64_t)sh_.pw_req[j][pw]) : 0;
      if (prm.filter_deny && grouped) {
        // ---- Filter's TTL writes (core.go:170-191), every node of the list offered (the batch form's rule, bs_batch_run): a node whose
        // Filter fails — getLeftResource nil (:545-548) or neither case 2 nor case 3 (:562-563) — deny-lists the group (:183-185) for the
        // gang's later pods; a node whose Filter passes leaves the pod's lastPermittedPod entry (:188).  The pod itself goes on.
        bool failed = false, passed = false;
        if (q.fl == BS_FL_EVALUATED) {
          if (!drained && stores_pending) { __syncthreads(); stores_pending = false; }      // the assume steps of earlier pods have landed
          drained = true;
          for (uint32_t n = threadIdx.x; n < N; n += kSeqBlock) {
            bool c2 = !(q.ff & 1u), c3h = !(q.ff & 2u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int64_t left = wsub(nd.alloc[(size_t)j * nd.stride + n], sq.nreq[(size_t)j * nd.stride + n]);   // getLeftResource :460-463
              c2 = c2 && left >= q.FR[j];
              c3h = c3h && left >= q.FM[j];
            }
            const bool ok = !(nd.flags[n] & (BS_NODE_NIL | BS_NODE_NO_NODE)) && (c2 || !c3h);
            failed = failed || !ok;
            passed = passed || ok;
          }
          const uint32_t par = hit_par & 1u;
          const uint32_t wbits = (__ballot(failed) ? 1u : 0u) | (__ballot(passed) ? 2u : 0u);
          if (lane_id() == 0) sh_.fdw[par][wave_id()] = wbits;
          lds_barrier();
          uint32_t all = 0;
#pragma unroll
          for (int ww = 0; ww < kSeqWaves; ++ww) all |= sh_.fdw[par][ww];
          all = uni32(all);
          failed = all & 1u;
          passed = all & 2u;
        } else if (q.fl < 16u) passed = N != 0u;                                             // case 1 / no MinResources: nil on every node
        if (failed) deny = true;                                                             // (written out with PreFilter's own entries below)
        if (passed && t0) sq.last_permitted[i] = 1;
      }
      SeqAssumed as;
      // first-fit cursor of the pod's request class (pods of a gang share a template: the search of the next one starts where this
      // one's ended).  Exact while (a) no assumed request has a negative lane — then free capacity only shrinks and a node that
      // could not hold this request cannot hold it later —, (b) the plugin's Filter does not gate the choice (its verdict moves
      // with the leader), (c) the cursor was left by the same fit class.
      const bool cur_ok = prm.use_cursor && mono && q.fl < 16u && q.fl != BS_FL_EVALUATED && q.pcls < nd.n_classes;
      uint32_t pc = 0, ce = 0, start = 0;
      if (cur_ok) {
        pc = uni32(sh_.pw_pclass[pw]);
        ce = ((pc * 2654435761u) ^ (q.pcls * 40503u)) >> (32 - kSeqCursorBits);
        const unsigned long long cv = sh_.cur_kn[ce];
        if ((uint32_t)(cv >> 32) == pc + 1u && sh_.cur_fc[ce] == q.pcls) start = (uint32_t)cv;
      }
      BS_SEQ_P(9);
      if (cur_ok && start >= nd.n) {
        at = BS_INF;                                         // an identical request found no node before, and nothing has been freed since
      } else {
        at = seq_pick<TS>(nd, sq, prm, sh_, q, slot_key, drained || !stores_pending, hit_par, as, n_tiles, start >> 6);
        if (!drained) stores_pending = false;                // (the search drained them itself)
        if (cur_ok && t0) {                                  // (every wave read the entry before the search's barrier)
          sh_.cur_kn[ce] = ((unsigned long long)(pc + 1u) << 32) | (at != BS_INF ? at : nd.n);
          sh_.cur_fc[ce] = q.pcls;
        }
      }
      if (at != BS_INF) {
#pragma unroll
        for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
          if (j < L && j != 3 && q.preq[j] < 0 && (j < 3 || ((q.ppres >> (j - 4)) & 1u))) mono = false;   // capacity was FREED: cursors are history
      }
      n_pick++;
      if (at != BS_INF) {
        stores_pending = true;
        // ---- the table summaries follow the assume step: the node's left values moved by the pod's request in every table the
        // node counts in (it is schedulable — no flag — so it is a row and has no taint error; what is left is the class's fit bit)
        if (prm.cache_slots) {
          int64_t dlt[BS_MAX_LANES];
          bool drop = false;                                 // a bound would break: negative request, or a scalar key the node's requests did not have
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) {
            dlt[j] = 0;
            if (j < L) {
              if (j < 3) dlt[j] = (j == BS_LANE_EPH && !gate) ? 0 : q.preq[j];
              else if (j == 3) dlt[j] = 1;
              else if ((q.ppres >> (j - 4)) & 1u) {
                if (!((as.ap >> (j - 4)) & 1u)) dlt[j] = 0;                        // allocatable lacks the key: the lane never exists for this node
                else if ((as.rp >> (j - 4)) & 1u) dlt[j] = q.preq[j];
                else drop = true;                                                  // the lane starts to exist at this node
              }
              if (dlt[j] < 0) drop = true;
            }
          }
          const uint32_t ta = at >> 6;
#pragma unroll
          for (uint32_t c = 0; c < kSeqCacheSlots; ++c) {
            if (c < prm.cache_slots && slot_key[c] != BS_INF && ((as.fitbits >> c) & 1u)) {
              if (drop) slot_key[c] = BS_INF;                // summarised afresh at its next use
              else {
                for (uint32_t t = ta + threadIdx.x; t < ch.T; t += kSeqBlock) {       // the node's tile (its total) and every tile behind it (their offsets)
                  unsigned long long* pb = (t == ta ? ch.tt : ch.off) + (size_t)c * L * ch.T + t;
                  unsigned long long cur[BS_MAX_LANES];      // every lane's word fetched before the first is written back (one LDS latency, not L)
#pragma unroll
                  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                    if (j < L) cur[j] = pb[(size_t)j * ch.T];
#pragma unroll
                  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
                    if (j < L) pb[(size_t)j * ch.T] = cur[j] - (unsigned long long)dlt[j];
                }
              }
            }
          }
        }
      }
    }
    BS_SEQ_T(4);
    BS_SEQ_P(14);
    if (deny) { gflags |= BS_GROUP_DENIED; own.flags = gflags; }
    if (threadIdx.x == kSeqResultThread) {                  // write-only results: not wave 0, whose thread 0 has the group and node bookkeeping to do
      sq.pf_code[i] = (uint8_t)code;
      if (sq.pf_first_k) sq.pf_first_k[i] = fk;
      if (sq.pf_leader) sq.pf_leader[i] = sop_leader;
    }
    if (t0 && deny) sq.g_flags[gi] = (uint8_t)gflags;
    BS_SEQ_P(15);
    if (at != BS_INF && !grouped) {                          // core.go:269-272: Permit lets it through at once
      if (t0) sq.pod_node[i] = (int32_t)at;
    } else if (at != BS_INF) {
      // ---- Permit, core.go:268-309.  Every wave applies the update to its copy of the group; thread 0 writes it out.
      const uint32_t mm = pod_mm;
      const uint32_t m1 = gmatched + 1u;                                                             // :290
      const bool ready = m1 >= (uint32_t)(mm - gsc);                                                 // :303
      // sendStartScheduleSignal -> StartBatchSchedule (batchscheduler.go:254-344) releases — unless the phase is none of PreScheduling /
      // Scheduling (:258-261): then the latch is all that happens and the pod waits on like any other
      const bool release = ready && !(gflags & BS_GROUP_PHASE_CLOSED);
      const uint32_t has_rec = own.nwait & kSeqHasRecord, nw = own.nwait & ~kSeqHasRecord;
      const uint32_t prev = own.head, k = nw + 1u;
      if (ready) gflags |= BS_GROUP_SCHEDULED_LATCH;                                                 // :305
      if (!release) {
        if (t0) {
          sq.g_matched[gi] = m1;
          if (gflags != own.flags) sq.g_flags[gi] = (uint8_t)gflags;
          sq.wait_rec[i] = ((unsigned long long)prev << 32) | at;
          sq.head[gi] = i + 1u;
          sq.nwait[gi] = k | has_rec;
          if (wl_ok && wl_cnt < kSeqWaitList) { sh_.wl_pod[wl_cnt] = i; sh_.wl_node[wl_cnt] = at; }
        }
        own.matched = m1;
        own.flags = gflags;
        own.head = i + 1u;
        own.nwait = k | has_rec;
        if (wl_cnt < kSeqWaitList) wl_cnt++; else wl_ok = false;
      } else {
        // EVERY entry of MatchedPodNodes is allowed (:292,:301-333), deleted (:326) and counted by PostBind (core.go:327): the pods this
        // pass placed — they bind in parallel from the LDS list when it holds them all — and the m1 - k that were waiting when it began
        const bool from_list = wl_ok && wl_cnt == nw;
        if (from_list) {
          for (uint32_t e = threadIdx.x; e < wl_cnt; e += kSeqBlock) sq.pod_node[sh_.wl_pod[e]] = (int32_t)sh_.wl_node[e];
        }
        const uint32_t scn = gsc + m1;                                                               // PostBind, core.go:327 (uint32)
        if (scn >= mm) gflags |= BS_GROUP_PHASE_CLOSED;                                              // core.go:329-330 phase Scheduled
        if (t0) {
          sq.g_matched[gi] = 0;
          sq.pod_node[i] = (int32_t)at;
          if (!from_list)
            for (uint32_t wv = prev; wv != 0u;) {
              const unsigned long long rec = sq.wait_rec[wv - 1u];
              sq.pod_node[wv - 1u] = (int32_t)(uint32_t)rec;
              wv = (uint32_t)(rec >> 32);
            }
          sq.head[gi] = 0;
          sq.nwait[gi] = kSeqHasRecord;
          sq.g_sc[gi] = scn;
          sq.g_flags[gi] = (uint8_t)gflags;
          if (!has_rec) {
            if (n_released < sq.cap) {
              sq.released_group[n_released] = (uint32_t)gi;
              sq.released_pods[n_released] = m1;
              sq.first_tick[n_released] = sq.t_first[gi];
              sq.ready_tick[n_released] = (unsigned long long)wall_clock64() - clk0;
              sq.slot_of[gi] = n_released;
            }
          } else if (sq.slot_of[gi] != BS_INF) {
            sq.released_pods[sq.slot_of[gi]] += m1;          // a second release of the gang (Status.Scheduled still below MinMember: only through uint32 wrap)
          }
        }
        own.matched = 0;
        own.head = 0;
        own.nwait = kSeqHasRecord;
        own.sc = scn;
        own.flags = gflags;
        wl_cnt = 0;
        wl_ok = true;
        if (!has_rec) n_released++;
      }
      {
        // matched moved: the group's progress changed.  findMaxPG's answer stands unless this key now beats the winner's, or the
        // winner itself fell back (released: no candidate any more).
        const unsigned long long key = seq_key((uint32_t)gi, own.flags, mm, own.sc, own.matched);
        if (t0) { if (prm.keys_in_lds) s_keys[gi] = key; else sq.keys[gi] = key; }
        if (fold_valid) {
          if (fold_top == 0ull || fold_panic) fold_valid = false;
          else if (gi == fold_leader) { if (key < fold_top || (key & 1ull) || key == ~0ull) fold_valid = false; else fold_top = key; }
          else if (key > fold_top) fold_valid = false;
        }
      }
    }
    if (grouped && gi == ldr_of) {                           // the leader's copy follows what this pod did to its group
      const uint32_t h = ldr.head;
      ldr = own;
      ldr.head = h;
    }
    BS_SEQ_T(5);
    BS_SEQ_P(16);
  }
  BS_SEQ_FULL_BARRIER();
  if (t0) {
    sq.info[0] = n_released;
    sq.info[1] = (unsigned long long)wall_clock64() - clk0;
    sq.info[2] = n_pick;
    sq.info[3] = n_scan;
    sq.info[4] = (unsigned long long)(uint32_t)(sop_leader + 1);
    sq.info[5] = n_rounds;
    sq.info[6] = n_tiles;
    sq.info[7] = n_folds | (n_builds << 40);
#ifdef BS_SEQ_PROBE
    for (int k = 0; k < 8; ++k) sq.info[8 + k] = ph[k];
    for (int k = 0; k < 8; ++k) { sq.info[16 + k] = g_seq_scan_ph[k]; g_seq_scan_ph[k] = 0; }
    for (int k = 0; k < 32; ++k) sq.info[32 + k] = sh_.ph2[k];
#endif
  }
}

}  // namespace bs
