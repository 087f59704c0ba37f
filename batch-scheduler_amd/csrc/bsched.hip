// bsched.hip — C ABI of include/bsched.h on top of the gfx950 kernels in bs_kernels.hpp.
//
// Host responsibilities only: device memory, one HIP stream, launch geometry, event timing,
// H2D/D2H staging.  Every decision is computed on the GPU; there is no CPU evaluation path here
// (the CPU restatement lives in oracle/ and is test infrastructure).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: librccl itself is dlopen'ed on demand (hosts without RCCL can still load the library)

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#ifndef BS_UNITY
#define BS_TU_MAIN
#endif
#include "bs_kernels.hpp"
#include "bs_fast.hpp"
#include "bs_epoch.hpp"
#include "bs_sort.hpp"
#include "bs_fit.hpp"
#include "bs_queue.hpp"
#include "bs_fdeny.hpp"
#include "bs_seq.hpp"
#include "bs_launch.hpp"
#ifdef BS_UNITY   // one translation unit (the probe builds: g_probe / g_seq_scan_ph are per translation unit)
#include "tu_fast.hip"
#include "tu_seq.hip"
#endif

using namespace bs;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EventPair { hipEvent_t a, b; uint32_t id; };

// Host-side packing of the fit-builder inputs: every array goes into one byte arena (16-byte aligned
// pieces) that is uploaded with a single copy.
struct FitArena {
  std::vector<uint8_t> bytes;
  size_t put(const void* src, size_t n) {
    const size_t at = (bytes.size() + 15) & ~(size_t)15;
    bytes.resize(at + n);
    if (n) std::memcpy(bytes.data() + at, src, n);
    return at;
  }
};
template <typename T> const T* at_dev(const void* base, size_t off) { return reinterpret_cast<const T*>((const uint8_t*)base + off); }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// element n of a device array that may not exist yet (null + n is undefined behaviour even if nobody follows the pointer)
template <class T>
T* at(T* p, size_t n) { return p ? p + n : nullptr; }

// One pod pack = the arrays of bs_pods_soa for exactly `p` pods (the part a load uploads in ONE copy, `in_bytes`) followed by
// the per-pod ids the library derives (request class, (group, class) pair): what travels with a pod when the queue is patched.
// The pinned staging buffer uses the same layout (input part only) — and its OWN copy of it (bs_pods_map must not disturb the
// resident queue).
struct PodLayout {
  size_t group = 0, req = 0, pres = 0, cls = 0, owner = 0, flags = 0, in_bytes = 0, pclass = 0, ppair = 0, bytes = 0;
  uint32_t p = 0;
};
PodLayout pod_layout(uint32_t P, uint32_t L) {
  const size_t n = std::max<uint32_t>(P, 1);
  PodLayout l;
  size_t o = 0;
  l.group = o; o = align256(o + n * 4);
  l.req = o; o = align256(o + n * L * 8);
  l.pres = o; o = align256(o + n * 4);
  l.cls = o; o = align256(o + n * 4);
  l.owner = o; o = align256(o + n * 8);
  l.flags = o; o = align256(o + n);
  l.in_bytes = o;
  l.pclass = o; o = align256(o + n * 4);
  l.ppair = o; o = align256(o + n * 4);
  l.bytes = o;
  l.p = P;
  return l;
}

}  // namespace

struct bs_ctx {
  bs_config cfg{};
  uint32_t L = 4, S = 0, LP = 4;
  hipStream_t stream = nullptr;
  std::string last_error;

  // ---- nodes
  bool have_nodes = false, have_fit = false, have_groups = false, have_pods = false;
  uint32_t N = 0, Ncap = 0, M = 0, C = 0, fit_words = 0;
  DevBuf d_alloc, d_nreq, d_apres, d_rpres, d_nflags, d_fit, d_kmap, d_m, d_left4, d_lglob;
  DevBuf d_fitarena, d_fitcols;                  // bs_fit_build inputs / label columns
  std::vector<int64_t> h_alloc, h_nreq;          // [L][N] mirrors (churn + read-back)
  std::vector<uint32_t> h_apres, h_rpres, h_kmap;
  std::vector<uint8_t> h_nflags;
  std::vector<uint32_t> h_fit;                   // [C][fit_words]

  // ---- groups
  uint32_t G = 0, n_uncaptured = 0;   // groups without a pod (first-pod capture possible)
  std::vector<uint32_t> h_gmatched, h_gcls;
  std::vector<uint8_t> h_gflags;
  int32_t steady_table = -1;        // the one table every reservation query uses when no capture can occur, -1 unknown
  // Speculation (bs_batch_run): a group patch re-runs findMaxPG on the device, and the host would have to wait for its answer (the
  // table id) before it can launch the chain — ~4 us of idle GPU and ~10 us of spinning per cycle.  In a steady state the answer is
  // nearly always the one of the cycle before, so the chain is launched on THAT and checked when the results are first asked for
  // (batch_settle): a wrong guess costs one re-run, a right one nothing.
  int32_t steady_prev = -1;          // steady_table of the last resolved analysis
  bool spec_active = false;          // the last batch ran on a guessed table that nobody has checked yet
  int32_t spec_table = -1;
  uint32_t spec_stages = 0;
  uint32_t no_spec = 0;              // BS_NO_SPECULATE=1
  uint64_t n_spec = 0, n_spec_miss = 0;
  // BS_HOST_PROBE=1: where bs_batch_run's host time goes (ns, accumulated; printed at bs_destroy)
  uint32_t host_probe = 0;
  uint64_t hp_ns[6] = {0, 0, 0, 0, 0, 0}, hp_n = 0, hp_t0 = 0, hp_t1 = 0;
  uint64_t early_filter_min = 200000000ull;   // pod x node pairs from which Filter overlaps the scan
  hipStream_t stream3 = nullptr;    // early Filter: runs beside the node scan when no capture can occur
  hipEvent_t ev_query = nullptr, ev_filter = nullptr;
  // groups live in ONE device allocation (one pinned-staged H2D per load); d_info / h_info carry what findMaxPG
  // found for the loaded state back to the host without a stream wait (see resolve_groups)
  DevBuf d_gpack, d_info, d_gdelta;
  size_t off_gmm = 0, off_gsc = 0, off_gmatched = 0, off_gflags = 0, off_gcls = 0, off_gminres = 0, off_gmrpres = 0, off_gocc = 0, gpack_bytes = 0;
  void* h_gstage = nullptr;          // pinned: groups pack, then deltas
  size_t h_gstage_cap = 0;
  hipEvent_t ev_gstage = nullptr;
  bool gstage_busy = false;
  int32_t* h_info = nullptr;         // pinned [8]: leader, panic, steady table, tag | K of the loaded pods, tag
  int32_t info_tag = 0, kinfo_tag = 0;
  bool info_pending = false, kinfo_pending = false;
  uint32_t max_group_cls = 0, max_pod_cls = 0;   // largest fit class any HAS_POD group / grouped pod names (checked against C per batch)
  uint32_t h_K = 0;                  // request classes of the loaded pods (valid after resolve_pods)
  uint32_t k_bound = 0;              // while kinfo_pending: an upper bound of the class count the device holds (last known K + pods inserted since)

  // ---- pods
  uint32_t P = 0;
  // pods live in ONE device allocation (one H2D per batch from a pinned staging buffer); outputs likewise (one D2H)
  DevBuf d_pack[2], d_outpack;       // two pod packs: bs_pods_apply compacts from the current one into the other
  PodLayout lay[2], stage_lay;       // their layouts, and the staging buffer's own
  uint32_t cur_pack = 0;
  void* h_stage = nullptr;           // pinned host staging
  hipEvent_t ev_stage = nullptr;     // the last H2D out of the staging buffer (bs_pods_load does not wait for it)
  bool stage_busy = false;
  bool last_use_classes = false;
  size_t h_stage_cap = 0;
  uint32_t map_p = 0;                // pods the staging buffer is currently mapped for (bs_pods_map), 0 = not mapped
  // queue-resident cycle (bs_pods_apply, bs_queue.hpp)
  DevBuf d_gstat2, d_cdir, d_pdir, d_ckeys, d_cpres, d_pkeys;
  uint32_t gstat_cur = 0;            // which of d_gstat / d_gstat2 holds the per-group minima of the resident queue
  uint32_t pair_cap = 0, dir_slots = 0;   // id space of classes / pairs between two derivations; hash slots of each directory
  uint32_t ids_used = 0;             // upper bound of the class / pair ids drawn since the last derivation
  bool rep_valid = false;            // d_cls_rep / d_cls_id / pair ids still name pods of the resident queue (no compaction since)
  bool dirs_ready = false;           // the directories match the resident queue's classes and pairs
  uint32_t id_room = 0;              // BS_ID_ROOM: ids beyond the queue length (0 = the default: as many again + 1024)
  uint32_t serial_insert_max = 2048; // more inserted pods than this: re-derive in parallel instead of the insert wave
  void* h_dstage = nullptr;          // pinned: the delta the apply kernel reads in place
  size_t h_dstage_cap = 0;
  bool dstage_busy = false;
  uint64_t n_applies = 0, n_rederives = 0;
  void* h_nstage = nullptr;          // pinned: node requests of bs_nodes_assume
  size_t h_nstage_cap = 0;
  hipEvent_t ev_nstage = nullptr;
  bool nstage_busy = false;
  size_t off_pf_code = 0, off_pf_first_k = 0, off_pf_leader = 0, off_fl_code = 0, off_fl_feasible = 0, off_fl_slot = 0, off_admit = 0, off_ready = 0, outpack_bytes = 0;

  // ---- batch scratch / outputs
  DevBuf d_first_elig, d_first_owner, d_first_reject, d_first_pod, d_cap_epoch;
  DevBuf d_epoch, d_nepochs, d_leader_epoch, d_panic_epoch;
  DevBuf d_tcode, d_stage, d_leader_raw, d_first_row, d_first_row64, d_scan_rec, d_feas_rec, d_chunk_rec, d_qreq_s, d_qflags_s, d_qpos;
  DevBuf d_needed, d_qcount, d_ticket, d_desc;
  bool scratch_armed = false;
  bool side_ready = false;      // desc[] of the steady-state table is in place for the next batch   // per-group minima are INF (k_init ran, or the previous batch's k_tally re-armed them)
  DevBuf d_tables, d_kp, d_stats, d_fparams, d_fflags, d_chunk_tot, d_blk_scratch, d_gmax, d_chunk_kp;
  // request slots (see BatchDev): classes of the loaded pods + per-batch slot arrays
  DevBuf d_cls_slots, d_cls_rep, d_cls_id, d_qtab_s, d_fu_slot, d_uparams, d_uflags, d_uclaim, d_fu_bitmap, d_fu_feas;
  DevBuf d_nodew;                    // node words of the batch (BatchDev::nodew): 3 tables x (W + 2) word pairs + the two leaders' maxSingle
  bool batch_void = false;           // the last batch's results must not be handed out (check_handover); cleared by the next bs_batch_run
  uint32_t tp_tmin = 768;            // BS_TP_TMIN: tiles of Filter slots from which the transposed items take pairs of tiles (x ranks on a sharded context)
  bool no_nodew = false;             // BS_NO_NODEW=1: the transposed Filter item derives the node-only masks of every block itself (rounds 4-5; A/B switch)
  uint32_t slot_keep = 0xFFFFFFFFu;   // BS_HASH_SLOT_BITS (tests): directory probes start at hash & slot_keep
  uint32_t cls_cap = 0, hash_keep = 0x7FFFFFFFu, n_nominres = 0, scan_slots_cap = 0, filter_slots_cap = 0;
  DevBuf d_fl_bitmap, d_admit, d_ready, d_gcount, d_admit64, d_own_start;
  bool owner_ready = false;          // own_start[] matches the resident queue and the group count (sharded contexts only)
  // fast path (bs_fast.hpp)
  DevBuf d_order_rank, d_sort;         // queue ordering: per-group order ranks; inputs | index ping-pong | permutation
  uint32_t order_g = 0;
  DevBuf d_gstat, d_pair_next, d_pair_firstq, d_first_reach, d_qstamp_s, d_fast_reject, d_epoch_group;
  bool pairs_ready = false;          // d_gstat / pairs match the loaded pods and G
  bool bitmap_valid = false;         // d_fl_bitmap holds the expanded rows of the last batch
  bool last_fast = false;
  bool batch_since_pods = false;     // a batch ran over the loaded pods (its slot mode is the one the rows have)
  // result staging (bs_batch_read) and, in latency mode, the pinned result pack the last launch writes itself
  void* h_rstage = nullptr;
  size_t h_rstage_cap = 0;
  uint8_t* h_hout = nullptr;         // [outpack layout | feas[hstride] | tag]
  uint64_t* h_hrows = nullptr;       // [W + 1][hstride]
  size_t h_hout_cap = 0, h_hrows_cap = 0, off_hfeas = 0, off_htag = 0;
  uint32_t hstride = 0;
  int32_t host_tag = 0;
  bool last_host_out = false;
  uint32_t no_fast = 0;
  // positional three-launch chain (bs_epoch.hpp): analysis of (groups, pods) kept across batches
  DevBuf d_run_of_epoch, d_run_leader, d_gslot, d_gfirstq;
  bool epochs_ready = false, einfo_pending = false;
  int32_t einfo_tag = 0;
  uint32_t h_R = 0, h_eflags = 0, no_epoch = 0;
  uint32_t last_chain = 0;           // 0 general chain, 1 steady-state chain, 2 positional chain
  uint32_t last_rows = 0;            // Filter slot rows of the last positional batch
  // single-query scratch
  DevBuf d_sq;
  DevBuf d_seq;                      // bs_seq_run: scaled allocatables, keys, per-gang / per-pod bookkeeping, results
  uint32_t table_slots = 0, table_mcap = 0;

  uint32_t rank = 0, nranks = 1;
  bool reduce_external = false;      // partitioned mode: tally only, the caller reduces and calls bs_batch_finish
  uint32_t* ext_admit = nullptr;     // caller-owned device memory for the admit counters
  int32_t sop_leader0 = -1;
  uint32_t last_stages = 0;
  // Batch counters.  batch_seq: monotonic, 64 bit (timing sampling, statistics).  stamp_ctr in [0, 65534]: slot stamps are
  // 1 + stamp_ctr (16 bits in the slot words); when it comes round to 0 the stamped arrays are zeroed, so a slot nobody wrote
  // for 65535 batches cannot look live again.  key_seq in [1, 0xFFFFFFFE]: the 64-bit atomicMin keys carry ~key_seq in the
  // high word ("a newer batch always wins", never all-ones = the 'none' the arrays are born with); when it runs out it
  // restarts at 1 behind a re-fill of the keyed arrays with 'none' (once per 2^32 - 2 batches).
  uint64_t batch_seq = 0;
  uint32_t stamp_ctr = 1, key_seq = 1;
  bool rekey_pending = false;
  bool batch_pending_finish = false;
  bool groups_launch_pending = false; // bs_groups_apply left its (inline) deltas + findMaxPG for the next launch: k_pods_apply takes them along, anything else flushes
  DeltaPack pending_dp{};
  uint32_t no_fuse_final = 0;        // BS_NO_FUSE_FINAL: launches B and C always as separate launches
  uint32_t tp_filter = 6;            // BS_TP_FILTER (throughput regime = more than 16 tiles of class slots): 0 = scan and Filter roles in one launch (k_fast_scan_filter: rounds 2-4),
                                     // 1..4 = k_fast_scan, then k_fast_filter<4,DB> / <2,DB> / <2,!DB> / k_fast_filter_w7 (109 / 93 / 75 / 72 VGPRs),
                                     // 5 = k_fast_scan, then k_fast_filter_t (the transposed item, bs_filter_t.hpp: 64 VGPRs),
                                     // 6 / 7 = one launch, Filter role by the transposed item (7: the Filter blocks first),
                                     // 8 = as 5, the two launches side by side on two streams
  uint32_t tp_share = 0;             // BS_TP_SHARE: scan shares per tile of class slots when launch B is not the fused form (at most);
                                     // 0 = 2 in the throughput regime, 64 otherwise (what the sweeps of profiles/r04c_* say)
  uint32_t tp_fwaves = 0;            // BS_TP_FWAVES: waves the Filter work of that regime is cut for; 0 = 16384 from 65 536 (tile, two node
                                     // blocks) units on, filter_waves below; an explicit BS_FILTER_WAVES rules
  bool filter_waves_env = false;
  uint32_t tp_split = 0;             // BS_TP_SPLIT: the transposed Filter items are cut for tp_split x the launched waves and dealt out tile quad by
                                     // tile quad (filter_loop_t, by_tile); 0 = 2 on a rank of a sharded context (bs_shard_set), 1 otherwise.
                                     // Class ids follow the queue (k_pod_class_ids), so all but 1 / nranks of the slot tiles are idle on a rank and
                                     // return at their first load; the live ones are cut finer so that they spread over more of the launched
                                     // waves.  cfg4 all-distinct, rank 0 of 8 (profiles/r05_shard_scaling.md): 60 us at x2, 63 at x4, 74 at x8
                                     // (an item's prologue — requests, bounds, first node block — is ~4 us whatever its length).
  int fused_blocks_resident = -1;    // whole-chip residency of k_fast_scan_filter_final (blocks), -1 = not asked yet
  int step_a_resident[2] = {-1, -1};          // ... of k_fast_step_a
  // The one-launch form of launch A + the scan / Filter roles (k_fast_step_a, then k_fast_final), where it applies (the latency regime: at most 256
  // classes, 64 table chunks, 4 scalar lanes; the second batch over a queue onwards).  BS_STEP_A=2, the DEFAULT since round 6: the class-slot form —
  // class_slots_block publishes every class's slots from the class directory, the pod blocks gate nobody: 19.05-19.35 us per cfg3/tail step against
  // 20.5-20.7 for the two-launch chain.  BS_STEP_A=1: round 5's form (every pod block publishes: 32 us, kept as a tested experiment).  BS_STEP_A=0: off.
  uint32_t step_a_form = 3;
  bool step_a_on = true;
  uint32_t step_shares = 8;          // BS_STEP_SHARES: blocks that share one table chunk's class slots (class-slot form, cfg3: 2 / 4 / 8 / 16 shares = 25.1 / 20.4 / 19.05 / 20.9 us per step)
  uint32_t test_timeout_after = 0;   // BS_TEST_HANDOVER_TIMEOUT=n (test hook): the n-th one-launch step reports a timed-out hand-over as the device would
  uint32_t tk_pods = 0, tk_tab = 0;  // values of ticket[8] / ticket[9] the next k_fast_step_a starts from (never reset: wrap-safe differences)
  uint32_t tk_p1 = 0, tk_done = 0;   // ... of the spread counter at kTkP1 (form 3: the pod blocks' first halves); tk_done: of the counter at kTkDone (large queues: every table / Filter block adds once)
  bool last_step_a = false;
  uint32_t scan_share_override = 0, no_fuse_filter = 0, early_forced = 0, target_waves = 8192, filter_waves = 8192, collect_stats = 0;
  uint32_t general_waves = 4096;     // scan grid cap of the general chain (tools/cold_sweep.py)
  bs_batch_stats stats{};

  // ---- timing
  std::vector<EventPair> events;
  size_t events_used = 0;
  bs_timing timing{};

  // ---- native RCCL (dlopen'ed on demand; entry points resolved once in bs_comm_init)
  void* rccl_handle = nullptr;
  void* comm = nullptr;
  decltype(&ncclAllReduce) rccl_allreduce = nullptr;
  decltype(&ncclCommDestroy) rccl_destroy = nullptr;
  uint32_t launches = 0;             // kernel launches of the last batch
  // BS_BATCH_FILTER_DENY (bs_fdeny.hpp)
  DevBuf d_fd_event, d_fd_in, d_fd_flag;
  bool fd_on = false;                // the run being launched replays Filter's deny entry
  bool fd_active = false;            // the last batch did, and nobody has looked at its flag words yet (fd_settle)
  uint32_t first_reach_hint = 0xFFFFFFFFu;   // bs_first_reach_hint (partitioned mode), reset by every queue load / patch
  bool fd_unsynced = false;          // a BS_BATCH_FILTER_DENY batch was launched and the stream has not been waited for since
  bool fd_in_live = false;           // a fixed-point re-run: the chains honour d_fd_in
  uint32_t fd_iter = 0, fd_stages = 0, fd_seq_inv = 0;
  uint64_t n_fd_reruns = 0;          // fixed-point re-runs so far (bs_batch_stats_get)
};

namespace {

const char* kKernelNames[BS_KERNEL_COUNT] = {"prepass", "leader", "query", "tables", "scan", "resolve", "filter", "tally"};

#define HIPCHK(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(_e);                      \
      return BS_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int timer_begin(bs_ctx* c, uint32_t id, size_t* slot, hipStream_t st = nullptr) {
  *slot = (size_t)-1;
  if (!c->cfg.enable_timing) return BS_OK;
  // mode 1: only the two dominant kernels, and only every 8th batch — an event pair costs a few
  // microseconds of stream time, sampling keeps the timed region representative
  if (c->cfg.enable_timing == 1 &&
      ((id != BS_KERNEL_QUERY && id != BS_KERNEL_SCAN && id != BS_KERNEL_RESOLVE && id != BS_KERNEL_FILTER) || (c->batch_seq & 7ull) != 0))
    return BS_OK;
  if (c->events_used == c->events.size()) {
    EventPair ep{};
    HIPCHK(c, hipEventCreate(&ep.a));
    HIPCHK(c, hipEventCreate(&ep.b));
    c->events.push_back(ep);
  }
  *slot = c->events_used++;
  c->events[*slot].id = id;
  HIPCHK(c, hipEventRecord(c->events[*slot].a, st ? st : c->stream));
  return BS_OK;
}
int timer_end(bs_ctx* c, size_t slot, hipStream_t st = nullptr) {
  if (slot == (size_t)-1) return BS_OK;
  HIPCHK(c, hipEventRecord(c->events[slot].b, st ? st : c->stream));
  return BS_OK;
}
int timer_collect(bs_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->events_used; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b));
    c->timing.total_ms[c->events[i].id] += ms;
    c->timing.launches[c->events[i].id] += 1;
  }
  c->events_used = 0;
  return BS_OK;
}

// a failed launch is reported with the kernel group it belongs to (bs_kernel_name)
#define LAUNCHCHK(ctx, id)                                                                         \
  do {                                                                                             \
    hipError_t _e = hipGetLastError();                                                             \
    if (_e != hipSuccess) {                                                                        \
      (ctx)->last_error = std::string("launch of kernel group '") + kKernelNames[id] + "': " + hipGetErrorString(_e); \
      return BS_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)
#define TIMED_ON(ctx, id, st, ...)                             \
  do {                                                         \
    size_t _slot;                                              \
    int _rc = timer_begin(ctx, id, &_slot, st);                \
    if (_rc) return _rc;                                       \
    __VA_ARGS__;                                               \
    LAUNCHCHK(ctx, id);                                        \
    _rc = timer_end(ctx, _slot, st);                           \
    if (_rc) return _rc;                                       \
  } while (0)
#define TIMED(ctx, id, ...) TIMED_ON(ctx, id, (hipStream_t) nullptr, __VA_ARGS__)

// one batch done: advance the three counters (see bs_ctx)
void advance_batch_seq(bs_ctx* c) {
  c->batch_seq++;
  c->stamp_ctr = (c->stamp_ctr + 1u) % 65535u;
  if (c->key_seq >= 0xFFFFFFFEu) { c->key_seq = 1; c->rekey_pending = true; }
  else c->key_seq++;
}

int flush_groups(bs_ctx* c);
// Every entry point starts here.  A group patch whose launch was deferred (bs_groups_apply) goes out now, unless the caller
// is the one call that can take it along in its own launch (bs_pods_apply).
int use_device(bs_ctx* c, bool flush = true) {
  HIPCHK(c, hipSetDevice(c->cfg.device));
  if (flush && c->groups_launch_pending) return flush_groups(c);
  return BS_OK;
}

NodesDev nodes_dev(const bs_ctx* c) {
  NodesDev nd{};
  nd.n = c->N;
  nd.stride = c->Ncap;
  nd.alloc = c->d_alloc.as<int64_t>();
  nd.req = c->d_nreq.as<int64_t>();
  nd.apres = c->d_apres.as<uint32_t>();
  nd.rpres = c->d_rpres.as<uint32_t>();
  nd.flags = c->d_nflags.as<uint8_t>();
  nd.fit = c->d_fit.as<uint32_t>();
  nd.fit_words = c->fit_words;
  nd.n_classes = c->C;
  nd.kmap = c->d_kmap.as<uint32_t>();
  nd.m = c->M;
  nd.left4 = c->d_left4.as<int64_t>();
  nd.lglob = c->d_lglob.as<int64_t>();
  return nd;
}
GroupsDev groups_dev(const bs_ctx* c) {
  GroupsDev g{};
  g.g = c->G;
  uint8_t* pk = c->d_gpack.as<uint8_t>();
  g.min_member = reinterpret_cast<uint32_t*>(at(pk, c->off_gmm));
  g.status_scheduled = reinterpret_cast<uint32_t*>(at(pk, c->off_gsc));
  g.matched = reinterpret_cast<uint32_t*>(at(pk, c->off_gmatched));
  g.flags = at(pk, c->off_gflags);
  g.cls = reinterpret_cast<uint32_t*>(at(pk, c->off_gcls));
  g.minres = reinterpret_cast<int64_t*>(at(pk, c->off_gminres));
  g.mrpres = reinterpret_cast<uint32_t*>(at(pk, c->off_gmrpres));
  g.occupied = reinterpret_cast<uint64_t*>(at(pk, c->off_gocc));
  return g;
}
PodsDev pods_dev(const bs_ctx* c) {
  PodsDev p{};
  p.p = c->P;
  uint8_t* pk = c->d_pack[c->cur_pack].as<uint8_t>();
  const PodLayout& l = c->lay[c->cur_pack];
  p.group = reinterpret_cast<int32_t*>(at(pk, l.group));
  p.req = reinterpret_cast<int64_t*>(at(pk, l.req));
  p.pres = reinterpret_cast<uint32_t*>(at(pk, l.pres));
  p.cls = reinterpret_cast<uint32_t*>(at(pk, l.cls));
  p.owner = reinterpret_cast<uint64_t*>(at(pk, l.owner));
  p.flags = at(pk, l.flags);
  return p;
}
uint32_t* pclass_dev(const bs_ctx* c) { uint8_t* p = c->d_pack[c->cur_pack].as<uint8_t>(); return p ? reinterpret_cast<uint32_t*>(p + c->lay[c->cur_pack].pclass) : nullptr; }
uint32_t* ppair_dev(const bs_ctx* c) { uint8_t* p = c->d_pack[c->cur_pack].as<uint8_t>(); return p ? reinterpret_cast<uint32_t*>(p + c->lay[c->cur_pack].ppair) : nullptr; }
uint32_t* gstat_dev(const bs_ctx* c) { return (c->gstat_cur ? c->d_gstat2 : c->d_gstat).as<uint32_t>(); }
BatchDev batch_dev(const bs_ctx* c) {
  BatchDev b{};
  b.first_elig = c->d_first_elig.as<uint32_t>();
  b.first_owner = c->d_first_owner.as<uint32_t>();
  b.first_reject = c->d_first_reject.as<uint32_t>();
  b.first_pod = c->d_first_pod.as<uint32_t>();
  b.cap_epoch = c->d_cap_epoch.as<uint32_t>();
  b.epoch = c->d_epoch.as<uint32_t>();
  b.nepochs = c->d_nepochs.as<uint32_t>();
  b.leader_epoch = c->d_leader_epoch.as<int32_t>();
  b.panic_epoch = c->d_panic_epoch.as<uint8_t>();
  b.tcode = c->d_tcode.as<uint8_t>();
  b.stage = c->d_stage.as<uint8_t>();
  b.leader_raw = c->d_leader_raw.as<int32_t>();
  b.first_row = c->d_first_row.as<uint32_t>();
  b.first_row64 = c->d_first_row64.as<unsigned long long>();
  b.scan_rec = c->d_scan_rec.as<unsigned long long>();
  b.feas_rec = c->d_feas_rec.as<unsigned long long>();
  b.chunk_rec = c->d_chunk_rec.as<unsigned long long>();
  b.qreq_s = c->d_qreq_s.as<int64_t>();
  b.qflags_s = c->d_qflags_s.as<uint32_t>();
  b.qpos = c->d_qpos.as<uint32_t>();
  b.needed = c->d_needed.as<uint32_t>();
  b.qcount = c->d_qcount.as<uint32_t>();
  b.ticket = c->d_ticket.as<uint32_t>();
  b.gcount = c->d_gcount.as<uint32_t>();
  b.own_start = c->d_own_start.as<uint32_t>();
  b.admit64 = c->d_admit64.as<unsigned long long>();
  b.desc = c->d_desc.as<TableDesc>();
  b.tables = c->d_tables.as<int64_t>();
  b.kp = c->d_kp.as<uint32_t>();
  b.stats = c->d_stats.as<uint64_t>();
  b.chunk_tot = c->d_chunk_tot.as<unsigned long long>();
  b.blk_scratch = c->d_blk_scratch.as<uint32_t>();
  b.gmax = c->d_gmax.as<int64_t>();
  b.chunk_kp = c->d_chunk_kp.as<uint32_t>();
  b.fparams = c->d_fparams.as<int64_t>();
  b.fflags = c->d_fflags.as<uint32_t>();
  b.pclass = pclass_dev(c);
  b.kclass = at(c->d_nepochs.as<uint32_t>(), 2);
  b.cls_slots = c->d_cls_slots.as<unsigned long long>();
  b.cls_mask = c->cls_cap ? c->cls_cap - 1 : 0;
  b.qtab_s = c->d_qtab_s.as<int32_t>();
  b.uparams = c->d_uparams.as<int64_t>();
  b.uflags = c->d_uflags.as<uint32_t>();
  b.uclaim = c->d_uclaim.as<uint32_t>();
  b.nodew = nullptr;                 // run_fast sets it for the batches whose launch A builds the node words
  b.nodew_stride = 0;
  b.tiles2_min = 0xFFFFFFFFu;
  b.fu_bitmap = c->d_fu_bitmap.as<uint64_t>();
  // per-slot feasible counts sit right behind the last row of the slot bitmap: one 2-D copy returns rows + counts
  b.fu_feas = reinterpret_cast<uint32_t*>(at(c->d_fu_bitmap.as<uint64_t>(), (size_t)cdiv(c->N, 64) * c->filter_slots_cap));
  b.qstamp_s = c->d_qstamp_s.as<uint32_t>();
  b.first_pod_s = gstat_dev(c);
  b.first_np_s = at(gstat_dev(c), (size_t)c->G);
  b.first_owner_s = at(gstat_dev(c), (size_t)2 * c->G);
  b.pair_head = reinterpret_cast<const unsigned long long*>(at(c->d_gstat.as<uint32_t>(), ((size_t)3 * c->G + 1) & ~(size_t)1));
  b.ppair = ppair_dev(c);
  b.pair_stride = c->pair_cap;
  b.pair_next = c->d_pair_next.as<unsigned long long>();
  b.pair_firstq = c->d_pair_firstq.as<unsigned long long>();
  b.first_reach64 = c->d_first_reach.as<unsigned long long>();
  b.fast_reject = c->d_fast_reject.as<uint32_t>();
  b.epoch_group = c->d_epoch_group.as<uint32_t>();
  b.h_err = c->h_info ? c->h_info + 12 : nullptr;
  b.fd_event = c->d_fd_event.as<unsigned long long>();
  b.fd_in = c->fd_in_live ? c->d_fd_in.as<uint32_t>() : nullptr;
  b.fd_flag = c->d_fd_flag.as<uint32_t>();
  b.h_fd = c->h_info ? c->h_info + 14 : nullptr;
  uint8_t* ok = c->d_outpack.as<uint8_t>();
  b.pf_code = at(ok, c->off_pf_code);
  b.pf_first_k = reinterpret_cast<uint32_t*>(at(ok, c->off_pf_first_k));
  b.pf_leader = reinterpret_cast<int32_t*>(at(ok, c->off_pf_leader));
  b.fl_code = at(ok, c->off_fl_code);
  b.fl_feasible = reinterpret_cast<uint32_t*>(at(ok, c->off_fl_feasible));
  b.fu_slot = reinterpret_cast<uint32_t*>(at(ok, c->off_fl_slot));     // per-pod Filter slot travels with the other per-pod results
  b.fl_bitmap = c->d_fl_bitmap.as<uint64_t>();
  b.admit = c->ext_admit ? c->ext_admit : reinterpret_cast<uint32_t*>(at(ok, c->off_admit));
  b.ready = at(ok, c->off_ready);
  return b;
}
BatchParams batch_params(const bs_ctx* c) {
  BatchParams p{};
  p.L = c->L; p.S = c->S; p.LP = c->LP; p.C = c->C;
  p.eph_gate = c->cfg.eph_gate;
  // partitioned mode (bs_reduce_external): the caller dealt whole groups to the ranks, every loaded pod is this rank's
  p.rank = c->reduce_external ? 0u : c->rank;
  p.nranks = c->reduce_external ? 1u : c->nranks;
  p.sop_leader0 = c->sop_leader0;
  p.run_filter = 0;
  p.hash_keep = c->hash_keep;
  p.use_classes = 0;
  p.scan_slots_cap = 0;
  p.filter_slots_cap = 0;
  p.collect_stats = c->collect_stats;
  p.mcap = c->table_mcap;
  p.filter_deny = c->fd_on ? 1u : 0u;
  p.fd_iter = c->fd_iter;
  p.first_reach_hint = c->reduce_external ? c->first_reach_hint : BS_INF;
  return p;
}

// reserve and, when the buffer was (re)allocated, fill it: stamped arrays must never start out as garbage
int reserve_filled(bs_ctx* c, DevBuf& d, size_t bytes, int byte_value) {
  const void* before = d.p;
  HIPCHK(c, d.reserve(bytes));
  if (d.p != before) HIPCHK(c, hipMemsetAsync(d.p, byte_value, d.cap, c->stream));
  return BS_OK;
}

// (re)allocate table storage once classes and node capacity are known
int ensure_tables(bs_ctx* c) {
  if (!c->have_nodes || !c->have_fit) return BS_OK;
  const uint32_t slots = 2 * c->C + 1;
  const size_t bytes = (size_t)slots * c->Ncap * c->LP * sizeof(int64_t);
  if (bytes > ((size_t)64 << 30)) { c->last_error = "table storage exceeds 64 GiB"; return BS_ERR_CAPACITY; }
  HIPCHK(c, c->d_tables.reserve(bytes));
  HIPCHK(c, c->d_kp.reserve((size_t)slots * 16 * sizeof(uint32_t)));
  HIPCHK(c, c->d_chunk_tot.reserve((size_t)slots * cdiv(c->Ncap, 256) * 16 * 8));
  HIPCHK(c, c->d_gmax.reserve((size_t)slots * cdiv(c->Ncap, 64) * c->LP * 8));
  HIPCHK(c, c->d_chunk_kp.reserve((size_t)slots * cdiv(c->Ncap, 256) * 16 * 4));
  HIPCHK(c, c->d_desc.reserve((size_t)slots * sizeof(TableDesc)));
  HIPCHK(c, c->d_needed.reserve((size_t)(2 * c->C + 1) * 4));
  HIPCHK(c, c->d_qcount.reserve(16));
  HIPCHK(c, c->d_stats.reserve(8 * sizeof(uint64_t)));
  HIPCHK(c, c->d_sq.reserve(4096));
  c->table_slots = slots;
  c->table_mcap = c->Ncap;
  c->epochs_ready = false;           // needed[] was written for another class count
  return BS_OK;
}

// Upload the node mirror from list index `lo` on (0 = everything) and re-derive kmap / left4 from there.
int upload_nodes(bs_ctx* c, uint32_t lo = 0) {
  const uint32_t N = c->N, L = c->L;
  if (N > c->Ncap || c->Ncap == 0) {
    c->Ncap = std::max<uint32_t>(64, N + N / 4 + 64);
    c->d_tables.release();   // row capacity changed: tables are re-reserved in ensure_tables
    lo = 0;
  }
  const size_t cap = c->Ncap;
  const size_t old_cap_bytes = c->d_alloc.cap;
  HIPCHK(c, c->d_alloc.reserve(cap * L * 8));
  if (c->d_alloc.cap != old_cap_bytes) lo = 0;     // (re)allocated: nothing resident yet
  HIPCHK(c, c->d_nreq.reserve(cap * L * 8));
  HIPCHK(c, c->d_left4.reserve(cap * 4 * 8 + 128 * 8));  // + two node blocks of padding: k_fast_filter_t's scalar loads run to the end of the last 64-node block, the lean loop one group (<= 16 nodes) further
  HIPCHK(c, c->d_lglob.reserve(64));
  HIPCHK(c, c->d_apres.reserve(cap * 4));
  HIPCHK(c, c->d_rpres.reserve(cap * 4));
  HIPCHK(c, c->d_nflags.reserve(cap));
  HIPCHK(c, c->d_kmap.reserve(cap * 4));
  HIPCHK(c, c->d_m.reserve(16));
  lo = std::min(lo, N);
  const uint32_t base0 = (lo / kScanBlock) * kScanBlock;         // derive restarts at a block boundary
  const uint32_t cnt = N - base0;
  if (cnt) {
    for (uint32_t j = 0; j < L; ++j) {
      HIPCHK(c, hipMemcpyAsync(c->d_alloc.as<int64_t>() + j * cap + base0, c->h_alloc.data() + (size_t)j * N + base0, (size_t)cnt * 8, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->d_nreq.as<int64_t>() + j * cap + base0, c->h_nreq.data() + (size_t)j * N + base0, (size_t)cnt * 8, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_apres.as<uint32_t>() + base0, c->h_apres.data() + base0, (size_t)cnt * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_rpres.as<uint32_t>() + base0, c->h_rpres.data() + base0, (size_t)cnt * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_nflags.as<uint8_t>() + base0, c->h_nflags.data() + base0, (size_t)cnt, hipMemcpyHostToDevice, c->stream));
  }
  // rows contributed by the unchanged nodes [0, base0): kmap is increasing, so a lower bound finds them
  uint32_t m_before = 0;
  if (base0) m_before = (uint32_t)(std::lower_bound(c->h_kmap.begin(), c->h_kmap.end(), base0) - c->h_kmap.begin());
  NodesDev nd = nodes_dev(c);
  hipLaunchKernelGGL(k_nodes_derive, dim3(1), dim3(kScanBlock), 0, c->stream, nd, c->d_kmap.as<uint32_t>(), c->d_m.as<uint32_t>(), c->d_left4.as<int64_t>(),
                     c->d_lglob.as<int64_t>(), base0, m_before);
  HIPCHK(c, hipGetLastError());
  uint32_t m = 0;
  HIPCHK(c, hipMemcpyAsync(&m, c->d_m.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->M = m;
  c->h_kmap.resize(m);
  if (m > m_before) HIPCHK(c, hipMemcpy(c->h_kmap.data() + m_before, c->d_kmap.as<uint32_t>() + m_before, (size_t)(m - m_before) * 4, hipMemcpyDeviceToHost));
  c->have_nodes = true;
  return ensure_tables(c);
}

int upload_fit(bs_ctx* c) {
  HIPCHK(c, c->d_fit.reserve(std::max<size_t>(4, c->h_fit.size() * 4)));
  if (!c->h_fit.empty())
    HIPCHK(c, hipMemcpyAsync(c->d_fit.p, c->h_fit.data(), c->h_fit.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_fit = true;
  return ensure_tables(c);
}

template <int S>
void launch_scan_s(bs_ctx* c, dim3 grid, const BatchDev& b, const BatchParams& p, uint32_t m, uint32_t nseg, uint32_t nslots, uint32_t ng, uint32_t ts,
                   bool local) {
  if (local) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<S, true>), grid, dim3(256), 0, c->stream, b, p, m, nseg, nslots, ng, ts);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<S, false>), grid, dim3(256), 0, c->stream, b, p, m, nseg, nslots, ng, ts);
}
// local: the tables are chunk-local (k_tables_local_nofix); otherwise final (k_tables_fix ran)
void launch_scan(bs_ctx* c, dim3 grid, const BatchDev& b, const BatchParams& p, uint32_t m, uint32_t nseg, uint32_t nslots, uint32_t ng, uint32_t ts,
                 bool local = false) {
  switch (c->S) {
    case 0: launch_scan_s<0>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 1: launch_scan_s<1>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 2: launch_scan_s<2>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 3: launch_scan_s<3>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 4: launch_scan_s<4>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 5: launch_scan_s<5>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 6: launch_scan_s<6>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 7: launch_scan_s<7>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 8: launch_scan_s<8>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 9: launch_scan_s<9>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 10: launch_scan_s<10>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    case 11: launch_scan_s<11>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
    default: launch_scan_s<12>(c, grid, b, p, m, nseg, nslots, ng, ts, local); break;
  }
}

void launch_tables_local(bs_ctx* c, hipStream_t st, dim3 grid, const NodesDev& nd, const BatchDev& b, const BatchParams& p, const TableDesc* forced) {
  const dim3 tb(kTblChunk);
  switch (c->S <= 4 ? (int)c->S : -1) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<0>), grid, tb, 0, st, nd, b, p, forced); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<1>), grid, tb, 0, st, nd, b, p, forced); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<2>), grid, tb, 0, st, nd, b, p, forced); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<3>), grid, tb, 0, st, nd, b, p, forced); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<4>), grid, tb, 0, st, nd, b, p, forced); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local<-1>), grid, tb, 0, st, nd, b, p, forced); break;
  }
}

template <int S>
void launch_scan_filter_s(bs_ctx* c, uint32_t scan_blocks, uint32_t filter_blocks, const PodsDev& pd, const NodesDev& nd, const BatchDev& b,
                          const BatchParams& p, uint32_t m, uint32_t jcap, uint32_t nslots, uint32_t ng, uint32_t ts, bool local) {
  if (local)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_filter<S, true>), dim3(scan_blocks + filter_blocks), dim3(256), 0, c->stream, pd, nd, b, p, m, jcap, nslots,
                       ng, ts, scan_blocks, c->filter_waves, c->filter_slots_cap);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_filter<S, false>), dim3(scan_blocks + filter_blocks), dim3(256), 0, c->stream, pd, nd, b, p, m, jcap, nslots,
                       ng, ts, scan_blocks, c->filter_waves, c->filter_slots_cap);
}
void launch_scan_filter(bs_ctx* c, uint32_t scan_blocks, uint32_t filter_blocks, const PodsDev& pd, const NodesDev& nd, const BatchDev& b,
                        const BatchParams& p, uint32_t m, uint32_t jcap, uint32_t nslots, uint32_t ng, uint32_t ts, bool local) {
  switch (c->S) {
    case 0: launch_scan_filter_s<0>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 1: launch_scan_filter_s<1>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 2: launch_scan_filter_s<2>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 3: launch_scan_filter_s<3>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 4: launch_scan_filter_s<4>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 5: launch_scan_filter_s<5>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 6: launch_scan_filter_s<6>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 7: launch_scan_filter_s<7>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 8: launch_scan_filter_s<8>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 9: launch_scan_filter_s<9>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 10: launch_scan_filter_s<10>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    case 11: launch_scan_filter_s<11>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
    default: launch_scan_filter_s<12>(c, scan_blocks, filter_blocks, pd, nd, b, p, m, jcap, nslots, ng, ts, local); break;
  }
}

void launch_tables_nofix(bs_ctx* c, dim3 grid, const NodesDev& nd, const BatchDev& b, const BatchParams& p, uint32_t nchunks) {
  const dim3 tb(kTblChunk);
  const uint32_t cs = cdiv(c->Ncap, 256), gs = cdiv(c->Ncap, 64);
  switch (c->S <= 4 ? (int)c->S : -1) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<0>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<1>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<2>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<3>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<4>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables_local_nofix<-1>), grid, tb, 0, c->stream, nd, b, p, nchunks, cs, gs); break;
  }
}

static FastLaunch fast_launch(const bs_ctx* c) {
  FastLaunch f{c->stream, c->S, c->M, c->P, c->filter_waves, c->filter_slots_cap, c->tp_filter, c->cfg.device};
  const uint32_t mul = c->tp_split ? c->tp_split : ((!c->reduce_external && c->nranks > 1u) ? 2u : 1u);
  if (mul > 1u) f.filter_split = (uint32_t)std::min<uint64_t>((uint64_t)c->filter_waves * mul, 1u << 22);
  return f;
}
static int fused_residency(bs_ctx* c) {
  if (c->fused_blocks_resident < 0) c->fused_blocks_resident = fused_residency_query(fast_launch(c));
  return c->fused_blocks_resident;
}
// Filter: the distinct requests (slots) against every node; fixed grid, the kernel splits the work itself.
void launch_filter(bs_ctx* c, hipStream_t st, const PodsDev& pd, const NodesDev& nd, const BatchDev& b, bool use_classes) {
  const uint32_t W = cdiv(c->N, 64), ptiles = cdiv(c->P, 64);
  if (!ptiles || !W) return;
  const uint32_t waves = std::min<uint32_t>(c->filter_waves, 2 * ptiles * std::max<uint32_t>(1, cdiv(W, 2)));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_filter<2>), dim3(cdiv(waves, 4)), dim3(256), 0, st, pd, nd, b, c->filter_waves, use_classes ? 1u : 0u,
                     c->filter_slots_cap, c->collect_stats);
}

// Cap on the shares (blocks of four waves, a quarter of every group's rows each) that deal the live 64-row groups of one tile
// of class slots among themselves.  Measured (tools/share_sweep.sh, cfg3, step in us at 8 / 16 / 32 / 64 shares): tail 23.7 / 24.2 /
// 23.8 / 24.2, busy 23.3 / 23.4 / 23.7 / 23.8, warm 23.5 / 24.2 / 27.4 / 28.6 — a share that has no group of its own still pays
// the launch's first round trip, and with many live groups the shares re-read each other's results: eight is enough.
uint32_t pick_scan_share(const bs_ctx* c) { return c->scan_share_override ? c->scan_share_override : 8u; }

// Build the running-sum table for (cls, pct) in the scratch slot (last one) — single queries.
int build_scratch_table(bs_ctx* c, uint32_t cls, float pct, uint32_t* slot_out) {
  if (!c->have_nodes || !c->have_fit) { c->last_error = "nodes / fit not loaded"; return BS_ERR_STATE; }
  if (cls >= c->C) { c->last_error = "class out of range"; return BS_ERR_INVALID; }
  const uint32_t slot = c->table_slots - 1;
  BatchDev b = batch_dev(c);
  BatchParams p = batch_params(c);
  TableDesc d{cls, pct};
  HIPCHK(c, hipMemcpyAsync(b.desc + slot, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
  const uint32_t nchunks = std::max<uint32_t>(1, cdiv(c->M, 256));
  // one slot: shift the table / kp / desc / chunk-total bases so that blockIdx.x 0 == slot
  BatchDev b2 = b;
  b2.tables = b.tables + (size_t)slot * p.mcap * p.LP;
  b2.kp = b.kp + (size_t)slot * 16;
  b2.desc = b.desc + slot;
  b2.chunk_tot = b.chunk_tot + (size_t)slot * cdiv(c->Ncap, 256) * 16;
  b2.gmax = b.gmax + (size_t)slot * cdiv(c->Ncap, 64) * p.LP;
  b2.chunk_kp = b.chunk_kp + (size_t)slot * cdiv(c->Ncap, 256) * 16;
  const TableDesc* forced = b.desc + slot;
  launch_tables_local(c, c->stream, dim3(1, nchunks), nodes_dev(c), b2, p, forced);
  if (nchunks > 1) hipLaunchKernelGGL(k_tables_fix, dim3(1, nchunks - 1), dim3(kTblChunk), 0, c->stream, nodes_dev(c), b2, p, forced);
  HIPCHK(c, hipGetLastError());
  *slot_out = slot;
  return BS_OK;
}

// After the group state (or the fit classes) changed: re-arm the general chain's scratch and run findMaxPG for
// the loaded state (k_leader_info).  For the no-capture case it also decides, on the device, which single table
// every reservation query of a batch will use — leader with matched > 0 means every other group's pods reserve
// for it at percent 0.7 against the leader's fit class (core.go:157-161) — and writes that table's descriptor.
// Leader, panic flag and table id come back through pinned memory; nothing waits here (resolve_groups does, at
// the next bs_batch_run, and then only if the copy has not landed yet).
int analyse_groups(bs_ctx* c, bool rearm_scratch = true, const bs_group_delta* deltas = nullptr, uint32_t ndeltas = 0, bool defer = false) {
  if (c->groups_launch_pending) { int rc = flush_groups(c); if (rc) return rc; }     // an earlier patch is still waiting: it goes first
  c->steady_table = -1;
  c->side_ready = false;
  c->info_pending = false;
  if (!c->G) { c->scratch_armed = false; return BS_OK; }
  GroupsDev gr = groups_dev(c);
  BatchDev b = batch_dev(c);
  if (rearm_scratch) {
    hipLaunchKernelGGL(k_init, dim3(cdiv(std::max<uint32_t>(c->G, 8), 256)), dim3(256), 0, c->stream, gr, b);
    c->scratch_armed = true;
  }
  DeltaPack dp;
  dp.n = ndeltas;
  static_assert(sizeof(bs_group_delta) == sizeof(GroupDelta), "delta layout");
  if (ndeltas) std::memcpy(dp.d, deltas, (size_t)ndeltas * sizeof(GroupDelta));
  c->info_tag++;
  c->info_pending = true;
  c->epochs_ready = false;
  if (defer) {                                       // the launch is left to whoever touches the stream next (k_pods_apply takes it along)
    c->pending_dp = dp;
    c->groups_launch_pending = true;
    return BS_OK;
  }
  hipLaunchKernelGGL(k_leader_info, dim3(1), dim3(kLeaderBlock), 0, c->stream, gr, b, (c->have_fit && c->have_nodes) ? c->C : 0u, c->info_tag, c->h_info,
                     dp, const_cast<uint32_t*>(gr.matched), const_cast<uint32_t*>(gr.status_scheduled), const_cast<uint8_t*>(gr.flags));
  LAUNCHCHK(c, BS_KERNEL_LEADER);
  return BS_OK;
}

// the deferred group patch as its own launch
int flush_groups(bs_ctx* c) {
  if (!c->groups_launch_pending) return BS_OK;
  c->groups_launch_pending = false;
  GroupsDev gr = groups_dev(c);
  BatchDev b = batch_dev(c);
  hipLaunchKernelGGL(k_leader_info, dim3(1), dim3(kLeaderBlock), 0, c->stream, gr, b, (c->have_fit && c->have_nodes) ? c->C : 0u, c->info_tag, c->h_info,
                     c->pending_dp, const_cast<uint32_t*>(gr.matched), const_cast<uint32_t*>(gr.status_scheduled), const_cast<uint8_t*>(gr.flags));
  LAUNCHCHK(c, BS_KERNEL_LEADER);
  return BS_OK;
}

// Wait until the kernel that wrote h_info[tag_at] = tag has done so (it writes pinned host memory directly).  By
// the time anybody asks, the kernel has normally finished long ago and this is one load.
int wait_host_tag(bs_ctx* c, int tag_at, int32_t tag, const int32_t* base = nullptr) {
  volatile const int32_t* info = base ? base : c->h_info;
  for (uint32_t spin = 0; info[tag_at] != tag; ++spin) {
    if (spin == 2000) (void)hipStreamQuery(c->stream);                    // make sure the launch has left the host
    if (spin > 20000000u) { HIPCHK(c, hipStreamSynchronize(c->stream)); if (info[tag_at] != tag) { c->last_error = "host info tag never arrived"; return BS_ERR_HIP; } }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return BS_OK;
}

int resolve_groups(bs_ctx* c) {
  if (!c->info_pending) return BS_OK;
  int rc = wait_host_tag(c, 3, c->info_tag);
  if (rc) return rc;
  c->info_pending = false;
  c->steady_table = (c->n_uncaptured == 0 && c->h_info[2] >= 0) ? c->h_info[2] : -1;
  c->steady_prev = c->steady_table;
  return BS_OK;
}

int resolve_pods(bs_ctx* c) {
  if (!c->kinfo_pending) return BS_OK;
  int rc = wait_host_tag(c, 5, c->kinfo_tag);
  if (rc) return rc;
  c->kinfo_pending = false;
  c->h_K = (uint32_t)c->h_info[4];
  c->k_bound = c->h_K;
  return BS_OK;
}

int ensure_gstage(bs_ctx* c, size_t bytes) {
  if (c->gstage_busy) { HIPCHK(c, hipEventSynchronize(c->ev_gstage)); c->gstage_busy = false; }
  if (bytes <= c->h_gstage_cap) return BS_OK;
  if (c->h_gstage) { (void)hipHostFree(c->h_gstage); c->h_gstage = nullptr; c->h_gstage_cap = 0; }
  HIPCHK(c, hipHostMalloc(&c->h_gstage, bytes, hipHostMallocDefault));
  c->h_gstage_cap = bytes;
  return BS_OK;
}

// What a batch needs from the pods alone, derived on the device from the RESIDENT queue: request classes (k_pod_class_a),
// then dense class ids, per-group minima and (group, class) pairs (k_pod_pairs); K reaches the host through pinned memory.
// Runs behind the upload of bs_pods_load, and again whenever ids cannot be patched any more (bs_pods_apply: id space used up,
// too many inserts for the insert wave, group count changed after a compaction).
// pairs_only: classes (d_cls_rep / d_cls_id) are still those of the resident queue, only G changed.
int derive_pods(bs_ctx* c, bool pairs_only) {
  const uint32_t G = c->G, P = c->P, L = c->L;
  HIPCHK(c, c->d_gstat.reserve((size_t)5 * std::max<uint32_t>(G, 1) * 4 + 16));
  HIPCHK(c, c->d_fast_reject.reserve((size_t)std::max<uint32_t>(G, 1) * 4));
  HIPCHK(c, c->d_gcount.reserve((size_t)std::max<uint32_t>(G, 1) * 4));
  HIPCHK(c, c->d_admit64.reserve((size_t)std::max<uint32_t>(G, 1) * 8));
  uint32_t* gcount = c->d_gcount.as<uint32_t>();
  const uint32_t gcn = c->have_groups ? G : 0u;                                              // (pods before groups: counted when the groups arrive)
  c->gstat_cur = 0;
  unsigned long long* ctab = c->d_cls_slots.as<unsigned long long>();
  unsigned long long* ptab = ctab + c->cls_cap;                                            // second half: the pair table
  uint32_t* kcount = c->d_nepochs.as<uint32_t>() + 2;
  const uint32_t ngstat = c->have_groups ? 5 * G + 1 : 0u;
  if (P) {
    if (pairs_only) {
      hipLaunchKernelGGL(k_pods_prep, dim3(64), dim3(256), 0, c->stream, ptab, c->cls_cap, c->d_gstat.as<uint32_t>(), ngstat, (uint32_t*)nullptr, gcount, gcn);
    } else {
      hipLaunchKernelGGL(k_pods_prep, dim3(64), dim3(256), 0, c->stream, ctab, 2 * c->cls_cap, c->d_gstat.as<uint32_t>(), ngstat, kcount, gcount, gcn);
      hipLaunchKernelGGL(k_pod_class_a, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, pods_dev(c), ctab, c->cls_cap - 1, c->hash_keep, L,
                         c->d_cls_rep.as<uint32_t>(), c->d_blk_scratch.as<uint32_t>());
      hipLaunchKernelGGL(k_pod_class_ids, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, P, c->d_cls_rep.as<uint32_t>(), c->d_blk_scratch.as<uint32_t>(),
                         c->d_cls_id.as<uint32_t>(), kcount);
    }
    LAUNCHCHK(c, BS_KERNEL_PREPASS);
  } else {
    if (!pairs_only) HIPCHK(c, hipMemsetAsync(kcount, 0, 4, c->stream));
    if (gcn) HIPCHK(c, hipMemsetAsync(gcount, 0, (size_t)gcn * 4, c->stream));
  }
  c->kinfo_tag++;
  hipLaunchKernelGGL(k_pod_pairs, dim3(std::max<uint32_t>(1, cdiv(P, 256))), dim3(256), 0, c->stream, pods_dev(c), G, c->d_cls_rep.as<uint32_t>(),
                     c->d_cls_id.as<uint32_t>(), pclass_dev(c), ptab, c->cls_cap - 1, c->hash_keep, c->d_gstat.as<uint32_t>(),
                     ppair_dev(c), c->d_pair_next.as<unsigned long long>(), kcount, c->kinfo_tag, c->h_info, gcount);
  LAUNCHCHK(c, BS_KERNEL_PREPASS);
  c->kinfo_pending = true;
  c->k_bound = P;                                                    // a fresh derivation: no more classes than pods
  c->pairs_ready = c->have_groups;
  c->owner_ready = false;
  c->epochs_ready = false;
  c->rep_valid = true;
  c->dirs_ready = false;
  c->ids_used = P;
  return BS_OK;
}

// the pairs are missing or were built for another group count: bring them up to date before a batch / an apply
int build_pairs(bs_ctx* c) {
  if (!c->rep_valid) c->n_rederives++;
  return derive_pods(c, c->rep_valid);
}

// Scratch and per-pod arrays sized by the queue length; `n` pods must fit.  Grows with headroom: a queue that gains a few
// pods per cycle must not reallocate per cycle.
int reserve_pod_scratch(bs_ctx* c, uint32_t P) {
  int rc;
  const size_t n = std::max<uint32_t>(P, 1);
  const size_t room = n + n / 4 + 256;
  auto grow = [&](DevBuf& d, size_t per) -> hipError_t { return d.cap >= n * per ? hipSuccess : d.reserve(room * per); };
  if (c->d_first_reach.cap < (n / kTblChunk + 2) * 8 && (rc = reserve_filled(c, c->d_first_reach, (room / kTblChunk + 2) * 8, 0xFF))) return rc;   // one candidate word per pod block of launch A
  HIPCHK(c, grow(c->d_epoch, 4));
  if (c->d_epoch_group.cap < (n + 2) * 4) HIPCHK(c, c->d_epoch_group.reserve((room + 2) * 4));
  HIPCHK(c, grow(c->d_tcode, 1));
  HIPCHK(c, grow(c->d_stage, 1));
  HIPCHK(c, grow(c->d_leader_raw, 4));
  HIPCHK(c, grow(c->d_qpos, 4));
  HIPCHK(c, grow(c->d_fparams, 64));
  HIPCHK(c, grow(c->d_fflags, 4));
  HIPCHK(c, grow(c->d_cls_rep, 4));
  HIPCHK(c, grow(c->d_cls_id, 4));
  if (c->d_blk_scratch.cap < (n / 256 + 2) * 4) HIPCHK(c, c->d_blk_scratch.reserve((room / 256 + 2) * 4));
  // id space of pairs / classes: a pair's id is its representative pod at derivation time, drawn numbers follow behind
  if (c->pair_cap < (c->id_room ? n + c->id_room : 2 * n + 1024)) {
    const uint32_t cap = (uint32_t)std::min<size_t>(c->id_room ? n + c->id_room : 2 * room + 1024, 0x7FFFFFF0u);
    HIPCHK(c, c->d_pair_next.reserve((size_t)cap * 8));
    c->pair_cap = cap;
    c->dirs_ready = false;
    c->epochs_ready = false;
  }
  if ((rc = reserve_filled(c, c->d_pair_firstq, (size_t)c->pair_cap * 8, 0xFF))) return rc;   // 64-bit minima keyed by ~batch_seq: never reset, only born as "none"
  {
    uint32_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    if (cap > c->cls_cap || !c->d_cls_slots.p) {
      HIPCHK(c, c->d_cls_slots.reserve((size_t)cap * 8 * 2));  // request-class table | (group, class) pair table
      c->cls_cap = cap;
    }
  }
  return BS_OK;
}

// Device layout of everything bs_batch_read returns in its first copy: per-pod arrays | admit[G] | ready[G].
int layout_out(bs_ctx* c) {
  const size_t n = std::max<uint32_t>(c->P, 1), g = std::max<uint32_t>(c->G, 1);
  size_t o = 0;
  c->off_pf_code = o; o = align256(o + n);
  c->off_pf_first_k = o; o = align256(o + n * 4);
  c->off_pf_leader = o; o = align256(o + n * 4);
  c->off_fl_code = o; o = align256(o + n);
  c->off_fl_feasible = o; o = align256(o + n * 4);
  c->off_fl_slot = o; o = align256(o + n * 4);
  c->off_admit = o; o = align256(o + g * 4);
  c->off_ready = o; o = align256(o + g);
  c->outpack_bytes = o;
  HIPCHK(c, c->d_outpack.reserve(o));
  return BS_OK;
}

// Positional analysis of the loaded (groups, pods) for the three-launch chain of bs_epoch.hpp: per-group minima gated by
// the group flags, capture epochs, findMaxPG per epoch, leader runs, the tables the state can ask for, groups in class
// order.  Five launches behind whatever was loaded last; runs / flags come back through pinned memory.
int analyse_epochs(bs_ctx* c) {
  const uint32_t G = c->G, P = c->P;
  int rc;
  HIPCHK(c, c->d_run_of_epoch.reserve((size_t)(G + 2) * 4));
  HIPCHK(c, c->d_run_leader.reserve(64));
  HIPCHK(c, c->d_gslot.reserve((size_t)std::max<uint32_t>(G, 1) * 4));
  if ((rc = reserve_filled(c, c->d_gfirstq, (size_t)std::max<uint32_t>(G, 1) * 8, 0xFF))) return rc;
  if ((rc = reserve_filled(c, c->d_pair_firstq, (size_t)std::max<uint32_t>(c->pair_cap, 1) * 8 * 2 * kMaxRuns, 0xFF))) return rc;
  GroupsDev gr = groups_dev(c);
  PodsDev pd = pods_dev(c);
  BatchDev b = batch_dev(c);
  EpochDev ep{};
  ep.run_of_epoch = c->d_run_of_epoch.as<uint32_t>();
  ep.run_leader = c->d_run_leader.as<int32_t>();
  ep.gslot = c->d_gslot.as<uint32_t>();
  ep.gfirstq = c->d_gfirstq.as<unsigned long long>();
  hipLaunchKernelGGL(k_epoch_groups, dim3(cdiv(G, 256)), dim3(256), 0, c->stream, gr, b);
  hipLaunchKernelGGL(k_epochs2_a, dim3(cdiv(P, kScanBlock)), dim3(kScanBlock), 0, c->stream, pd, gr, b);
  hipLaunchKernelGGL(k_epochs2_b, dim3(cdiv(P, kScanBlock)), dim3(kScanBlock), 0, c->stream, pd, gr, b);
  hipLaunchKernelGGL(k_leader_scan, dim3(1), dim3(kLeaderBlock), 0, c->stream, gr, b);
  c->einfo_tag++;
  hipLaunchKernelGGL(k_epoch_views, dim3(1), dim3(kLeaderBlock), 0, c->stream, pd, gr, b, ep, c->C, c->einfo_tag, c->h_info + 8);
  LAUNCHCHK(c, BS_KERNEL_LEADER);
  c->einfo_pending = true;
  c->epochs_ready = true;
  c->scratch_armed = true;           // the per-group minima hold exactly what the general chain's pre-pass would derive again
  return BS_OK;
}

// Both sides loaded and the state positional for sure (captures or MinResources defaults possible): analyse now, so the
// first batch does not wait for it.  (A captured state whose leader has no matched pod is analysed by its first batch.)
int maybe_analyse_epochs(bs_ctx* c) {
  if (c->epochs_ready || c->no_fast || c->no_epoch) return BS_OK;
  if (!c->have_groups || !c->have_pods || !c->have_fit || !c->have_nodes || !c->pairs_ready || !c->P || !c->G) return BS_OK;
  if (c->n_uncaptured == 0 && c->n_nominres == 0) return BS_OK;
  if (c->groups_launch_pending) { int rc = flush_groups(c); if (rc) return rc; }
  return analyse_epochs(c);
}

template <int TS>
void launch_epoch_a(bs_ctx* c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchParams& prm,
                           const EpochDev& ep, uint32_t nchunks, uint32_t qb, uint32_t ntab) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_epoch_query_tables<TS>), grid, dim3(kTblChunk), 0, c->stream, pd, gr, nd, b, prm, ep, nchunks, cdiv(c->Ncap, 256),
                     cdiv(c->Ncap, 64), qb, ntab);
}
template <int S>
void launch_epoch_b(bs_ctx* c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& b, const BatchParams& prm, const EpochDev& ep,
                           uint32_t nseg, uint32_t scan_blocks, uint32_t filter_slots) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_epoch_scan_filter<S>), grid, dim3(256), 0, c->stream, pd, nd, b, prm, ep, c->M, nseg, c->G, scan_blocks,
                     c->filter_waves, c->filter_slots_cap, filter_slots);
}

int resolve_epochs(bs_ctx* c) {
  if (!c->einfo_pending) return BS_OK;
  int rc = wait_host_tag(c, 11, c->einfo_tag);
  if (rc) return rc;
  c->einfo_pending = false;
  c->h_R = (uint32_t)c->h_info[8];
  c->h_eflags = (uint32_t)c->h_info[9];
  return BS_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

// A batch launched on a guessed table (speculation) or with BS_BATCH_FILTER_DENY is only final once fd_settle has looked at it — it may
// have to run again.  Every call that changes what a batch reads (nodes, fit, groups) or that runs over the state itself (bs_seq_run)
// settles it FIRST, against the state it was launched on; its results stay readable afterwards.
static int fd_settle(bs_ctx* c);
static int settle_pending(bs_ctx* c) { return (c->fd_active || c->spec_active) ? fd_settle(c) : BS_OK; }

uint32_t bs_abi_version(void) { return BS_ABI_VERSION; }

const char* bs_strerror(int status) {
  switch (status) {
    case BS_OK: return "ok";
    case BS_ERR_INVALID: return "invalid argument";
    case BS_ERR_NO_DEVICE: return "no usable gfx950 device";
    case BS_ERR_HIP: return "HIP runtime error";
    case BS_ERR_STATE: return "call order: snapshot / groups / pods not loaded";
    case BS_ERR_CAPACITY: return "capacity exceeded";
    case BS_ERR_NOMEM: return "out of memory";
    case BS_ERR_COMM: return "RCCL error";
    case BS_ERR_RETRY: return "the last batch's results are void and the cause is repaired: run the batch again";
    default: return "unknown status";
  }
}

const char* bs_last_error(const bs_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* bs_kernel_name(uint32_t id) { return id < BS_KERNEL_COUNT ? kKernelNames[id] : "?"; }

int bs_create(const bs_config* cfg, bs_ctx** out) {
  if (!cfg || !out) return BS_ERR_INVALID;
  *out = nullptr;
  if (cfg->abi_version != BS_ABI_VERSION || cfg->scalar_lanes > BS_MAX_SCALARS) return BS_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return BS_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return BS_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BS_ERR_NO_DEVICE;   // gfx950-only code object
  bs_ctx* c = new (std::nothrow) bs_ctx();
  if (!c) return BS_ERR_NOMEM;
  c->cfg = *cfg;
  c->S = cfg->scalar_lanes;
  c->L = BS_FIXED_LANES + c->S;
  c->LP = c->S == 0 ? 4 : (c->S <= 4 ? 8 : 16);
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return BS_ERR_NO_DEVICE;
  }
  if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_query, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_filter, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return BS_ERR_NO_DEVICE;
  }
  // small counters every load / batch path touches: owned by the context from the start, so no call order
  // between bs_pods_load and bs_groups_load is implied
  if (c->d_nepochs.reserve(64) != hipSuccess || hipMemset(c->d_nepochs.p, 0, 64) != hipSuccess || c->d_ticket.reserve(kTkWords * 4) != hipSuccess ||
      hipMemset(c->d_ticket.p, 0, kTkWords * 4) != hipSuccess) {
    delete c;
    return BS_ERR_NOMEM;
  }
  if (c->d_info.reserve(64) != hipSuccess || c->d_first_reach.reserve(64) != hipSuccess || hipMemset(c->d_first_reach.p, 0xFF, 64) != hipSuccess ||
      hipHostMalloc((void**)&c->h_info, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||   // kernels write it directly
      hipEventCreateWithFlags(&c->ev_gstage, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return BS_ERR_NOMEM;
  }
  std::memset(c->h_info, 0, 64);
  c->batch_seq = 1;
  // test hooks (tests/test_gpu_soak.py): start the wrapping counters just below their wrap points
  if (const char* e = std::getenv("BS_STAMP_START")) c->stamp_ctr = (uint32_t)(std::strtoul(e, nullptr, 0) % 65535u);
  if (const char* e = std::getenv("BS_KEYSEQ_START")) { c->key_seq = (uint32_t)std::strtoul(e, nullptr, 0); if (c->key_seq == 0u || c->key_seq == 0xFFFFFFFFu) c->key_seq = 1; }
  if (const char* e = std::getenv("BS_NO_FAST")) c->no_fast = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_NO_EPOCH")) c->no_epoch = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_NO_FUSE_FILTER")) c->no_fuse_filter = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_NO_FUSE_FINAL")) c->no_fuse_final = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_TP_FILTER")) c->tp_filter = (uint32_t)std::min(8, std::max(0, std::atoi(e)));
  if (const char* e = std::getenv("BS_TEST_HANDOVER_TIMEOUT")) c->test_timeout_after = (uint32_t)std::atoi(e);
  if (const char* e = std::getenv("BS_STEP_A")) { c->step_a_on = std::atoi(e) != 0; c->step_a_form = (uint32_t)std::atoi(e); }
  if (const char* e = std::getenv("BS_STEP_SHARES")) c->step_shares = (uint32_t)std::min(32, std::max(1, std::atoi(e)));
  if (const char* e = std::getenv("BS_TP_SHARE")) c->tp_share = (uint32_t)std::min(64, std::max(1, std::atoi(e)));
  if (const char* e = std::getenv("BS_TP_FWAVES")) c->tp_fwaves = (uint32_t)std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BS_TP_SPLIT")) c->tp_split = (uint32_t)std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BS_NO_NODEW")) c->no_nodew = std::atoi(e) != 0;
  if (const char* e = std::getenv("BS_TP_TMIN")) c->tp_tmin = (uint32_t)std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BS_NO_SPECULATE")) c->no_spec = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_HOST_PROBE")) c->host_probe = std::atoi(e) ? 1u : 0u;
  if (const char* e = std::getenv("BS_HASH_SLOT_BITS")) { const int hb = std::atoi(e); c->slot_keep = hb >= 32 ? 0xFFFFFFFFu : ((1u << std::max(0, hb)) - 1u); }
  if (const char* e = std::getenv("BS_HASH_BITS")) { const int hb = std::atoi(e); c->hash_keep = hb >= 31 ? 0x7FFFFFFFu : ((1u << std::max(0, hb)) - 1u); }
  if (const char* e = std::getenv("BS_EARLY_FILTER_MIN")) { c->early_filter_min = std::strtoull(e, nullptr, 10); c->early_forced = 1; }
  if (const char* e = std::getenv("BS_SCAN_SHARE")) c->scan_share_override = (uint32_t)std::max(0, std::atoi(e));
  if (const char* e = std::getenv("BS_TARGET_WAVES")) { c->target_waves = std::max(1, std::atoi(e)); c->general_waves = c->target_waves; }
  if (const char* e = std::getenv("BS_FILTER_WAVES")) { c->filter_waves = std::max(1, std::atoi(e)); c->filter_waves_env = true; }
  if (const char* e = std::getenv("BS_SERIAL_INSERT_MAX")) c->serial_insert_max = (uint32_t)std::max(0, std::atoi(e));
  if (const char* e = std::getenv("BS_ID_ROOM")) c->id_room = (uint32_t)std::max(1, std::atoi(e));   // tests: a tiny id space forces re-derivations
  *out = c;
  return BS_OK;
}

int bs_destroy(bs_ctx* c) {
  if (!c) return BS_OK;
  if (c->host_probe && c->hp_n)
    std::fprintf(stderr, "[bs host probe] %llu fast-chain batches: to chain choice %.2f us | to launch A %.2f | launch A call %.2f | to launch B %.2f | launch B call %.2f | rest %.2f\n",
                 (unsigned long long)c->hp_n, c->hp_ns[0] / 1e3 / c->hp_n, c->hp_ns[1] / 1e3 / c->hp_n, c->hp_ns[2] / 1e3 / c->hp_n, c->hp_ns[3] / 1e3 / c->hp_n,
                 c->hp_ns[4] / 1e3 / c->hp_n, c->hp_ns[5] / 1e3 / c->hp_n);
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); }
  for (auto& e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->comm && c->rccl_destroy) (void)c->rccl_destroy((ncclComm_t)c->comm);
  if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev_gstage) (void)hipEventDestroy(c->ev_gstage);
  if (c->h_info) (void)hipHostFree(c->h_info);
  if (c->h_rstage) (void)hipHostFree(c->h_rstage);
  if (c->h_hout) (void)hipHostFree(c->h_hout);
  if (c->h_hrows) (void)hipHostFree(c->h_hrows);
  if (c->h_gstage) (void)hipHostFree(c->h_gstage);
  if (c->h_dstage) (void)hipHostFree(c->h_dstage);
  if (c->h_nstage) (void)hipHostFree(c->h_nstage);
  if (c->ev_nstage) (void)hipEventDestroy(c->ev_nstage);
  if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
  if (c->ev_query) (void)hipEventDestroy(c->ev_query);
  if (c->ev_filter) (void)hipEventDestroy(c->ev_filter);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return BS_OK;
}

int bs_nodes_load(bs_ctx* c, const bs_nodes_soa* nodes) {
  if (!c || !nodes) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  c->steady_prev = -1;                              // (a guess must name a table of the state that is being loaded, not of the one before; behind the settle, whose resolve_groups writes the field)
  const uint32_t N = nodes->n, L = c->L;
  if (N && (!nodes->allocatable || !nodes->requested || !nodes->allocatable_present || !nodes->requested_present || !nodes->flags))
    return BS_ERR_INVALID;
  if (N > 0x7FFFFFF0u) return BS_ERR_CAPACITY;
  c->N = N;
  c->h_alloc.assign(nodes->allocatable, nodes->allocatable + (size_t)L * N);
  c->h_nreq.assign(nodes->requested, nodes->requested + (size_t)L * N);
  c->h_apres.assign(nodes->allocatable_present, nodes->allocatable_present + N);
  c->h_rpres.assign(nodes->requested_present, nodes->requested_present + N);
  c->h_nflags.assign(nodes->flags, nodes->flags + N);
  c->have_fit = false;      // fit columns belong to a node list
  return upload_nodes(c);
}

int bs_nodes_count(const bs_ctx* c, uint32_t* n_out) {
  if (!c || !n_out) return BS_ERR_INVALID;
  *n_out = c->N;
  return BS_OK;
}

int bs_fit_load(bs_ctx* c, uint32_t n_classes, const uint32_t* fit_bits) {
  if (!c || n_classes == 0 || !fit_bits) return BS_ERR_INVALID;
  if (!c->have_nodes) { c->last_error = "bs_fit_load before bs_nodes_load"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  c->steady_prev = -1;                              // (a guess must name a table of the state that is being loaded, not of the one before; behind the settle, whose resolve_groups writes the field)
  c->C = n_classes;
  c->fit_words = cdiv(c->N, 32);
  c->h_fit.assign(fit_bits, fit_bits + (size_t)n_classes * c->fit_words);
  rc = upload_fit(c);
  if (rc == BS_OK && c->have_groups) rc = analyse_groups(c);
  return rc;
}

int bs_fit_build(bs_ctx* c, const bs_node_labels* nl, const bs_fit_templates* tp) {
  if (!c || !nl || !tp || tp->c == 0) return BS_ERR_INVALID;
  if (!c->have_nodes) { c->last_error = "bs_fit_build before bs_nodes_load"; return BS_ERR_STATE; }
  if (nl->n != c->N) { c->last_error = "bs_fit_build: label table size differs from the snapshot"; return BS_ERR_INVALID; }
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  c->steady_prev = -1;                              // (a guess must name a table of the state that is being loaded, not of the one before; behind the settle, whose resolve_groups writes the field)
  const uint32_t N = c->N, C = tp->c;
  const uint32_t nlab = N ? nl->label_off[N] : 0, ntaint = N ? nl->taint_off[N] : 0;
  const uint32_t nsel = tp->sel_off[C], nterm = tp->term_off[C], ntol = tp->tol_off[C];
  const uint32_t nex = tp->exprs.count, nfl = tp->fields.count;
  if ((nterm && (tp->term_expr_off[nterm] != nex || tp->term_field_off[nterm] != nfl)) || (!nterm && (nex || nfl))) {
    c->last_error = "bs_fit_build: term offsets do not cover the requirement tables";
    return BS_ERR_INVALID;
  }
  // the label keys any template mentions -> dense columns
  std::vector<uint32_t> keys(tp->sel_key, tp->sel_key + nsel);
  keys.insert(keys.end(), tp->exprs.key, tp->exprs.key + nex);
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  auto col_of = [&](uint32_t k) { return (uint32_t)(std::lower_bound(keys.begin(), keys.end(), k) - keys.begin()); };
  std::vector<uint32_t> sel_col(nsel), ex_col(nex);
  for (uint32_t i = 0; i < nsel; ++i) sel_col[i] = col_of(tp->sel_key[i]);
  for (uint32_t i = 0; i < nex; ++i) ex_col[i] = col_of(tp->exprs.key[i]);
  const uint32_t K = (uint32_t)keys.size(), Npad = (std::max<uint32_t>(N, 1) + 63u) & ~63u;

  FitArena A;
  const uint32_t zero_off[2] = {0, 0};
  const size_t o_name = A.put(nl->name, (size_t)N * 4);
  const size_t o_loff = A.put(N ? nl->label_off : zero_off, (size_t)(N + 1) * 4);
  const size_t o_lkey = A.put(nl->label_key, (size_t)nlab * 4), o_lval = A.put(nl->label_val, (size_t)nlab * 4);
  const size_t o_lint = A.put(nl->label_int, (size_t)nlab * 8), o_lok = A.put(nl->label_int_ok, nlab);
  const size_t o_toff = A.put(N ? nl->taint_off : zero_off, (size_t)(N + 1) * 4);
  const size_t o_tkey = A.put(nl->taint_key, (size_t)ntaint * 4), o_tval = A.put(nl->taint_val, (size_t)ntaint * 4);
  const size_t o_teff = A.put(nl->taint_effect, ntaint);
  const size_t o_keys = A.put(keys.data(), (size_t)K * 4);
  const size_t o_flags = A.put(tp->flags, C);
  const size_t o_soff = A.put(tp->sel_off, (size_t)(C + 1) * 4), o_scol = A.put(sel_col.data(), (size_t)nsel * 4);
  const size_t o_sval = A.put(tp->sel_val, (size_t)nsel * 4);
  const size_t o_moff = A.put(tp->term_off, (size_t)(C + 1) * 4);
  const size_t o_meo = A.put(nterm ? tp->term_expr_off : zero_off, (size_t)(nterm + 1) * 4);
  const size_t o_mfo = A.put(nterm ? tp->term_field_off : zero_off, (size_t)(nterm + 1) * 4);
  auto put_req = [&](const bs_requirements& r, const uint32_t* key, size_t (&o)[6]) {
    const uint32_t nv = r.count ? r.val_off[r.count] : 0;
    o[0] = A.put(key, (size_t)r.count * 4);
    o[1] = A.put(r.op, r.count);
    o[2] = A.put(r.count ? r.val_off : zero_off, (size_t)(r.count + 1) * 4);
    o[3] = A.put(r.val, (size_t)nv * 4);
    o[4] = A.put(r.val_int, (size_t)nv * 8);
    o[5] = A.put(r.val_int_ok, nv);
  };
  size_t o_ex[6], o_fl[6];
  put_req(tp->exprs, ex_col.data(), o_ex);
  put_req(tp->fields, tp->fields.key, o_fl);
  const size_t o_ooff = A.put(tp->tol_off, (size_t)(C + 1) * 4);
  const size_t o_okey = A.put(tp->tol_key, (size_t)ntol * 4), o_oval = A.put(tp->tol_val, (size_t)ntol * 4);
  const size_t o_oop = A.put(tp->tol_op, ntol), o_oeff = A.put(tp->tol_effect, ntol);

  HIPCHK(c, c->d_fitarena.reserve(A.bytes.size() + 16));
  HIPCHK(c, hipMemcpyAsync(c->d_fitarena.p, A.bytes.data(), A.bytes.size(), hipMemcpyHostToDevice, c->stream));
  const size_t colcells = (size_t)std::max<uint32_t>(K, 1) * Npad;
  const size_t o_cval = 0, o_cival = (colcells * 4 + 15) & ~(size_t)15, o_cflag = o_cival + colcells * 8;
  HIPCHK(c, c->d_fitcols.reserve(o_cflag + colcells));
  HIPCHK(c, hipMemsetAsync(c->d_fitcols.p, 0, o_cflag + colcells, c->stream));

  const void* B = c->d_fitarena.p;
  FitNodesDev nd{};
  nd.n = N;
  nd.name = at_dev<uint32_t>(B, o_name);
  nd.label_off = at_dev<uint32_t>(B, o_loff); nd.label_key = at_dev<uint32_t>(B, o_lkey); nd.label_val = at_dev<uint32_t>(B, o_lval);
  nd.label_int = at_dev<int64_t>(B, o_lint); nd.label_int_ok = at_dev<uint8_t>(B, o_lok);
  nd.taint_off = at_dev<uint32_t>(B, o_toff); nd.taint_key = at_dev<uint32_t>(B, o_tkey); nd.taint_val = at_dev<uint32_t>(B, o_tval);
  nd.taint_effect = at_dev<uint8_t>(B, o_teff);
  nd.nflags = c->d_nflags.as<uint8_t>();
  FitCols cols{};
  cols.K = K; cols.Npad = Npad;
  cols.keys = at_dev<uint32_t>(B, o_keys);
  cols.val = (uint32_t*)((uint8_t*)c->d_fitcols.p + o_cval);
  cols.ival = (int64_t*)((uint8_t*)c->d_fitcols.p + o_cival);
  cols.flag = (uint8_t*)c->d_fitcols.p + o_cflag;
  FitTplDev td{};
  td.c = C; td.field_name_key = tp->field_name_key;
  td.flags = at_dev<uint8_t>(B, o_flags);
  td.sel_off = at_dev<uint32_t>(B, o_soff); td.sel_col = at_dev<uint32_t>(B, o_scol); td.sel_val = at_dev<uint32_t>(B, o_sval);
  td.term_off = at_dev<uint32_t>(B, o_moff); td.term_expr_off = at_dev<uint32_t>(B, o_meo); td.term_field_off = at_dev<uint32_t>(B, o_mfo);
  auto req_dev = [&](const size_t (&o)[6]) {
    FitReqDev r{};
    r.key = at_dev<uint32_t>(B, o[0]); r.op = at_dev<uint8_t>(B, o[1]); r.val_off = at_dev<uint32_t>(B, o[2]);
    r.val = at_dev<uint32_t>(B, o[3]); r.val_int = at_dev<int64_t>(B, o[4]); r.val_int_ok = at_dev<uint8_t>(B, o[5]);
    return r;
  };
  td.ex = req_dev(o_ex); td.fl = req_dev(o_fl);
  td.tol_off = at_dev<uint32_t>(B, o_ooff); td.tol_key = at_dev<uint32_t>(B, o_okey); td.tol_val = at_dev<uint32_t>(B, o_oval);
  td.tol_op = at_dev<uint8_t>(B, o_oop); td.tol_effect = at_dev<uint8_t>(B, o_oeff);

  c->C = C;
  c->fit_words = cdiv(N, 32);
  c->h_fit.assign((size_t)C * c->fit_words, 0);
  HIPCHK(c, c->d_fit.reserve(std::max<size_t>(4, c->h_fit.size() * 4)));
  if (N) {
    const uint32_t nb = cdiv(N, 256);
    hipLaunchKernelGGL(k_fit_cols, dim3(nb), dim3(256), 0, c->stream, nd, cols);
    hipLaunchKernelGGL(k_fit_match, dim3(nb, std::min<uint32_t>(C, 1024)), dim3(256), 0, c->stream, nd, td, cols, c->d_fit.as<uint32_t>(), c->fit_words);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_fit.data(), c->d_fit.p, c->h_fit.size() * 4, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_fit = true;
  rc = ensure_tables(c);
  if (rc == BS_OK && c->have_groups) rc = analyse_groups(c);
  return rc;
}

int bs_fit_read(bs_ctx* c, uint32_t* out) {
  if (!c || !out) return BS_ERR_INVALID;
  if (!c->have_fit) { c->last_error = "bs_fit_read before a fit load / build"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  const size_t bytes = (size_t)c->C * c->fit_words * 4;
  if (bytes) HIPCHK(c, hipMemcpyAsync(out, c->d_fit.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BS_OK;
}

int bs_groups_load(bs_ctx* c, const bs_groups_soa* g) {
  if (!c || !g) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  c->steady_prev = -1;                              // (a guess must name a table of the state that is being loaded, not of the one before; behind the settle, whose resolve_groups writes the field)
  c->first_reach_hint = 0xFFFFFFFFu;                // (which pod reaches findMaxPG first depends on the groups' deny entries and OccupiedBy)
  const uint32_t G = g->g, L = c->L;
  if (G && (!g->min_member || !g->status_scheduled || !g->matched || !g->flags || !g->cls || !g->min_resources || !g->min_resources_present ||
            !g->occupied_by))
    return BS_ERR_INVALID;
  const size_t n = std::max<uint32_t>(G, 1);
  // group arrays: one allocation, one pinned-staged transfer, no wait
  size_t o = 0;
  c->off_gmm = o; o = align256(o + n * 4);
  c->off_gsc = o; o = align256(o + n * 4);
  c->off_gmatched = o; o = align256(o + n * 4);
  c->off_gflags = o; o = align256(o + n);
  c->off_gcls = o; o = align256(o + n * 4);
  c->off_gminres = o; o = align256(o + n * L * 8);
  c->off_gmrpres = o; o = align256(o + n * 4);
  c->off_gocc = o; o = align256(o + n * 8);
  c->gpack_bytes = o;
  HIPCHK(c, c->d_gpack.reserve(o));
  HIPCHK(c, c->d_first_elig.reserve(n * 4));
  HIPCHK(c, c->d_first_owner.reserve(n * 4));
  HIPCHK(c, c->d_first_reject.reserve(n * 4));
  HIPCHK(c, c->d_first_pod.reserve(n * 4));
  HIPCHK(c, c->d_cap_epoch.reserve(n * 4));
  HIPCHK(c, c->d_leader_epoch.reserve((n + 1) * 4));
  HIPCHK(c, c->d_panic_epoch.reserve(n + 1));
  if (G != c->G) { c->pairs_ready = false; c->dirs_ready = false; }   // the per-group arrays of the pod load are sized by G
  c->G = G;
  if ((rc = layout_out(c))) return rc;
  c->n_uncaptured = 0;
  c->n_nominres = 0;
  c->max_group_cls = 0;
  if (G) {
    rc = ensure_gstage(c, c->gpack_bytes);
    if (rc) return rc;
    uint8_t* st = reinterpret_cast<uint8_t*>(c->h_gstage);
    std::memcpy(st + c->off_gmm, g->min_member, (size_t)G * 4);
    std::memcpy(st + c->off_gsc, g->status_scheduled, (size_t)G * 4);
    std::memcpy(st + c->off_gmatched, g->matched, (size_t)G * 4);
    std::memcpy(st + c->off_gflags, g->flags, (size_t)G);
    std::memcpy(st + c->off_gcls, g->cls, (size_t)G * 4);
    std::memcpy(st + c->off_gminres, g->min_resources, (size_t)G * L * 8);
    std::memcpy(st + c->off_gmrpres, g->min_resources_present, (size_t)G * 4);
    std::memcpy(st + c->off_gocc, g->occupied_by, (size_t)G * 8);
    HIPCHK(c, hipMemcpyAsync(c->d_gpack.p, st, c->gpack_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_gstage, c->stream));
    c->gstage_busy = true;
    for (uint32_t i = 0; i < G; ++i) {
      if (!(g->flags[i] & BS_GROUP_HAS_POD)) c->n_uncaptured++;
      else c->max_group_cls = std::max(c->max_group_cls, g->cls[i]);
      if (!(g->flags[i] & BS_GROUP_HAS_MINRES)) c->n_nominres++;
    }
    c->h_gflags.assign(g->flags, g->flags + G);
  } else {
    c->h_gflags.clear();
  }
  c->have_groups = true;
  if ((rc = analyse_groups(c))) return rc;
  return maybe_analyse_epochs(c);
}

int bs_groups_apply(bs_ctx* c, const bs_group_delta* deltas, uint32_t count) {
  if (!c || (count && !deltas)) return BS_ERR_INVALID;
  if (!c->have_groups) { c->last_error = "bs_groups_apply before bs_groups_load"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  c->first_reach_hint = 0xFFFFFFFFu;                // (which pod reaches findMaxPG first depends on the groups' deny entries and OccupiedBy)
  if (!count) return BS_OK;
  const uint8_t keep = BS_GROUP_HAS_POD | BS_GROUP_HAS_MINRES;
  for (uint32_t d = 0; d < count; ++d) {            // validate everything before touching anything
    const bs_group_delta& x = deltas[d];
    if (x.index >= c->G || x.flags > 0xFFu) { c->last_error = "bs_groups_apply: index / flags out of range"; return BS_ERR_INVALID; }
    if (((uint8_t)x.flags & keep) != (c->h_gflags[x.index] & keep)) {
      c->last_error = "bs_groups_apply: HAS_POD / HAS_MINRES may not change (use bs_groups_load)";
      return BS_ERR_INVALID;
    }
  }
  {
    // every delta is applied by its own thread: two deltas for one group would leave whichever thread wrote last
    std::vector<uint32_t> seen(count);
    for (uint32_t d = 0; d < count; ++d) seen[d] = deltas[d].index;
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { c->last_error = "bs_groups_apply: a group index appears twice"; return BS_ERR_INVALID; }
  }
  for (uint32_t d = 0; d < count; ++d) c->h_gflags[deltas[d].index] = (uint8_t)deltas[d].flags;
  // HAS_POD is unchanged, so the capture epochs the general chain's scratch holds stay valid: no re-arm.
  // Few deltas (the per-cycle case) ride in the arguments of the findMaxPG launch: ONE launch, no copy, no wait.
  if (count <= (uint32_t)kInlineDeltas) {
    // the launch itself is deferred: a bs_pods_apply in the same cycle takes the patch along in its own launch, any other
    // call sends it first (use_device)
    if ((rc = analyse_groups(c, false, deltas, count, true))) return rc;
    return maybe_analyse_epochs(c);                  // (positional state: flushes the patch, the analysis reads the patched groups)
  }
  const size_t bytes = (size_t)count * sizeof(bs_group_delta);
  rc = ensure_gstage(c, std::max(bytes, c->gpack_bytes));
  if (rc) return rc;
  HIPCHK(c, c->d_gdelta.reserve(bytes));
  std::memcpy(c->h_gstage, deltas, bytes);
  HIPCHK(c, hipMemcpyAsync(c->d_gdelta.p, c->h_gstage, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_gstage, c->stream));
  c->gstage_busy = true;
  GroupsDev gr = groups_dev(c);
  hipLaunchKernelGGL(k_groups_apply, dim3(cdiv(count, 256)), dim3(256), 0, c->stream, c->d_gdelta.as<GroupDelta>(), count,
                     const_cast<uint32_t*>(gr.matched), const_cast<uint32_t*>(gr.status_scheduled), const_cast<uint8_t*>(gr.flags));
  LAUNCHCHK(c, BS_KERNEL_LEADER);
  if ((rc = analyse_groups(c, false))) return rc;
  return maybe_analyse_epochs(c);
}

int bs_groups_read(bs_ctx* c, bs_groups_soa* g) {
  if (!c || !g) return BS_ERR_INVALID;
  if (!c->have_groups) return BS_ERR_STATE;
  if (g->g != c->G) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t G = c->G, L = c->L;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (G) {
    const uint8_t* pk = c->d_gpack.as<uint8_t>();
    HIPCHK(c, hipMemcpy(g->min_member, pk + c->off_gmm, (size_t)G * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->status_scheduled, pk + c->off_gsc, (size_t)G * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->matched, pk + c->off_gmatched, (size_t)G * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->flags, pk + c->off_gflags, (size_t)G, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->cls, pk + c->off_gcls, (size_t)G * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->min_resources, pk + c->off_gminres, (size_t)G * L * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->min_resources_present, pk + c->off_gmrpres, (size_t)G * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g->occupied_by, pk + c->off_gocc, (size_t)G * 8, hipMemcpyDeviceToHost));
  }
  return BS_OK;
}

static int ensure_stage(bs_ctx* c, size_t bytes) {
  // the previous upload may still be reading the staging buffer
  if (c->stage_busy) { HIPCHK(c, hipEventSynchronize(c->ev_stage)); c->stage_busy = false; }
  if (bytes <= c->h_stage_cap) return BS_OK;
  if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
  HIPCHK(c, hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault));
  c->h_stage_cap = bytes;
  return BS_OK;
}
int bs_pods_map(bs_ctx* c, uint32_t p, bs_pods_soa* view) {
  if (!c || !view || !p) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  const PodLayout l = pod_layout(p, c->L);           // the staging buffer's own layout: the resident queue is not touched
  rc = ensure_stage(c, l.in_bytes);                  // waits until the previous upload has left the buffer
  if (rc) return rc;
  c->stage_lay = l;
  uint8_t* st = reinterpret_cast<uint8_t*>(c->h_stage);
  view->p = p;
  view->group = reinterpret_cast<const int32_t*>(st + l.group);
  view->req = reinterpret_cast<const int64_t*>(st + l.req);
  view->req_present = reinterpret_cast<const uint32_t*>(st + l.pres);
  view->cls = reinterpret_cast<const uint32_t*>(st + l.cls);
  view->owner = reinterpret_cast<const uint64_t*>(st + l.owner);
  view->flags = st + l.flags;
  c->map_p = p;
  return BS_OK;
}

// after the resident queue changed length: result pack layout, scratch
static int resize_queue(bs_ctx* c, uint32_t P) {
  int rc;
  c->P = P;
  if ((rc = layout_out(c))) return rc;
  return reserve_pod_scratch(c, P);
}

int bs_pods_load(bs_ctx* c, const bs_pods_soa* pods) {
  if (!c || !pods) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t P = pods->p, L = c->L;
  if (P && (!pods->group || !pods->req || !pods->req_present || !pods->cls || !pods->owner || !pods->flags)) return BS_ERR_INVALID;
  const PodLayout l = pod_layout(P, L);
  // the caller's arrays ARE the mapped staging buffer (bs_pods_map): nothing to pack.  Pointers into the staging buffer that
  // do not match the mapping (another p, a stale view) would make the packing copy overlap itself: refused.
  const uint8_t* sb = reinterpret_cast<const uint8_t*>(c->h_stage);
  const bool inside = P && sb && (const uint8_t*)pods->group >= sb && (const uint8_t*)pods->group < sb + c->h_stage_cap;
  const bool mapped = inside && c->map_p == P && (const uint8_t*)pods->group == sb + c->stage_lay.group && (const uint8_t*)pods->req == sb + c->stage_lay.req &&
                      (const uint8_t*)pods->req_present == sb + c->stage_lay.pres && (const uint8_t*)pods->cls == sb + c->stage_lay.cls &&
                      (const uint8_t*)pods->owner == sb + c->stage_lay.owner && pods->flags == sb + c->stage_lay.flags;
  if (inside && !mapped) { c->last_error = "bs_pods_load: pointers into the mapped staging buffer, but not the view bs_pods_map handed out for this p"; return BS_ERR_INVALID; }
  if (!mapped) {
    rc = ensure_stage(c, l.in_bytes);
    if (rc) return rc;
  }
  c->map_p = 0;
  // pod arrays: one allocation, one transfer (the load always lands in pack 0)
  c->cur_pack = 0;
  if (c->d_pack[0].cap < l.bytes) HIPCHK(c, c->d_pack[0].reserve(l.bytes + l.bytes / 4));
  c->lay[0] = l;
  // per-pod outputs (+ the per-group ones): one allocation, one transfer back
  if ((rc = resize_queue(c, P))) return rc;
  c->pairs_ready = false;
  c->epochs_ready = false;
  c->batch_since_pods = false;
  c->first_reach_hint = 0xFFFFFFFFu;
  c->max_pod_cls = 0;
  for (uint32_t i = 0; i < P; ++i)
    if (pods->group[i] >= 0) c->max_pod_cls = std::max(c->max_pod_cls, pods->cls[i]);
  if (P) {
    uint8_t* st = reinterpret_cast<uint8_t*>(c->h_stage);
    if (!mapped) {                                   // (bs_pods_map: the caller marshalled the queue in place)
      std::memcpy(st + l.group, pods->group, (size_t)P * 4);
      std::memcpy(st + l.req, pods->req, (size_t)P * L * 8);
      std::memcpy(st + l.pres, pods->req_present, (size_t)P * 4);
      std::memcpy(st + l.cls, pods->cls, (size_t)P * 4);
      std::memcpy(st + l.owner, pods->owner, (size_t)P * 8);
      std::memcpy(st + l.flags, pods->flags, (size_t)P);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_pack[0].p, st, l.in_bytes, hipMemcpyHostToDevice, c->stream));
  }
  // request classes, per-group minima and (group, class) pairs of the pods: three launches behind the upload
  // (reset | first half of the class builder | second half + pairs); K reaches the host through pinned memory
  rc = derive_pods(c, false);
  if (rc) return rc;
  // no wait here: the batch that follows is ordered behind the upload on the same stream; only the staging
  // buffer must not be touched again before the copy has left it (ensure_stage / bs_batch_read wait for that)
  if (!c->ev_stage) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_stage, c->stream));
  c->stage_busy = true;
  c->have_pods = true;
  return maybe_analyse_epochs(c);
}

// ---- bs_pods_apply: the resident queue patched on the device (bs_queue.hpp) ----------------------------------------------
static int ensure_dstage(bs_ctx* c, size_t bytes) {
  // the previous apply may still be reading the blob.  No event per apply (a record costs the cycle a microsecond of host time):
  // the flag is cleared whenever the host has seen something later on the stream complete (the batch's completion word, a
  // stream wait); two applies without that in between wait for the stream here.
  if (c->dstage_busy) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->dstage_busy = false; }
  if (bytes <= c->h_dstage_cap) return BS_OK;
  if (c->h_dstage) { (void)hipHostFree(c->h_dstage); c->h_dstage = nullptr; c->h_dstage_cap = 0; }
  const size_t want = std::max<size_t>(bytes + bytes / 2, 64 << 10);
  HIPCHK(c, hipHostMalloc(&c->h_dstage, want, hipHostMallocDefault));
  c->h_dstage_cap = want;
  return BS_OK;
}

static QueueDirs queue_dirs(const bs_ctx* c) {
  QueueDirs q{};
  q.slot_keep = c->slot_keep;
  q.cdir = c->d_cdir.as<unsigned long long>();
  q.pdir = c->d_pdir.as<unsigned long long>();
  q.cmask = q.pmask = c->dir_slots ? c->dir_slots - 1 : 0;
  q.ckeys = c->d_ckeys.as<int64_t>();
  q.cpres = c->d_cpres.as<uint32_t>();
  q.kcap = c->pair_cap;
  q.pkeys = c->d_pkeys.as<unsigned long long>();
  q.kcount = c->d_nepochs.as<uint32_t>() + 2;
  q.paircount = c->d_nepochs.as<uint32_t>() + 4;
  q.pair_head = reinterpret_cast<unsigned long long*>(c->d_gstat.as<uint32_t>() + (((size_t)3 * c->G + 1) & ~(size_t)1));
  q.pair_next = c->d_pair_next.as<unsigned long long>();
  q.pcap = c->pair_cap;
  q.overflow = c->h_info ? c->h_info + 13 : nullptr;
  return q;
}

// Directories (hash -> class id, hash -> pair id) of a queue whose ids still name its own pods: the first apply after a
// derivation files every class / pair representative; later applies only add to them.
static int build_dirs(bs_ctx* c) {
  const uint32_t P = c->P, G = c->G;
  uint32_t slots = 2048;
  while (slots < 2 * c->pair_cap) slots <<= 1;
  HIPCHK(c, c->d_cdir.reserve((size_t)slots * 8));
  HIPCHK(c, c->d_pdir.reserve((size_t)slots * 8));
  HIPCHK(c, c->d_ckeys.reserve((size_t)c->pair_cap * c->L * 8));
  HIPCHK(c, c->d_cpres.reserve((size_t)c->pair_cap * 4));
  HIPCHK(c, c->d_pkeys.reserve((size_t)c->pair_cap * 8));
  HIPCHK(c, c->d_gstat2.reserve((size_t)3 * std::max<uint32_t>(G, 1) * 4 + 16));
  c->dir_slots = slots;
  HIPCHK(c, hipMemsetAsync(c->d_cdir.p, 0, (size_t)slots * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_pdir.p, 0, (size_t)slots * 8, c->stream));
  HIPCHK(c, hipMemsetAsync((c->gstat_cur ? c->d_gstat : c->d_gstat2).p, 0xFF, (size_t)3 * std::max<uint32_t>(G, 1) * 4, c->stream));   // the spare minima
  const uint32_t pc = P;                                                                    // pair ids below P belong to the derivation
  HIPCHK(c, hipMemcpyAsync(c->d_nepochs.as<uint32_t>() + 4, &pc, 4, hipMemcpyHostToDevice, c->stream));
  if (P) {
    hipLaunchKernelGGL(k_dirs_build, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, pods_dev(c), G, c->L, c->d_cls_rep.as<uint32_t>(), c->d_cls_id.as<uint32_t>(),
                       ppair_dev(c), queue_dirs(c), c->hash_keep);
    LAUNCHCHK(c, BS_KERNEL_PREPASS);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));      // (`pc` is a stack word; once per derivation)
  c->dirs_ready = true;
  return BS_OK;
}

int bs_pods_apply_stats(const bs_ctx* c, uint64_t* applies, uint64_t* rederives) {
  if (!c) return BS_ERR_INVALID;
  if (applies) *applies = c->n_applies;
  if (rederives) *rederives = c->n_rederives;
  return BS_OK;
}

int bs_pods_count(const bs_ctx* c, uint32_t* p_out) {
  if (!c || !p_out) return BS_ERR_INVALID;
  if (!c->have_pods) return BS_ERR_STATE;
  *p_out = c->P;
  return BS_OK;
}

int bs_pods_read(bs_ctx* c, const bs_pods_out* out) {
  if (!c || !out) return BS_ERR_INVALID;
  if (!c->have_pods) return BS_ERR_STATE;
  if (out->p != c->P) { c->last_error = "bs_pods_read: out->p does not match the resident queue"; return BS_ERR_INVALID; }
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t P = c->P, L = c->L;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!P) return BS_OK;
  const PodsDev pd = pods_dev(c);
  if (out->group) HIPCHK(c, hipMemcpy(out->group, pd.group, (size_t)P * 4, hipMemcpyDeviceToHost));
  if (out->req) HIPCHK(c, hipMemcpy(out->req, pd.req, (size_t)P * L * 8, hipMemcpyDeviceToHost));
  if (out->req_present) HIPCHK(c, hipMemcpy(out->req_present, pd.pres, (size_t)P * 4, hipMemcpyDeviceToHost));
  if (out->cls) HIPCHK(c, hipMemcpy(out->cls, pd.cls, (size_t)P * 4, hipMemcpyDeviceToHost));
  if (out->owner) HIPCHK(c, hipMemcpy(out->owner, pd.owner, (size_t)P * 8, hipMemcpyDeviceToHost));
  if (out->flags) HIPCHK(c, hipMemcpy(out->flags, pd.flags, (size_t)P, hipMemcpyDeviceToHost));
  return BS_OK;
}

int bs_pods_apply(bs_ctx* c, const bs_pods_delta* d) {
  if (!c || !d) return BS_ERR_INVALID;
  if (!c->have_pods) { c->last_error = "bs_pods_apply before bs_pods_load"; return BS_ERR_STATE; }
  int rc = use_device(c, false);                     // a deferred group patch rides in this call's launch when it can
  if (rc) return rc;
  const uint32_t P = c->P, L = c->L, R = d->n_remove, F = d->n_flags, I = d->insert.p;
  // ---- validate everything before touching anything
  if ((R && !d->remove) || (F && (!d->flag_index || !d->flag_value))) return BS_ERR_INVALID;
  const bs_pods_soa& in = d->insert;
  if (I && (!in.group || !in.req || !in.req_present || !in.cls || !in.owner || !in.flags)) return BS_ERR_INVALID;
  if (R > P) { c->last_error = "bs_pods_apply: more removals than pods"; return BS_ERR_INVALID; }
  for (uint32_t i = 0; i < R; ++i)
    if (d->remove[i] >= P || (i && d->remove[i] <= d->remove[i - 1])) { c->last_error = "bs_pods_apply: remove[] must be strictly ascending indices of the current queue"; return BS_ERR_INVALID; }
  for (uint32_t i = 0; i < F; ++i)
    if (d->flag_index[i] >= P || (i && d->flag_index[i] <= d->flag_index[i - 1])) { c->last_error = "bs_pods_apply: flag_index[] must be strictly ascending indices of the current queue"; return BS_ERR_INVALID; }
  if ((uint64_t)P - R + I > 0x7FFFFFF0ull) { c->last_error = "bs_pods_apply: queue too long"; return BS_ERR_CAPACITY; }
  const uint32_t Pn = P - R + I;
  if (d->insert_at)
    for (uint32_t i = 0; i < I; ++i)
      if (d->insert_at[i] >= Pn || (i && d->insert_at[i] <= d->insert_at[i - 1])) { c->last_error = "bs_pods_apply: insert_at[] must be strictly ascending positions of the new queue"; return BS_ERR_INVALID; }
  if (!R && !F && !I) return BS_OK;
  if (c->batch_pending_finish) { c->last_error = "bs_pods_apply between a sharded batch and bs_batch_finish"; return BS_ERR_STATE; }
  c->n_applies++;

  // ---- the delta goes into pinned memory as ONE tightly packed blob the kernel stages into LDS with one bulk read per block:
  // [remove | insert_at | flag_index | flag_value] (what every gather block needs), then the inserted pods (SoA)
  auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t o = 0;
  const size_t o_rem = o; o = a16(o + (size_t)R * 4);
  const size_t o_at = o; o = a16(o + (size_t)I * 4);
  const size_t o_fi = o; o = a16(o + (size_t)F * 4);
  const size_t o_fv = o; o = a16(o + (size_t)F);
  const size_t lists_bytes = o;
  const size_t i_group = o; o = a16(o + (size_t)I * 4);
  const size_t i_req = o; o = a16(o + (size_t)I * L * 8);
  const size_t i_pres = o; o = a16(o + (size_t)I * 4);
  const size_t i_cls = o; o = a16(o + (size_t)I * 4);
  const size_t i_owner = o; o = a16(o + (size_t)I * 8);
  const size_t i_flags = o; o = a16(o + (size_t)I);
  if ((rc = ensure_dstage(c, o + 16))) return rc;
  uint8_t* st = reinterpret_cast<uint8_t*>(c->h_dstage);
  if (R) std::memcpy(st + o_rem, d->remove, (size_t)R * 4);
  if (I) {
    uint32_t* at = reinterpret_cast<uint32_t*>(st + o_at);
    if (d->insert_at) std::memcpy(at, d->insert_at, (size_t)I * 4);
    else for (uint32_t i = 0; i < I; ++i) at[i] = P - R + i;
    std::memcpy(st + i_group, in.group, (size_t)I * 4);
    std::memcpy(st + i_req, in.req, (size_t)I * L * 8);
    std::memcpy(st + i_pres, in.req_present, (size_t)I * 4);
    std::memcpy(st + i_cls, in.cls, (size_t)I * 4);
    std::memcpy(st + i_owner, in.owner, (size_t)I * 8);
    std::memcpy(st + i_flags, in.flags, (size_t)I);
    for (uint32_t i = 0; i < I; ++i)
      if (in.group[i] >= 0) c->max_pod_cls = std::max(c->max_pod_cls, in.cls[i]);
  }
  if (F) { std::memcpy(st + o_fi, d->flag_index, (size_t)F * 4); std::memcpy(st + o_fv, d->flag_value, (size_t)F); }

  // ---- ids can be patched when the pairs are current, the directories can be had, the id space has room and the insert
  // wave is not asked to do a parallel job; otherwise: copy only, then derive everything from the new resident queue
  if ((!c->pairs_ready || !c->dirs_ready) && c->groups_launch_pending && (rc = flush_groups(c))) return rc;   // other launches come first: the patch too
  if (!c->pairs_ready && c->have_groups && c->rep_valid && (rc = build_pairs(c))) return rc;       // (G changed since the pods were derived)
  bool derive = c->pairs_ready && c->have_groups && I <= c->serial_insert_max && (c->dirs_ready || c->rep_valid);
  const uint32_t old_pair_cap = c->pair_cap;
  const PodsDev old = pods_dev(c);
  const uint32_t* old_pclass = pclass_dev(c);
  const uint32_t* old_ppair = ppair_dev(c);
  if (derive && (uint64_t)c->ids_used + I > c->pair_cap) derive = false;
  if (derive && c->ids_used > 2 * (P - R + I) + 1024) derive = false;      // most ids name classes whose pods are gone: start over
  if (derive && !c->dirs_ready && (rc = build_dirs(c))) return rc;
  // ---- the other pack takes the new queue
  const uint32_t np = c->cur_pack ^ 1u;
  const PodLayout nl = pod_layout(Pn, L);
  if (c->d_pack[np].cap < nl.bytes) HIPCHK(c, c->d_pack[np].reserve(nl.bytes + nl.bytes / 4));
  c->lay[np] = nl;
  // from here to the launch the context describes the NEW queue length; a failure on the way puts the old one back
  struct Restore {
    bs_ctx* c; uint32_t P; bool armed;
    ~Restore() { if (armed) { c->P = P; (void)layout_out(c); } }
  } restore{c, P, true};
  if ((rc = resize_queue(c, Pn))) return rc;
  if (c->pair_cap != old_pair_cap) derive = false;                   // the id space was re-sized: directories are gone
  PodsMut nw{};
  uint8_t* nb = c->d_pack[np].as<uint8_t>();
  nw.group = reinterpret_cast<int32_t*>(nb + nl.group);
  nw.req = reinterpret_cast<int64_t*>(nb + nl.req);
  nw.pres = reinterpret_cast<uint32_t*>(nb + nl.pres);
  nw.cls = reinterpret_cast<uint32_t*>(nb + nl.cls);
  nw.owner = reinterpret_cast<uint64_t*>(nb + nl.owner);
  nw.flags = nb + nl.flags;
  nw.pclass = reinterpret_cast<uint32_t*>(nb + nl.pclass);
  nw.ppair = reinterpret_cast<uint32_t*>(nb + nl.ppair);
  nw.p = Pn;
  PodDeltaDev dd{};
  dd.n_remove = R; dd.n_insert = I; dd.n_flags = F;
  dd.remove = reinterpret_cast<const uint32_t*>(st + o_rem);
  dd.insert_at = reinterpret_cast<const uint32_t*>(st + o_at);
  dd.flag_index = reinterpret_cast<const uint32_t*>(st + o_fi);
  dd.flag_value = st + o_fv;
  dd.ins.p = I;
  dd.ins.group = reinterpret_cast<const int32_t*>(st + i_group);
  dd.ins.req = reinterpret_cast<const int64_t*>(st + i_req);
  dd.ins.pres = reinterpret_cast<const uint32_t*>(st + i_pres);
  dd.ins.cls = reinterpret_cast<const uint32_t*>(st + i_cls);
  dd.ins.owner = reinterpret_cast<const uint64_t*>(st + i_owner);
  dd.ins.flags = st + i_flags;
  dd.blob = st;
  dd.blob_bytes = (uint32_t)std::min<size_t>(o, 0xFFFFFFF0u);
  dd.lists_bytes = (uint32_t)std::min<size_t>(lists_bytes, 0xFFFFFFF0u);
  const uint32_t gb = cdiv(std::max<uint32_t>(Pn, 1), kApplyBlock);
  uint32_t* g_new = (c->gstat_cur ? c->d_gstat : c->d_gstat2).as<uint32_t>();
  uint32_t* g_next = (c->gstat_cur ? c->d_gstat2 : c->d_gstat).as<uint32_t>();
  if (derive) c->kinfo_tag++;
  GroupPatch gp{};
  const GroupsDev grp = c->have_groups ? groups_dev(c) : GroupsDev{};
  if (c->groups_launch_pending) {                                    // this cycle's group patch + findMaxPG: one more block of this launch
    gp.on = 1;
    gp.C = (c->have_fit && c->have_nodes) ? c->C : 0u;
    gp.tag = c->info_tag;
    gp.info = c->h_info;
    gp.matched = const_cast<uint32_t*>(grp.matched);
    gp.status_scheduled = const_cast<uint32_t*>(grp.status_scheduled);
    gp.flags = const_cast<uint8_t*>(grp.flags);
    gp.dp = c->pending_dp;
  }
  hipLaunchKernelGGL(k_pods_apply, dim3(gb + 1 + gp.on), dim3(kApplyBlock), 0, c->stream, old, old_pclass, old_ppair, nw, dd, c->G, L, g_new, g_next,
                     derive ? queue_dirs(c) : QueueDirs{}, c->hash_keep, derive ? 1u : 0u, gb, c->kinfo_tag, c->h_info, grp, gp.on ? batch_dev(c) : BatchDev{}, gp,
                     c->d_gcount.as<uint32_t>());
  LAUNCHCHK(c, BS_KERNEL_PREPASS);
  restore.armed = false;                                             // the launch is out: the new queue is the queue
  if (gp.on) c->groups_launch_pending = false;                       // (the group patch went with it)
  c->dstage_busy = true;
  c->cur_pack = np;
  c->owner_ready = false;
  c->rep_valid = false;                                              // the queue was compacted: pod indices of the derivation are history
  c->epochs_ready = false;
  c->batch_since_pods = false;
  c->first_reach_hint = 0xFFFFFFFFu;
  c->bitmap_valid = false;
  if (derive) {
    c->gstat_cur ^= 1u;
    c->ids_used += I;
    c->kinfo_pending = true;
    c->k_bound += I;                                                 // every inserted pod may bring a request nobody had
  } else {
    c->n_rederives++;
    if ((rc = derive_pods(c, false))) return rc;
  }
  return maybe_analyse_epochs(c);
}

// -------------------------------------------------------------------------------------------------
// batched queue ordering (bs_sort.hpp)
// -------------------------------------------------------------------------------------------------
int bs_queue_order_load(bs_ctx* c, uint32_t g, const uint32_t* order_rank) {
  if (!c || (g && !order_rank)) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c, c->d_order_rank.reserve(std::max<size_t>(4, (size_t)g * 4)));
  if (g) HIPCHK(c, hipMemcpyAsync(c->d_order_rank.p, order_rank, (size_t)g * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->order_g = g;
  return BS_OK;
}

int bs_queue_sort(bs_ctx* c, uint32_t p, const int32_t* priority, const int32_t* group, const int64_t* queue_ts, uint32_t* perm_out) {
  if (!c || (p && (!priority || !group || !queue_ts || !perm_out))) return BS_ERR_INVALID;
  if (!p) return BS_OK;
  int rc = use_device(c);
  if (rc) return rc;
  const size_t n = p;
  const size_t o_prio = 0, o_grp = align256(n * 4), o_ts = o_grp + align256(n * 4), o_a = o_ts + align256(n * 8), o_b = o_a + align256(n * 4),
               o_perm = o_b + align256(n * 4), total = o_perm + align256(n * 4);
  HIPCHK(c, c->d_sort.reserve(total));
  uint8_t* base = c->d_sort.as<uint8_t>();
  HIPCHK(c, hipMemcpyAsync(base + o_prio, priority, n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + o_grp, group, n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + o_ts, queue_ts, n * 8, hipMemcpyHostToDevice, c->stream));
  SortIn in{};
  in.p = p;
  in.g = c->order_g;                                 // groups the order ranks cover; a label beyond it is a lister error
  in.prio = reinterpret_cast<const int32_t*>(base + o_prio);
  in.group = reinterpret_cast<const int32_t*>(base + o_grp);
  in.ts = reinterpret_cast<const int64_t*>(base + o_ts);
  in.order_rank = c->d_order_rank.as<uint32_t>();
  hipLaunchKernelGGL(k_queue_sort, dim3(1), dim3(kSortBlock), 0, c->stream, in, reinterpret_cast<uint32_t*>(base + o_a),
                     reinterpret_cast<uint32_t*>(base + o_b), reinterpret_cast<uint32_t*>(base + o_perm));
  LAUNCHCHK(c, BS_KERNEL_PREPASS);
  HIPCHK(c, hipMemcpyAsync(perm_out, base + o_perm, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BS_OK;
}

int bs_shard_set(bs_ctx* c, uint32_t rank, uint32_t nranks) {
  if (!c || nranks == 0 || rank >= nranks) return BS_ERR_INVALID;
  c->rank = rank;
  c->nranks = nranks;
  return BS_OK;
}

int bs_stream(bs_ctx* c, void** stream) {
  if (!c || !stream) return BS_ERR_INVALID;
  *stream = (void*)c->stream;
  return BS_OK;
}

int bs_reduce_external(bs_ctx* c, uint32_t on) {
  if (!c) return BS_ERR_INVALID;
  c->reduce_external = on != 0;
  return BS_OK;
}

int bs_first_reach_hint(bs_ctx* c, uint32_t local_index) {
  if (!c) return BS_ERR_INVALID;
  if (!c->have_pods) { c->last_error = "bs_first_reach_hint before bs_pods_load"; return BS_ERR_STATE; }
  c->first_reach_hint = local_index;
  return BS_OK;
}

int bs_group_admit_bind(bs_ctx* c, void* dptr) {
  if (!c) return BS_ERR_INVALID;
  c->ext_admit = reinterpret_cast<uint32_t*>(dptr);
  return BS_OK;
}

int bs_group_admit_devptr(bs_ctx* c, void** dptr, uint32_t* count) {
  if (!c || !dptr || !count) return BS_ERR_INVALID;
  if (!c->have_groups) return BS_ERR_STATE;
  *dptr = (void*)batch_dev(c).admit;      // inside the result pack: re-query after a load that grows it
  *count = c->G;
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
// After the last kernel of a batch: the cross-rank reduction of the admit counters (native RCCL) or the hand-over
// to the caller's collective.
static int batch_collective(bs_ctx* c, uint32_t stages, const GroupsDev& gr, const BatchDev& b) {
  c->batch_pending_finish = false;
  if (!(stages & BS_STAGE_TALLY)) return BS_OK;
  if (c->nranks > 1 || c->reduce_external) {
    if (c->comm) {
      // native RCCL: one all-reduce(sum) of the per-group admit counters on the context stream
      if (!c->rccl_allreduce || c->rccl_allreduce(b.admit, b.admit, c->G, ncclUint32, ncclSum, (ncclComm_t)c->comm, c->stream) != ncclSuccess) {
        c->last_error = "ncclAllReduce failed";
        return BS_ERR_COMM;
      }
      if (c->G) hipLaunchKernelGGL(k_ready, dim3(cdiv(c->G, 256)), dim3(256), 0, c->stream, gr, b);
      LAUNCHCHK(c, BS_KERNEL_TALLY);
    } else {
      c->batch_pending_finish = true;   // caller reduces bs_group_admit_devptr, then bs_batch_finish
    }
  }
  return BS_OK;
}

// slot arrays shared by both chains (scan = classes + groups or one per pod, Filter = 2 x classes or one per pod)
static int reserve_slots(bs_ctx* c, bool run_filter) {
  const uint32_t P = c->P, G = c->G;
  // slots are named after request classes, and class ids outlive their pods between two derivations (bs_pods_apply): size for the ids
  const uint32_t ids = std::max(P, c->ids_used);
  uint32_t scan_cap = ids + G + 64, filter_cap = 2 * ids + 64;
  // the id bound creeps up with every bs_pods_apply: keep the capacities once they suffice, grow them by half when they do not
  // (a reallocation is a device-wide wait)
  if (scan_cap <= c->scan_slots_cap && c->d_first_row.cap >= (size_t)c->scan_slots_cap * 4) scan_cap = c->scan_slots_cap; else scan_cap += scan_cap / 2;
  if (filter_cap <= c->filter_slots_cap && (!run_filter || c->d_uflags.cap >= (size_t)c->filter_slots_cap * 4) &&
      (!run_filter || c->d_fu_bitmap.cap >= (size_t)(cdiv(c->N, 64) + 1) * c->filter_slots_cap * 8))
    filter_cap = c->filter_slots_cap;
  else
    filter_cap += filter_cap / 2;
  int rc;
  HIPCHK(c, c->d_first_row.reserve((size_t)scan_cap * 4));
  if ((rc = reserve_filled(c, c->d_first_row64, (size_t)scan_cap * 8, 0xFF))) return rc;      // 64-bit minima keyed by ~batch_seq: born as "none", never reset
  if ((rc = reserve_filled(c, c->d_scan_rec, (size_t)kStepSlotsMax * kScanRecChunks * 8, 0))) return rc;       // } tagged result words of the whole-step launch
  if ((rc = reserve_filled(c, c->d_feas_rec, (size_t)2 * kStepSlotsMax * kFeasRecGroups * 8, 0))) return rc;   // } (tag = ~batch_seq, never 0)
  if ((rc = reserve_filled(c, c->d_chunk_rec, (size_t)64 * kRecStride * 8, 0))) return rc;    // tagged words (tag = ~batch_seq, never 0)
  HIPCHK(c, c->d_qreq_s.reserve((size_t)scan_cap * c->LP * 8));
  HIPCHK(c, c->d_qflags_s.reserve((size_t)scan_cap * 4));
  HIPCHK(c, c->d_qtab_s.reserve((size_t)scan_cap * 4));
  if ((rc = reserve_filled(c, c->d_qstamp_s, (size_t)scan_cap * 4, 0))) return rc;
  if (run_filter) {
    HIPCHK(c, c->d_fu_bitmap.reserve((size_t)(cdiv(c->N, 64) + 1) * filter_cap * 8));      // slot rows, then the per-slot feasible counts
    HIPCHK(c, c->d_uparams.reserve((size_t)filter_cap * 64));
    if ((rc = reserve_filled(c, c->d_uflags, (size_t)filter_cap * 4, 0))) return rc;
    if ((rc = reserve_filled(c, c->d_uclaim, (size_t)filter_cap * 4, 0))) return rc;
    HIPCHK(c, c->d_nodew.reserve(((size_t)6 * (cdiv(c->N, 64) + 2) + 16) * 8));
  }
  c->scan_slots_cap = scan_cap;
  c->filter_slots_cap = filter_cap;
  return BS_OK;
}

// Latency mode (BS_BATCH_HOST_RESULTS): pinned, GPU-visible result pack the last launch writes itself.
// Layout = the device result pack (layout_out) | feas[hstride] | completion word; rows [W + 1][hstride] beside it.
static int ensure_hout(bs_ctx* c) {
  const uint32_t hs = std::min<uint32_t>(c->filter_slots_cap, 1024u);
  const size_t off_feas = align256(c->outpack_bytes), off_tag = off_feas + align256((size_t)hs * 4), need = off_tag + 256;
  const size_t need_rows = (size_t)(cdiv(c->N, 64) + 1) * hs * 8;
  if (need > c->h_hout_cap) {
    if (c->h_hout) (void)hipHostFree(c->h_hout);
    c->h_hout = nullptr; c->h_hout_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_hout, need, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_hout, 0, need);
    c->h_hout_cap = need;
  }
  if (need_rows > c->h_hrows_cap) {
    if (c->h_hrows) (void)hipHostFree(c->h_hrows);
    c->h_hrows = nullptr; c->h_hrows_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_hrows, need_rows, hipHostMallocMapped | hipHostMallocCoherent));
    c->h_hrows_cap = need_rows;
  }
  c->hstride = hs;
  c->off_hfeas = off_feas;
  c->off_htag = off_tag;
  return BS_OK;
}

// Latency mode (BS_BATCH_HOST_RESULTS) of the two three-launch chains: the final launch mirrors every result into pinned host
// memory and publishes a completion word; this points the batch view at that memory.
static int setup_host_out(bs_ctx* c, uint32_t stages, bool run_filter, BatchDev& b, BatchParams& prm) {
  const bool host_out = (stages & BS_BATCH_HOST_RESULTS) && c->nranks == 1 && !c->reduce_external && !c->ext_admit && !(stages & BS_BATCH_COMMIT);
  c->last_host_out = host_out;
  if (!host_out) return BS_OK;
  int rc = ensure_hout(c);
  if (rc) return rc;
  c->host_tag = c->host_tag == 0x7FFFFFFF ? 1 : c->host_tag + 1;
  prm.host_tag = c->host_tag;
  uint8_t* h = c->h_hout;
  b.h_pf_code = h + c->off_pf_code;
  b.h_pf_first_k = reinterpret_cast<uint32_t*>(h + c->off_pf_first_k);
  b.h_pf_leader = reinterpret_cast<int32_t*>(h + c->off_pf_leader);
  b.h_fl_code = h + c->off_fl_code;
  b.h_fl_feasible = reinterpret_cast<uint32_t*>(h + c->off_fl_feasible);
  b.h_fl_slot = reinterpret_cast<uint32_t*>(h + c->off_fl_slot);
  b.h_admit = reinterpret_cast<uint32_t*>(h + c->off_admit);
  b.h_ready = h + c->off_ready;
  b.h_feas = reinterpret_cast<uint32_t*>(h + c->off_hfeas);
  b.h_tag = reinterpret_cast<int32_t*>(h + c->off_htag);
  b.h_rows = run_filter ? c->h_hrows : nullptr;
  b.hstride = c->hstride;
  return BS_OK;
}

// BS_BATCH_FILTER_DENY | BS_BATCH_COMMIT, after the stream has been waited for: did the device gate this run's commit kernels off
// (its flag word is non-zero: the run is not the fixed point, fd_resolve runs the batch again)?
static bool fd_commit_gated(const bs_ctx* c) {
  volatile const int32_t* hf = c->h_info + 14;
  return c->fd_on && (hf[0] || hf[1]);
}

// BS_BATCH_FILTER_DENY: the two launches behind a chain's last one (bs_fdeny.hpp).  tail: this chain left tally and completion
// word to k_fd_apply.
static int launch_filter_deny(bs_ctx* c, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, BatchParams prm, bool tail) {
  if (!c->P) return BS_OK;
  prm.seq_inv = ~c->key_seq;
  c->fd_seq_inv = prm.seq_inv;
  const dim3 grid(cdiv(c->P, 256)), blk(256);
  TIMED(c, BS_KERNEL_RESOLVE, {
    // (the final blocks of the steady-state and positional chains have already taken the event minima)
    if (!tail) hipLaunchKernelGGL(k_fd_events, grid, blk, 0, c->stream, pd, gr, nd, b, prm);
    hipLaunchKernelGGL(k_fd_apply, grid, blk, 0, c->stream, pd, gr, nd, b, prm, tail ? 1u : 0u);
  });
  c->launches += tail ? 1 : 2;
  return BS_OK;
}
// ... and in front of a chain's commit kernel: Filter's deny entries join the group's first rejected pod
static int launch_filter_deny_marks(bs_ctx* c, const BatchDev& b, BatchParams prm, uint32_t* reject) {
  if (!c->G) return BS_OK;
  prm.seq_inv = c->fd_seq_inv;
  hipLaunchKernelGGL(k_fd_reject, dim3(cdiv(c->G, 256)), dim3(256), 0, c->stream, b, prm, reject, c->G);
  LAUNCHCHK(c, BS_KERNEL_RESOLVE);
  c->launches++;
  return BS_OK;
}

static inline uint64_t host_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
// k_fast_step_a (bs_fast.hpp): the one-launch form of launch A + the scan / Filter roles — the latency regime only (K known on the
// host and <= 256 class slots, <= 64 table chunks), no instrumentation that needs the legacy launches (work counters, per-launch stamps),
// and a context that has not seen an in-launch hand-over time out.  BS_STEP_A=0 switches it off (bs_ctx::step_a_form).
static bool step_a_possible(const bs_ctx* c, uint32_t stages, uint32_t k, uint32_t nchunks) {
  return c->step_a_on && !c->no_fuse_final && k > 0 && k <= kStepSlotsMax && nchunks <= 64 && c->M > 0 && c->P > 0 && !c->collect_stats &&
         c->cfg.enable_timing < 2 && c->S <= 4;
}
static int step_a_residency(bs_ctx* c, bool whole) {
  int& r = c->step_a_resident[whole ? 1 : 0];
  if (r < 0) r = step_a_residency_query(fast_launch(c), whole);
  return r;
}

// The steady-state chain (bs_fast.hpp): three launches, nothing reset, no wait.
static int run_fast(bs_ctx* c, uint32_t stages) {
  int rc;
  const uint32_t P = c->P, G = c->G, N = c->N;
  const uint32_t W = cdiv(N, 64);
  const bool run_filter = stages & BS_STAGE_FILTER;
  const bool commit = stages & BS_BATCH_COMMIT;
  NodesDev nd = nodes_dev(c);
  GroupsDev gr = groups_dev(c);
  PodsDev pd = pods_dev(c);
  BatchDev b = batch_dev(c);
  BatchParams prm = batch_params(c);
  prm.run_filter = run_filter;
  prm.use_classes = 1;
  prm.fuse_filter = run_filter ? 1u : 0u;
  prm.scan_slots_cap = c->scan_slots_cap;
  prm.filter_slots_cap = c->filter_slots_cap;
  prm.stamp = 1u + c->stamp_ctr;
  prm.seq_inv = ~c->key_seq;
  prm.commit = commit ? 1u : 0u;
  prm.do_tally = (stages & BS_STAGE_TALLY) ? 1u : 0u;
  prm.do_ready = (prm.do_tally && c->nranks == 1 && !c->reduce_external) ? 1u : 0u;
  if ((rc = setup_host_out(c, stages, run_filter, b, prm))) return rc;
  const uint32_t side_slot = (uint32_t)c->steady_table;
  const uint32_t nchunks = std::max<uint32_t>(1, cdiv(c->M, 256));
  BatchDev bt = b;                                   // the batch view shifted to the steady table's slot (slot index 0)
  bt.tables = b.tables + (size_t)side_slot * prm.mcap * prm.LP;
  bt.kp = b.kp + (size_t)side_slot * 16;
  bt.chunk_tot = b.chunk_tot + (size_t)side_slot * cdiv(c->Ncap, 256) * 16;
  bt.chunk_kp = b.chunk_kp + (size_t)side_slot * cdiv(c->Ncap, 256) * 16;
  bt.gmax = b.gmax + (size_t)side_slot * cdiv(c->Ncap, 64) * prm.LP;
  const TableDesc* forced = b.desc + side_slot;
  const int ts = c->S <= 4 ? (int)c->S : -1;
  if (commit && G) HIPCHK(c, hipMemsetAsync(c->d_fast_reject.p, 0xFF, (size_t)G * 4, c->stream));

  // K is known on the host once its copy from the pod load / queue patch has landed — never waited for here; when it is, the
  // kernels need not fetch it in front of everything else
  if (c->kinfo_pending && ((volatile int32_t*)c->h_info)[5] == c->kinfo_tag && (rc = resolve_pods(c))) return rc;
  prm.k_host = c->kinfo_pending ? 0u : c->h_K;
  const uint64_t hp1 = c->host_probe ? host_ns() : 0;
  // ---- launch A and the scan / Filter roles of launch B as ONE launch (k_fast_step_a; round 6: its class-slot form is the default), then k_fast_final
  // K not on the host yet (the queue was patched a moment ago and nobody has waited for the patch's word): the whole-step form sizes its grid for a
  // BOUND of the class count (last known K + pods inserted since) and its blocks read K on the device; the other forms want it known
  const bool k_known = !c->kinfo_pending;
  const uint32_t k_step = k_known ? prm.k_host : ((c->step_a_form >= 3u && c->dirs_ready) ? c->k_bound : 0u);
  if (step_a_possible(c, stages, k_step, nchunks) && (c->step_a_form == 1u || c->dirs_ready || (c->batch_since_pods && c->pairs_ready && c->rep_valid && c->have_groups))) {
    const uint32_t qb = cdiv(P, kTblChunk);
    const uint32_t K = k_step;
    const uint32_t nshares = std::max<uint32_t>(1, std::min<uint32_t>(c->step_shares, cdiv(K, 8)));
    const uint32_t fblocks = run_filter ? cdiv(std::min<uint32_t>(c->filter_waves, cdiv(2 * K, 64) * std::max<uint32_t>(1, cdiv(W, 2))), 4) : 0u;
    // the class-slot form needs the class directory (ckeys / cpres: class id -> request lanes).  It is built once per derivation of the queue — like bs_pods_apply's first call — and only when the
    // queue's classes and pairs are already resolved on the host (pairs_ready / rep_valid: the pod load's class count has landed; never waited for here): a caller that re-uploads its queue and
    // scores it at once takes the two-launch chain and pays neither the directory kernel nor a stream wait.  So the FIRST batch over a fresh queue is of either form (both equal the oracle).
    uint32_t pb = 0;
    if (c->step_a_form >= 2u) {
      if (!c->dirs_ready && c->batch_since_pods && c->pairs_ready && c->rep_valid && c->have_groups && (rc = build_dirs(c))) return rc;
      if (c->dirs_ready) pb = cdiv(K, kTblChunk);
    }
    // form 3: the pod blocks go on to the final verdicts inside the same launch (fast_final_block<true>) — one launch for the whole step; its table rows
    // are the nodes in list order (skipped nodes are rows that add nothing), not the compacted rows
    const uint32_t nch_nodes = std::max<uint32_t>(1, cdiv(N, 256));
    const uint32_t whole = (c->step_a_form >= 3u && pb && nch_nodes <= 64 && c->d_first_row64.p && c->d_scan_rec.p && c->d_feas_rec.p && c->d_chunk_rec.p) ? 1u : 0u;
    const uint32_t nchunks_s = whole ? nch_nodes : nchunks;
    const uint32_t grid = qb + pb + nchunks_s * nshares + fblocks;
    if ((k_known || whole) && (int)grid <= step_a_residency(c, whole != 0)) {
      TIMED(c, BS_KERNEL_QUERY, {
        launch_fast_step_a(fast_launch(c), dim3(grid), pd, gr, nd, b, bt, prm, forced, nchunks_s, qb, nshares, fblocks, c->tk_pods, c->tk_tab, pb,
                           c->d_ckeys.as<int64_t>(), c->d_cpres.as<uint32_t>(), c->pair_cap, whole, c->tk_p1, c->tk_done, side_slot - c->C);
      });
      c->tk_pods += pb ? pb : qb;
      if (!whole) c->tk_tab += nchunks_s;                              // (the whole-step form hands the chunk totals over as tagged words: no ticket)
      if (whole) {
        c->tk_p1 += qb;
        if (qb > kGatherDirectBlocks) c->tk_done += nchunks_s * nshares + fblocks;   // (a small queue's producers leave tagged words, no counter)
      } else {
        TIMED(c, BS_KERNEL_RESOLVE, hipLaunchKernelGGL(k_fast_final, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, pd, gr, nd, b, prm, cdiv(P, kTblChunk)));
      }
      c->launches = whole ? 1 : 2;
      c->last_step_a = true;
      if (c->test_timeout_after && --c->test_timeout_after == 0 && c->h_info) {      // test hook: what a block whose wait ran out does (bs_fast.hpp, kSpinBound)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ((volatile int32_t*)c->h_info)[12] = 1;
      }
      if (prm.filter_deny && (rc = launch_filter_deny(c, pd, gr, nd, b, prm, true))) return rc;
      if (commit) {
        if (prm.filter_deny && (rc = launch_filter_deny_marks(c, b, prm, b.fast_reject))) return rc;
        if (G) hipLaunchKernelGGL(k_fast_commit, dim3(cdiv(G, 256)), dim3(256), 0, c->stream, pd, b, const_cast<uint8_t*>(gr.flags),
                                  const_cast<uint64_t*>(gr.occupied), G, prm.filter_deny ? b.fd_flag : nullptr);
        LAUNCHCHK(c, BS_KERNEL_RESOLVE);
        int32_t last = -1;
        HIPCHK(c, hipMemcpyAsync(&last, b.pf_leader + (P - 1), 4, hipMemcpyDeviceToHost, c->stream));
        if (G) HIPCHK(c, hipMemcpyAsync(c->h_gflags.data(), gr.flags, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!fd_commit_gated(c)) c->sop_leader0 = last;
        c->launches++;
      }
      return batch_collective(c, stages, gr, b);
    }
  }
  c->last_step_a = false;
  // the throughput regime (more than 16 tiles of class slots) is known before launch A: when launch B will take the transposed Filter item,
  // launch A's grid carries the node-word blocks (node_words_block) behind the table chunks
  const uint32_t k_est = std::min<uint32_t>(P, c->kinfo_pending ? std::max<uint32_t>(2 * c->h_K, 1024) : std::max<uint32_t>(c->h_K, 1));
  const bool node_words = run_filter && cdiv(k_est, 64) > 16 && c->tp_filter >= 5u && !c->no_nodew && c->d_nodew.p;
  if (node_words) {
    b.nodew = bt.nodew = c->d_nodew.as<uint64_t>();
    b.nodew_stride = bt.nodew_stride = W + 2;
    b.tiles2_min = bt.tiles2_min = c->tp_tmin * std::max<uint32_t>(1u, c->nranks);
  }
  // ---- launch A: per-pod decisions, scan / Filter slots | chunk-local running sums of the table
  TIMED(c, BS_KERNEL_QUERY, {
    const uint32_t qb = cdiv(P, kTblChunk);
    const dim3 qg(qb + nchunks + (node_words ? cdiv(N, kTblChunk) : 0u)), blk(kTblChunk);
    switch (ts) {
      case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<0>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<1>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<2>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<3>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<4>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_query_tables<-1>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
    }
  });
  const uint64_t hp2 = c->host_probe ? host_ns() : 0;
  bool fused = false, tp_split = false;
  uint64_t hp3 = 0;
  // ---- launch B: node scan over the class slots | Filter evaluation over the Filter slots
  // The work loops size themselves on the device (the class count lives there); the grid only has to be large enough.
  TIMED(c, BS_KERNEL_SCAN, {
    const uint32_t nseg = pick_scan_share(c);
    // few tiles (the latency regime): one scan item (tile of 64 class slots x share) per BLOCK, its four waves take a quarter of
    // every group's rows each; many tiles (thousands of distinct requests): one item per wave, 4 per block
    const uint32_t tiles = cdiv(k_est, 64);
    const bool throughput = tiles > 16;
    // scan shares per tile when an item is one wave's: thousands of tiles are parallelism enough, and every item pays the table's
    // first fetch (cfg3 all-distinct 48.5 -> 38.5 us, cfg4 with the transposed Filter role 220 -> 192 us from 64 / 8 shares to 2)
    const uint32_t share_b = c->tp_share ? c->tp_share : (throughput ? 2u : 64u);
    // the Filter work of the transposed item is cut for twice the waves from 65 536 (tile, two node blocks) units on (half of the
    // slot tiles belong to the carried leader and are usually idle: 8192 items leave half of the wave slots without work there)
    struct FwGuard { bs_ctx* c; uint32_t saved; ~FwGuard() { c->filter_waves = saved; } } fw_guard{c, c->filter_waves};
    if (throughput && c->tp_filter >= 5u && !c->filter_waves_env)
      c->filter_waves = c->tp_fwaves ? c->tp_fwaves
                                     : ((uint64_t)cdiv(2 * k_est, 64) * std::max<uint32_t>(1, cdiv(W, 2)) >= 65536u ? std::max<uint32_t>(c->filter_waves, 16384u) : c->filter_waves);
    const uint32_t fblocks = run_filter ? cdiv(std::min<uint32_t>(c->filter_waves, cdiv(2 * k_est, 64) * std::max<uint32_t>(1, cdiv(W, 2))), 4) : 0u;
    auto scan_grid = [&](uint32_t nsub) {
      const uint32_t items = tiles * std::min<uint32_t>(nsub == 4u ? nseg : share_b, cdiv(c->M, 64));
      return std::max<uint32_t>(1, std::min<uint32_t>(cdiv(c->target_waves, 4), nsub == 4u ? items : cdiv(items, 4)));
    };
    // the fused form (final blocks wait for the producers INSIDE the launch) only in the latency regime, and only when every block
    // of its grid is resident at once (see fused_residency); BS_NO_FUSE_FINAL=1 forces the separate launches
    fused = tiles <= 16 && !c->no_fuse_final && (int)(scan_grid(4u) + fblocks + cdiv(P, 256)) <= fused_residency(c);
    prm.scan_nsub = fused ? 4u : 1u;
    const uint32_t scan_blocks = scan_grid(prm.scan_nsub);
    hp3 = c->host_probe ? host_ns() : 0;
    if (fused) {
      // ---- ... and launch C in the same launch: final codes, Filter code / slot / feasible count per pod, admit counts, quorum
      const dim3 grid(scan_blocks + fblocks + cdiv(P, 256));
      launch_fast_bc(fast_launch(c), grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, fblocks);
    } else if (c->tp_filter == 8u && fblocks && throughput) {
      // the two roles as launches of their own, SIDE BY SIDE: the scan on the side stream (its items are chains of dependent loads,
      // its 130-odd VGPRs its own business), the transposed Filter item at seven waves per SIMD on the main one; both read what launch
      // A wrote and write disjoint outputs for the final launch, which waits for both
      FastLaunch side = fast_launch(c);
      side.stream = c->stream3;
      HIPCHK(c, hipEventRecord(c->ev_query, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_query, 0));
      launch_fast_scan(side, dim3(scan_blocks), bt, prm, share_b);
      HIPCHK(c, hipEventRecord(c->ev_filter, c->stream3));
      launch_fast_filter(fast_launch(c), dim3(fblocks), pd, nd, bt, prm);
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_filter, 0));
      tp_split = true;
    } else if (c->tp_filter >= 6u && fblocks && throughput && c->S <= 4u) {
      // both roles in one launch, the Filter role by the transposed item (the default of the throughput regime)
      launch_fast_bt(fast_launch(c), dim3(scan_blocks + fblocks), nd, bt, prm, share_b, scan_blocks);
    } else if (c->tp_filter && fblocks && throughput) {
      // the throughput regime with the two roles as launches of their own: the scan at its register footprint, the Filter loop at a
      // leaner one (more resident waves); same stream, the scan first — its items are dependent-load chains that would otherwise sit
      // in the wave slots the Filter loop can fill
      launch_fast_scan(fast_launch(c), dim3(scan_blocks), bt, prm, share_b);
      launch_fast_filter(fast_launch(c), dim3(fblocks), pd, nd, bt, prm);
      tp_split = true;
    } else {
      launch_fast_b(fast_launch(c), dim3(scan_blocks + fblocks), pd, nd, bt, prm, share_b, scan_blocks);
    }
  });
  if (c->host_probe) {
    const uint64_t hp4 = host_ns();
    c->hp_ns[0] += c->hp_t1 - c->hp_t0; c->hp_ns[1] += hp1 - c->hp_t1; c->hp_ns[2] += hp2 - hp1; c->hp_ns[3] += hp3 - hp2; c->hp_ns[4] += hp4 - hp3;
    c->hp_n++;
  }
  c->launches = 2;
  if (!fused) {                                      // the throughput regime (or a grid the chip cannot hold at once): launch C on its own
    TIMED(c, BS_KERNEL_RESOLVE, hipLaunchKernelGGL(k_fast_final, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, pd, gr, nd, b, prm, cdiv(P, kTblChunk)));
    c->launches = tp_split ? 4 : 3;
  }
  if (prm.filter_deny && (rc = launch_filter_deny(c, pd, gr, nd, b, prm, true))) return rc;
  if (commit) {
    if (prm.filter_deny && (rc = launch_filter_deny_marks(c, b, prm, b.fast_reject))) return rc;
    if (G) hipLaunchKernelGGL(k_fast_commit, dim3(cdiv(G, 256)), dim3(256), 0, c->stream, pd, b, const_cast<uint8_t*>(gr.flags),
                              const_cast<uint64_t*>(gr.occupied), G, prm.filter_deny ? b.fd_flag : nullptr);
    LAUNCHCHK(c, BS_KERNEL_RESOLVE);
    int32_t last = -1;
    HIPCHK(c, hipMemcpyAsync(&last, b.pf_leader + (P - 1), 4, hipMemcpyDeviceToHost, c->stream));
    if (G) HIPCHK(c, hipMemcpyAsync(c->h_gflags.data(), gr.flags, G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!fd_commit_gated(c)) c->sop_leader0 = last;   // findMaxPG ignores deny entries and OccupiedBy: the group analysis stays valid
    c->launches++;
  }
  return batch_collective(c, stages, gr, b);
}

// The positional chain (bs_epoch.hpp): three launches for batches in which captures, MinResources defaults or a leader
// without matched pods make a pod's decision depend on its queue position.  `taken` = false: the analysis says this batch
// is not for this chain (too many leader runs, slot capacity) and nothing was launched.
// After k_commit / k_fast_commit changed the group state on the device: the leader carried into the next batch and the
// host's view of the flags (captures / MinResources defaults may be gone now).
static int commit_readback(bs_ctx* c, const GroupsDev& gr, const BatchDev& b) {
  const uint32_t P = c->P, G = c->G;
  std::vector<uint32_t> cls(G);
  if (P) {
    int32_t last = -1;
    HIPCHK(c, hipMemcpyAsync(&last, b.pf_leader + (P - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (fd_commit_gated(c)) return BS_OK;            // this run is not the fixed point: nothing was committed, nothing is carried over
    c->sop_leader0 = last;
  }
  if (G) {
    HIPCHK(c, hipMemcpy(c->h_gflags.data(), gr.flags, G, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cls.data(), gr.cls, (size_t)G * 4, hipMemcpyDeviceToHost));
  }
  c->n_uncaptured = 0;
  c->n_nominres = 0;
  c->max_group_cls = 0;
  for (uint32_t i = 0; i < G; ++i) {
    if (!(c->h_gflags[i] & BS_GROUP_HAS_POD)) c->n_uncaptured++;
    else c->max_group_cls = std::max(c->max_group_cls, cls[i]);
    if (!(c->h_gflags[i] & BS_GROUP_HAS_MINRES)) c->n_nominres++;
  }
  return BS_OK;
}

static int run_epoch(bs_ctx* c, uint32_t stages, bool* taken) {
  int rc;
  *taken = false;
  if (!c->epochs_ready && (rc = analyse_epochs(c))) return rc;
  if ((rc = resolve_pods(c)) || (rc = resolve_epochs(c))) return rc;
  const uint32_t P = c->P, G = c->G, N = c->N, C = c->C, K = c->h_K, R = c->h_R;
  const uint32_t W = cdiv(N, 64);
  const bool run_filter = stages & BS_STAGE_FILTER;
  if ((c->h_eflags & 4u) || R == 0 || R > kMaxRuns || K == 0) return BS_OK;
  const uint32_t GB = cdiv(2 * R * K, 64) * 64;                  // (view, class) slots, then the group slots from a tile boundary on
  const uint32_t filter_slots = (R + 1) * K;
  if ((uint64_t)GB + G > c->scan_slots_cap || filter_slots > c->filter_slots_cap || filter_slots > P) return BS_OK;
  *taken = true;
  NodesDev nd = nodes_dev(c);
  GroupsDev gr = groups_dev(c);
  PodsDev pd = pods_dev(c);
  BatchDev b = batch_dev(c);
  BatchParams prm = batch_params(c);
  prm.run_filter = run_filter;
  prm.use_classes = 0;
  prm.fuse_filter = run_filter ? 1u : 0u;
  prm.scan_slots_cap = c->scan_slots_cap;
  prm.filter_slots_cap = c->filter_slots_cap;
  prm.stamp = 1u + c->stamp_ctr;
  prm.seq_inv = ~c->key_seq;
  prm.do_tally = (stages & BS_STAGE_TALLY) ? 1u : 0u;
  prm.do_ready = (prm.do_tally && c->nranks == 1 && !c->reduce_external) ? 1u : 0u;
  EpochDev ep{};
  ep.run_of_epoch = c->d_run_of_epoch.as<uint32_t>();
  ep.run_leader = c->d_run_leader.as<int32_t>();
  ep.gslot = c->d_gslot.as<uint32_t>();
  ep.gfirstq = c->d_gfirstq.as<unsigned long long>();
  ep.R = R; ep.K = K; ep.GB = GB;
  ep.has_first = c->h_eflags & 1u;
  ep.has_reserve = (c->h_eflags >> 1) & 1u;
  if ((rc = setup_host_out(c, stages, run_filter, b, prm))) return rc;
  c->last_rows = filter_slots;
  const uint32_t nchunks = std::max<uint32_t>(1, cdiv(c->M, 256));
  const int ts = c->S <= 4 ? (int)c->S : -1;
  // ---- launch A: per-pod decisions, scan / Filter slots | chunk-local running sums of the tables the state can ask for
  TIMED(c, BS_KERNEL_QUERY, {
    const uint32_t qb = cdiv(P, kTblChunk);
    const uint32_t ntab = ep.has_reserve ? 2 * C : (ep.has_first ? C : 0u);       // percent-0.7 tables only behind a leader with matched pods
    const dim3 qg(qb + cdiv(ntab, kTableGroup) * nchunks);
    switch (ts) {
      case 0: launch_epoch_a<0>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
      case 1: launch_epoch_a<1>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
      case 2: launch_epoch_a<2>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
      case 3: launch_epoch_a<3>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
      case 4: launch_epoch_a<4>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
      default: launch_epoch_a<-1>(c, qg, pd, gr, nd, b, prm, ep, nchunks, qb, ntab); break;
    }
  });
  // ---- launch B: node scan over the live slots | Filter over the Filter slots
  TIMED(c, BS_KERNEL_SCAN, {
    // A tile here is 64 DIFFERENT queries against one table (groups in class order): it is done when the slowest of them
    // is, so its live 64-row groups are dealt over many waves (there are only tens of tiles: waves are not scarce, the
    // length of one wave's chain of groups is what the launch waits for).
    const uint32_t tiles = (ep.has_reserve ? cdiv(2 * R * K, 64) : 0u) + (ep.has_first ? cdiv(G, 64) : 0u);
    const uint32_t wave_cap = 4 * c->general_waves;
    const uint32_t nseg = c->scan_share_override ? c->scan_share_override
                                                 : std::max<uint32_t>(2, std::min<uint32_t>(64, wave_cap / (2 * std::max<uint32_t>(tiles, 1))));
    const uint32_t scan_blocks = std::max<uint32_t>(1, cdiv(std::min<uint32_t>(wave_cap, 2 * std::max<uint32_t>(tiles, 1) * std::min<uint32_t>(nseg, cdiv(c->M, 64))), 4));
    const uint32_t fblocks = run_filter ? std::max<uint32_t>(1, cdiv(std::min<uint32_t>(c->filter_waves, cdiv(filter_slots, 64) * std::max<uint32_t>(1, cdiv(W, 2))), 4)) : 0u;
    const dim3 grid(scan_blocks + fblocks);
    switch (c->S) {
      case 0: launch_epoch_b<0>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 1: launch_epoch_b<1>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 2: launch_epoch_b<2>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 3: launch_epoch_b<3>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 4: launch_epoch_b<4>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 5: launch_epoch_b<5>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 6: launch_epoch_b<6>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 7: launch_epoch_b<7>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 8: launch_epoch_b<8>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 9: launch_epoch_b<9>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 10: launch_epoch_b<10>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      case 11: launch_epoch_b<11>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
      default: launch_epoch_b<12>(c, grid, pd, nd, b, prm, ep, nseg, scan_blocks, filter_slots); break;
    }
  });
  // ---- launch C: final codes, stale leader, Filter code / slot / feasible count per pod, admit counts, quorum
  TIMED(c, BS_KERNEL_RESOLVE, hipLaunchKernelGGL(k_epoch_final, dim3(cdiv(P, 256)), dim3(256), 0, c->stream, pd, gr, nd, b, prm, ep));
  c->launches = 3;
  if (prm.filter_deny && (rc = launch_filter_deny(c, pd, gr, nd, b, prm, true))) return rc;
  if (stages & BS_BATCH_COMMIT) {
    // persist what the sequential PreFilter calls would have left behind (k_commit): captures and MinResources defaults from
    // the analysis, OccupiedBy, and the deny entries = every group's first rejected pod
    hipLaunchKernelGGL(k_epoch_reject_groups, dim3(cdiv(G, 256)), dim3(256), 0, c->stream, gr, b, prm, ep, P);
    if (prm.filter_deny && (rc = launch_filter_deny_marks(c, b, prm, b.first_reject))) return rc;
    hipLaunchKernelGGL(k_commit, dim3(cdiv(G, 256)), dim3(256), 0, c->stream, pd, b, prm, const_cast<uint8_t*>(gr.flags), const_cast<uint32_t*>(gr.cls),
                       const_cast<int64_t*>(gr.minres), const_cast<uint32_t*>(gr.mrpres), const_cast<uint64_t*>(gr.occupied), G,
                       prm.filter_deny ? b.fd_flag : nullptr);
    LAUNCHCHK(c, BS_KERNEL_RESOLVE);
    c->launches += 2;
    if ((rc = commit_readback(c, gr, b))) return rc;
    if ((rc = analyse_groups(c))) return rc;              // (findMaxPG for the committed state; the positional analysis is redone by whoever needs it)
  }
  return batch_collective(c, stages, gr, b);
}

static int batch_run_inner(bs_ctx* c, uint32_t stages) {
  int rc = BS_OK;
  if (c->host_probe) c->hp_t0 = host_ns();
  c->fd_on = (stages & BS_BATCH_FILTER_DENY) != 0;
  if (c->fd_on) {
    const size_t g1 = std::max<uint32_t>(c->G, 1);
    if ((rc = reserve_filled(c, c->d_fd_event, g1 * 8, 0xFF)) || (rc = reserve_filled(c, c->d_fd_in, g1 * 4, 0xFF)) || (rc = reserve_filled(c, c->d_fd_flag, 16, 0))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_fd_flag.p, 0, 4, c->stream));
    ((volatile int32_t*)c->h_info)[14] = 0;
    ((volatile int32_t*)c->h_info)[15] = 0;
  }
  const uint32_t P = c->P, G = c->G, N = c->N, C = c->C;
  // fit-class indices address fit rows and running-sum tables on the device: out of range = refuse the batch
  if ((G > c->n_uncaptured && c->max_group_cls >= C) || (P && c->max_pod_cls >= C)) {
    c->last_error = "fit class index out of range (groups.cls / pods.cls vs the loaded fit classes)";
    return BS_ERR_INVALID;
  }
  // findMaxPG's answer for the patched groups: taken if it has landed; guessed (last cycle's table) if it has not and the state is a
  // steady one — the guess is checked when the results are first asked for (batch_settle)
  c->spec_active = false;
  if (c->info_pending && ((volatile int32_t*)c->h_info)[3] != c->info_tag && !c->no_spec && c->steady_prev >= 0 && (uint32_t)c->steady_prev < 2 * c->C && c->n_uncaptured == 0 &&
      c->n_nominres == 0 && !(stages & BS_BATCH_COMMIT) && c->cfg.enable_timing < 2 && !c->collect_stats && c->nranks == 1 && !c->reduce_external &&
      c->fd_iter == 0 && !c->groups_launch_pending && !c->ext_admit) {
    c->steady_table = c->steady_prev;
    c->spec_active = true;
    c->spec_table = c->steady_prev;
    c->spec_stages = stages;
    c->n_spec++;
  } else if ((rc = resolve_groups(c))) {
    return rc;
  }
  if (!c->pairs_ready && (rc = build_pairs(c))) return rc;
  if (c->nranks > 1 && !c->owner_ready && P) {       // sharded: who owns what (balanced by pod count, whole groups)
    HIPCHK(c, c->d_own_start.reserve((size_t)P * 4));
    hipLaunchKernelGGL(k_owner_starts, dim3(1), dim3(kScanBlock), 0, c->stream, pods_dev(c), G, gstat_dev(c), c->d_gcount.as<uint32_t>(),
                       c->d_own_start.as<uint32_t>());
    LAUNCHCHK(c, BS_KERNEL_PREPASS);
    c->owner_ready = true;
  }
  const uint32_t W = cdiv(N, 64);
  const bool run_filter = stages & BS_STAGE_FILTER;
  if ((rc = reserve_slots(c, run_filter))) return rc;
  const uint32_t scan_cap = c->scan_slots_cap, filter_cap = c->filter_slots_cap;
  (void)W;
  // slot stamps are 1 + stamp_ctr: when the stamp starts over, a slot nobody wrote for 65535 batches would look live again
  if (c->stamp_ctr == 0) {
    if (c->d_qstamp_s.p) HIPCHK(c, hipMemsetAsync(c->d_qstamp_s.p, 0, c->d_qstamp_s.cap, c->stream));
    if (c->d_uflags.p) HIPCHK(c, hipMemsetAsync(c->d_uflags.p, 0, c->d_uflags.cap, c->stream));
    if (c->d_uclaim.p) HIPCHK(c, hipMemsetAsync(c->d_uclaim.p, 0, c->d_uclaim.cap, c->stream));
  }
  // the key sequence ran out: every keyed 64-bit minimum goes back to 'none' before ~key_seq starts over
  if (c->rekey_pending) {
    if (c->d_pair_firstq.p) HIPCHK(c, hipMemsetAsync(c->d_pair_firstq.p, 0xFF, c->d_pair_firstq.cap, c->stream));
    if (c->d_first_reach.p) HIPCHK(c, hipMemsetAsync(c->d_first_reach.p, 0xFF, c->d_first_reach.cap, c->stream));
    if (c->d_first_row64.p) HIPCHK(c, hipMemsetAsync(c->d_first_row64.p, 0xFF, c->d_first_row64.cap, c->stream));
    if (c->d_scan_rec.p) HIPCHK(c, hipMemsetAsync(c->d_scan_rec.p, 0, c->d_scan_rec.cap, c->stream));
    if (c->d_feas_rec.p) HIPCHK(c, hipMemsetAsync(c->d_feas_rec.p, 0, c->d_feas_rec.cap, c->stream));
    if (c->d_chunk_rec.p) HIPCHK(c, hipMemsetAsync(c->d_chunk_rec.p, 0, c->d_chunk_rec.cap, c->stream));
    if (c->d_gfirstq.p) HIPCHK(c, hipMemsetAsync(c->d_gfirstq.p, 0xFF, c->d_gfirstq.cap, c->stream));
    if (c->d_fd_event.p) HIPCHK(c, hipMemsetAsync(c->d_fd_event.p, 0xFF, c->d_fd_event.cap, c->stream));
    c->rekey_pending = false;
  }

  const bool captures_possible = c->n_uncaptured > 0 && P > 0;
  // request classes stand for the pods when nothing a pod derives can depend on its queue position:
  // no first-pod capture and no MinResources default (core.go:486-493) left to happen
  const bool use_classes = !captures_possible && c->n_nominres == 0;
  c->last_use_classes = use_classes;
  c->last_stages = stages;
  c->bitmap_valid = false;
  c->batch_since_pods = true;
  // No capture possible and the leader has matched pods: every scan query of the batch uses ONE known
  // table (k_leader_info).
  const bool inline_tables = !captures_possible && c->steady_table >= 0 && c->M && P && c->cfg.enable_timing < 2;
  // No capture possible: the Filter inputs do not depend on the node scan (k_fparams_early), so Filter can
  // run on its own stream beside scan / reject / final.  Joining a second stream costs ~10-20 us of
  // cross-queue signalling, so only when Filter is long enough (slot = pod batches), or when forced.
  const bool early_filter = run_filter && !captures_possible && P && N && !(stages & BS_BATCH_COMMIT) && c->cfg.enable_timing < 2 &&
                            (uint64_t)P * N >= c->early_filter_min && (!use_classes || c->early_forced);
  // ---- steady state: the three-launch chain
  c->last_fast = use_classes && inline_tables && !early_filter && N && !c->no_fast && !c->no_fuse_filter;
  c->last_chain = c->last_fast ? 1u : 0u;
  if (c->last_fast) {
    if (c->host_probe) c->hp_t1 = host_ns();
    rc = run_fast(c, stages);
    advance_batch_seq(c);
    return rc;
  }
  // ---- positional state: the three-launch chain over (view, class) and group slots
  if (!c->no_fast && !c->no_epoch && !c->no_fuse_filter && P && G && N && c->M && !early_filter && c->cfg.enable_timing < 2) {
    bool taken = false;
    rc = run_epoch(c, stages, &taken);
    if (rc) return rc;
    if (taken) {
      c->last_chain = 2;
      advance_batch_seq(c);
      return BS_OK;
    }
  }
  c->epochs_ready = false;            // the general chain re-derives (and its tally re-arms) the per-group minima

  // ---- general chain (first-pod captures, MinResources defaults, leader without matched pods, early Filter)
  c->last_host_out = false;
  NodesDev nd = nodes_dev(c);
  GroupsDev gr = groups_dev(c);
  PodsDev pd = pods_dev(c);
  BatchDev b = batch_dev(c);
  BatchParams prm = batch_params(c);
  prm.run_filter = run_filter;
  const uint32_t tiles_est = cdiv(std::max<uint32_t>(P, 1), 64);
  // J waves share the live 64-row groups of one tile (k_scan deals them round-robin).  Measured on the cold batches
  // (tools/cold_sweep.py): almost every query ends inside the first one or two live groups, and every extra share is a
  // whole extra wave that scans a group for nothing — 13 us at J = 1 against 40 us at J = 8 (cfg3 cold), 28 us against
  // 319 us (cfg4 cold).  So: at most two shares, and a grid that just covers tiles x tables.
  const uint32_t nseg = c->scan_share_override ? c->scan_share_override : 2u;
  const dim3 blk(256);
  prm.use_classes = use_classes ? 1u : 0u;
  prm.scan_slots_cap = scan_cap;
  prm.filter_slots_cap = filter_cap;
  const int ts = c->S <= 4 ? (int)c->S : -1;
  bool commit_dirty = false;
  const uint32_t side_slot = inline_tables ? (uint32_t)c->steady_table : 0u;
  prm.early_filter = early_filter ? 1u : 0u;
  const bool local_ready = c->nranks == 1 && !c->reduce_external;
  // the tally's last pass re-arms the per-group minima and capture epochs for the next batch (saves k_init)
  const bool rearm = !(stages & BS_BATCH_COMMIT);      // (a committing batch changes the group flags: k_init re-derives the capture epochs)
  // class mode without early Filter: k_query fills the Filter slots and ONE launch does node scan + Filter evaluation
  const bool fuse_filter = run_filter && use_classes && !early_filter && P && N && !c->no_fuse_filter;
  prm.fuse_filter = fuse_filter ? 1u : 0u;
  const uint32_t nchunks = std::max<uint32_t>(1, cdiv(c->M, 256));
  uint32_t launches = 0;
  // one known table: it is built inside the first two launches (k_prepass_tables / k_query_tables)
  BatchDev bt = b;                                                       // the batch view shifted to the table's slot (slot index 0)
  const TableDesc* forced = nullptr;
  if (inline_tables) {
    bt.tables = b.tables + (size_t)side_slot * prm.mcap * prm.LP;
    bt.kp = b.kp + (size_t)side_slot * 16;
    bt.chunk_tot = b.chunk_tot + (size_t)side_slot * cdiv(c->Ncap, 256) * 16;
    bt.gmax = b.gmax + (size_t)side_slot * cdiv(c->Ncap, 64) * prm.LP;
    bt.chunk_kp = b.chunk_kp + (size_t)side_slot * cdiv(c->Ncap, 256) * 16;
    forced = b.desc + side_slot;
  }

  // ---- per-batch resets + eligibility (+ findMaxPG when no first-pod capture can occur)
  TIMED(c, BS_KERNEL_PREPASS, {
    if (!c->scratch_armed) { hipLaunchKernelGGL(k_init, dim3(cdiv(std::max<uint32_t>(G, 4), 256)), blk, 0, c->stream, gr, b); launches++; }
    const uint32_t span = std::max(std::max(P, G), (2 * C + 1) * 16);
    const uint32_t fused = captures_possible ? 0u : 1u;
    launches++;
    if (inline_tables) {
      const uint32_t pre = cdiv(span, kPrepassBlock);
      const dim3 pg(pre + 1 + nchunks), pb(kPrepassBlock);
      switch (ts) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<0>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<1>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<2>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<3>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<4>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prepass_tables<-1>), pg, pb, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, pre); break;
      }
    } else
    hipLaunchKernelGGL(k_prepass, dim3(cdiv(span, kPrepassBlock) + fused), dim3(kPrepassBlock), 0, c->stream, pd, gr, b, prm,
                       captures_possible ? 0u : 1u, fused);
    if (captures_possible) {
      hipLaunchKernelGGL(k_epochs_a, dim3(cdiv(P, kScanBlock)), dim3(kScanBlock), 0, c->stream, pd, gr, b);
      hipLaunchKernelGGL(k_epochs_b, dim3(cdiv(P, kScanBlock)), dim3(kScanBlock), 0, c->stream, pd, gr, b);
      launches += 2;
    }
  });
  c->scratch_armed = false;
  c->side_ready = false;
  if (captures_possible) {
    // findMaxPG for every capture epoch: one block, prefix maximum over the epochs (k_leader_scan)
    TIMED(c, BS_KERNEL_LEADER, hipLaunchKernelGGL(k_leader_scan, dim3(1), dim3(kLeaderBlock), 0, c->stream, gr, b));
    launches++;
  }
  // ---- decisions that need no node scan, request vectors, scan tiles
  TIMED(c, BS_KERNEL_QUERY, {
    if (P && inline_tables) {
      const uint32_t qb = cdiv(P, 256);
      const dim3 qg(qb + (nchunks > 1 ? nchunks - 1 : 0));
      switch (ts) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<0>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<1>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<2>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<3>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<4>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_tables<-1>), qg, blk, 0, c->stream, pd, gr, nd, b, bt, prm, forced, nchunks, qb); break;
      }
      launches++;
    } else if (P) {
      const dim3 qg(cdiv(P, 256));
      switch (ts) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<0>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<1>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<2>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<3>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<4>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<-1>), qg, blk, 0, c->stream, pd, gr, b, prm); break;
      }
      launches++;
    }
  });
  if (early_filter) {
    HIPCHK(c, hipEventRecord(c->ev_query, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_query, 0));
    const dim3 fg(cdiv(P, 256));
    switch (ts) {
      case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<0>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
      case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<1>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
      case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<2>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
      case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<3>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
      case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<4>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
      default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fparams_early<-1>), fg, blk, 0, c->stream3, pd, gr, b, prm); break;
    }
    TIMED_ON(c, BS_KERNEL_FILTER, c->stream3, launch_filter(c, c->stream3, pd, nd, b, use_classes));
    HIPCHK(c, hipEventRecord(c->ev_filter, c->stream3));
    launches += 2;
  }
  // ---- running-sum tables of the (class, percent) pairs some query uses
  if (c->M && P) {
    if (!inline_tables) {
      // chunk-local running sums of every table in use; the scan adds the chunk offsets (no fix-up pass)
      TIMED(c, BS_KERNEL_TABLES, launch_tables_nofix(c, dim3(2 * C, nchunks), nd, b, prm, nchunks));
      launches++;
    }
    const bool local = !inline_tables;
    const uint32_t tsplit = inline_tables ? 1u : std::min<uint32_t>(32, 2 * C);
    // the kernel finds out which tiles can be live (per-pod slots, group slots) and deals shares accordingly; the grid
    // only has to offer about one wave per (tile that can be live, table split): the smaller of the two slot ranges is a
    // good guess for cold batches (only first checks) and warm ones (only reservation checks) alike
    const uint32_t tiles_guess = std::max(cdiv(G, 64), std::min(tiles_est, cdiv(G, 64) * 4)) + 1;
    const uint32_t scan_blocks = std::max<uint32_t>(1, cdiv(std::min<uint32_t>(c->general_waves, tiles_guess * tsplit * std::min<uint32_t>(nseg, cdiv(c->M, 64))), 4));
    if (fuse_filter) {
      const uint32_t fblocks = cdiv(std::min<uint32_t>(c->filter_waves, 2 * cdiv(P, 64) * std::max<uint32_t>(1, cdiv(W, 2))), 4);
      TIMED(c, BS_KERNEL_SCAN, launch_scan_filter(c, scan_blocks, fblocks, pd, nd, b, prm, c->M, nseg, P, G, tsplit, local));
    } else {
      TIMED(c, BS_KERNEL_SCAN, launch_scan(c, dim3(scan_blocks), b, prm, c->M, nseg, P, G, tsplit, local));
    }
    launches++;
  } else if (fuse_filter) {
    // no schedulable node: nothing to scan, Filter still has its slots to evaluate
    const uint32_t fblocks = cdiv(std::min<uint32_t>(c->filter_waves, 2 * cdiv(P, 64) * std::max<uint32_t>(1, cdiv(W, 2))), 4);
    TIMED(c, BS_KERNEL_SCAN, launch_scan_filter(c, 1u, fblocks, pd, nd, b, prm, 0u, 1u, P, G, 1u, false));
    launches++;
  }
  // ---- REJECT codes, deny replay, stale-leader propagation, Filter parameters
  TIMED(c, BS_KERNEL_RESOLVE, {
    if (P) {
      hipLaunchKernelGGL(k_reject, dim3(cdiv(P, 256)), blk, 0, c->stream, pd, b);
      const dim3 fg(cdiv(P, 256));
      switch (ts) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<0>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<1>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<2>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<3>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<4>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<-1>), fg, blk, 0, c->stream, pd, gr, nd, b, prm); break;
      }
      launches += 2;
    }
  });
  if (run_filter && P && !early_filter) {
    if (!fuse_filter && W) {       // slot = pod (or class slots without the fused launch): Filter evaluation of the slots k_final filled
      TIMED(c, BS_KERNEL_FILTER, launch_filter(c, c->stream, pd, nd, b, use_classes));
      launches++;
    }
  } else if (early_filter) {
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_filter, 0));
    hipLaunchKernelGGL(k_void_rows, dim3(cdiv(P, 256)), blk, 0, c->stream, pd, b);
    LAUNCHCHK(c, BS_KERNEL_FILTER);
    launches++;
  } else if (P) {
    HIPCHK(c, hipMemsetAsync(b.fl_code, BS_FL_NOT_RUN, P, c->stream));
  }
  // ---- BS_BATCH_FILTER_DENY: Filter's deny entries, in front of the tally
  if (prm.filter_deny && P) {
    c->launches = launches;
    if ((rc = launch_filter_deny(c, pd, gr, nd, b, prm, false))) return rc;
    launches = c->launches;
  }
  // ---- per-pod feasible counts from the slots, per-group admit counts, quorum (last block), re-arm
  const bool do_tally = stages & BS_STAGE_TALLY;
  if (P ? (run_filter || do_tally) : do_tally) {
    TIMED(c, BS_KERNEL_TALLY, {
      hipLaunchKernelGGL(k_tally, dim3(std::max<uint32_t>(1, cdiv(P, kTallyBlock))), dim3(kTallyBlock), 0, c->stream, pd, gr, nd, b,
                         run_filter ? 1u : 0u, do_tally ? 1u : 0u, (do_tally && local_ready) ? 1u : 0u, (do_tally && rearm) ? 1u : 0u);
    });
    launches++;
  }
  if (stages & BS_BATCH_COMMIT) {
    if (prm.filter_deny && P) {
      c->launches = launches;
      if ((rc = launch_filter_deny_marks(c, b, prm, b.first_reject))) return rc;
      launches = c->launches;
    }
    if (G) hipLaunchKernelGGL(k_commit, dim3(cdiv(G, 256)), blk, 0, c->stream, pd, b, prm, const_cast<uint8_t*>(gr.flags), const_cast<uint32_t*>(gr.cls),
                              const_cast<int64_t*>(gr.minres), const_cast<uint32_t*>(gr.mrpres), const_cast<uint64_t*>(gr.occupied), G,
                              (prm.filter_deny && P) ? b.fd_flag : nullptr);
    LAUNCHCHK(c, BS_KERNEL_RESOLVE);
    launches++;
    if ((rc = commit_readback(c, gr, b))) return rc;      // the committed capture may have given every group a pod
    commit_dirty = true;
  }
  c->launches = launches;
  advance_batch_seq(c);
  if (do_tally) {
    c->scratch_armed = rearm;
    c->side_ready = rearm && inline_tables;
  }
  rc = batch_collective(c, stages, gr, b);
  if (rc) return rc;
  if (commit_dirty) return analyse_groups(c);
  return BS_OK;
}

// BS_BATCH_FILTER_DENY, the rare half (bs_fdeny.hpp): the run that just completed turned away a pod somebody else needed (bit 0 of
// its flag words), so its results are not the sequential ones.  Fixed-point iteration: the events a run found are the input of
// the next (k_fd_next -> fd_in, honoured by every chain's classification and by the positional analysis) until a run finds
// exactly the events it was given.  A verdict only depends on events in FRONT of the pod, so positions settle in queue order;
// P + 2 runs is the bound nobody gets near (two runs in practice: one that finds, one that confirms).  A committing batch's
// commit kernels are gated on the device by the same flag word: only the run that is the fixed point commits.
// The results of a run that is thrown away (a wrong guess, a superseded fixed-point iteration) take their hand-over error word with them:
// the re-run reports its own.  The lesson is kept: separate launches from now on.
static void discard_handover(bs_ctx* c) {
  if (c->h_info && ((volatile int32_t*)c->h_info)[12]) {
    ((volatile int32_t*)c->h_info)[12] = 0;
    c->no_fuse_final = 1;
  }
}

static int fd_resolve(bs_ctx* c) {
  if (!c->fd_active) return BS_OK;
  c->fd_active = false;
  c->fd_unsynced = false;                            // (every caller has waited for the batch)
  volatile int32_t* hf = c->h_info + 14;
  if (!hf[0]) return BS_OK;
  int rc = BS_OK;
  bool settled = false;
  const uint32_t stages = c->fd_stages;
  for (uint32_t iter = 1; iter <= c->P + 2 && !settled; ++iter) {
    BatchDev b = batch_dev(c);
    b.fd_in = c->d_fd_in.as<uint32_t>();
    BatchParams prm = batch_params(c);
    prm.seq_inv = c->fd_seq_inv;
    hipLaunchKernelGGL(k_fd_next, dim3(cdiv(std::max<uint32_t>(c->G, 1), 256)), dim3(256), 0, c->stream, b, prm, c->G);
    LAUNCHCHK(c, BS_KERNEL_RESOLVE);
    c->fd_in_live = true;
    c->fd_iter = iter;
    c->epochs_ready = false;                         // the positional analysis depends on fd_in
    discard_handover(c);
    rc = batch_run_inner(c, stages);
    if (rc == BS_OK) rc = hipStreamSynchronize(c->stream) == hipSuccess ? BS_OK : BS_ERR_HIP;
    c->n_fd_reruns++;
    if (rc) break;
    settled = !hf[1];
  }
  c->fd_in_live = false;
  c->fd_iter = 0;
  c->fd_on = false;
  c->epochs_ready = false;                           // (analysed against fd_in: the next batch derives its own)
  if (rc) return rc;
  if (!settled) { c->last_error = "BS_BATCH_FILTER_DENY: the fixed-point iteration did not settle"; return BS_ERR_STATE; }
  return BS_OK;
}

// Every reader of a batch's results comes through here first.  A batch launched on a GUESSED table (see bs_ctx, speculation): wait
// for it, take findMaxPG's real answer (long landed by then) and run the batch again if the guess was wrong.  A
// BS_BATCH_FILTER_DENY batch: wait for it and look at its flag words (fd_resolve).
static int fd_settle(bs_ctx* c) {
  if (!c->fd_active && !c->spec_active) return BS_OK;
  if (!c->batch_since_pods) { c->fd_active = false; c->spec_active = false; return BS_OK; }      // the queue changed since: those results are history
  int rc;
  if (c->last_host_out) {
    if ((rc = wait_host_tag(c, 0, c->host_tag, reinterpret_cast<const int32_t*>(c->h_hout + c->off_htag)))) return rc;
  } else {
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (c->spec_active) {
    c->spec_active = false;
    if ((rc = resolve_groups(c))) return rc;
    if (c->steady_table != c->spec_table) {            // wrong guess: the batch again, on the real answer (nothing of a what-if batch sticks)
      c->n_spec_miss++;
      const bool fd = c->fd_active;
      discard_handover(c);
      if ((rc = batch_run_inner(c, c->spec_stages))) return rc;
      c->fd_active = fd;
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  return fd_resolve(c);
}

int bs_batch_run(bs_ctx* c, uint32_t stages) {
  if (!c) return BS_ERR_INVALID;
  if (!c->have_nodes || !c->have_fit || !c->have_groups || !c->have_pods) {
    c->last_error = "bs_batch_run needs nodes, fit, groups and pods loaded";
    return BS_ERR_STATE;
  }
  if (!(stages & BS_STAGE_PREFILTER)) { c->last_error = "PREFILTER stage is mandatory"; return BS_ERR_INVALID; }
  if ((stages & BS_BATCH_COMMIT) && c->nranks > 1) { c->last_error = "COMMIT is single-rank only"; return BS_ERR_STATE; }
  if (stages & BS_BATCH_FILTER_DENY) {
    if (!(stages & BS_STAGE_FILTER)) { c->last_error = "BS_BATCH_FILTER_DENY needs BS_STAGE_FILTER"; return BS_ERR_INVALID; }
    if (c->nranks > 1 || c->reduce_external) { c->last_error = "BS_BATCH_FILTER_DENY is single-rank only"; return BS_ERR_STATE; }
  }
  int rc = use_device(c);
  if (rc) return rc;
  if ((stages & BS_BATCH_FILTER_DENY) && c->fd_unsynced) {
    // the verdict words of a BS_BATCH_FILTER_DENY batch are pinned and untagged: an earlier such batch nobody has waited for may still
    // store into them (k_fd_apply) after batch_run_inner has reset them from the host — wait for it first
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fd_unsynced = false;
  }
  c->fd_active = false;
  c->fd_iter = 0;
  c->fd_in_live = false;
  c->batch_void = false;                             // a new batch: the verdict on the one before is history
  rc = batch_run_inner(c, stages);
  if (rc == BS_OK && (stages & BS_BATCH_FILTER_DENY) && c->P) c->fd_unsynced = true;
  if (rc == BS_OK && (stages & BS_BATCH_FILTER_DENY) && c->P) {
    c->fd_active = true;
    c->fd_stages = stages;
    if (stages & BS_BATCH_COMMIT) {                  // a committing batch has already waited for its results (commit_readback): settle it now
      HIPCHK(c, hipStreamSynchronize(c->stream));
      rc = fd_resolve(c);
    }
  }
  return rc;
}

int bs_batch_finish(bs_ctx* c) {
  if (!c) return BS_ERR_INVALID;
  if (!c->have_groups) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  if (c->G) hipLaunchKernelGGL(k_ready, dim3(cdiv(c->G, 256)), dim3(256), 0, c->stream, groups_dev(c), batch_dev(c));
  HIPCHK(c, hipGetLastError());
  c->batch_pending_finish = false;
  return BS_OK;
}

// a final block of a fused launch gave up waiting for its producers: the batch's results are not to be trusted
static int check_handover(bs_ctx* c) {
  if (c->h_info && ((volatile int32_t*)c->h_info)[13]) {   // the insert wave of a queue patch ran out of ids (the accounting should make that impossible)
    ((volatile int32_t*)c->h_info)[13] = 0;
    c->pairs_ready = false;                            // classes, pairs and directories are derived again from the resident queue
    c->dirs_ready = false;
    c->rep_valid = false;
    c->ids_used = c->pair_cap;
    const int rc2 = derive_pods(c, false);
    c->n_rederives++;
    c->batch_void = true;                              // whatever ran over the overflowed queue is void until the next bs_batch_run
    c->last_error = "bs_pods_apply: class / pair id space overflowed on the device; the queue was re-derived, run the batch again";
    return rc2 ? rc2 : BS_ERR_RETRY;
  }
  if (c->h_info && ((volatile int32_t*)c->h_info)[12]) {
    ((volatile int32_t*)c->h_info)[12] = 0;
    c->no_fuse_final = 1;                              // from now on: separate launches
    c->batch_void = true;
    c->last_error = "in-launch hand-over timed out (producer blocks not resident): batch void, run it again (the context now uses separate launches)";
    return BS_ERR_RETRY;
  }
  if (c->batch_void) {                                 // (the word was consumed by an earlier look — bs_seq_run's, or a reader's — and no batch has run since)
    c->last_error = "the last batch's results are void (hand-over time-out or queue re-derivation, reported earlier): run the batch again";
    return BS_ERR_RETRY;
  }
  return BS_OK;
}

int bs_batch_sync(bs_ctx* c) {
  if (!c) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = fd_settle(c))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->dstage_busy = false;
  return check_handover(c);
}

// rows the last batch's Filter slots occupy: 2 x request classes (the batch's leader | the leader carried into
// the batch) when classes stand for the pods, one per pod otherwise
static int filter_rows_of(bs_ctx* c, uint32_t* rows) {
  int rc = resolve_pods(c);
  if (rc) return rc;
  // the mode of the last batch over these pods if there was one, else of the batch the loaded state would run
  const bool classes = c->batch_since_pods ? c->last_use_classes : (c->n_uncaptured == 0 && c->n_nominres == 0);
  *rows = c->P ? (classes ? 2 * c->h_K : c->P) : 0;
  if (c->batch_since_pods && c->last_chain == 2) *rows = c->last_rows;        // (leader run, class) rows of the positional chain: never more than P
  return BS_OK;
}

int bs_filter_rows_count(bs_ctx* c, uint32_t* rows) {
  if (!c || !rows) return BS_ERR_INVALID;
  if (!c->have_pods || !c->have_groups) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  return filter_rows_of(c, rows);
}

int bs_batch_read(bs_ctx* c, const bs_batch_out* out) {
  if (!c || !out) return BS_ERR_INVALID;
  if (!c->have_pods || !c->have_groups) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = fd_settle(c))) return rc;
  const uint32_t P = c->P, G = c->G, W = cdiv(c->N, 64);
  const bool filtered = c->last_stages & BS_STAGE_FILTER;
  const bool want_pod = P && (out->pf_code || out->pf_first_k || out->pf_leader || out->fl_code || out->fl_feasible || out->fl_slot);
  const bool want_grp = G && (c->last_stages & BS_STAGE_TALLY) && (out->group_admit || out->group_ready);
  // Filter rows: the slots of the last batch, word-major, compacted to the rows in use
  uint32_t nrows = 0;
  if (out->fl_rows || out->fl_rows_feasible || out->fl_rows_n) {
    if (filtered && (rc = filter_rows_of(c, &nrows))) return rc;
    if (out->fl_rows_n) *out->fl_rows_n = nrows;
    if ((out->fl_rows || out->fl_rows_feasible) && nrows > out->fl_rows_cap) {
      c->last_error = "bs_batch_read: fl_rows_cap is smaller than the rows of the batch";
      return BS_ERR_CAPACITY;
    }
  }
  const bool want_rows = nrows && W && out->fl_rows, want_rfeas = nrows && out->fl_rows_feasible;
  if (nrows > c->filter_slots_cap) { c->last_error = "bs_batch_read: row count exceeds the slot capacity of the batch"; return BS_ERR_STATE; }
  // the pods x nodes bitmap exists only when somebody asks for it
  if (P && W && out->fl_bitmap && filtered && !c->bitmap_valid) {
    HIPCHK(c, c->d_fl_bitmap.reserve(std::max<size_t>(8, (size_t)W * P * 8)));
    hipLaunchKernelGGL(k_filter_expand, dim3(cdiv(P, 256), std::max<uint32_t>(1, cdiv(W, kExpandWords))), dim3(256), 0, c->stream, pods_dev(c),
                       nodes_dev(c), batch_dev(c), W, c->filter_slots_cap);
    LAUNCHCHK(c, BS_KERNEL_FILTER);
    c->bitmap_valid = true;
  }
  // latency mode: the batch wrote its results into pinned host memory itself — poll the completion word, copy out
  if (c->last_host_out && c->batch_since_pods && !(P && out->fl_bitmap && W && filtered)) {
    rc = wait_host_tag(c, 0, c->host_tag, reinterpret_cast<const int32_t*>(c->h_hout + c->off_htag));
    if (rc) return rc;
    if ((rc = check_handover(c))) return rc;
    c->dstage_busy = false;                            // (the batch ran behind every earlier apply)
    const uint8_t* st = c->h_hout;
    if (want_pod) {
      if (out->pf_code) std::memcpy(out->pf_code, st + c->off_pf_code, P);
      if (out->pf_first_k) std::memcpy(out->pf_first_k, st + c->off_pf_first_k, (size_t)P * 4);
      if (out->pf_leader) std::memcpy(out->pf_leader, st + c->off_pf_leader, (size_t)P * 4);
      if (out->fl_code) std::memcpy(out->fl_code, st + c->off_fl_code, P);
      if (out->fl_feasible) std::memcpy(out->fl_feasible, st + c->off_fl_feasible, (size_t)P * 4);
      if (out->fl_slot) std::memcpy(out->fl_slot, st + c->off_fl_slot, (size_t)P * 4);
    }
    if (want_grp) {
      if (out->group_admit) std::memcpy(out->group_admit, st + c->off_admit, (size_t)G * 4);
      if (out->group_ready) std::memcpy(out->group_ready, st + c->off_ready, G);
    }
    if ((want_rows || want_rfeas) && nrows <= c->hstride) {
      if (want_rfeas) std::memcpy(out->fl_rows_feasible, st + c->off_hfeas, (size_t)nrows * 4);
      if (want_rows)
        for (uint32_t w = 0; w < W; ++w) std::memcpy(out->fl_rows + (size_t)w * out->fl_rows_cap, c->h_hrows + (size_t)w * c->hstride, (size_t)nrows * 8);
      return BS_OK;
    }
    if (!(want_rows || want_rfeas)) return BS_OK;
    // more rows than the pinned window holds: the rows (only) take the copy path below
  }
  // ONE wait and at most two copies: the result pack (per-pod arrays | admit | ready, contiguous on the device) and the
  // Filter rows with their feasible counts (a strided window of the slot bitmap; the counts sit behind its last row)
  const bool ext = c->ext_admit != nullptr;
  const size_t pack_bytes = want_grp ? c->outpack_bytes : c->off_admit;
  const size_t off_xadmit = align256(c->outpack_bytes);
  const size_t off_rows = off_xadmit + align256((size_t)G * 4);
  const bool any_rows = want_rows || want_rfeas;
  const size_t rows_h = any_rows ? (size_t)W + 1 : 0, rows_bytes = rows_h * nrows * 8;
  if (off_rows + rows_bytes + 256 > c->h_rstage_cap) {
    if (c->h_rstage) (void)hipHostFree(c->h_rstage);
    c->h_rstage = nullptr; c->h_rstage_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_rstage, off_rows + rows_bytes + 256, hipHostMallocDefault));
    c->h_rstage_cap = off_rows + rows_bytes + 256;
  }
  uint8_t* st = reinterpret_cast<uint8_t*>(c->h_rstage);
  if (want_pod || want_grp) HIPCHK(c, hipMemcpyAsync(st, c->d_outpack.p, pack_bytes, hipMemcpyDeviceToHost, c->stream));
  if (want_grp && ext && out->group_admit) HIPCHK(c, hipMemcpyAsync(st + off_xadmit, c->ext_admit, (size_t)G * 4, hipMemcpyDeviceToHost, c->stream));
  if (any_rows)
    HIPCHK(c, hipMemcpy2DAsync(st + off_rows, (size_t)nrows * 8, c->d_fu_bitmap.p, (size_t)c->filter_slots_cap * 8, (size_t)nrows * 8, rows_h,
                               hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stage_busy = false;                             // (the stream is idle: the pod upload has left its buffer too)
  c->dstage_busy = false;
  if ((rc = check_handover(c))) return rc;
  if (want_pod) {
    if (out->pf_code) std::memcpy(out->pf_code, st + c->off_pf_code, P);
    if (out->pf_first_k) std::memcpy(out->pf_first_k, st + c->off_pf_first_k, (size_t)P * 4);
    if (out->pf_leader) std::memcpy(out->pf_leader, st + c->off_pf_leader, (size_t)P * 4);
    if (out->fl_code) std::memcpy(out->fl_code, st + c->off_fl_code, P);
    if (out->fl_feasible) std::memcpy(out->fl_feasible, st + c->off_fl_feasible, (size_t)P * 4);
    if (out->fl_slot) std::memcpy(out->fl_slot, st + c->off_fl_slot, (size_t)P * 4);
  }
  if (want_grp) {
    if (out->group_admit) std::memcpy(out->group_admit, st + (ext ? off_xadmit : c->off_admit), (size_t)G * 4);
    if (out->group_ready) std::memcpy(out->group_ready, st + c->off_ready, G);
  }
  if (want_rfeas) std::memcpy(out->fl_rows_feasible, st + off_rows + (size_t)W * nrows * 8, (size_t)nrows * 4);
  if (want_rows)
    for (uint32_t w = 0; w < W; ++w) std::memcpy(out->fl_rows + (size_t)w * out->fl_rows_cap, st + off_rows + (size_t)w * nrows * 8, (size_t)nrows * 8);
  if (P && out->fl_bitmap && W) {
    if (filtered) HIPCHK(c, hipMemcpy(out->fl_bitmap, c->d_fl_bitmap.p, (size_t)W * P * 8, hipMemcpyDeviceToHost));
    else std::memset(out->fl_bitmap, 0, (size_t)W * P * 8);
  }
  return BS_OK;
}

// Zero-copy results of a latency-mode batch: pointers into the pinned memory the last launch wrote.
int bs_batch_map(bs_ctx* c, bs_batch_view* v) {
  if (!c || !v) return BS_ERR_INVALID;
  if (!c->have_pods || !c->have_groups) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = fd_settle(c))) return rc;                  // (a BS_BATCH_FILTER_DENY batch that had to be re-run may have left the three-launch chains)
  if (!c->last_host_out || !c->batch_since_pods) {
    c->last_error = "bs_batch_map: the last batch did not write host results (BS_BATCH_HOST_RESULTS on a three-launch chain of a single-rank context)";
    return BS_ERR_STATE;
  }
  const uint32_t P = c->P, G = c->G, W = cdiv(c->N, 64);
  const bool filtered = c->last_stages & BS_STAGE_FILTER;
  uint32_t nrows = 0;
  if (filtered && (rc = filter_rows_of(c, &nrows))) return rc;
  if ((rc = wait_host_tag(c, 0, c->host_tag, reinterpret_cast<const int32_t*>(c->h_hout + c->off_htag)))) return rc;
  if ((rc = check_handover(c))) return rc;
  c->dstage_busy = false;                              // (the batch ran behind every earlier apply)
  const uint8_t* st = c->h_hout;
  std::memset(v, 0, sizeof(*v));
  v->p = P; v->g = G; v->words = W;
  v->pf_code = st + c->off_pf_code;
  v->pf_first_k = reinterpret_cast<const uint32_t*>(st + c->off_pf_first_k);
  v->pf_leader = reinterpret_cast<const int32_t*>(st + c->off_pf_leader);
  v->fl_code = st + c->off_fl_code;
  v->fl_feasible = reinterpret_cast<const uint32_t*>(st + c->off_fl_feasible);
  v->fl_slot = reinterpret_cast<const uint32_t*>(st + c->off_fl_slot);
  if (c->last_stages & BS_STAGE_TALLY) {
    v->group_admit = reinterpret_cast<const uint32_t*>(st + c->off_admit);
    v->group_ready = st + c->off_ready;
  }
  v->fl_rows_n = nrows;
  v->fl_rows_stride = c->hstride;
  if (filtered && nrows && nrows <= c->hstride) {
    v->fl_rows = c->h_hrows;
    v->fl_rows_feasible = reinterpret_cast<const uint32_t*>(st + c->off_hfeas);
  }
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
// single queries
// -------------------------------------------------------------------------------------------------
int bs_node_left(bs_ctx* c, uint32_t cls, float percent, int64_t* left, uint32_t* present) {
  if (!c || !left || !present) return BS_ERR_INVALID;
  if (!c->have_nodes || !c->have_fit) return BS_ERR_STATE;
  if (cls >= c->C) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t N = c->N, L = c->L;
  if (!N) return BS_OK;
  DevBuf d_left, d_pres;
  HIPCHK(c, d_left.reserve((size_t)N * L * 8));
  HIPCHK(c, d_pres.reserve((size_t)N * 4));
  hipLaunchKernelGGL(k_node_left, dim3(cdiv(N, 256)), dim3(256), 0, c->stream, nodes_dev(c), cls, percent, L, d_left.as<int64_t>(), d_pres.as<uint32_t>());
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(left, d_left.p, (size_t)N * L * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(present, d_pres.p, (size_t)N * 4, hipMemcpyDeviceToHost));
  return BS_OK;
}

int bs_scan_prefix(bs_ctx* c, uint32_t cls, float percent, int64_t* prefix, uint32_t* present, uint32_t* node_index, uint32_t* rows) {
  if (!c || !prefix || !present || !node_index || !rows) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  uint32_t slot;
  rc = build_scratch_table(c, cls, percent, &slot);
  if (rc) return rc;
  const uint32_t M = c->M, L = c->L, LP = c->LP, N = c->N;
  std::vector<int64_t> t((size_t)M * LP);
  uint32_t kp[16];
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (M) HIPCHK(c, hipMemcpy(t.data(), c->d_tables.as<int64_t>() + (size_t)slot * c->table_mcap * LP, (size_t)M * LP * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(kp, c->d_kp.as<uint32_t>() + (size_t)slot * 16, sizeof(kp), hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < M; ++k) {
    for (uint32_t j = 0; j < L; ++j) prefix[(size_t)j * N + k] = t[(size_t)k * LP + j];
    uint32_t pr = 0;
    for (uint32_t s = 0; s < c->S; ++s) if (k >= kp[s]) pr |= 1u << s;
    present[k] = pr;
    node_index[k] = c->h_kmap[k];
  }
  *rows = M;
  return BS_OK;
}

int bs_cluster_total(bs_ctx* c, uint32_t cls, int64_t* total, uint32_t* present) {
  if (!c || !total || !present) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  uint32_t slot;
  rc = build_scratch_table(c, cls, 1.0f, &slot);
  if (rc) return rc;
  const uint32_t M = c->M, L = c->L, LP = c->LP;
  std::vector<int64_t> row(LP, 0);
  uint32_t kp[16];
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (M) HIPCHK(c, hipMemcpy(row.data(), c->d_tables.as<int64_t>() + ((size_t)slot * c->table_mcap + (M - 1)) * LP, (size_t)LP * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(kp, c->d_kp.as<uint32_t>() + (size_t)slot * 16, sizeof(kp), hipMemcpyDeviceToHost));
  uint32_t pr = 0;
  for (uint32_t j = 0; j < L; ++j) total[j] = M ? row[j] : 0;
  for (uint32_t s = 0; s < c->S; ++s) if (M && kp[s] != BS_INF) pr |= 1u << s;
  *present = pr;
  return BS_OK;
}

int bs_cluster_fits(bs_ctx* c, uint32_t cls, float percent, const int64_t* req, uint32_t req_present, uint8_t* fits, uint32_t* first_k) {
  if (!c || !req || !fits) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  uint32_t slot;
  rc = build_scratch_table(c, cls, percent, &slot);
  if (rc) return rc;
  const uint32_t L = c->L, LP = c->LP, S = c->S, M = c->M;
  // scratch layout (bytes): 128 qtab | 256 qflags | 320 first_row | 512 qreq[LP]   (one request slot)
  uint8_t* sq = c->d_sq.as<uint8_t>();
  const int32_t qtab = (int32_t)slot;
  uint32_t inf = BS_INF, qflags = 0;
  int64_t q[BS_MAX_LANES];
  for (uint32_t j = 0; j < LP; ++j) q[j] = INT64_MIN;
  for (uint32_t j = 0; j < 4; ++j) q[j] = req[j];
  if (!c->cfg.eph_gate) q[BS_LANE_EPH] = 0;    // a Resource built by Add never carries ephemeral-storage with the gate off
  uint32_t absok = 0;
  for (uint32_t s = 0; s < S; ++s) {
    const bool pres = req_present & (1u << s);
    if (!pres || req[4 + s] == 0) absok |= 1u << s;
    q[4 + s] = pres ? req[4 + s] : INT64_MIN;
  }
  qflags = (req_present & 0xFFFu) | (absok << 16);
  HIPCHK(c, hipMemcpyAsync(sq + 128, &qtab, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sq + 256, &qflags, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sq + 320, &inf, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sq + 512, q, (size_t)LP * 8, hipMemcpyHostToDevice, c->stream));
  BatchDev b = batch_dev(c);
  b.qtab_s = reinterpret_cast<int32_t*>(sq + 128);
  b.qflags_s = reinterpret_cast<uint32_t*>(sq + 256);
  b.first_row = reinterpret_cast<uint32_t*>(sq + 320);
  b.qreq_s = reinterpret_cast<int64_t*>(sq + 512);
  BatchParams prm = batch_params(c);
  prm.collect_stats = 0;
  if (M) {
    const uint32_t nseg = std::min<uint32_t>(cdiv(M, 64), 64);      // waves sharing the live groups of the one query
    launch_scan(c, dim3(cdiv(nseg, 4)), b, prm, M, nseg, 1u, 0u, 1u);
    HIPCHK(c, hipGetLastError());
  }
  uint32_t row = BS_INF;
  HIPCHK(c, hipMemcpyAsync(&row, sq + 320, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *fits = row != BS_INF;
  if (first_k) *first_k = row == BS_INF ? BS_K_NONE : c->h_kmap[row];
  (void)L;
  return BS_OK;
}

int bs_find_max_pg(bs_ctx* c, int32_t* leader, uint32_t* finished, uint8_t* panic) {
  if (!c || !leader || !panic) return BS_ERR_INVALID;
  if (!c->have_groups) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  GroupsDev gr = groups_dev(c);
  BatchDev b = batch_dev(c);
  BatchParams prm = batch_params(c);
  prm.C = 0;
  prm.collect_stats = 0;
  const uint32_t one = 1;
  hipLaunchKernelGGL(k_init, dim3(cdiv(std::max<uint32_t>(c->G, 8), 256)), dim3(256), 0, c->stream, gr, b);
  c->scratch_armed = false;
  HIPCHK(c, hipMemcpyAsync(b.nepochs, &one, 4, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_leader, dim3(1), dim3(kLeaderBlock), 0, c->stream, gr, b);
  HIPCHK(c, hipGetLastError());
  int32_t l = -1;
  uint8_t pn = 0;
  HIPCHK(c, hipMemcpyAsync(&l, b.leader_epoch, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&pn, b.panic_epoch, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *leader = l;
  *panic = pn;
  if (finished) *finished = 0;   // maxFinished is unused by every caller in the reference (core.go:120 discards it)
  return BS_OK;
}

int bs_filter_one(bs_ctx* c, int32_t pod_group, const int64_t* pod_req, uint32_t pod_req_present, int32_t leader,
                  uint32_t node, uint8_t* fl_code, uint8_t* fn_code) {
  // Implemented on top of the batch path: a one-pod batch whose PreFilter is forced to pass is not
  // expressible, so this entry point evaluates through k_filter_params/k_filter with a one-pod view.
  if (!c || !pod_req || !fl_code || !fn_code) return BS_ERR_INVALID;
  if (!c->have_nodes || !c->have_groups) return BS_ERR_STATE;
  if (leader >= 0 && (uint32_t)leader >= c->G) { c->last_error = "bs_filter_one: leader out of range"; return BS_ERR_INVALID; }
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t L = c->L, N = c->N;
  // one-pod SoA in scratch device memory
  DevBuf d;
  HIPCHK(c, d.reserve(4096 + (size_t)cdiv(std::max<uint32_t>(N, 1), 64) * 8));
  uint8_t* base = d.as<uint8_t>();
  int32_t grp = pod_group;
  uint8_t pf = BS_PF_PASS_NO_MAX, zero8 = 0;
  uint32_t zero32 = 0;
  uint64_t zero64 = 0;
  HIPCHK(c, hipMemcpyAsync(base + 0, &grp, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 64, pod_req, (size_t)L * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 256, &pod_req_present, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 320, &zero32, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 384, &zero64, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 448, &zero8, 1, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 512, &pf, 1, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + 576, &leader, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(base + 640, 0, 256, c->stream));
  PodsDev pd{};
  pd.p = 1;
  pd.group = reinterpret_cast<int32_t*>(base + 0);
  pd.req = reinterpret_cast<int64_t*>(base + 64);
  pd.pres = reinterpret_cast<uint32_t*>(base + 256);
  pd.cls = reinterpret_cast<uint32_t*>(base + 320);
  pd.owner = reinterpret_cast<uint64_t*>(base + 384);
  pd.flags = base + 448;
  BatchDev b = batch_dev(c);
  b.pf_code = base + 512;
  b.pf_leader = reinterpret_cast<int32_t*>(base + 576);
  b.fl_code = base + 640;
  b.fflags = reinterpret_cast<uint32_t*>(base + 704);
  b.fl_feasible = reinterpret_cast<uint32_t*>(base + 768);
  b.fparams = reinterpret_cast<int64_t*>(base + 832);
  b.fl_bitmap = reinterpret_cast<uint64_t*>(base + 4096);
  b.fu_slot = reinterpret_cast<uint32_t*>(base + 960);
  b.uparams = b.fparams;                           // one pod, one slot: the slot arrays are the pod's
  b.uflags = b.fflags;
  b.fu_bitmap = b.fl_bitmap;
  b.fu_feas = b.fl_feasible;
  // first_elig must not redirect MinResources for a stand-alone query: use INF for every group
  DevBuf fe;
  HIPCHK(c, fe.reserve(std::max<size_t>(4, (size_t)c->G * 4)));
  HIPCHK(c, hipMemsetAsync(fe.p, 0xFF, std::max<size_t>(4, (size_t)c->G * 4), c->stream));
  b.first_elig = fe.as<uint32_t>();
  BatchParams prm = batch_params(c);
  GroupsDev gr = groups_dev(c);
  NodesDev nd = nodes_dev(c);
  hipLaunchKernelGGL(k_filter_params, dim3(1), dim3(256), 0, c->stream, pd, gr, b, prm);
  const uint32_t W = cdiv(N, 64);
  if (W) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_filter<2>), dim3(1), dim3(64), 0, c->stream, pd, nd, b, 1u, 0u, 1u, 0u);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_filter_one, dim3(1), dim3(64), 0, c->stream, nd, b, node, base + 768 + 16);
  HIPCHK(c, hipGetLastError());
  uint8_t fl = 0, fn = 0;
  uint64_t word = 0;
  HIPCHK(c, hipMemcpyAsync(&fl, base + 640, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&fn, base + 768 + 16, 1, hipMemcpyDeviceToHost, c->stream));
  if (node < N) HIPCHK(c, hipMemcpyAsync(&word, base + 4096 + (size_t)(node >> 6) * 8, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *fl_code = fl;
  *fn_code = fn;
  if (fl == BS_FL_EVALUATED) {
    // the batched bitmap and the per-pair kernel must agree on pass / fail
    const bool pass = node < N && ((word >> (node & 63)) & 1ull);
    if (pass != (fn < 16u)) { c->last_error = "bitmap / filter_one disagreement"; return BS_ERR_HIP; }
  }
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
// churn (BASELINE config 5): list edits on the host mirror, then a suffix re-upload
// -------------------------------------------------------------------------------------------------
int bs_nodes_apply(bs_ctx* c, const bs_node_delta* deltas, uint32_t count) {
  if (!c || (count && !deltas)) return BS_ERR_INVALID;
  if (!c->have_nodes || !c->have_fit) return BS_ERR_STATE;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  const uint32_t L = c->L, C = c->C;
  // unpack fit bits to one byte vector per class for easy insert/erase
  uint32_t N = c->N;
  std::vector<std::vector<uint8_t>> fit(C, std::vector<uint8_t>(N));
  for (uint32_t cl = 0; cl < C; ++cl)
    for (uint32_t n = 0; n < N; ++n) fit[cl][n] = (c->h_fit[(size_t)cl * c->fit_words + (n >> 5)] >> (n & 31)) & 1u;
  std::vector<std::vector<int64_t>> al(L), rq(L);
  for (uint32_t j = 0; j < L; ++j) {
    al[j].assign(c->h_alloc.begin() + (size_t)j * N, c->h_alloc.begin() + (size_t)(j + 1) * N);
    rq[j].assign(c->h_nreq.begin() + (size_t)j * N, c->h_nreq.begin() + (size_t)(j + 1) * N);
  }
  // presence / flag mirrors are edited in copies too: an invalid delta anywhere in the list leaves the context untouched
  std::vector<uint32_t> apres = c->h_apres, rpres = c->h_rpres;
  std::vector<uint8_t> nflags = c->h_nflags;
  uint32_t lo = N;                     // first list index whose content changes
  for (uint32_t d = 0; d < count; ++d) {
    const bs_node_delta& x = deltas[d];
    lo = std::min(lo, x.kind == BS_DELTA_APPEND ? N : x.index);
    if (x.kind == BS_DELTA_REMOVE) {
      if (x.index >= N) return BS_ERR_INVALID;
      for (uint32_t j = 0; j < L; ++j) { al[j].erase(al[j].begin() + x.index); rq[j].erase(rq[j].begin() + x.index); }
      apres.erase(apres.begin() + x.index);
      rpres.erase(rpres.begin() + x.index);
      nflags.erase(nflags.begin() + x.index);
      for (uint32_t cl = 0; cl < C; ++cl) fit[cl].erase(fit[cl].begin() + x.index);
      --N;
      continue;
    }
    uint32_t at = x.index;
    if (x.kind == BS_DELTA_APPEND) {
      at = N++;
      for (uint32_t j = 0; j < L; ++j) { al[j].push_back(0); rq[j].push_back(0); }
      apres.push_back(0); rpres.push_back(0); nflags.push_back(0);
      for (uint32_t cl = 0; cl < C; ++cl) fit[cl].push_back(0);
    } else if (x.kind != BS_DELTA_UPDATE || at >= N) {
      return BS_ERR_INVALID;
    }
    for (uint32_t j = 0; j < L; ++j) { al[j][at] = x.allocatable[j]; rq[j][at] = x.requested[j]; }
    apres[at] = x.allocatable_present;
    rpres[at] = x.requested_present;
    nflags[at] = (uint8_t)x.flags;
    if (x.n_fit_exceptions > 8) return BS_ERR_INVALID;
    for (uint32_t cl = 0; cl < C; ++cl) fit[cl][at] = x.fit_default ? 1 : 0;
    for (uint32_t e = 0; e < x.n_fit_exceptions; ++e) {
      if (x.fit_exceptions[e] >= C) return BS_ERR_INVALID;
      fit[x.fit_exceptions[e]][at] = x.fit_default ? 0 : 1;
    }
  }
  c->N = N;
  c->h_apres.swap(apres);
  c->h_rpres.swap(rpres);
  c->h_nflags.swap(nflags);
  c->h_alloc.resize((size_t)L * N);
  c->h_nreq.resize((size_t)L * N);
  for (uint32_t j = 0; j < L; ++j) {
    std::copy(al[j].begin(), al[j].end(), c->h_alloc.begin() + (size_t)j * N);
    std::copy(rq[j].begin(), rq[j].end(), c->h_nreq.begin() + (size_t)j * N);
  }
  c->fit_words = cdiv(N, 32);
  c->h_fit.assign((size_t)C * c->fit_words, 0);
  for (uint32_t cl = 0; cl < C; ++cl)
    for (uint32_t n = 0; n < N; ++n)
      if (fit[cl][n]) c->h_fit[(size_t)cl * c->fit_words + (n >> 5)] |= 1u << (n & 31);
  rc = upload_nodes(c, lo);            // suffix upload + re-derive from the first changed index
  if (rc) return rc;
  rc = upload_fit(c);
  if (rc == BS_OK && c->have_groups) rc = analyse_groups(c);
  return rc;
}

int bs_nodes_assume(bs_ctx* c, const bs_node_request* reqs, uint32_t count) {
  if (!c || (count && !reqs)) return BS_ERR_INVALID;
  if (!c->have_nodes) { c->last_error = "bs_nodes_assume before bs_nodes_load"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  if (!count) return BS_OK;
  const uint32_t N = c->N, L = c->L;
  {
    std::vector<uint32_t> seen(count);
    for (uint32_t d = 0; d < count; ++d) {
      if (reqs[d].index >= N) { c->last_error = "bs_nodes_assume: node index out of range"; return BS_ERR_INVALID; }
      if (reqs[d].requested_present >> c->S) { c->last_error = "bs_nodes_assume: present bit beyond the scalar lanes"; return BS_ERR_INVALID; }
      seen[d] = reqs[d].index;
    }
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { c->last_error = "bs_nodes_assume: a node index appears twice"; return BS_ERR_INVALID; }
  }
  static_assert(sizeof(bs_node_request) == sizeof(NodeRequest), "node request layout");
  const size_t bytes = (size_t)count * sizeof(bs_node_request);
  if (c->nstage_busy) { HIPCHK(c, hipEventSynchronize(c->ev_nstage)); c->nstage_busy = false; }
  if (bytes > c->h_nstage_cap) {
    if (c->h_nstage) (void)hipHostFree(c->h_nstage);
    c->h_nstage = nullptr; c->h_nstage_cap = 0;
    const size_t want = std::max<size_t>(2 * bytes, 16 << 10);
    HIPCHK(c, hipHostMalloc(&c->h_nstage, want, hipHostMallocDefault));
    c->h_nstage_cap = want;
  }
  std::memcpy(c->h_nstage, reqs, bytes);
  for (uint32_t d = 0; d < count; ++d) {                          // the host mirror a later bs_nodes_apply starts from
    for (uint32_t j = 0; j < L; ++j) c->h_nreq[(size_t)j * N + reqs[d].index] = reqs[d].requested[j];
    c->h_rpres[reqs[d].index] = reqs[d].requested_present;
  }
  hipLaunchKernelGGL(k_nodes_assume, dim3(cdiv(count, 256)), dim3(256), 0, c->stream, reinterpret_cast<const NodeRequest*>(c->h_nstage), count, L, c->Ncap,
                     c->d_alloc.as<int64_t>(), c->d_nreq.as<int64_t>(), c->d_rpres.as<uint32_t>(), c->d_nflags.as<uint8_t>(), c->d_left4.as<int64_t>(),
                     c->d_lglob.as<int64_t>());
  LAUNCHCHK(c, BS_KERNEL_PREPASS);
  if (!c->ev_nstage) HIPCHK(c, hipEventCreateWithFlags(&c->ev_nstage, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_nstage, c->stream));
  c->nstage_busy = true;
  c->bitmap_valid = false;
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
// the sequential scheduling pass (bs_seq.hpp): one persistent workgroup walks the resident queue
// -------------------------------------------------------------------------------------------------
int bs_nodes_read(bs_ctx* c, int64_t* requested, uint32_t* requested_present) {
  if (!c || !requested || !requested_present) return BS_ERR_INVALID;
  if (!c->have_nodes) { c->last_error = "bs_nodes_read before bs_nodes_load"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  const uint32_t N = c->N, L = c->L;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (N) {
    HIPCHK(c, hipMemcpy2D(requested, (size_t)N * 8, c->d_nreq.p, (size_t)c->Ncap * 8, (size_t)N * 8, L, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(requested_present, c->d_rpres.p, (size_t)N * 4, hipMemcpyDeviceToHost));
  }
  return BS_OK;
}

int bs_seq_run(bs_ctx* c, uint32_t stages, bs_seq_out* out) {
  if (!c || !out) return BS_ERR_INVALID;
  if (!c->have_nodes || !c->have_fit || !c->have_groups || !c->have_pods) {
    c->last_error = "bs_seq_run needs nodes, fit, groups and pods loaded";
    return BS_ERR_STATE;
  }
  if (!(stages & BS_STAGE_PREFILTER)) { c->last_error = "PREFILTER stage is mandatory"; return BS_ERR_INVALID; }
  if ((stages & BS_BATCH_FILTER_DENY) && !(stages & BS_STAGE_FILTER)) { c->last_error = "BS_BATCH_FILTER_DENY needs BS_STAGE_FILTER"; return BS_ERR_INVALID; }
  if (c->nranks > 1 || c->reduce_external) { c->last_error = "bs_seq_run is single-rank only (a sequential pass does not shard)"; return BS_ERR_STATE; }
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = settle_pending(c))) return rc;
  const uint32_t P = c->P, G = c->G, N = c->N, C = c->C, L = c->L;
  if ((G > c->n_uncaptured && c->max_group_cls >= C) || (P && c->max_pod_cls >= C)) {
    c->last_error = "fit class index out of range (groups.cls / pods.cls vs the loaded fit classes)";
    return BS_ERR_INVALID;
  }
  if (G > 0x7FFFFFF0u) return BS_ERR_CAPACITY;
  // the first-fit cursors are keyed by the resident queue's request classes: a queue patch whose insert wave ran out of class ids
  // (h_info[13], set by the device) left them unusable until the queue is re-derived — check_handover does that
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // An unread batch's error words are looked at here (the id overflow concerns this pass: check_handover re-derives the queue) but they stay
  // that batch's: batch_void keeps every later read of it failing with BS_ERR_RETRY until bs_batch_run starts a new one.  The pass itself
  // reads no batch result and goes on.
  if ((rc = check_handover(c)) && rc != BS_ERR_RETRY) return rc;
  if (rc == BS_ERR_RETRY) c->last_error.clear();            // (nothing failed for THIS call)
  out->n_released = 0;
  out->total_ns = 0;
  out->node_picks = out->node_scans = out->scan_rounds = out->pick_rounds = out->leader_folds = out->table_builds = 0;
  // ---- scratch: one allocation
  const size_t nP = std::max<uint32_t>(P, 1), nG = std::max<uint32_t>(G, 1), cap = std::max<uint32_t>(out->cap, 1), stride = std::max<uint32_t>(c->Ncap, 1);
  size_t o = 0;
  const size_t o_sc07 = o; o = align256(o + stride * L * 8);
  const size_t o_sc10 = o; o = align256(o + stride * L * 8);
  const size_t o_meta = o; o = align256(o + stride * 4);
  const size_t o_keys = o; o = align256(o + nG * 8);
  const size_t o_wait = o; o = align256(o + nP * 8);
  const size_t o_head = o; o = align256(o + nG * 4);
  const size_t o_nwait = o; o = align256(o + nG * 4);
  const size_t o_slot = o; o = align256(o + nG * 4);
  const size_t o_tfirst = o; o = align256(o + nG * 8);
  const size_t o_res = o;                                   // results: one D2H
  const size_t o_code = o; o = align256(o + nP);
  const size_t o_node = o; o = align256(o + nP * 4);
  const size_t o_fk = o; o = align256(o + nP * 4);
  const size_t o_leader = o; o = align256(o + nP * 4);
  const size_t o_lperm = o; o = align256(o + nP);
  const size_t o_rg = o; o = align256(o + cap * 4);
  const size_t o_rp = o; o = align256(o + cap * 4);
  const size_t o_ft = o; o = align256(o + cap * 8);
  const size_t o_rt = o; o = align256(o + cap * 8);
  const size_t o_info = o; o = align256(o + 512);
  HIPCHK(c, c->d_seq.reserve(o));
  uint8_t* base = c->d_seq.as<uint8_t>();
  GroupsDev gr = groups_dev(c);
  SeqDev sq{};
  sq.nreq = c->d_nreq.as<int64_t>();
  sq.rpres = c->d_rpres.as<uint32_t>();
  sq.g_matched = const_cast<uint32_t*>(gr.matched);
  sq.g_sc = const_cast<uint32_t*>(gr.status_scheduled);
  sq.g_flags = const_cast<uint8_t*>(gr.flags);
  sq.g_cls = const_cast<uint32_t*>(gr.cls);
  sq.g_minres = const_cast<int64_t*>(gr.minres);
  sq.g_mrpres = const_cast<uint32_t*>(gr.mrpres);
  sq.g_occ = const_cast<uint64_t*>(gr.occupied);
  sq.left07 = reinterpret_cast<int64_t*>(base + o_sc07);
  sq.left10 = reinterpret_cast<int64_t*>(base + o_sc10);
  sq.nmeta = reinterpret_cast<uint32_t*>(base + o_meta);
  sq.keys = reinterpret_cast<unsigned long long*>(base + o_keys);
  sq.wait_rec = reinterpret_cast<unsigned long long*>(base + o_wait);
  sq.head = reinterpret_cast<uint32_t*>(base + o_head);
  sq.nwait = reinterpret_cast<uint32_t*>(base + o_nwait);
  sq.slot_of = reinterpret_cast<uint32_t*>(base + o_slot);
  sq.t_first = reinterpret_cast<unsigned long long*>(base + o_tfirst);
  sq.pclass = pclass_dev(c);
  sq.pf_code = base + o_code;
  sq.pod_node = reinterpret_cast<int32_t*>(base + o_node);
  sq.pf_first_k = reinterpret_cast<uint32_t*>(base + o_fk);
  sq.pf_leader = reinterpret_cast<int32_t*>(base + o_leader);
  sq.last_permitted = base + o_lperm;
  sq.released_group = reinterpret_cast<uint32_t*>(base + o_rg);
  sq.released_pods = reinterpret_cast<uint32_t*>(base + o_rp);
  sq.first_tick = reinterpret_cast<unsigned long long*>(base + o_ft);
  sq.ready_tick = reinterpret_cast<unsigned long long*>(base + o_rt);
  sq.cap = out->cap;
  sq.info = reinterpret_cast<unsigned long long*>(base + o_info);
  SeqParams prm{};
  prm.S = c->S;
  prm.eph_gate = c->cfg.eph_gate;
  prm.run_filter = (stages & BS_STAGE_FILTER) ? 1u : 0u;
  prm.filter_deny = (stages & BS_BATCH_FILTER_DENY) ? 1u : 0u;
  prm.C = C;
  prm.sop_leader0 = c->sop_leader0;
  prm.keys_in_lds = G <= kSeqKeysLds ? 1u : 0u;
  prm.prune = cdiv(N, 64) <= kSeqPruneTiles ? 1u : 0u;
  size_t lds = prm.keys_in_lds ? align256((size_t)nG * 8) : 0;
  {
    // table summaries: as many slots as the CU's LDS holds behind the static arrays and the key window (one thread per tile: <= 1024 tiles)
    const size_t T = cdiv(N, 64), per_slot = T * ((size_t)L * 24 + 8), query = 0;
    const size_t budget = (size_t)160 * 1024 - sizeof(SeqShared) - 2048;
    uint32_t K = 0;
    if (T && T <= (size_t)kSeqPruneTiles && budget > lds + query + per_slot) K = (uint32_t)std::min<size_t>(kSeqCacheSlots, (budget - lds - query) / per_slot);
    if (const char* e = std::getenv("BS_SEQ_CACHE_SLOTS")) K = std::min<uint32_t>(K, (uint32_t)std::max(0, std::atoi(e)));   // tests: 0 = the round scan, 1 = thrash one slot
    prm.cache_slots = K;
    prm.cache_off = (uint32_t)lds;
    if (K) lds += align256(K * per_slot + query + 64);
  }
  // first-fit cursors per request class (bs_seq.hpp, seq_pick): BS_SEQ_NO_CURSOR=1 = every search starts at the head of the list
  prm.use_cursor = (P && sq.pclass && !(std::getenv("BS_SEQ_NO_CURSOR") && std::atoi(std::getenv("BS_SEQ_NO_CURSOR")))) ? 1u : 0u;
  HIPCHK(c, hipMemsetAsync(base + o_info, 0, 512, c->stream));
  const PodsDev pd = pods_dev(c);
  const NodesDev nd = nodes_dev(c);
  launch_seq(c->stream, c->S, lds, pd, gr, nd, sq, prm);
  LAUNCHCHK(c, BS_KERNEL_QUERY);
  // ---- results: one copy of the whole result block, then the caller's arrays
  std::vector<uint8_t> res(o - o_res);
  HIPCHK(c, hipMemcpyAsync(res.data(), base + o_res, res.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint8_t* rb = res.data() - o_res;
  const unsigned long long* info = reinterpret_cast<const unsigned long long*>(rb + o_info);
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz
  auto to_ns = [&](unsigned long long ticks) { return (int64_t)((long double)ticks * 1.0e6L / (long double)khz); };
  out->n_released = (uint32_t)info[0];
  out->total_ns = to_ns(info[1]);
  out->node_picks = info[2];
  out->node_scans = info[3];
  out->scan_rounds = info[5];
  out->pick_rounds = info[6];
  out->leader_folds = info[7] & ((1ull << 40) - 1ull);
  out->table_builds = info[7] >> 40;
  if (const char* e = std::getenv("BS_SEQ_PROBE_PRINT")) {   // probe build: cycles per phase (see bs_seq.hpp)
    if (std::atoi(e)) std::fprintf(stderr, "seq probe cycles: control %llu capture %llu fold %llu scan %llu pick %llu permit %llu top-barrier %llu\n", info[8], info[9],
                                   info[10], info[11], info[12], info[13], info[14]);
    if (std::atoi(e)) std::fprintf(stderr, "  scan rounds (thread 0): issue-next-loads %llu select %llu wave-scans %llu lds-writes %llu barrier %llu fk-check %llu offsets+compare %llu tail %llu\n",
                                   info[16], info[17], info[18], info[19], info[20], info[21], info[22], info[23]);
    if (std::atoi(e)) {                                      // the finer split of thread 0's time (BS_SEQ_P in bs_seq.hpp)
      static const char* nm[17] = {"top-barrier", "group-loads", "control", "scan:drain", "scan:slot", "scan:candidates", "scan:tiles", "scan:barrier+min", "scan:tail",
                                   "pick:request", "pick:drain", "pick:tiles", "pick:barrier+min", "pick:assume", "summaries", "result-stores", "permit"};
      std::fprintf(stderr, "  thread 0, cycles:");
      for (int k2 = 0; k2 < 17; ++k2) std::fprintf(stderr, " %s %llu |", nm[k2], info[32 + k2]);
      std::fprintf(stderr, "\n");
    }
  }
  if (P) {
    if (out->pf_code) std::memcpy(out->pf_code, rb + o_code, P);
    if (out->pod_node) std::memcpy(out->pod_node, rb + o_node, (size_t)P * 4);
    if (out->pf_first_k) std::memcpy(out->pf_first_k, rb + o_fk, (size_t)P * 4);
    if (out->pf_leader) std::memcpy(out->pf_leader, rb + o_leader, (size_t)P * 4);
    if (out->last_permitted) { if (prm.filter_deny) std::memcpy(out->last_permitted, rb + o_lperm, P); else std::memset(out->last_permitted, 0, P); }
    c->sop_leader0 = (int32_t)(uint32_t)info[4] - 1;         // sop.maxFinishedPG as the pass left it
  }
  const uint32_t k = std::min(out->n_released, out->cap);
  if (k) {
    if (out->released_group) std::memcpy(out->released_group, rb + o_rg, (size_t)k * 4);
    if (out->released_pods) std::memcpy(out->released_pods, rb + o_rp, (size_t)k * 4);
    const unsigned long long* ft = reinterpret_cast<const unsigned long long*>(rb + o_ft);
    const unsigned long long* rt = reinterpret_cast<const unsigned long long*>(rb + o_rt);
    for (uint32_t i = 0; i < k; ++i) {
      if (out->first_ns) out->first_ns[i] = to_ns(ft[i]);
      if (out->ready_ns) out->ready_ns[i] = to_ns(rt[i]);
    }
  }
  // ---- the host mirrors and everything derived from the state the pass rewrote
  if (N && P) {
    HIPCHK(c, hipMemcpy2D(c->h_nreq.data(), (size_t)N * 8, c->d_nreq.p, (size_t)c->Ncap * 8, (size_t)N * 8, L, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(c->h_rpres.data(), c->d_rpres.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_nodes_derive, dim3(1), dim3(kScanBlock), 0, c->stream, nd, c->d_kmap.as<uint32_t>(), c->d_m.as<uint32_t>(), c->d_left4.as<int64_t>(),
                       c->d_lglob.as<int64_t>(), 0u, 0u);   // left4 / cluster bounds follow the requests (flags, hence kmap, are unchanged)
    LAUNCHCHK(c, BS_KERNEL_PREPASS);
  }
  c->bitmap_valid = false;
  if (G && P) {
    const BatchDev b = batch_dev(c);
    std::vector<uint32_t> cls(G);
    HIPCHK(c, hipMemcpy(c->h_gflags.data(), gr.flags, G, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cls.data(), gr.cls, (size_t)G * 4, hipMemcpyDeviceToHost));
    (void)b;
    c->n_uncaptured = 0;
    c->n_nominres = 0;
    c->max_group_cls = 0;
    for (uint32_t i = 0; i < G; ++i) {
      if (!(c->h_gflags[i] & BS_GROUP_HAS_POD)) c->n_uncaptured++;
      else c->max_group_cls = std::max(c->max_group_cls, cls[i]);
      if (!(c->h_gflags[i] & BS_GROUP_HAS_MINRES)) c->n_nominres++;
    }
    if ((rc = analyse_groups(c))) return rc;
    if ((rc = maybe_analyse_epochs(c))) return rc;
  }
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
// flat-argument forms (cgo: no Go-allocated struct of Go pointers crosses by pointer): the structs are built HERE, on the C stack
// -------------------------------------------------------------------------------------------------
int bs_nodes_load_flat(bs_ctx* c, uint32_t n, const int64_t* allocatable, const int64_t* requested, const uint32_t* allocatable_present,
                       const uint32_t* requested_present, const uint8_t* flags) {
  bs_nodes_soa s{};
  s.n = n; s.allocatable = allocatable; s.requested = requested; s.allocatable_present = allocatable_present; s.requested_present = requested_present;
  s.flags = flags;
  return bs_nodes_load(c, &s);
}
int bs_groups_load_flat(bs_ctx* c, uint32_t g, const uint32_t* min_member, const uint32_t* status_scheduled, const uint32_t* matched, const uint8_t* flags,
                        const uint32_t* cls, const int64_t* min_resources, const uint32_t* min_resources_present, const uint64_t* occupied_by) {
  bs_groups_soa s{};                                   // (the struct serves bs_groups_read too, hence non-const members: the load only reads)
  s.g = g;
  s.min_member = const_cast<uint32_t*>(min_member); s.status_scheduled = const_cast<uint32_t*>(status_scheduled); s.matched = const_cast<uint32_t*>(matched);
  s.flags = const_cast<uint8_t*>(flags); s.cls = const_cast<uint32_t*>(cls); s.min_resources = const_cast<int64_t*>(min_resources);
  s.min_resources_present = const_cast<uint32_t*>(min_resources_present); s.occupied_by = const_cast<uint64_t*>(occupied_by);
  return bs_groups_load(c, &s);
}
int bs_groups_read_flat(bs_ctx* c, uint32_t g, uint32_t* min_member, uint32_t* status_scheduled, uint32_t* matched, uint8_t* flags, uint32_t* cls,
                        int64_t* min_resources, uint32_t* min_resources_present, uint64_t* occupied_by) {
  bs_groups_soa s{};
  s.g = g; s.min_member = min_member; s.status_scheduled = status_scheduled; s.matched = matched; s.flags = flags; s.cls = cls;
  s.min_resources = min_resources; s.min_resources_present = min_resources_present; s.occupied_by = occupied_by;
  return bs_groups_read(c, &s);
}
int bs_pods_load_flat(bs_ctx* c, uint32_t p, const int32_t* group, const int64_t* req, const uint32_t* req_present, const uint32_t* cls, const uint64_t* owner,
                      const uint8_t* flags) {
  bs_pods_soa s{};
  s.p = p; s.group = group; s.req = req; s.req_present = req_present; s.cls = cls; s.owner = owner; s.flags = flags;
  return bs_pods_load(c, &s);
}
int bs_pods_apply_flat(bs_ctx* c, uint32_t n_remove, const uint32_t* remove, uint32_t n_flags, const uint32_t* flag_index, const uint8_t* flag_value,
                       uint32_t n_insert, const int32_t* group, const int64_t* req, const uint32_t* req_present, const uint32_t* cls, const uint64_t* owner,
                       const uint8_t* flags, const uint32_t* insert_at) {
  bs_pods_delta d{};
  d.n_remove = n_remove; d.remove = remove; d.n_flags = n_flags; d.flag_index = flag_index; d.flag_value = flag_value;
  d.insert.p = n_insert; d.insert.group = group; d.insert.req = req; d.insert.req_present = req_present; d.insert.cls = cls; d.insert.owner = owner;
  d.insert.flags = flags;
  d.insert_at = insert_at;
  return bs_pods_apply(c, &d);
}
int bs_pods_read_flat(bs_ctx* c, uint32_t p, int32_t* group, int64_t* req, uint32_t* req_present, uint32_t* cls, uint64_t* owner, uint8_t* flags) {
  bs_pods_out o{};
  o.p = p; o.group = group; o.req = req; o.req_present = req_present; o.cls = cls; o.owner = owner; o.flags = flags;
  return bs_pods_read(c, &o);
}
int bs_batch_read_flat(bs_ctx* c, uint8_t* pf_code, uint32_t* pf_first_k, int32_t* pf_leader, uint8_t* fl_code, uint32_t* fl_feasible, uint64_t* fl_bitmap,
                       uint32_t* group_admit, uint8_t* group_ready, uint32_t* fl_slot, uint64_t* fl_rows, uint32_t* fl_rows_feasible, uint32_t fl_rows_cap,
                       uint32_t* fl_rows_n) {
  bs_batch_out o{};
  o.pf_code = pf_code; o.pf_first_k = pf_first_k; o.pf_leader = pf_leader; o.fl_code = fl_code; o.fl_feasible = fl_feasible; o.fl_bitmap = fl_bitmap;
  o.group_admit = group_admit; o.group_ready = group_ready; o.fl_slot = fl_slot; o.fl_rows = fl_rows; o.fl_rows_feasible = fl_rows_feasible;
  o.fl_rows_cap = fl_rows_cap; o.fl_rows_n = fl_rows_n;
  return bs_batch_read(c, &o);
}
int bs_seq_run_flat(bs_ctx* c, uint32_t stages, uint8_t* pf_code, uint32_t* pf_first_k, int32_t* pf_leader, int32_t* pod_node, uint32_t cap,
                    uint32_t* released_group, uint32_t* released_pods, int64_t* first_ns, int64_t* ready_ns, int64_t* scalars_out,
                    uint8_t* last_permitted) {
  bs_seq_out o{};
  o.last_permitted = last_permitted;
  o.pf_code = pf_code; o.pf_first_k = pf_first_k; o.pf_leader = pf_leader; o.pod_node = pod_node; o.cap = cap; o.released_group = released_group;
  o.released_pods = released_pods; o.first_ns = first_ns; o.ready_ns = ready_ns;
  const int rc = bs_seq_run(c, stages, &o);
  if (scalars_out) {
    scalars_out[0] = o.n_released; scalars_out[1] = o.total_ns; scalars_out[2] = (int64_t)o.node_picks; scalars_out[3] = (int64_t)o.node_scans;
    scalars_out[4] = (int64_t)o.scan_rounds; scalars_out[5] = (int64_t)o.pick_rounds; scalars_out[6] = (int64_t)o.leader_folds;
    scalars_out[7] = (int64_t)o.table_builds;
  }
  return rc;
}
int bs_fit_build_flat(bs_ctx* c, uint32_t n, const uint32_t* name, const uint32_t* label_off, const uint32_t* label_key, const uint32_t* label_val,
                      const int64_t* label_int, const uint8_t* label_int_ok, const uint32_t* taint_off, const uint32_t* taint_key, const uint32_t* taint_val,
                      const uint8_t* taint_effect, uint32_t cn, uint32_t field_name_key, const uint8_t* tpl_flags, const uint32_t* sel_off,
                      const uint32_t* sel_key, const uint32_t* sel_val, const uint32_t* term_off, const uint32_t* term_expr_off, const uint32_t* term_field_off,
                      uint32_t ex_count, const uint32_t* ex_key, const uint8_t* ex_op, const uint32_t* ex_val_off, const uint32_t* ex_val,
                      const int64_t* ex_val_int, const uint8_t* ex_val_int_ok, uint32_t fd_count, const uint32_t* fd_key, const uint8_t* fd_op,
                      const uint32_t* fd_val_off, const uint32_t* fd_val, const int64_t* fd_val_int, const uint8_t* fd_val_int_ok, const uint32_t* tol_off,
                      const uint32_t* tol_key, const uint32_t* tol_val, const uint8_t* tol_op, const uint8_t* tol_effect) {
  bs_node_labels nl{};
  nl.n = n; nl.name = name; nl.label_off = label_off; nl.label_key = label_key; nl.label_val = label_val; nl.label_int = label_int;
  nl.label_int_ok = label_int_ok; nl.taint_off = taint_off; nl.taint_key = taint_key; nl.taint_val = taint_val; nl.taint_effect = taint_effect;
  bs_fit_templates tp{};
  tp.c = cn; tp.field_name_key = field_name_key; tp.flags = tpl_flags; tp.sel_off = sel_off; tp.sel_key = sel_key; tp.sel_val = sel_val;
  tp.term_off = term_off; tp.term_expr_off = term_expr_off; tp.term_field_off = term_field_off;
  tp.exprs.count = ex_count; tp.exprs.key = ex_key; tp.exprs.op = ex_op; tp.exprs.val_off = ex_val_off; tp.exprs.val = ex_val; tp.exprs.val_int = ex_val_int;
  tp.exprs.val_int_ok = ex_val_int_ok;
  tp.fields.count = fd_count; tp.fields.key = fd_key; tp.fields.op = fd_op; tp.fields.val_off = fd_val_off; tp.fields.val = fd_val;
  tp.fields.val_int = fd_val_int; tp.fields.val_int_ok = fd_val_int_ok;
  tp.tol_off = tol_off; tp.tol_key = tol_key; tp.tol_val = tol_val; tp.tol_op = tol_op; tp.tol_effect = tol_effect;
  return bs_fit_build(c, &nl, &tp);
}

// -------------------------------------------------------------------------------------------------
// native RCCL (for hosts without torch): librccl is dlopen'ed on first use
// -------------------------------------------------------------------------------------------------
static void* open_rccl() {
  static void* h = nullptr;
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  return h;
}

int bs_comm_unique_id(uint8_t id[128]) {
  if (!id) return BS_ERR_INVALID;
  static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the RCCL unique id over as 128 bytes");
  void* h = open_rccl();
  if (!h) return BS_ERR_COMM;
  auto f = (decltype(&ncclGetUniqueId))dlsym(h, "ncclGetUniqueId");
  ncclUniqueId u;
  if (!f || f(&u) != ncclSuccess) return BS_ERR_COMM;
  std::memcpy(id, u.internal, 128);
  return BS_OK;
}

int bs_comm_init(bs_ctx* c, const uint8_t id[128], uint32_t rank, uint32_t nranks) {
  if (!c || !id || nranks == 0 || rank >= nranks) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  void* h = open_rccl();
  if (!h) { c->last_error = std::string("dlopen librccl: ") + (dlerror() ? dlerror() : "?"); return BS_ERR_COMM; }
  ncclUniqueId u;
  std::memcpy(u.internal, id, 128);
  auto init = (decltype(&ncclCommInitRank))dlsym(h, "ncclCommInitRank");
  c->rccl_allreduce = (decltype(&ncclAllReduce))dlsym(h, "ncclAllReduce");
  c->rccl_destroy = (decltype(&ncclCommDestroy))dlsym(h, "ncclCommDestroy");
  ncclComm_t comm = nullptr;
  if (!init || !c->rccl_allreduce || !c->rccl_destroy) { c->last_error = "librccl lacks ncclCommInitRank / ncclAllReduce / ncclCommDestroy"; return BS_ERR_COMM; }
  const ncclResult_t r = init(&comm, (int)nranks, u, (int)rank);
  if (r != ncclSuccess) { c->last_error = std::string("ncclCommInitRank failed: code ") + std::to_string((int)r); return BS_ERR_COMM; }
  c->rccl_handle = h;
  c->comm = comm;
  c->rank = rank;
  c->nranks = nranks;
  return BS_OK;
}

// -------------------------------------------------------------------------------------------------
#ifdef BS_PROBE
// probe build only: the stamps of the launches since the last call ([kernel][block][stamp] clock ticks at 100 MHz; 0 = not written)
int bs_probe_read(bs_ctx* c, unsigned long long* out) {
  if (!c || !out) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(g_probe)));
  static std::vector<unsigned long long> zero(sizeof(g_probe) / 8, 0ull);
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero.data(), sizeof(g_probe)));
  return BS_OK;
}
#endif

int bs_timing_reset(bs_ctx* c) {
  if (!c) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  rc = timer_collect(c);
  std::memset(&c->timing, 0, sizeof(c->timing));
  return rc;
}

int bs_timing_get(bs_ctx* c, bs_timing* out) {
  if (!c || !out) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  rc = timer_collect(c);
  *out = c->timing;
  return rc;
}

int bs_speculation_stats(const bs_ctx* c, uint64_t* launched, uint64_t* missed) {
  if (!c || !launched || !missed) return BS_ERR_INVALID;
  *launched = c->n_spec;
  *missed = c->n_spec_miss;
  return BS_OK;
}

int bs_filter_deny_stats(const bs_ctx* c, uint64_t* reruns) {
  if (!c || !reruns) return BS_ERR_INVALID;
  *reruns = c->n_fd_reruns;
  return BS_OK;
}

int bs_batch_stats_get(bs_ctx* c, bs_batch_stats* out) {
  // Re-runs nothing: the counters are filled by the next bs_batch_run after this call arms them, and
  // read back here.  Usage: bs_batch_stats_get(arm) -> bs_batch_run -> bs_batch_stats_get(read).
  if (!c || !out) return BS_ERR_INVALID;
  int rc = use_device(c);
  if (rc) return rc;
  std::memset(out, 0, sizeof(*out));
  if (!c->collect_stats) {
    c->collect_stats = 1;
    return BS_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  uint64_t raw[8] = {0};
  uint32_t nt = 0, nq = 0;
  if (c->d_stats.p) HIPCHK(c, hipMemcpy(raw, c->d_stats.p, sizeof(raw), hipMemcpyDeviceToHost));
  if (c->d_qcount.p) HIPCHK(c, hipMemcpy(&nq, c->d_qcount.p, 4, hipMemcpyDeviceToHost));
  if (c->d_needed.p && c->C) {
    std::vector<uint32_t> need(2 * c->C);
    HIPCHK(c, hipMemcpy(need.data(), c->d_needed.p, need.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t x : need) nt += x ? 1u : 0u;
  }
  out->scan_rows_executed = raw[0];
  out->scan_evals_executed = raw[1];
  out->scan_queries = raw[4];
  out->scan_queries_logical = raw[2];
  out->filter_lane_blocks = raw[5];
  out->filter_tile_blocks = raw[6];
  out->class_mode = c->last_use_classes ? 1 : 0;
  out->fast_path = c->last_fast ? 1 : 0;
  out->chain = c->last_chain;
  out->launches = c->launches;
  out->tables_built = c->last_fast ? 1 : nt;
  out->logical_evals = (uint64_t)c->P * c->N;
  out->filter_evals = (c->last_stages & BS_STAGE_FILTER) ? (uint64_t)c->P * c->N : 0;
  if (c->last_stages & BS_STAGE_FILTER) {
    out->filter_distinct = raw[3];
    out->filter_evals_executed = raw[3] * c->N;
  }
  c->collect_stats = 0;
  return BS_OK;
}

}  // extern "C"
