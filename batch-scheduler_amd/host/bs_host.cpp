// bs_host.cpp — host-side mirror of the reference's ScheduleOperation (pkg/scheduler/core/core.go)
// and of the batch-release step of the plugin (pkg/scheduler/batch/batchscheduler.go), in C++ above
// the C ABI of include/bsched.h.
//
// This is the sequential, one-pod-at-a-time mode: the same entry points, argument meaning and error
// behaviour as the Go code (PreFilter / Filter / Permit / PostBind / Compare), with every piece of
// node arithmetic delegated to the HIP library through the ABI:
//     findMaxPG                           -> bs_find_max_pg          (core.go:701-739)
//     compareClusterResourceAndRequire    -> bs_cluster_fits         (core.go:595-632)
//     computeResourceSatisfied            -> bs_filter_one           (core.go:514-564)
// What stays on the host is what stays in Go in a real deployment: the TTL caches (go-cache v2.1.0
// semantics, virtual clock), the PodGroup cache bookkeeping, string-free label / owner identities.
// There is no arithmetic fallback: without the HIP library nothing here can decide anything.
//
// It exists for (a) the reference's own end-to-end scene (README.md:78-188, BASELINE config 1) and
// (b) a gang-admit-latency number for the "sequential replay with the GPU node loop" variant
// (SURVEY.md 8(d)).  The batched path (bs_batch_run) is the product's fast path.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bsched.h"
#include "../../include/bsched_host.h"

namespace {

constexpr int64_t kSecond = 1000000000LL;
constexpr uint32_t BSH_START_BATCH_TOO_SMALL = 0xFFFFFFFFu;   // bsh_start_batch: the gang does not fit the caller's buffers

// patrickmn/go-cache v2.1.0 (go.mod:67): Set overwrites; Add fails while a live item exists; an item is
// expired when now > expiration (strictly); Items() returns the unexpired ones.
struct TtlCache {
  struct Item { uint64_t val; int64_t exp; uint64_t aux; };
  std::unordered_map<uint64_t, Item> items;
  void Set(uint64_t k, uint64_t v, int64_t now, int64_t ttl, uint64_t aux = 0) { items[k] = Item{v, ttl > 0 ? now + ttl : 0, aux}; }
  bool Get(uint64_t k, int64_t now, uint64_t* v = nullptr, uint64_t* aux = nullptr) const {
    auto it = items.find(k);
    if (it == items.end()) return false;
    if (it->second.exp > 0 && now > it->second.exp) return false;
    if (v) *v = it->second.val;
    if (aux) *aux = it->second.aux;
    return true;
  }
  bool Add(uint64_t k, uint64_t v, int64_t now, int64_t ttl) {
    if (Get(k, now)) return false;
    Set(k, v, now, ttl);
    return true;
  }
  void Delete(uint64_t k) { items.erase(k); }
  uint32_t Count(int64_t now) const {
    uint32_t c = 0;
    for (auto& kv : items)
      if (!(kv.second.exp > 0 && now > kv.second.exp)) c++;
    return c;
  }
  std::vector<uint64_t> Keys(int64_t now) const {
    std::vector<uint64_t> k;
    for (auto& kv : items)
      if (!(kv.second.exp > 0 && now > kv.second.exp)) k.push_back(kv.first);
    return k;
  }
};

// PodGroupPhase, types.go:28-56: the values of include/bsched_host.h (the phase machine proper — syncHandler, the merge patch — is bs_phase.cpp)
enum Phase : uint8_t { Pending = BSH_PHASE_PENDING, PreScheduling = BSH_PHASE_PRESCHEDULING, Scheduling = BSH_PHASE_SCHEDULING, Scheduled = BSH_PHASE_SCHEDULED,
                       Running = BSH_PHASE_RUNNING, Finished = BSH_PHASE_FINISHED, Failed = BSH_PHASE_FAILED };

struct Group {   // cache.PodGroupMatchStatus + PodGroup spec/status (cache.go:52-67, types.go:79-130)
  uint32_t min_member = 0, status_scheduled = 0;
  bool scheduled_latch = false, has_pod = false, has_minres = false;
  uint32_t cls = 0;
  int64_t minres[BS_MAX_LANES] = {0};
  uint32_t minres_present = 0;
  uint64_t occupied_by = 0;
  uint8_t phase = Pending;
  int64_t max_schedule_time_ns = -1;   // Spec.MaxScheduleTime, -1 = nil
  int64_t creation_ts = 0;             // CreationTimestamp (Compare)
  uint64_t name_rank = 0;              // order-isomorphic stand-in for the group NAME string (Compare :404)
  TtlCache matched;                    // MatchedPodNodes: uid -> node
  TtlCache name_uids;                  // PodNameUIDs: pod name id -> uid
};

struct Pod {
  uint64_t uid, name;
  int32_t group;          // index, BS_POD_NOT_GROUPED, BS_POD_GROUP_MISSING
  int64_t req[BS_MAX_LANES];
  uint32_t req_present, cls;
  uint64_t owner;
  int32_t priority;
  int64_t queue_ts;
};

}  // namespace

struct bsh_sop {
  bs_ctx* ctx = nullptr;
  uint32_t L = 4, S = 0;
  std::vector<Group> groups;
  TtlCache last_denied_pg;     // core.go:71   New(30s, 3s)
  TtlCache last_permitted;     // core.go:72   New(3s, 3s)
  int64_t now = 0;             // virtual clock, ns
  int64_t max_sche_time_ns = 60 * kSecond;
  int32_t max_finished_pg = -1;   // sop.maxFinishedPG (core.go:58)
  bool has_max_status = false;    // sop.maxPGStatus != nil (core.go:59)
  bool groups_dirty = true;
  uint64_t gpu_calls = 0;
  std::string last_error;

  int64_t wait_time(const Group& g) const {          // util/k8s.go:82-91
    if (g.max_schedule_time_ns >= 0) return g.max_schedule_time_ns;
    return max_sche_time_ns;
  }

  // push the PodGroup cache to the device (what bs_groups_load takes: flat counters)
  int sync_groups() {
    if (!groups_dirty) return BS_OK;
    const uint32_t G = (uint32_t)groups.size();
    std::vector<uint32_t> mm(G), sc(G), ma(G), cl(G), mp(G);
    std::vector<uint8_t> fl(G);
    std::vector<int64_t> mr((size_t)L * std::max<uint32_t>(G, 1));
    std::vector<uint64_t> oc(G);
    for (uint32_t g = 0; g < G; ++g) {
      const Group& x = groups[g];
      mm[g] = x.min_member; sc[g] = x.status_scheduled; ma[g] = x.matched.Count(now); cl[g] = x.cls; mp[g] = x.minres_present; oc[g] = x.occupied_by;
      fl[g] = (x.scheduled_latch ? BS_GROUP_SCHEDULED_LATCH : 0) | (x.has_pod ? BS_GROUP_HAS_POD : 0) | (x.has_minres ? BS_GROUP_HAS_MINRES : 0) |
              (bsh_phase_closed(x.phase) ? BS_GROUP_PHASE_CLOSED : 0);   // batchscheduler.go:258-261 (bs_phase.cpp decides)
      for (uint32_t j = 0; j < L; ++j) mr[(size_t)j * G + g] = x.minres[j];
    }
    bs_groups_soa s{G, mm.data(), sc.data(), ma.data(), fl.data(), cl.data(), mr.data(), mp.data(), oc.data()};
    int rc = bs_groups_load(ctx, &s);
    if (rc == BS_OK) groups_dirty = false;
    gpu_calls++;
    return rc;
  }

  // getPodResourceRequire(pod).ResourceList() -> MinResources (core.go:489-493): Add drops eph without the gate;
  // the gate lives in the library config, the shim passes what Go's Add produced.
  void fill_occupied(Group& g, const Pod& p, bool* err) {   // core.go:477-512
    *err = false;
    if (!g.has_pod) { g.has_pod = true; g.cls = p.cls; groups_dirty = true; }
    if (!g.has_minres) {
      for (uint32_t j = 0; j < L; ++j) g.minres[j] = p.req[j];
      g.minres_present = p.req_present;
      g.has_minres = true;
      groups_dirty = true;
    }
    if (g.occupied_by == 0) {
      if (p.owner != 0) { g.occupied_by = p.owner; groups_dirty = true; }
      return;
    }
    if (p.owner == 0 || p.owner != g.occupied_by) *err = true;
  }

  // getPreAllocatedResource (core.go:774-793)
  void pre_allocated(const Group& g, int64_t matched, int64_t* out, uint32_t* present) {
    for (uint32_t j = 0; j < L; ++j) out[j] = 0;
    *present = 0;
    const int64_t nf = matched != 0 ? (int64_t)g.min_member - matched : (int64_t)g.min_member - (int64_t)g.status_scheduled;
    if (nf > 0 && g.has_minres) {
      for (uint32_t j = 0; j < L; ++j) out[j] = (int64_t)((uint64_t)g.minres[j] * (uint64_t)nf);
      *present = g.minres_present;
    }
    if (out[BS_LANE_PODS] == 0) out[BS_LANE_PODS] = (int64_t)g.min_member + 1;
  }

  // ScheduleOperation.PreFilter, core.go:88-167.  Returns a BS_PF_* code.
  int PreFilter(const Pod& p, uint32_t* first_k) {
    if (first_k) *first_k = BS_K_NOT_SCANNED;
    if (p.group == BS_POD_NOT_GROUPED) return BS_PF_PASS_NOT_GROUPED;                 // :89-92
    if (last_permitted.Get(p.uid, now)) return BS_PF_PASS_LAST_PERMITTED;             // :95-98
    if (p.group < 0 || (size_t)p.group >= groups.size()) return BS_PF_ERR_PG_NOT_FOUND;   // :100-103
    if (last_denied_pg.Get((uint64_t)p.group, now)) return BS_PF_ERR_DENIED;          // :105-110
    Group& pgs = groups[p.group];
    bool occ_err = false;
    fill_occupied(pgs, p, &occ_err);                                                  // :113-115
    if (occ_err) return BS_PF_ERR_OCCUPIED;
    if (sync_groups() != BS_OK) return -1;
    int32_t leader = -1;
    uint32_t fin = 0;
    uint8_t panic = 0;
    if (bs_find_max_pg(ctx, &leader, &fin, &panic) != BS_OK) return -1;                // :118-123
    gpu_calls++;
    if (panic) return BS_PF_PANIC_DIV0;
    max_finished_pg = leader;
    has_max_status = leader >= 0;
    if (leader < 0) return BS_PF_PASS_NO_MAX;                                         // :127-130
    const int64_t matched = (int64_t)groups[leader].matched.Count(now);                // :132-135
    int64_t req[BS_MAX_LANES];
    uint32_t present = 0;
    uint8_t fits = 0;
    uint32_t fk = BS_K_NONE;
    if (matched == 0) {                                                                // :136-147
      pre_allocated(pgs, 0, req, &present);
      if (bs_cluster_fits(ctx, pgs.cls, 1.0f, req, present, &fits, &fk) != BS_OK) return -1;
      gpu_calls++;
      if (first_k) *first_k = fk;
      if (!fits) { last_denied_pg.Add((uint64_t)p.group, 0, now, 20 * kSecond); return BS_PF_REJECT_FIRST; }   // :142, Add: window not extended
      return BS_PF_PASS_FIRST_FITS;
    }
    if (max_finished_pg == p.group) return BS_PF_PASS_IS_MAX;                          // :150-155
    pre_allocated(groups[leader], matched, req, &present);                             // :157
    for (uint32_t j = 0; j < L; ++j) req[j] = (int64_t)((uint64_t)req[j] + (uint64_t)p.req[j]);   // :158-159
    present |= p.req_present;
    if (bs_cluster_fits(ctx, groups[leader].cls, 0.7f, req, present, &fits, &fk) != BS_OK) return -1;   // :161
    gpu_calls++;
    if (first_k) *first_k = fk;
    if (!fits) { last_denied_pg.Add((uint64_t)p.group, 0, now, 20 * kSecond); return BS_PF_REJECT_RESERVE; }
    return BS_PF_PASS_RESERVE_FITS;
  }

  // ScheduleOperation.Filter, core.go:170-191.  fl = BS_FL_*, fn = BS_FN_* (valid iff fl == EVALUATED).
  int Filter(const Pod& p, uint32_t node, uint8_t* fl, uint8_t* fn) {
    *fn = BS_FN_PASS_CASE2;
    if (p.group == BS_POD_NOT_GROUPED) { *fl = BS_FL_PASS_NOT_GROUPED; return BS_OK; }
    if (p.group < 0 || (size_t)p.group >= groups.size()) { *fl = BS_FL_ERR_PG_NOT_FOUND; return BS_OK; }
    if (sync_groups() != BS_OK) return -1;
    const int32_t leader = has_max_status ? max_finished_pg : -1;
    if (bs_filter_one(ctx, p.group, p.req, p.req_present, leader, node, fl, fn) != BS_OK) return -1;
    gpu_calls++;
    const bool err = *fl >= 16 || (*fl == BS_FL_EVALUATED && *fn >= 16);
    if (*fl == BS_FL_PANIC_NIL_MAX) return BS_OK;                       // the Go process would have panicked (:525)
    if (err) last_denied_pg.Add((uint64_t)p.group, 0, now, 20 * kSecond);   // :183-186
    else last_permitted.Add(p.uid, 0, now, 2 * kSecond);                  // :188
    return BS_OK;
  }

  // ScheduleOperation.Permit, core.go:268-309.  ready / code: 0 ok(ready), 1 ErrorWaiting, 2 ErrorNotMatched, 3 not found
  int Permit(const Pod& p, uint32_t node, bool* ready) {
    *ready = false;
    if (p.group == BS_POD_NOT_GROUPED) { *ready = true; return 2; }        // :269-272
    if (p.group < 0 || (size_t)p.group >= groups.size()) return 3;          // :274-277
    Group& pgs = groups[p.group];
    pgs.phase = (uint8_t)bsh_pg_permit(pgs.phase);                         // :279-281
    const int64_t wait = wait_time(pgs);                                   // :289
    pgs.matched.Set(p.uid, node, now, wait, p.name);                       // :290
    uint64_t old_uid = 0;
    if (pgs.name_uids.Get(p.name, now, &old_uid)) pgs.matched.Delete(old_uid);   // :291-296 (also when old_uid == uid, Q13)
    pgs.name_uids.Set(p.name, p.uid, now, wait);                           // :300
    groups_dirty = true;
    const uint32_t have = pgs.matched.Count(now);
    if (have >= (uint32_t)(pgs.min_member - pgs.status_scheduled)) {        // :303 uint32 wrap
      pgs.scheduled_latch = true;                                          // :305
      *ready = true;
      return 0;
    }
    return 1;
  }

  // in-memory part of PostBind, core.go:312-362 (the API PATCH is the Go side's business)
  void PostBind(const Pod& p) {
    if (p.group < 0 || (size_t)p.group >= groups.size()) return;
    Group& pgs = groups[p.group];
    pgs.status_scheduled++;                                                // :327,:359
    pgs.phase = pgs.status_scheduled >= pgs.min_member ? Scheduled : Scheduling;   // :329-336,:356
    groups_dirty = true;
  }

  // ScheduleOperation.Compare, core.go:368-411
  bool Compare(const Pod& a, const Pod& b) const {
    const bool ga = a.group != BS_POD_NOT_GROUPED, gb = b.group != BS_POD_NOT_GROUPED;
    if (a.priority > b.priority) return true;                              // :379-381
    if (a.priority == b.priority) {
      if (!ga && !gb) return a.queue_ts < b.queue_ts;                      // :384-386
      if (!ga) return true;                                                // :388-390
      if (!gb) return false;                                               // :391-393
    }
    // lister lookups: an ungrouped or unknown group is a lister error -> false (:395-399)
    if (a.group < 0 || b.group < 0 || (size_t)a.group >= groups.size() || (size_t)b.group >= groups.size()) return false;
    const Group& g1 = groups[a.group];
    const Group& g2 = groups[b.group];
    if (a.priority == b.priority && g1.creation_ts < g2.creation_ts) return true;                              // :400-402
    if (a.priority == b.priority && g1.creation_ts == g2.creation_ts && g1.name_rank > g2.name_rank) return true;   // :404-406
    return a.priority == b.priority && g1.creation_ts == g2.creation_ts && g1.name_rank == g2.name_rank && a.queue_ts < b.queue_ts;
  }
};

namespace {
Pod make_pod(const bsh_sop* s, uint64_t uid, uint64_t name, int32_t group, const int64_t* req, uint32_t present, uint32_t cls, uint64_t owner,
             int32_t prio, int64_t ts) {
  Pod p{};
  p.uid = uid; p.name = name; p.group = group; p.req_present = present; p.cls = cls; p.owner = owner; p.priority = prio; p.queue_ts = ts;
  for (uint32_t j = 0; j < s->L; ++j) p.req[j] = req ? req[j] : 0;
  return p;
}
}  // namespace

extern "C" {

bsh_sop* bsh_create(bs_ctx* ctx, uint32_t scalar_lanes, int64_t max_sche_time_ns) {
  if (!ctx) return nullptr;
  bsh_sop* s = new bsh_sop();
  s->ctx = ctx;
  s->S = scalar_lanes;
  s->L = 4 + scalar_lanes;
  if (max_sche_time_ns > 0) s->max_sche_time_ns = max_sche_time_ns;
  return s;
}
void bsh_destroy(bsh_sop* s) { delete s; }
void bsh_set_time(bsh_sop* s, int64_t now_ns) { s->now = now_ns; s->groups_dirty = true; }
int64_t bsh_time(const bsh_sop* s) { return s->now; }
uint64_t bsh_gpu_calls(const bsh_sop* s) { return s->gpu_calls; }

// PodGroup cache entry as the controller creates it (controller.go:314-335): counters zero, TTL maps empty
int32_t bsh_add_group(bsh_sop* s, uint32_t min_member, uint32_t status_scheduled, int64_t max_schedule_time_ns, int64_t creation_ts,
                      uint64_t name_rank, const int64_t* min_resources /*nullable*/, uint32_t min_resources_present) {
  Group g;
  g.min_member = min_member;
  g.status_scheduled = status_scheduled;
  g.max_schedule_time_ns = max_schedule_time_ns;
  g.creation_ts = creation_ts;
  g.name_rank = name_rank;
  if (min_resources) {
    for (uint32_t j = 0; j < s->L; ++j) g.minres[j] = min_resources[j];
    g.minres_present = min_resources_present;
    g.has_minres = true;
  }
  s->groups.push_back(std::move(g));
  s->groups_dirty = true;
  return (int32_t)s->groups.size() - 1;
}

int bsh_prefilter(bsh_sop* s, uint64_t uid, uint64_t name, int32_t group, const int64_t* req, uint32_t present, uint32_t cls, uint64_t owner,
                  uint32_t* first_k) {
  return s->PreFilter(make_pod(s, uid, name, group, req, present, cls, owner, 0, 0), first_k);
}
int bsh_filter(bsh_sop* s, uint64_t uid, int32_t group, const int64_t* req, uint32_t present, uint32_t node, uint8_t* fl, uint8_t* fn) {
  return s->Filter(make_pod(s, uid, 0, group, req, present, 0, 0, 0, 0), node, fl, fn);
}
int bsh_permit(bsh_sop* s, uint64_t uid, uint64_t name, int32_t group, uint32_t node, uint8_t* ready) {
  bool r = false;
  const int code = s->Permit(make_pod(s, uid, name, group, nullptr, 0, 0, 0, 0, 0), node, &r);
  *ready = r ? 1 : 0;
  return code;
}
void bsh_postbind(bsh_sop* s, int32_t group) { s->PostBind(make_pod(s, 0, 0, group, nullptr, 0, 0, 0, 0, 0)); }
int bsh_less(const bsh_sop* s, int32_t group1, int32_t prio1, int64_t ts1, int32_t group2, int32_t prio2, int64_t ts2) {
  return s->Compare(make_pod(s, 0, 0, group1, nullptr, 0, 0, 0, prio1, ts1), make_pod(s, 0, 0, group2, nullptr, 0, 0, 0, prio2, ts2)) ? 1 : 0;
}

// StartBatchSchedule, batchscheduler.go:254-344, in-memory part: when the quorum still holds, every
// matched pod is allowed (returned to the caller, who binds it) and leaves MatchedPodNodes.
// out_uids / out_nodes: capacity `cap`; returns the number released, 0 when the phase / quorum gate closes.
// A gang larger than `cap` is NOT released in part (the surplus would leave the cache without ever being
// bound): nothing is touched and BSH_START_BATCH_TOO_SMALL is returned; size the buffers from bsh_group_matched.
uint32_t bsh_start_batch(bsh_sop* s, int32_t group, uint64_t* out_uids, uint32_t* out_nodes, uint32_t cap) {
  if (group < 0 || (size_t)group >= s->groups.size()) return 0;
  Group& pgs = s->groups[group];
  if (pgs.phase != PreScheduling && pgs.phase != Scheduling) return 0;            // :258-261
  const uint32_t have = pgs.matched.Count(s->now);
  if (have < (uint32_t)(pgs.min_member - pgs.status_scheduled)) return 0;          // :303-305
  if (have > cap) return BSH_START_BATCH_TOO_SMALL;
  uint32_t n = 0;
  std::vector<uint64_t> keys = pgs.matched.Keys(s->now);
  std::sort(keys.begin(), keys.end());     // Go map order is random; any order releases the same set
  for (uint64_t uid : keys) {
    uint64_t node = 0;
    pgs.matched.Get(uid, s->now, &node);
    out_uids[n] = uid;
    out_nodes[n] = (uint32_t)node;
    n++;
    pgs.matched.Delete(uid);               // :332 (pendingPodNameIDs.Delete(uid) at :333 uses the wrong key: a no-op)
  }
  s->groups_dirty = true;
  return n;
}

// Push the PodGroup cache to the device now (bs_groups_load): what a shim does before handing a queue to
// bs_batch_run, so that the batch starts from exactly the state the per-pod entry points have built up.
int bsh_sync(bsh_sop* s) { return s->sync_groups(); }

// observers for tests
uint32_t bsh_group_matched(const bsh_sop* s, int32_t g) { return s->groups[g].matched.Count(s->now); }
uint32_t bsh_group_status_scheduled(const bsh_sop* s, int32_t g) { return s->groups[g].status_scheduled; }
uint32_t bsh_group_flags(const bsh_sop* s, int32_t g) {
  const Group& x = s->groups[g];
  return (x.scheduled_latch ? 1u : 0u) | (x.has_pod ? 2u : 0u) | (x.has_minres ? 4u : 0u) | ((uint32_t)x.phase << 8);
}
int bsh_group_denied(const bsh_sop* s, int32_t g) { return s->last_denied_pg.Get((uint64_t)g, s->now) ? 1 : 0; }

// stand-alone TTL cache (CPU-only unit tests of the go-cache semantics)
void* bsh_ttl_new() { return new TtlCache(); }
void bsh_ttl_free(void* t) { delete (TtlCache*)t; }
void bsh_ttl_set(void* t, uint64_t k, uint64_t v, int64_t now, int64_t ttl) { ((TtlCache*)t)->Set(k, v, now, ttl); }
int bsh_ttl_add(void* t, uint64_t k, uint64_t v, int64_t now, int64_t ttl) { return ((TtlCache*)t)->Add(k, v, now, ttl) ? 0 : -1; }
int bsh_ttl_get(void* t, uint64_t k, int64_t now, uint64_t* v) { return ((TtlCache*)t)->Get(k, now, v) ? 1 : 0; }
void bsh_ttl_delete(void* t, uint64_t k) { ((TtlCache*)t)->Delete(k); }
uint32_t bsh_ttl_count(void* t, int64_t now) { return ((TtlCache*)t)->Count(now); }

}  // extern "C"
