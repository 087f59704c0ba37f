// bs_queue.hpp — the pending queue stays resident: bs_pods_apply patches it on the device.
//
// The reference calls PreFilter (core.go:88) once per pod per scheduling cycle on a queue that changes by a few pods
// between cycles.  bs_pods_load re-uploads and re-hashes the whole queue; bs_pods_apply keeps pods, request classes,
// (group, request class) pairs and the per-group pod minima on the device and applies a delta:
//     stable removal of pods  |  insertion of new pods at given queue positions (append = the tail)  |  new flag bytes
// in ONE launch, the delta read straight from pinned host memory (no copy command):
//   gather blocks   every thread owns one position of the NEW queue, finds its source (a retained pod of the old queue or an
//                   inserted one) with two binary searches over the delta's index lists (staged in LDS), copies the pod with
//                   its derived class / pair ids from the old pack to the new one (ping-pong: order-preserving compaction
//                   cannot run in place) and contributes to the per-group minima of the new queue
//   insert wave     ONE wave classifies the inserted pods against the class / pair directories (hash -> id, keys stored by
//                   id — the load path's tables name representative PODS, whose indices do not survive a compaction):
//                   lookup; lanes that miss elect one lane per distinct hash, that lane draws the next id and inserts;
//                   the others look again.  One wave, so there is no race and no id is ever drawn twice for one key.
// Class and pair ids only grow between rebuilds (a class whose last pod left keeps its id; its slot is never stamped by a
// batch again); the host re-derives everything from the resident queue when the id space is used up.
#pragma once

#include "bs_kernels.hpp"

namespace bs {

constexpr int kApplyBlock = 256;
constexpr uint32_t kApplyLds = 1024;              // entries of each index list a block stages in LDS (more: searched in place)

struct PodsMut {                                  // a pod pack as a write target, with the derived per-pod ids
  int32_t* group; int64_t* req; uint32_t* pres; uint32_t* cls; uint64_t* owner; uint8_t* flags; uint32_t* pclass; uint32_t* ppair;
  uint32_t p;                                     // pods (lane stride of req)
};

struct PodDeltaDev {
  uint32_t n_remove, n_insert, n_flags;
  const uint32_t* remove;                         // [n_remove] old queue indices, strictly ascending
  const uint32_t* insert_at;                      // [n_insert] positions in the NEW queue, strictly ascending
  const uint32_t* flag_index;                     // [n_flags] old queue indices, strictly ascending
  const uint8_t* flag_value;                      // [n_flags]
  PodsDev ins;                                    // the inserted pods (ins.p == n_insert)
};

struct QueueDirs {                                // hash -> id directories for the insert wave
  unsigned long long* cdir; uint32_t cmask;       // request classes: slot = 1 << 63 | hash31 << 32 | class id
  int64_t* ckeys; uint32_t* cpres; uint32_t kcap; // [L][kcap] request lanes and [kcap] present bits by class id
  unsigned long long* pdir; uint32_t pmask;       // (group, class) pairs: slot = 1 << 63 | hash31 << 32 | pair id
  unsigned long long* pkeys;                      // [pair cap] group << 32 | class by pair id
  uint32_t* kcount;                               // classes drawn so far
  uint32_t* paircount;                            // pair ids drawn so far
  unsigned long long* pair_head;                  // [G] chain heads (class << 32 | pair id), low word BS_INF = none
  unsigned long long* pair_next;                  // [pair cap]
};

__device__ __forceinline__ uint64_t class_hash(const PodsDev& pods, uint32_t i, uint32_t L) {
  uint64_t h = mix64((uint64_t)pods.pres[i] + 0x9e3779b97f4a7c15ull);
  for (uint32_t j = 0; j < L; ++j) h = mix64(h ^ (uint64_t)pods.req[(size_t)j * pods.p + i]);
  return h;
}
__device__ __forceinline__ uint64_t pair_hash(uint32_t g, uint32_t c) { return mix64(((uint64_t)g << 32) | c); }

// entries <= x in an ascending list (the list itself, or the list minus its own index when SHIFT — see k_pods_apply)
template <bool SHIFT>
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t* a, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint32_t v = SHIFT ? a[mid] - mid : a[mid];
    if (v <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Directories of a freshly loaded queue (first bs_pods_apply after a bs_pods_load): every class representative files its
// class under its hash, every pair representative its pair.  All inserted keys are distinct: no comparison, the first
// empty slot on the probe path is taken.
__global__ void k_dirs_build(PodsDev pods, uint32_t G, uint32_t L, const uint32_t* rep, const uint32_t* id, const uint32_t* ppair, QueueDirs q, uint32_t hash_keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pods.p) return;
  if (rep[i] == i) {
    const uint32_t c = id[i];
    const uint64_t h = class_hash(pods, i, L);
    for (uint32_t j = 0; j < L; ++j) q.ckeys[(size_t)j * q.kcap + c] = pods.req[(size_t)j * pods.p + i];
    q.cpres[c] = pods.pres[i];
    const unsigned long long mine = (1ull << 63) | ((unsigned long long)(((uint32_t)(h >> 32)) & hash_keep & 0x7FFFFFFFu) << 32) | c;
    for (uint32_t sl = (uint32_t)h & q.cmask;; sl = (sl + 1u) & q.cmask)
      if (atomicCAS(&q.cdir[sl], 0ull, mine) == 0ull) break;
  }
  if (ppair[i] == i) {
    const uint32_t g = (uint32_t)pods.group[i], c = id[rep[i]];
    const uint64_t h = pair_hash(g, c);
    q.pkeys[i] = ((unsigned long long)g << 32) | c;
    const unsigned long long mine = (1ull << 63) | ((unsigned long long)(((uint32_t)(h >> 32)) & hash_keep & 0x7FFFFFFFu) << 32) | i;
    for (uint32_t sl = (uint32_t)h & q.pmask;; sl = (sl + 1u) & q.pmask)
      if (atomicCAS(&q.pdir[sl], 0ull, mine) == 0ull) break;
  }
}

// agent-scope accessors for the directories: the insert wave reads back what its own lanes stored a moment ago
__device__ __forceinline__ unsigned long long ld_dir(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dir(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The insert wave: class and pair ids of the inserted pods (written at their positions in the new pack).
// Everything a lane needs from memory is fetched in as few dependent round trips as the data allows: the pod's request
// (pinned host memory: one PCIe round trip for all lanes), the slot of its hash, all key lanes of the class the slot names
// (issued together, compared afterwards — no short-circuit chain of loads), then the same for the pair.
__device__ __forceinline__ void apply_insert_wave(const PodDeltaDev& d, const PodsMut& nw, uint32_t G, uint32_t L, const QueueDirs& q, uint32_t hash_keep) {
  const int lane = lane_id();
  for (uint32_t base = 0; base < d.n_insert; base += 64u) {
    const uint32_t k = base + (uint32_t)lane;
    const bool valid = k < d.n_insert;
    // ---- the pod: request lanes, present bits, group, target position — one round trip
    int64_t rq[BS_MAX_LANES];
    uint32_t pres = 0, at = 0;
    int32_t gi = -1;
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j) rq[j] = (valid && j < L) ? d.ins.req[(size_t)j * d.ins.p + k] : 0;
    if (valid) { pres = d.ins.pres[k]; gi = d.ins.group[k]; at = d.insert_at[k]; }
    // ---- request class: equal (request lanes, present bits) <=> equal class
    uint64_t h = mix64((uint64_t)pres + 0x9e3779b97f4a7c15ull);
#pragma unroll
    for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
      if (j < L) h = mix64(h ^ (uint64_t)rq[j]);
    const uint32_t tag = (((uint32_t)(h >> 32)) & hash_keep & 0x7FFFFFFFu) | 0x80000000u;
    uint32_t cls = BS_INF;
    bool pending = valid;
    while (__ballot(pending)) {
      uint32_t sl = (uint32_t)h & q.cmask;
      if (pending) {                                           // lookup: probe until the key or an empty slot
        for (;; sl = (sl + 1u) & q.cmask) {
          const unsigned long long cur = ld_dir(&q.cdir[sl]);
          if (cur == 0ull) break;
          if ((uint32_t)(cur >> 32) != tag) continue;
          const uint32_t c = (uint32_t)cur;
          unsigned long long kv[BS_MAX_LANES];
          const uint32_t cp = __hip_atomic_load(&q.cpres[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) kv[j] = j < L ? ld_dir(reinterpret_cast<const unsigned long long*>(&q.ckeys[(size_t)j * q.kcap + c])) : 0ull;
          bool same = cp == pres;
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) same = same && (j >= L || (int64_t)kv[j] == rq[j]);
          if (same) { cls = c; pending = false; break; }
        }
      }
      // one lane per distinct hash among the lanes that missed draws the next class id and files it in the slot its probe
      // ended on; lanes with the same hash look again (same key: they find it; another key behind the same hash: they miss
      // again and one of them is elected in the next round)
      unsigned long long todo = __ballot(pending);
      bool elected = false;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint64_t h0 = bcast64(h, leader);
        if (lane == leader) elected = true;
        todo &= ~__ballot(pending && h == h0);
      }
      // elected lanes with DIFFERENT hashes can still end on the same empty slot: claim it with a CAS, losers probe on
      if (elected) {
        const uint32_t c = atomicAdd(q.kcount, 1u);
        if (c < q.kcap) {
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
            if (j < L) st_dir(reinterpret_cast<unsigned long long*>(&q.ckeys[(size_t)j * q.kcap + c]), (unsigned long long)rq[j]);
          __hip_atomic_store(&q.cpres[c], pres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __threadfence();
          const unsigned long long mine = ((unsigned long long)tag << 32) | c;
          for (;; sl = (sl + 1u) & q.cmask)
            if (atomicCAS(&q.cdir[sl], 0ull, mine) == 0ull) break;
        }
        cls = c;                                               // (c >= kcap cannot happen: the host re-derives before the id space runs out)
        pending = false;
      }
      __threadfence();
    }
    // ---- (group, request class) pair
    const bool grouped = valid && gi >= 0 && (uint32_t)gi < G;
    uint32_t pid = BS_INF;
    const uint64_t ph = grouped ? pair_hash((uint32_t)gi, cls) : 0ull;
    const uint32_t ptag = (((uint32_t)(ph >> 32)) & hash_keep & 0x7FFFFFFFu) | 0x80000000u;
    const unsigned long long pkey = ((unsigned long long)(uint32_t)gi << 32) | cls;
    pending = grouped;
    while (__ballot(pending)) {
      uint32_t sl = (uint32_t)ph & q.pmask;
      if (pending) {
        for (;; sl = (sl + 1u) & q.pmask) {
          const unsigned long long cur = ld_dir(&q.pdir[sl]);
          if (cur == 0ull) break;
          if ((uint32_t)(cur >> 32) != ptag) continue;
          if (ld_dir(&q.pkeys[(uint32_t)cur]) == pkey) { pid = (uint32_t)cur; pending = false; break; }
        }
      }
      unsigned long long todo = __ballot(pending);
      bool elected = false;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint64_t k0 = bcast64(pkey, leader);
        if (lane == leader) elected = true;
        todo &= ~__ballot(pending && pkey == k0);
      }
      if (elected) {
        const uint32_t np = atomicAdd(q.paircount, 1u);
        st_dir(&q.pkeys[np], pkey);
        // chain link: (class << 32) | pair id, pushed at the head of the group's chain (core.go:105-110 replay walks it)
        q.pair_next[np] = atomicExch(&q.pair_head[gi], ((unsigned long long)cls << 32) | np);
        __threadfence();
        const unsigned long long mine = ((unsigned long long)ptag << 32) | np;
        for (;; sl = (sl + 1u) & q.pmask)
          if (atomicCAS(&q.pdir[sl], 0ull, mine) == 0ull) break;
        pid = np;
        pending = false;
      }
      __threadfence();
    }
    if (valid) {
      nw.pclass[at] = cls;
      nw.ppair[at] = pid;
    }
  }
}

// gstat_new: [3][G] minima of the NEW queue (all ones on entry: the previous apply / load reset them);
// gstat_next: the other buffer, reset here for the apply after this one.  derive == 0: copy only (the host re-derives
// classes, pairs and minima from the resident queue afterwards).
__global__ __launch_bounds__(kApplyBlock) void k_pods_apply(PodsDev old, const uint32_t* old_pclass, const uint32_t* old_ppair, PodsMut nw, PodDeltaDev d,
                                                           uint32_t G, uint32_t L, uint32_t* gstat_new, uint32_t* gstat_next, QueueDirs q, uint32_t hash_keep,
                                                           uint32_t derive, uint32_t gather_blocks, int32_t tag, int32_t* hinfo) {
  __shared__ uint32_t s_rem[kApplyLds], s_at[kApplyLds], s_fi[kApplyLds];
  if (blockIdx.x >= gather_blocks) {                     // the insert block: wave 0 classifies, the block re-arms the spare minima
    if (derive) {
      for (uint32_t i = threadIdx.x; i < 3u * G; i += kApplyBlock) gstat_next[i] = BS_INF;
      if (threadIdx.x < 64) {
        apply_insert_wave(d, nw, G, L, q, hash_keep);
        if (threadIdx.x == 0 && hinfo) {
          hinfo[4] = (int32_t)__hip_atomic_load(q.kcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&hinfo[5], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    return;
  }
  const bool lds_rem = d.n_remove <= kApplyLds, lds_at = d.n_insert <= kApplyLds, lds_fi = d.n_flags <= kApplyLds;
  if (lds_rem) for (uint32_t i = threadIdx.x; i < d.n_remove; i += kApplyBlock) s_rem[i] = d.remove[i];
  if (lds_at) for (uint32_t i = threadIdx.x; i < d.n_insert; i += kApplyBlock) s_at[i] = d.insert_at[i];
  if (lds_fi) for (uint32_t i = threadIdx.x; i < d.n_flags; i += kApplyBlock) s_fi[i] = d.flag_index[i];
  __syncthreads();
  const uint32_t* rem = lds_rem ? s_rem : d.remove;
  const uint32_t* at = lds_at ? s_at : d.insert_at;
  const uint32_t* fi = lds_fi ? s_fi : d.flag_index;
  const uint32_t pn = blockIdx.x * kApplyBlock + threadIdx.x;
  if (pn >= nw.p) return;
  // inserts at positions <= pn; the pod here is inserted iff the last of them sits exactly here
  const uint32_t kk = upper_bound_u32<false>(at, d.n_insert, pn);
  int32_t gi;
  uint8_t fl;
  uint64_t own;
  if (kk && at[kk - 1] == pn) {
    const uint32_t k = kk - 1;
    gi = d.ins.group[k];
    fl = d.ins.flags[k];
    own = d.ins.owner[k];
    nw.group[pn] = gi;
    for (uint32_t j = 0; j < L; ++j) nw.req[(size_t)j * nw.p + pn] = d.ins.req[(size_t)j * d.ins.p + k];
    nw.pres[pn] = d.ins.pres[k];
    nw.cls[pn] = d.ins.cls[k];
    nw.owner[pn] = own;
    nw.flags[pn] = fl;                                   // pclass / ppair: the insert wave
  } else {
    // rank r among the retained pods -> old index o = r + (removed pods before o) = r + #{m : remove[m] - m <= r}
    const uint32_t r = pn - kk;
    const uint32_t o = r + upper_bound_u32<true>(rem, d.n_remove, r);
    gi = old.group[o];
    own = old.owner[o];
    fl = old.flags[o];
    const uint32_t f = upper_bound_u32<false>(fi, d.n_flags, o);
    if (f && fi[f - 1] == o) fl = d.flag_value[f - 1];
    nw.group[pn] = gi;
    for (uint32_t j = 0; j < L; ++j) nw.req[(size_t)j * nw.p + pn] = old.req[(size_t)j * old.p + o];
    nw.pres[pn] = old.pres[o];
    nw.cls[pn] = old.cls[o];
    nw.owner[pn] = own;
    nw.flags[pn] = fl;
    nw.pclass[pn] = old_pclass[o];
    nw.ppair[pn] = old_ppair[o];
  }
  if (derive && gi >= 0 && (uint32_t)gi < G) {           // per-group minima of the new queue (k_pod_pairs' rule)
    atomicMin(&gstat_new[gi], pn);
    if (!(fl & BS_POD_LAST_PERMITTED)) {
      atomicMin(&gstat_new[(size_t)G + gi], pn);
      if (own != 0) atomicMin(&gstat_new[(size_t)2 * G + gi], pn);
    }
  }
}

// bs_nodes_assume: node `index` gets a new requested vector.  left4 (getLeftResource lanes, core.go:460-463) follows; the
// cluster-wide bounds of left4 are only ever WIDENED here (they are pruning bounds: Filter skips a resource lane when even
// the smallest left covers every request of a tile, and gives up on case 2 when the largest is below every request — a
// bound that is too wide prunes less, never wrongly).  bs_nodes_load / bs_nodes_apply compute them exactly again.
struct NodeRequest { uint32_t index, requested_present; int64_t requested[BS_MAX_LANES]; };
__global__ void k_nodes_assume(const NodeRequest* reqs, uint32_t count, uint32_t L, uint32_t stride, const int64_t* alloc, int64_t* req, uint32_t* rpres,
                               const uint8_t* flags, int64_t* left4, int64_t* lglob) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const uint32_t n = reqs[t].index;
  for (uint32_t j = 0; j < L; ++j) req[(size_t)j * stride + n] = reqs[t].requested[j];
  rpres[n] = reqs[t].requested_present;
  const bool ok = !(flags[n] & (BS_NODE_NIL | BS_NODE_NO_NODE));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t v = wsub(alloc[(size_t)j * stride + n], reqs[t].requested[j]);
    left4[(size_t)j * stride + n] = v;
    if (ok) {
      atomicMin(reinterpret_cast<long long*>(&lglob[j]), (long long)v);
      atomicMax(reinterpret_cast<long long*>(&lglob[4 + j]), (long long)v);
    }
  }
}

}  // namespace bs
