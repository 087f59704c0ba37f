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
#include "bs_fast.hpp"

namespace bs {

constexpr int kApplyBlock = kLeaderBlock;        // one block of this launch can be the findMaxPG block of a group patch (leader_info_block)

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
  const uint8_t* blob;                            // every pointer above points into this pinned blob: [lists | inserted records]
  uint32_t blob_bytes, lists_bytes;               // both multiples of 16
};

struct QueueDirs {                                // hash -> id directories for the insert wave
  uint32_t slot_keep;                             // test knob (BS_HASH_SLOT_BITS): probes start at hash & slot_keep & mask — long probe paths on demand
  unsigned long long* cdir; uint32_t cmask;       // request classes: slot = 1 << 63 | hash31 << 32 | class id
  int64_t* ckeys; uint32_t* cpres; uint32_t kcap; // [L][kcap] request lanes and [kcap] present bits by class id
  unsigned long long* pdir; uint32_t pmask;       // (group, class) pairs: slot = 1 << 63 | hash31 << 32 | pair id
  unsigned long long* pkeys;                      // [pair cap] group << 32 | class by pair id
  uint32_t* kcount;                               // classes drawn so far
  uint32_t* paircount;                            // pair ids drawn so far
  unsigned long long* pair_head;                  // [G] chain heads (class << 32 | pair id), low word BS_INF = none
  unsigned long long* pair_next;                  // [pair cap]
  uint32_t pcap;                                  // pair ids the arrays hold
  int32_t* overflow;                              // pinned host word: the insert wave drew an id beyond kcap / pcap (the host's accounting
                                                  // makes that impossible; if it ever happens nothing is written out of bounds and the host
                                                  // refuses the next results and re-derives classes and pairs from the resident queue)
};

__device__ __forceinline__ uint64_t class_hash(const PodsDev& pods, uint32_t i, uint32_t L) {
  uint64_t h = mix64((uint64_t)pods.pres[i] + 0x9e3779b97f4a7c15ull);
  for (uint32_t j = 0; j < L; ++j) h = mix64(h ^ (uint64_t)pods.req[(size_t)j * pods.p + i]);
  return h;
}
// a pair is filed under (group, hash of the request), so its directory slot does not depend on the class id
__device__ __forceinline__ uint64_t pair_hash(uint32_t g, uint64_t request_hash) { return mix64(request_hash ^ mix64((uint64_t)g + 0x632be59bd9b4e019ull)); }

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
    for (uint32_t sl = (uint32_t)h & q.slot_keep & q.cmask;; sl = (sl + 1u) & q.cmask)
      if (atomicCAS(&q.cdir[sl], 0ull, mine) == 0ull) break;
  }
  if (ppair[i] == i) {
    const uint32_t g = (uint32_t)pods.group[i], c = id[rep[i]];
    const uint64_t h = pair_hash(g, class_hash(pods, i, L));      // the pair's representative carries the pair's request
    q.pkeys[i] = ((unsigned long long)g << 32) | c;
    const unsigned long long mine = (1ull << 63) | ((unsigned long long)(((uint32_t)(h >> 32)) & hash_keep & 0x7FFFFFFFu) << 32) | i;
    for (uint32_t sl = (uint32_t)h & q.slot_keep & q.pmask;; sl = (sl + 1u) & q.pmask)
      if (atomicCAS(&q.pdir[sl], 0ull, mine) == 0ull) break;
  }
}

// agent-scope accessors for the directories: the insert wave reads back what its own lanes stored a moment ago
__device__ __forceinline__ unsigned long long ld_dir(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dir(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ONE wave reads and writes the directories: program order + "my stores have been performed" is all the ordering there is to
// keep (directory words and keys are agent-scope accesses: they are performed at the coherence point, not in this XCD's L2).
// A release / acquire fence here is an L2 write-back + invalidate per round — several microseconds of this launch.
__device__ __forceinline__ void dir_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The insert wave: class and pair ids of the inserted pods (written at their positions in the new pack).
// The dependent chain is as short as the data allows: the pod's record (pinned host memory: one PCIe round trip for all
// lanes), then class directory and pair directory are probed TOGETHER — a pair is filed under (group, request hash), not
// (group, class id), so its slot is known before the class is — then the keys both slots name (class key lanes and the
// pair's (group, class id) word, issued together, compared afterwards).  An insert of pods of known gangs and templates (the
// per-cycle case) is three round trips deep.  Lanes that miss elect one lane per distinct key; that lane draws the next id
// and files it in the slot its probe ended on; the others look again.
__device__ __forceinline__ void apply_insert_wave(const PodDeltaDev& d, const PodsMut& nw, uint32_t G, uint32_t L, const QueueDirs& q, uint32_t hash_keep) {
  const int lane = lane_id();
  enum : uint32_t { PROBE = 0, CAND = 1, MISS = 2, DONE = 3 };
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
    const uint32_t ctag = (((uint32_t)(h >> 32)) & hash_keep & 0x7FFFFFFFu) | 0x80000000u;
    const bool grouped = valid && gi >= 0 && (uint32_t)gi < G;
    const uint64_t ph = grouped ? pair_hash((uint32_t)gi, h) : 0ull;
    const uint32_t ptag = (((uint32_t)(ph >> 32)) & hash_keep & 0x7FFFFFFFu) | 0x80000000u;
    uint32_t cls = BS_INF, pid = BS_INF;
    uint32_t cs = valid ? PROBE : DONE, ps = grouped ? PROBE : DONE;
    uint32_t slc = (uint32_t)h & q.slot_keep & q.cmask, slp = (uint32_t)ph & q.slot_keep & q.pmask;
    unsigned long long cand_pk = 0;                              // (group, class id) word of the pair candidate
    for (;;) {
      // ---- round trip 1: the directory words both probes stand on
      unsigned long long curc = 0, curp = 0;
      if (cs == PROBE) curc = ld_dir(&q.cdir[slc]);
      if (ps == PROBE) curp = ld_dir(&q.pdir[slp]);
      // ---- round trip 2: the keys those words name
      const bool ccand = cs == PROBE && curc != 0ull && (uint32_t)(curc >> 32) == ctag;
      const bool pcand = ps == PROBE && curp != 0ull && (uint32_t)(curp >> 32) == ptag;
      unsigned long long kv[BS_MAX_LANES];
      uint32_t cp = 0;
      if (ccand) {
        const uint32_t c = (uint32_t)curc;
        cp = __hip_atomic_load(&q.cpres[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (uint32_t j = 0; j < BS_MAX_LANES; ++j) kv[j] = j < L ? ld_dir(reinterpret_cast<const unsigned long long*>(&q.ckeys[(size_t)j * q.kcap + c])) : 0ull;
      }
      if (pcand) cand_pk = ld_dir(&q.pkeys[(uint32_t)curp]);
      if (cs == PROBE) {
        if (curc == 0ull) cs = MISS;
        else if (!ccand) slc = (slc + 1u) & q.cmask;
        else {
          bool same = cp == pres;
#pragma unroll
          for (uint32_t j = 0; j < BS_MAX_LANES; ++j) same = same && (j >= L || (int64_t)kv[j] == rq[j]);
          if (same) { cls = (uint32_t)curc; cs = DONE; } else slc = (slc + 1u) & q.cmask;
        }
      }
      if (ps == PROBE) {
        if (curp == 0ull) ps = MISS;
        else if (!pcand) slp = (slp + 1u) & q.pmask;
        else { ps = CAND; pid = (uint32_t)curp; }
      }
      // a pair candidate is the pair iff it names this group and the class id the class probe ended on
      if (ps == CAND && cs == DONE) {
        if (cand_pk == (((unsigned long long)(uint32_t)gi << 32) | cls)) ps = DONE;
        else { ps = PROBE; pid = BS_INF; slp = (slp + 1u) & q.pmask; }
      }
      if (__ballot(cs == PROBE || ps == PROBE)) continue;
      // ---- nobody can probe on: classes that are not filed yet draw their ids (one lane per distinct request) ...
      if (__ballot(cs == MISS)) {
        unsigned long long todo = __ballot(cs == MISS);
        bool elected = false;
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint64_t h0 = bcast64(h, leader);
          if (lane == leader) elected = true;
          todo &= ~__ballot(cs == MISS && h == h0);
        }
        if (elected) {
          uint32_t c = atomicAdd(q.kcount, 1u);
          if (c >= q.kcap) {                                     // (cannot happen: the host re-derives before the id space runs out — guarded all the same)
            if (q.overflow) __hip_atomic_store(q.overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            c = 0;                                               // an id inside the arrays; the results of the next batch are refused by the host
          } else {
#pragma unroll
            for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
              if (j < L) st_dir(reinterpret_cast<unsigned long long*>(&q.ckeys[(size_t)j * q.kcap + c]), (unsigned long long)rq[j]);
            __hip_atomic_store(&q.cpres[c], pres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dir_drain();
            // elected lanes with DIFFERENT requests can end on the same empty slot: claim it with a CAS, losers probe on
            const unsigned long long mine = ((unsigned long long)ctag << 32) | c;
            for (;; slc = (slc + 1u) & q.cmask)
              if (atomicCAS(&q.cdir[slc], 0ull, mine) == 0ull) break;
          }
          cls = c;
          cs = DONE;
        } else if (cs == MISS) {
          cs = PROBE;                                            // same request as an elected lane, or another one behind the same hash: look again
        }
        dir_drain();
        // a pair of a class that was drawn just now cannot be filed: its candidate (if any) is somebody else's
        if (ps == CAND && cs == DONE) {
          if (cand_pk == (((unsigned long long)(uint32_t)gi << 32) | cls)) ps = DONE;
          else { ps = PROBE; pid = BS_INF; slp = (slp + 1u) & q.pmask; }
        }
        continue;
      }
      // ---- ... then the pairs (every class is known now)
      if (__ballot(ps == MISS)) {
        const unsigned long long pkey = ((unsigned long long)(uint32_t)gi << 32) | cls;
        unsigned long long todo = __ballot(ps == MISS);
        bool elected = false;
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint64_t k0 = bcast64(pkey, leader);
          if (lane == leader) elected = true;
          todo &= ~__ballot(ps == MISS && pkey == k0);
        }
        if (elected) {
          uint32_t np = atomicAdd(q.paircount, 1u);
          if (np >= q.pcap) {                                    // (the same guard for the pair ids)
            if (q.overflow) __hip_atomic_store(q.overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            np = 0;
          } else {
            st_dir(&q.pkeys[np], pkey);
            // chain link: (class << 32) | pair id, pushed at the head of the group's chain (core.go:105-110 replay walks it)
            q.pair_next[np] = atomicExch(&q.pair_head[gi], ((unsigned long long)cls << 32) | np);
            dir_drain();
            const unsigned long long mine = ((unsigned long long)ptag << 32) | np;
            for (;; slp = (slp + 1u) & q.pmask)
              if (atomicCAS(&q.pdir[slp], 0ull, mine) == 0ull) break;
          }
          pid = np;
          ps = DONE;
        } else if (ps == MISS) {
          ps = PROBE;
        }
        dir_drain();
        continue;
      }
      break;                                                     // every lane: class DONE, pair DONE (or not grouped)
    }
    if (valid) {
      nw.pclass[at] = cls;
      nw.ppair[at] = grouped ? pid : BS_INF;
    }
  }
}

// gstat_new: [3][G] minima of the NEW queue (all ones on entry: the previous apply / load reset them);
// gstat_next: the other buffer, reset here for the apply after this one.  derive == 0: copy only (the host re-derives
// classes, pairs and minima from the resident queue afterwards).
// A group patch that arrived in the same cycle (bs_groups_apply, deltas in `gp`) rides along: block gather_blocks + 1 applies it
// and runs findMaxPG (leader_info_block) — one launch and one launch boundary less on the cycle's critical path.
struct GroupPatch { uint32_t on, C; int32_t tag; int32_t* info; uint32_t* matched; uint32_t* status_scheduled; uint8_t* flags; DeltaPack dp; };

// The delta as the kernel sees it in LDS: the host lays it out as ONE blob in pinned memory ([remove | insert_at | flag_index |
// flag_value] = the lists every gather block needs, then the inserted pods' records, which only the insert block needs), and a
// block fetches its part with ONE bulk read (16 bytes per lane, all lanes at once).  A kernel-side read of pinned host memory
// is a PCIe round trip of ~4 us on this box however small it is — and field-by-field reads queue up behind each other (the
// eight loads of a pod record took 9 us): one trip per block, everything else from LDS.
constexpr uint32_t kDeltaLds = 32768;             // bytes of LDS a block stages the blob (or its list part) into; larger deltas are read in place

template <class T>
__device__ __forceinline__ const T* rebase(const T* p, const uint8_t* from, const uint8_t* to) {
  return reinterpret_cast<const T*>(to + (reinterpret_cast<const uint8_t*>(p) - from));
}
__device__ __forceinline__ void stage_blob(uint4* dst, const uint8_t* src, uint32_t bytes) {      // bytes: a multiple of 16
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  for (uint32_t i = threadIdx.x; i < (bytes >> 4); i += kApplyBlock) dst[i] = s4[i];
  __syncthreads();
}
__device__ __forceinline__ void group_minima(uint32_t* gstat_new, uint32_t G, int32_t gi, uint8_t fl, uint64_t own, uint32_t pn) {
  if (gi < 0 || (uint32_t)gi >= G) return;             // per-group minima of the new queue (k_pod_pairs' rule)
  atomicMin(&gstat_new[gi], pn);
  if (!(fl & BS_POD_LAST_PERMITTED)) {
    atomicMin(&gstat_new[(size_t)G + gi], pn);
    if (own != 0) atomicMin(&gstat_new[(size_t)2 * G + gi], pn);
  }
}

__global__ __launch_bounds__(kApplyBlock) void k_pods_apply(PodsDev old, const uint32_t* old_pclass, const uint32_t* old_ppair, PodsMut nw, PodDeltaDev d,
                                                           uint32_t G, uint32_t L, uint32_t* gstat_new, uint32_t* gstat_next, QueueDirs q, uint32_t hash_keep,
                                                           uint32_t derive, uint32_t gather_blocks, int32_t tag, int32_t* hinfo, GroupsDev gr, BatchDev bd,
                                                           GroupPatch gp, uint32_t* gcount) {
  __shared__ uint4 s_blob[kDeltaLds / 16];
  const uint8_t* s_bytes = reinterpret_cast<const uint8_t*>(s_blob);
  BS_STAMP(0, 0);
  if (blockIdx.x == gather_blocks + 1) {                 // the group patch of this cycle + findMaxPG (only launched when gp.on)
    leader_info_block(gr, bd, gp.C, gp.tag, gp.info, gp.dp, gp.matched, gp.status_scheduled, gp.flags);
    BS_STAMP(0, 7);
    return;
  }
  if (blockIdx.x == gather_blocks) {
    // the insert block: the inserted pods are its business alone — records into the new pack (all threads), class / pair ids
    // (wave 0), their share of the per-group minima — and it re-arms the spare minima for the next apply
    if (derive) for (uint32_t i = threadIdx.x; i < 3u * G; i += kApplyBlock) gstat_next[i] = BS_INF;
    if (d.n_insert) {
      PodDeltaDev dl = d;
      if (d.blob_bytes <= kDeltaLds) {
        stage_blob(s_blob, d.blob, d.blob_bytes);
        dl.insert_at = rebase(d.insert_at, d.blob, s_bytes);
        dl.ins.group = rebase(d.ins.group, d.blob, s_bytes);
        dl.ins.req = rebase(d.ins.req, d.blob, s_bytes);
        dl.ins.pres = rebase(d.ins.pres, d.blob, s_bytes);
        dl.ins.cls = rebase(d.ins.cls, d.blob, s_bytes);
        dl.ins.owner = rebase(d.ins.owner, d.blob, s_bytes);
        dl.ins.flags = rebase(d.ins.flags, d.blob, s_bytes);
      }
      BS_STAMP(0, 1);
      for (uint32_t k = threadIdx.x; k < d.n_insert; k += kApplyBlock) {
        const uint32_t pn = dl.insert_at[k];
        const int32_t gi = dl.ins.group[k];
        const uint8_t fl = dl.ins.flags[k];
        const uint64_t own = dl.ins.owner[k];
        nw.group[pn] = gi;
        for (uint32_t j = 0; j < L; ++j) nw.req[(size_t)j * nw.p + pn] = dl.ins.req[(size_t)j * d.ins.p + k];
        nw.pres[pn] = dl.ins.pres[k];
        nw.cls[pn] = dl.ins.cls[k];
        nw.owner[pn] = own;
        nw.flags[pn] = fl;
        if (derive) {
          group_minima(gstat_new, G, gi, fl, own, pn);
          if (gi >= 0 && (uint32_t)gi < G) atomicAdd(&gcount[gi], 1u);   // pods per group: + the inserted ones (here), - the removed ones (gather block 0)
        }
      }
      if (derive && threadIdx.x < 64) apply_insert_wave(dl, nw, G, L, q, hash_keep);
    }
    BS_STAMP(0, 6);
    if (derive && threadIdx.x == 0 && hinfo) {
      // K and the tag that says "K is there" in ONE 64-bit store (hinfo[4], hinfo[5]: 8-byte aligned): nothing to order, so no system-scope release
      // (an L2 write-back of everything the block's XCD holds dirty, 1.4 us at the end of the cycle's first launch)
      const uint32_t kc = __hip_atomic_load(q.kcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&hinfo[4]), ((unsigned long long)(uint32_t)tag << 32) | kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    BS_STAMP(0, 7);
    return;
  }
  // ---- gather blocks: the index lists, staged in one bulk read when they fit
  const uint32_t* rem = d.remove;
  const uint32_t* at = d.insert_at;
  const uint32_t* fi = d.flag_index;
  const uint8_t* fv = d.flag_value;
  if (d.lists_bytes && d.lists_bytes <= kDeltaLds) {
    stage_blob(s_blob, d.blob, d.lists_bytes);
    rem = rebase(d.remove, d.blob, s_bytes);
    at = rebase(d.insert_at, d.blob, s_bytes);
    fi = rebase(d.flag_index, d.blob, s_bytes);
    fv = rebase(d.flag_value, d.blob, s_bytes);
  }
  BS_STAMP(0, 1);
  if (derive && blockIdx.x == 0) {
    for (uint32_t k = threadIdx.x; k < d.n_remove; k += kApplyBlock) {
      const int32_t gq = old.group[rem[k]];
      if (gq >= 0 && (uint32_t)gq < G) atomicSub(&gcount[gq], 1u);
    }
  }
  const uint32_t pn = blockIdx.x * kApplyBlock + threadIdx.x;
  if (pn >= nw.p) { BS_STAMP(0, 7); return; }
  // inserts at positions <= pn; the pod here is an inserted one iff the last of them sits exactly here (the insert block's)
  const uint32_t kk = upper_bound_u32<false>(at, d.n_insert, pn);
  if (kk && at[kk - 1] == pn) { BS_STAMP(0, 7); return; }
  // rank r among the retained pods -> old index o = r + (removed pods before o) = r + #{m : remove[m] - m <= r}
  const uint32_t r = pn - kk;
  const uint32_t o = r + upper_bound_u32<true>(rem, d.n_remove, r);
  const int32_t gi = old.group[o];
  const uint64_t own = old.owner[o];
  uint8_t fl = old.flags[o];
  const uint32_t pr = old.pres[o], cl = old.cls[o], pc = old_pclass[o], pp = old_ppair[o];
  int64_t rq[BS_MAX_LANES];
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j) rq[j] = j < L ? old.req[(size_t)j * old.p + o] : 0;     // every load of the pod in flight together
  const uint32_t f = upper_bound_u32<false>(fi, d.n_flags, o);
  if (f && fi[f - 1] == o) fl = fv[f - 1];
  BS_STAMP(0, 2);
  nw.group[pn] = gi;
#pragma unroll
  for (uint32_t j = 0; j < BS_MAX_LANES; ++j)
    if (j < L) nw.req[(size_t)j * nw.p + pn] = rq[j];
  nw.pres[pn] = pr;
  nw.cls[pn] = cl;
  nw.owner[pn] = own;
  nw.flags[pn] = fl;
  nw.pclass[pn] = pc;
  nw.ppair[pn] = pp;
  if (derive) group_minima(gstat_new, G, gi, fl, own, pn);
  BS_STAMP(0, 7);
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
