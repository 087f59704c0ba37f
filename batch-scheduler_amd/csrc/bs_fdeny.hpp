// bs_fdeny.hpp — BS_BATCH_FILTER_DENY: the deny entry a failing Filter writes (core.go:183-185), replayed inside the batch.
//
// Filter(pod, node) returns an error when computeResourceSatisfied finds neither case 2 nor case 3 on that node (:551-563);
// Filter then calls AddToDenyCache(group) (:184) and every later pod of the group that gets to the deny check of PreFilter
// (:105-110: grouped, group known, not let through on its lastPermittedPod entry) is turned away with "last failed in 20s" —
// no fillOccupiedObj, no findMaxPG, no Filter for it.
//
// The three chains compute the batch as if that entry were never written, and their results already hold the event: a pod with
// fl_code == EVALUATED and fewer feasible nodes than the list has.  Per group, the FIRST such pod in queue order is real (nothing in
// front of it turns it away) and what lies behind it is mechanical, so one short launch behind the chain's last one
// (two behind the general chain) finishes the job:
//
//   k_fd_events   per pod: an event -> 64-bit keyed minimum per group (~key sequence << 32 | queue position: never reset);
//                 general chain only — the final blocks of the steady-state and positional chains do it themselves
//   k_fd_apply    per pod: behind the group's first event and at the deny check -> ERR_DENIED, Filter not run, no feasible node
//                 (device results and the pinned host mirrors); then the tally the chain's last launch left to this one
//                 (tally_tail: admit counts, quorum, completion word).
//   k_fd_reject   BS_BATCH_COMMIT only, in front of the chain's commit kernel: the event is a deny entry to persist.
//
// That is exact unless a pod that is turned away was needed by somebody else:
//   (1) it was the pod that brought a changed findMaxPG result into sop.maxFinishedPG (its stale-leader value differs from the
//       pod's in front of it) — the pods let through on lastPermittedPod entries behind it would read another leader in Filter;
//   (2) it was the first pod of its group to reach fillOccupiedObj and the group still lacked its pod or its MinResources — the
//       capture (core.go:486-493) does not happen, findMaxPG's candidates and the group's own request change.
// Both only happen behind a pod that was let through on its lastPermittedPod entry and failed Filter (or behind a first pod that
// fillOccupiedObj refused).  k_fd_apply detects them (bit 0 of the flag word) and the host resolves the batch by fixed-point
// iteration (bsched.hip fd_resolve): the events found become the INPUT of a re-run — fd_in[g] = queue position behind which the
// group's pods are turned away at the deny check, honoured by every chain's per-pod classification and by the positional analysis —
// until the events a run finds equal the ones it was given (bit 1 = they differ).  A pod's verdict only depends on events in front
// of it, so the positions settle in queue order and the fixed point is the sequential result.
#pragma once

#include "bs_epoch.hpp"

namespace bs {

__device__ __forceinline__ uint32_t fd_event_pos(const BatchDev& b, const BatchParams& prm, uint32_t g) {
  const unsigned long long e = b.fd_event[g];
  return (uint32_t)(e >> 32) == prm.seq_inv ? (uint32_t)e : BS_INF;
}

__global__ __launch_bounds__(256) void k_fd_events(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pods.p) return;
  const int32_t gi = pods.group[i];
  if (gi < 0 || (uint32_t)gi >= gr.g) return;
  if (((b.fflags[i] >> 8) & 0xFFu) != BS_FL_EVALUATED) return;
  if (pod_feasible(nd, b, i) < nd.n) atomicMin(&b.fd_event[gi], ((unsigned long long)prm.seq_inv << 32) | i);
}

// tail: 1 = the chain's last launch left tally / completion word to this one (steady-state and positional chains)
__global__ __launch_bounds__(256) void k_fd_apply(PodsDev pods, GroupsDev gr, NodesDev nd, BatchDev b, BatchParams prm, uint32_t tail) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool valid = i < pods.p;
  int32_t gi = BS_POD_NOT_GROUPED;
  uint8_t code = BS_PF_PASS_NOT_GROUPED, fl = BS_FL_NOT_RUN;
  uint32_t feasible = 0, flags = 0;
  if (valid) {
    gi = pods.group[i];
    code = b.pf_code[i];
    flags = b.fflags[i];
    fl = (uint8_t)((flags >> 8) & 0xFFu);
    feasible = fl == BS_FL_EVALUATED ? b.fu_feas[b.fu_slot[i]] : (fl < 16u ? nd.n : 0u);
  }
  const bool grouped = valid && gi >= 0 && (uint32_t)gi < gr.g;
  uint32_t trouble = 0;
  if (grouped) {
    const uint32_t ev = fd_event_pos(b, prm, (uint32_t)gi);
    const bool at_check = code != BS_PF_PASS_LAST_PERMITTED && code != BS_PF_ERR_PG_NOT_FOUND && code != BS_PF_NOT_OWNED && code != BS_PF_PASS_NOT_GROUPED;
    if (ev < i && at_check && code != BS_PF_ERR_DENIED) {
      // was this pod needed by somebody else?
      if (code != BS_PF_ERR_OCCUPIED) {                                  // it got to findMaxPG (every other code lies behind core.go:118)
        const int32_t before = i ? b.pf_leader[i - 1] : prm.sop_leader0;
        if (b.pf_leader[i] != before) trouble |= 1u;
      }
      const uint8_t gf = gr.flags[gi];
      if (b.first_np_s[gi] == i && (gf & (BS_GROUP_HAS_POD | BS_GROUP_HAS_MINRES)) != (BS_GROUP_HAS_POD | BS_GROUP_HAS_MINRES)) trouble |= 1u;
      if (prm.fd_iter) trouble |= 2u;                                    // a re-run was GIVEN its events: nobody is turned away here at the fixed point
      code = BS_PF_ERR_DENIED;
      fl = BS_FL_NOT_RUN;
      feasible = 0;
      b.pf_code[i] = code;
      b.pf_first_k[i] = BS_K_NOT_SCANNED;
      b.fl_code[i] = fl;
      b.fl_feasible[i] = 0;
      b.fflags[i] = (flags & 0xFFu) | ((uint32_t)fl << 8);
      b.fu_slot[i] = 0;
      if (prm.host_tag) { b.h_pf_code[i] = code; b.h_pf_first_k[i] = BS_K_NOT_SCANNED; b.h_fl_code[i] = fl; b.h_fl_feasible[i] = 0; b.h_fl_slot[i] = 0; }
    } else if (!tail && fl == BS_FL_EVALUATED) {
      b.fl_feasible[i] = feasible;                                       // (general chain: k_tally writes it again, same value)
    }
  }
  // fixed-point re-runs: the events found have to be the events given
  if (prm.fd_iter) {
    for (uint32_t g = i; g < gr.g; g += gridDim.x * 256u)
      if (fd_event_pos(b, prm, g) != b.fd_in[g]) trouble |= 2u;
  }
  if (__ballot(trouble != 0)) {
    uint32_t t = trouble;
#pragma unroll
    for (int o = 32; o; o >>= 1) t |= (uint32_t)__shfl_xor((int)t, o);
    if (lane_id() == 0) {
      atomicOr(b.fd_flag, t);                                            // device word: gates the commit kernels of this run
      if (t & 1u) __hip_atomic_store(b.h_fd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);        // pinned host words (plain stores: no PCIe atomics)
      if (t & 2u) __hip_atomic_store(b.h_fd + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (!tail) return;
  const bool pass = code != BS_PF_NOT_OWNED && BS_PF_IS_PASS(code);
  const bool admit = grouped && pass && (!prm.run_filter || feasible > 0);
  tally_tail(gr, b, prm, grouped, grouped ? (uint32_t)gi : 0u, admit, gridDim.x);
}

// BS_BATCH_COMMIT: AddToDenyCache of a failing Filter joins the group's first rejected pod — what the chain's commit kernel turns into
// BS_GROUP_DENIED (and what decides whether the first owner got to write OccupiedBy: the event's pod passed fillOccupiedObj, or was
// let through in front of it without calling it)
__global__ void k_fd_reject(BatchDev b, BatchParams prm, uint32_t* reject, uint32_t G) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t ev = fd_event_pos(b, prm, g);
  if (ev < reject[g]) reject[g] = ev;
}

// the events of the batch that just ran become the input of the next run
__global__ void k_fd_next(BatchDev b, BatchParams prm, uint32_t G) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) b.fd_in[g] = fd_event_pos(b, prm, g);
}

}  // namespace bs
