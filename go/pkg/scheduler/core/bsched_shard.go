// +build cgo

/*
Sharding the pending-pod axis over several GPUs (one process per GPU; SURVEY.md section 8(e), include/bsched.h "The three
multi-rank modes").  Nothing here is compiled in this repository's image (no Go toolchain); the rules are the ones
batch-scheduler_amd/dist.py states and tests/test_dist_first_reach.py + tests/test_host_cpu.py pin against the oracle and
against the device (owner_ranks == k_owner_starts, first_reach_thresholds == the single-context batch).
*/
package core

/*
#include "bsched.h"
*/
import "C"

// setShard: replicated mode — the whole queue stays resident on every rank, the device decides ownership (whole groups,
// balanced by pod count).  Remembered in g.ranks: runBatch leaves BS_BATCH_FILTER_DENY (single-rank) out on a sharded context.
func (g *gpuCore) setShard(rank, nranks int) error {
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.check("bs_shard_set", C.bs_shard_set(g.ctx, C.uint32_t(rank), C.uint32_t(nranks))); err != nil {
		return err
	}
	g.ranks = nranks
	return nil
}

// setPartitioned: partitioned mode — this rank loads only the pods ownerRanks gives it and the caller (or bs_comm_init) reduces
// the admit counters.  firstReach = firstReachThreshold(...)[rank], set again behind every queue / group change (bsched.h).
func (g *gpuCore) setPartitioned(nranks int, firstReach uint32) error {
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.check("bs_reduce_external", C.bs_reduce_external(g.ctx, 1)); err != nil {
		return err
	}
	g.ranks = nranks
	return g.check("bs_first_reach_hint", C.bs_first_reach_hint(g.ctx, C.uint32_t(firstReach)))
}

// ownerRanks: the rank that evaluates each pod of the queue.  Walking the queue, the first pod of every group carries the
// weight of the group's pods (a pod outside the loaded groups carries 1); the running weight in front of a pod's anchor (the
// first pod of its group, or the pod itself) times nranks over the queue length is its rank.  Mirrors dist.owner_ranks and
// k_owner_starts bit for bit.
func ownerRanks(group []int32, nGroups, nranks int) []int {
	p := len(group)
	first := make([]int, nGroups)
	count := make([]int, nGroups)
	for i := range first {
		first[i] = p
	}
	valid := func(gi int32) bool { return gi >= 0 && int(gi) < nGroups }
	for i, gi := range group {
		if valid(gi) {
			if first[gi] == p {
				first[gi] = i
			}
			count[gi]++
		}
	}
	start := make([]int, p) // pods owned before queue position i
	run := 0
	for i, gi := range group {
		start[i] = run
		switch {
		case !valid(gi):
			run++
		case first[gi] == i:
			run += count[gi]
		}
	}
	out := make([]int, p)
	for i, gi := range group {
		anchor := i
		if valid(gi) {
			anchor = first[gi]
		}
		out[i] = start[anchor] * nranks / maxInt(p, 1)
	}
	return out
}

func maxInt(a, b int) int {
	if a > b {
		return a
	}
	return b
}

// shardGroup / shardPod: what firstReachThreshold needs of the group state and the queue (the same fields the ABI carries)
type shardGroup struct {
	flags           uint8 // BS_GROUP_*
	minMember       uint32
	statusScheduled uint32
	occupiedBy      uint64 // interned OccupiedBy, 0 == ""
}
type shardPod struct {
	group int32
	flags uint8  // BS_POD_*
	owner uint64 // interned joined owner UIDs, 0 == none
}

// firstReachThreshold: per rank, how many of ITS pods stand in front of the whole queue's first pod that reaches findMaxPG
// (core.go:118-123) — the argument of bs_first_reach_hint; 0xFFFFFFFF for every rank when no pod does.  A pod reaches when it is
// labelled with a known group (core.go:100-103), holds no lastPermittedPod entry (:95-98), its group is not deny-listed (:105-110),
// OccupiedBy agrees (:494-511: against the group's entry or, while the group has none, against the first pod of the group that
// brings owner references) and findMaxPG does not divide by zero (:716-717: then nobody reaches).  Restates
// dist.first_reach_thresholds (pinned on the oracle, tests/test_dist_first_reach.py).
func firstReachThreshold(pods []shardPod, groups []shardGroup, ranks []int, nranks int) []uint32 {
	none := make([]uint32, nranks)
	for r := range none {
		none[r] = 0xFFFFFFFF
	}
	p, G := len(pods), len(groups)
	if p == 0 || G == 0 {
		return none
	}
	for _, gr := range groups {
		cand := gr.flags&C.BS_GROUP_SCHEDULED_LATCH == 0 && gr.flags&C.BS_GROUP_HAS_POD != 0
		if cand && gr.minMember == 0 && gr.statusScheduled != 0 {
			return none // uint32 division by zero in findMaxPG: every pod panics there
		}
	}
	firstOwner := make([]int, G) // first pod of the group without a lastPermittedPod entry that has owner references
	for i := range firstOwner {
		firstOwner[i] = p
	}
	grouped := func(q shardPod) bool { return q.group >= 0 && int(q.group) < G }
	for i, q := range pods {
		if grouped(q) && q.flags&C.BS_POD_LAST_PERMITTED == 0 && q.owner != 0 && firstOwner[q.group] == p {
			firstOwner[q.group] = i
		}
	}
	first := -1
	for i, q := range pods {
		if !grouped(q) || q.flags&C.BS_POD_LAST_PERMITTED != 0 || groups[q.group].flags&C.BS_GROUP_DENIED != 0 {
			continue
		}
		occErr := false
		if occ := groups[q.group].occupiedBy; occ != 0 {
			occErr = q.owner == 0 || q.owner != occ
		} else if fo := firstOwner[q.group]; fo < p && i > fo {
			occErr = q.owner == 0 || q.owner != pods[fo].owner
		}
		if !occErr {
			first = i
			break
		}
	}
	if first < 0 {
		return none
	}
	out := make([]uint32, nranks)
	for i := 0; i < first; i++ {
		out[ranks[i]]++
	}
	return out
}
