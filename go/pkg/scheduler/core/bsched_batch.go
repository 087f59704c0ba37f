// Package core — the batched mode of the cgo shim: one scheduling cycle = patch the groups that changed, hand over
// the drained queue, run, read; the plugin hooks then answer from tables with no cgo crossing.
//
// SOURCE ONLY (see bsched_cgo.go).
package core

/*
#include "bsched.h"
*/
import "C"

import (
	"unsafe"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/types"
	"k8s.io/kubernetes/pkg/scheduler/nodeinfo"

	pgv1 "github.com/tenstack/batch-scheduler/pkg/apis/podgroup/v1"
	"github.com/tenstack/batch-scheduler/pkg/scheduler/cache"
	"github.com/tenstack/batch-scheduler/pkg/util"
)

// loadGroups flattens the PodGroup cache (cache.go:45-67) in a FIXED iteration order (findMaxPG ranges over a Go map,
// core.go:703; parity is defined for the order handed over here, e.g. sorted by full name).  Full load: after a
// first-pod capture or when groups come and go; per-cycle counter changes go through patchGroups.
func (g *gpuCore) loadGroups(order []string, pgCache map[string]*cache.PodGroupMatchStatus, denied func(string) bool) error {
	G, L := len(order), 4+len(g.scalars)
	mm, sc, matched := make([]C.uint32_t, G+1), make([]C.uint32_t, G+1), make([]C.uint32_t, G+1)
	flags, cls := make([]C.uint8_t, G+1), make([]C.uint32_t, G+1)
	minres, mrp, occ := make([]C.int64_t, L*G+1), make([]C.uint32_t, G+1), make([]C.uint64_t, G+1)
	for i, name := range order {
		pgs := pgCache[name]
		mm[i] = C.uint32_t(pgs.PodGroup.Spec.MinMember)
		sc[i] = C.uint32_t(pgs.PodGroup.Status.Scheduled)
		matched[i] = C.uint32_t(len(pgs.MatchedPodNodes.Items())) // core.go:716
		flags[i] = C.uint8_t(g.groupFlags(name, pgs, denied))
		if pgs.Pod != nil {
			cls[i] = C.uint32_t(g.classOf(pgs.Pod))
		}
		if mr := pgs.PodGroup.Spec.MinResources; mr != nil { // core.go:489-493
			var r nodeinfo.Resource
			r.Add(*mr)
			mrp[i] = g.lanes(&r, minres, G, i)
		}
		occ[i] = C.uint64_t(g.intern64(pgs.PodGroup.Status.OccupiedBy)) // 0 == ""
	}
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.ensureFitRows(); err != nil { // classOf(pgs.Pod) above may have met a new pod template
		return err
	}
	// flat form: every array its own argument (cgo pointer rule, see bsched_cgo.go)
	if err := g.check("bs_groups_load_flat", C.bs_groups_load_flat(g.ctx, C.uint32_t(G), &mm[0], &sc[0], &matched[0], &flags[0], &cls[0],
		&minres[0], &mrp[0], &occ[0])); err != nil {
		return err
	}
	g.groups = G
	return nil
}

func (g *gpuCore) groupFlags(name string, pgs *cache.PodGroupMatchStatus, denied func(string) bool) uint32 {
	var f uint32
	if pgs.Scheduled { // core.go:305
		f |= C.BS_GROUP_SCHEDULED_LATCH
	}
	if pgs.Pod != nil {
		f |= C.BS_GROUP_HAS_POD
	}
	if pgs.PodGroup.Spec.MinResources != nil {
		f |= C.BS_GROUP_HAS_MINRES
	}
	if denied(name) { // live lastDeniedPG entry, core.go:105
		f |= C.BS_GROUP_DENIED
	}
	// StartBatchSchedule releases only in phase PreScheduling / Scheduling (batchscheduler.go:258-261; Permit turns Pending into
	// PreScheduling first, core.go:279-281): every other phase is "closed" for the pod-by-pod pass (bs_seq_run)
	switch pgs.PodGroup.Status.Phase {
	case pgv1.PodGroupPending, pgv1.PodGroupPreScheduling, pgv1.PodGroupScheduling, "":
	default:
		f |= C.BS_GROUP_PHASE_CLOSED
	}
	return f
}

// patchGroups: what one scheduling cycle changes — Permit adds to MatchedPodNodes (core.go:290), PostBind moves pods to
// Status.Scheduled (core.go:327), the quorum latch (core.go:305), deny entries appear and expire.  One kernel launch on
// the library side, nothing waited for.
func (g *gpuCore) patchGroups(index map[string]uint32, changed []string, pgCache map[string]*cache.PodGroupMatchStatus, denied func(string) bool) error {
	if len(changed) == 0 {
		return nil
	}
	d := make([]C.bs_group_delta, len(changed))
	for i, name := range changed {
		pgs := pgCache[name]
		d[i] = C.bs_group_delta{index: C.uint32_t(index[name]), matched: C.uint32_t(len(pgs.MatchedPodNodes.Items())),
			status_scheduled: C.uint32_t(pgs.PodGroup.Status.Scheduled), flags: C.uint32_t(g.groupFlags(name, pgs, denied))}
	}
	g.mu.Lock()
	defer g.mu.Unlock()
	return g.check("bs_groups_apply", C.bs_groups_apply(g.ctx, &d[0], C.uint32_t(len(d))))
}

// batchResult: Go-owned result arrays of one batch.  Filter is answered from the slot rows (distinct requests), never
// from a pods x nodes bitmap.
type batchResult struct {
	index     map[types.UID]int // pod -> queue position
	nodeIndex map[string]int    // node name -> list index of the snapshot
	n         int
	pfCode    []C.uint8_t
	pfFirstK  []C.uint32_t
	flCode    []C.uint8_t
	flSlot    []C.uint32_t
	rows      []C.uint64_t // [ceil(n/64)][rowsCap]
	rowsCap   int
	admit     []C.uint32_t
	ready     []C.uint8_t
}

// filterPasses: Filter(pod i, node k) of core.go:170-191 as a bit test.
func (r *batchResult) filterPasses(i, k int) bool {
	switch code := r.flCode[i]; {
	case code == C.BS_FL_EVALUATED:
		return r.rows[(k>>6)*r.rowsCap+int(r.flSlot[i])]>>(uint(k)&63)&1 == 1
	default:
		return code < 16 // pass on every node (not grouped / leader itself / no MinResources) or on none (error codes)
	}
}

// runBatch scores the drained scheduling queue (already ordered by Less) in one call.
func (g *gpuCore) runBatch(queue []*corev1.Pod, groupIndex map[string]uint32, permitted func(*corev1.Pod) bool) (*batchResult, error) {
	P, L := len(queue), 4+len(g.scalars)
	grp, req := make([]C.int32_t, P+1), make([]C.int64_t, L*P+1)
	pres, cls, owner, flags := make([]C.uint32_t, P+1), make([]C.uint32_t, P+1), make([]C.uint64_t, P+1), make([]C.uint8_t, P+1)
	res := &batchResult{index: make(map[types.UID]int, P), nodeIndex: g.nodeIdx, n: g.nodes}
	for i, p := range queue {
		res.index[p.UID] = i
		name, ok := util.VerifyPodLabelSatisfied(p) // util/k8s.go:62-70
		switch gi, found := groupIndex[p.Namespace+"/"+name]; {
		case !ok:
			grp[i] = C.BS_POD_NOT_GROUPED // core.go:89-92
		case !found:
			grp[i] = C.BS_POD_GROUP_MISSING // core.go:100-103
		default:
			grp[i] = C.int32_t(gi)
		}
		pres[i] = g.lanes(getPodResourceRequire(p), req, P, i) // core.go:761-772
		cls[i] = C.uint32_t(g.classOf(p))
		owner[i] = C.uint64_t(g.intern64(joinedOwnerUIDs(p))) // core.go:498-499, 0 = none
		if permitted(p) {                                    // live lastPermittedPod entry, core.go:95
			flags[i] |= C.BS_POD_LAST_PERMITTED
		}
	}
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.ensureFitRows(); err != nil { // a pod template the loaded fit rows do not cover yet: one new template must not fail the cycle
		return nil, err
	}
	if err := g.check("bs_pods_load_flat", C.bs_pods_load_flat(g.ctx, C.uint32_t(P), &grp[0], &req[0], &pres[0], &cls[0], &owner[0], &flags[0])); err != nil {
		return nil, err
	}
	// Filter is on in this form: BS_BATCH_FILTER_DENY replays the deny entry a failing Filter writes (core.go:183-185) inside the
	// batch, on the device — the codes are those of PreFilter + Filter-on-every-node, pod by pod (round 3 did that in a Go pass here)
	stages := C.uint32_t(C.BS_STAGE_ALL)
	if g.ranks <= 1 { // the deny replay is single-rank (bs_batch_run answers BS_ERR_STATE on sharded / externally reduced contexts)
		stages |= C.BS_BATCH_FILTER_DENY
	}
	if err := g.check("bs_batch_run", C.bs_batch_run(g.ctx, stages)); err != nil {
		return nil, err
	}
	var rowsNeeded C.uint32_t
	if err := g.check("bs_filter_rows_count", C.bs_filter_rows_count(g.ctx, &rowsNeeded)); err != nil {
		return nil, err
	}
	G, W := g.groups, (g.nodes+63)/64 // the library writes one admit / ready entry per LOADED group
	res.rowsCap = int(rowsNeeded) + 1
	res.pfCode, res.pfFirstK = make([]C.uint8_t, P+1), make([]C.uint32_t, P+1)
	res.flCode, res.flSlot = make([]C.uint8_t, P+1), make([]C.uint32_t, P+1)
	res.rows = make([]C.uint64_t, W*res.rowsCap+1)
	res.admit, res.ready = make([]C.uint32_t, G+1), make([]C.uint8_t, G+1)
	var rowsN C.uint32_t
	// (pf_leader, fl_feasible, fl_bitmap, fl_rows_feasible not wanted: NULL)
	return res, g.check("bs_batch_read_flat", C.bs_batch_read_flat(g.ctx, &res.pfCode[0], &res.pfFirstK[0], nil, &res.flCode[0], nil, nil,
		&res.admit[0], &res.ready[0], &res.flSlot[0], &res.rows[0], nil, C.uint32_t(res.rowsCap), &rowsN)) // the one stream wait of the cycle
}

// queueDelta: what changed in the pending queue since the last cycle, in queue positions (bs_pods_delta): pods that left
// (bound, deleted: strictly ascending indices of the OLD queue), pods whose lastPermittedPod entry appeared or expired, new
// pods with their positions in the NEW queue (nil = appended).  The scheduling queue is a heap ordered by Less, so the shim
// keeps the drained order of the previous cycle and diffs against it.
type queueDelta struct {
	remove    []uint32
	flagIndex []uint32
	flagValue []uint8
	insert    []*corev1.Pod
	insertAt  []uint32
}

// cycleView: the results of a latency-mode batch, read IN PLACE from the pinned memory the last launch wrote (bs_batch_map):
// no device-to-host copy, no stream wait, no host-side copy.  Valid until the next cycle runs.
type cycleView struct {
	v    C.bs_batch_view
	copy *cycleCopy // set when bs_batch_map declined (BS_ERR_STATE): the results were copied out with bs_batch_read_flat instead
}

// cycleCopy: the cycle's results in Go memory (the fall-back of runCycle): everything the accessors below hand out
type cycleCopy struct {
	pfCode  []C.uint8_t
	ready   []C.uint8_t
	flCode  []C.uint8_t
	flSlot  []C.uint32_t
	rows    []C.uint64_t // [W][rowsCap]
	rowsCap int
}

func (c *cycleView) pfCode(i int) uint8 {
	if c.copy != nil {
		return uint8(c.copy.pfCode[i])
	}
	return uint8(*(*C.uint8_t)(unsafe.Pointer(uintptr(unsafe.Pointer(c.v.pf_code)) + uintptr(i))))
}
func (c *cycleView) groupReady(g int) bool {
	if c.copy != nil {
		return c.copy.ready[g] != 0
	}
	return *(*C.uint8_t)(unsafe.Pointer(uintptr(unsafe.Pointer(c.v.group_ready)) + uintptr(g))) != 0
}
func (c *cycleView) filterPasses(i, k int) bool { // Filter(pod i, node k), core.go:170-191, as a bit test in the rows (mapped, or copied out)
	if c.copy != nil {
		code := c.copy.flCode[i]
		if code != C.BS_FL_EVALUATED {
			return code < 16
		}
		word := c.copy.rows[(k>>6)*c.copy.rowsCap+int(c.copy.flSlot[i])]
		return word>>(uint(k)&63)&1 == 1
	}
	code := *(*C.uint8_t)(unsafe.Pointer(uintptr(unsafe.Pointer(c.v.fl_code)) + uintptr(i)))
	if code != C.BS_FL_EVALUATED {
		return code < 16
	}
	slot := *(*C.uint32_t)(unsafe.Pointer(uintptr(unsafe.Pointer(c.v.fl_slot)) + uintptr(i)*4))
	word := *(*C.uint64_t)(unsafe.Pointer(uintptr(unsafe.Pointer(c.v.fl_rows)) + (uintptr(k>>6)*uintptr(c.v.fl_rows_stride)+uintptr(slot))*8))
	return word>>(uint(k)&63)&1 == 1
}

// runCycle: one scheduling cycle on the RESIDENT queue — patch the groups whose counters moved (patchGroups; its launch rides
// in the queue patch's), patch the queue (bs_pods_apply: stable removal, flag flips, insertion, all on the device), run the
// batch in latency mode, map the results.  tests/test_c11_client.py performs this very sequence through the same header.
func (g *gpuCore) runCycle(d *queueDelta, groupIndex map[string]uint32, permitted func(*corev1.Pod) bool) (*cycleView, error) {
	I, L := len(d.insert), 4+len(g.scalars)
	grp, req := make([]C.int32_t, I+1), make([]C.int64_t, L*I+1)
	pres, cls, owner, flags := make([]C.uint32_t, I+1), make([]C.uint32_t, I+1), make([]C.uint64_t, I+1), make([]C.uint8_t, I+1)
	for i, p := range d.insert {
		name, ok := util.VerifyPodLabelSatisfied(p)
		switch gi, found := groupIndex[p.Namespace+"/"+name]; {
		case !ok:
			grp[i] = C.BS_POD_NOT_GROUPED
		case !found:
			grp[i] = C.BS_POD_GROUP_MISSING
		default:
			grp[i] = C.int32_t(gi)
		}
		pres[i] = g.lanes(getPodResourceRequire(p), req, I, i)
		cls[i] = C.uint32_t(g.classOf(p))
		owner[i] = C.uint64_t(g.intern64(joinedOwnerUIDs(p)))
		if permitted(p) {
			flags[i] |= C.BS_POD_LAST_PERMITTED
		}
	}
	// every array of the delta its own argument (bs_pods_apply_flat): no Go-allocated bs_pods_delta, no Go pointer to Go pointers
	var remove, flagIndex, insertAt *C.uint32_t
	var flagValue *C.uint8_t
	if len(d.remove) > 0 {
		remove = (*C.uint32_t)(unsafe.Pointer(&d.remove[0]))
	}
	if len(d.flagIndex) > 0 {
		flagIndex = (*C.uint32_t)(unsafe.Pointer(&d.flagIndex[0]))
		flagValue = (*C.uint8_t)(unsafe.Pointer(&d.flagValue[0]))
	}
	if len(d.insertAt) > 0 {
		insertAt = (*C.uint32_t)(unsafe.Pointer(&d.insertAt[0]))
	}
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.ensureFitRows(); err != nil {
		return nil, err
	}
	if err := g.check("bs_pods_apply_flat", C.bs_pods_apply_flat(g.ctx, C.uint32_t(len(d.remove)), remove, C.uint32_t(len(d.flagIndex)), flagIndex, flagValue,
		C.uint32_t(I), &grp[0], &req[0], &pres[0], &cls[0], &owner[0], &flags[0], insertAt)); err != nil {
		return nil, err
	}
	// BS_ERR_RETRY (ABI v7): the batch's results are void for a reason the library has repaired by the time it says so (the id space
	// of the queue patch overflowed and the queue was re-derived; an in-launch hand-over timed out) — run the batch again, nothing of
	// the caller's state is wrong.  BS_ERR_STATE from bs_batch_map is something else: a VALID batch that wrote no host results.
	for attempt := 0; ; attempt++ {
		if err := g.check("bs_batch_run", C.bs_batch_run(g.ctx, C.BS_STAGE_PREFILTER|C.BS_STAGE_TALLY|C.BS_BATCH_HOST_RESULTS)); err != nil {
			return nil, err
		}
		res := &cycleView{}
		rc := C.bs_batch_map(g.ctx, &res.v)
		if rc == C.BS_ERR_RETRY && attempt < 2 {
			continue
		}
		if rc == C.BS_ERR_STATE {
			// not a failed cycle: the batch left the three-launch chains (more than sixteen leader runs, or a re-run behind a wrong
			// table guess ended on the general chain), which write no host results — the results are valid, copy ALL of them out
			// (every accessor of cycleView reads the copy then)
			var pc, rowsNeeded, rowsN C.uint32_t
			if err := g.check("bs_pods_count", C.bs_pods_count(g.ctx, &pc)); err != nil {
				return nil, err
			}
			if err := g.check("bs_filter_rows_count", C.bs_filter_rows_count(g.ctx, &rowsNeeded)); err != nil {
				return nil, err
			}
			P, W := int(pc), (g.nodes+63)/64
			cp := &cycleCopy{pfCode: make([]C.uint8_t, P+1), ready: make([]C.uint8_t, g.groups+1), flCode: make([]C.uint8_t, P+1),
				flSlot: make([]C.uint32_t, P+1), rowsCap: int(rowsNeeded) + 1}
			cp.rows = make([]C.uint64_t, W*cp.rowsCap+1)
			rc = C.bs_batch_read_flat(g.ctx, &cp.pfCode[0], nil, nil, &cp.flCode[0], nil, nil, nil, &cp.ready[0],
				&cp.flSlot[0], &cp.rows[0], nil, C.uint32_t(cp.rowsCap), &rowsN)
			if rc == C.BS_ERR_RETRY && attempt < 2 {
				continue
			}
			if err := g.check("bs_batch_read_flat", rc); err != nil {
				return nil, err
			}
			res.copy = cp
			return res, nil
		}
		if err := g.check("bs_batch_map", rc); err != nil {
			return nil, err
		}
		return res, nil
	}
}

// seqPass: the reference's own order of events on the device (bs_seq_run): PreFilter -> node choice -> assume -> Permit -> release,
// pod by pod over the resident queue, one launch.  The decisions are the ones core.go:88-167 / :268-309 would take in sequence
// (bit-identical to the CPU replay, tests/test_gpu_seq.py); the plugin binds the released pods to podNode[i].
type seqResult struct {
	pfCode        []C.uint8_t
	podNode       []C.int32_t
	releasedGroup []C.uint32_t
	releasedPods  []C.uint32_t
	readyNs       []C.int64_t
	lastPermitted []C.uint8_t // withFilter: 1 = Filter passed somewhere -> the plugin adds the pod to lastPermittedPod (core.go:188, 2 s TTL)
	nReleased     int
}

func (g *gpuCore) seqPass(P int, withFilter bool) (*seqResult, error) {
	cap := g.groups + 1
	r := &seqResult{pfCode: make([]C.uint8_t, P+1), podNode: make([]C.int32_t, P+1), releasedGroup: make([]C.uint32_t, cap),
		releasedPods: make([]C.uint32_t, cap), readyNs: make([]C.int64_t, cap), lastPermitted: make([]C.uint8_t, P+1)}
	firstNs := make([]C.int64_t, cap)
	var scalars [8]C.int64_t
	stages := C.uint32_t(C.BS_STAGE_PREFILTER)
	if withFilter { // Filter as the reference runs it: it gates the node choice AND writes its deny / lastPermittedPod entries (core.go:183-188)
		stages |= C.BS_STAGE_FILTER | C.BS_BATCH_FILTER_DENY
	}
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.check("bs_seq_run_flat", C.bs_seq_run_flat(g.ctx, stages, &r.pfCode[0], nil, nil, &r.podNode[0], C.uint32_t(cap),
		&r.releasedGroup[0], &r.releasedPods[0], &firstNs[0], &r.readyNs[0], &scalars[0], &r.lastPermitted[0])); err != nil {
		return nil, err
	}
	r.nReleased = int(scalars[0])
	return r, nil
}

// All slices live only for the duration of the calls (cgo pointer rules: the library copies and keeps no Go pointer).
//
// Plugin hooks over a batchResult (batchscheduler.go:102,151,165) — table look-ups, no cgo crossing:
//
//	func (bs *batchSchedulingPlugin) PreFilter(ctx context.Context, st *framework.CycleState, p *corev1.Pod) *framework.Status {
//		i, ok := bs.batch.index[p.UID]
//		if !ok { // not part of the scored queue: fall back to the sequential path, never to pod 0
//			return bs.preFilterSequential(ctx, st, p)
//		}
//		code := bs.batch.pfCode[i]
//		if code >= 16 { // !BS_PF_IS_PASS
//			if code == C.BS_PF_REJECT_FIRST || code == C.BS_PF_REJECT_RESERVE {
//				bs.operation.AddToDenyCache(fullName(p)) // the 20 s TTL clock stays in Go, core.go:423
//			}
//			return framework.NewStatus(framework.Unschedulable, pfMessage[code]) // same strings as core.go:102,107,143,164
//		}
//		return framework.NewStatus(framework.Success, "")
//	}
//
//	func (bs *batchSchedulingPlugin) Filter(ctx context.Context, st *framework.CycleState, p *corev1.Pod, ni *nodeinfo.NodeInfo) *framework.Status {
//		i, okp := bs.batch.index[p.UID]
//		k, okn := bs.batch.nodeIndex[ni.Node().Name]
//		if !okp || !okn { // a pod or node the batch has not seen: answer through bs_filter_one, never from row 0
//			return bs.filterSequential(ctx, st, p, ni)
//		}
//		if !bs.batch.filterPasses(i, k) {
//			return framework.NewStatus(framework.Unschedulable, util.ErrorResourceNotEnough.Error())
//		}
//		return framework.NewStatus(framework.Success, "")
//	}
//
// Permit keeps its TTL-map bookkeeping (core.go:284-300).  bs_batch_out.group_ready is the quorum predicate (:303) for
// "every pod of the queue that passes is permitted" against the FROZEN snapshot: a pre-screen, not a reservation — the
// batch does not charge the capacity one gang takes to the next (tests/test_batch_vs_sequential.py states the relation);
// after admitting a gang the shim patches the node requests (bs_nodes_apply) and group counters (patchGroups) and re-runs
// the batch for the rest of the queue (tens of microseconds).

func joinedOwnerUIDs(p *corev1.Pod) string { // core.go:494-499: sorted, comma-joined OwnerReferences UIDs
	if len(p.OwnerReferences) == 0 {
		return ""
	}
	ids := make([]string, 0, len(p.OwnerReferences))
	for _, o := range p.OwnerReferences {
		ids = append(ids, string(o.UID))
	}
	sortStrings(ids)
	out := ids[0]
	for _, s := range ids[1:] {
		out += "," + s
	}
	return out
}

func sortStrings(a []string) {
	for i := 1; i < len(a); i++ {
		for j := i; j > 0 && a[j] < a[j-1]; j-- {
			a[j], a[j-1] = a[j-1], a[j]
		}
	}
}
