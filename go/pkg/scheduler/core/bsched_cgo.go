// Package core — cgo binding of libbsched.so (include/bsched.h, ABI v6) for tenstack/batch-scheduler.
//
// SOURCE ONLY.  The image this library is developed in has no Go toolchain and k8s.io/kubernetes v1.17.5 is
// not vendored, so this file has been through neither `go build` nor `go vet`; it is kept as a real file
// (instead of prose in INTEGRATION.md) so that whoever has Go can gofmt / vet / build it.  Copy the files of
// this directory next to pkg/scheduler/core/core.go and build with
//
//	CGO_ENABLED=1 CGO_CFLAGS="-I$BSCHED/include" \
//	CGO_LDFLAGS="-L$BSCHED/batch-scheduler_amd -lbsched -Wl,-rpath,$BSCHED/batch-scheduler_amd" go build ./...
//
// (the reference builds with CGO_ENABLED=0, Makefile:28).
//
// cgo pointer rules, both halves: (1) the library keeps no pointer after a call returns; (2) no Go pointer passed to C points at
// Go memory that holds Go pointers — so the struct-taking entry points of bsched.h (bs_nodes_load, bs_groups_load, bs_pods_load,
// bs_pods_apply, bs_batch_read, bs_seq_run, bs_fit_build) are NEVER called from Go: their *_flat forms take every array as its
// own argument (tests/test_c11_client.py fails on any `C.bs_*(… &soa|&delta|&out …)` in this directory).  The only struct passed
// by pointer is bs_config (no pointers inside) and bs_batch_view (filled BY the library with pointers into ITS pinned memory).
//
// What stays in Go: the plugin surface (batchscheduler.go:102-216), label lookup (util/k8s.go:62), the TTL caches
// (core.go:71-72, cache.go:57-59 — they cross as flags and counts), resource.Quantity parsing, string interning,
// queue ordering (core.go:368-411), API-server I/O.  What crosses: flat int64 / uint32 / uint8 arrays.
package core

/*
#include <stdlib.h>
#include "bsched.h"
*/
import "C"

import (
	"fmt"
	"sort"
	"strings"
	"sync"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/kubernetes/pkg/scheduler/nodeinfo"
)

// gpuCore owns one bs_ctx; all mutating calls are serialised (bsched.h "Conventions").
type gpuCore struct {
	mu       sync.Mutex
	ctx      *C.bs_ctx
	scalars  []corev1.ResourceName // lane 4+s  <->  extended resource name
	classes  map[string]uint32     // pod-template signature -> fit class
	reps     []*corev1.Pod         // one representative pod per fit class
	interned map[string]uint64     // strings that are only compared (OccupiedBy, joined owner UIDs): 0 == ""
	nodes    int                   // nodes of the loaded snapshot
	infos    []*nodeinfo.NodeInfo  // the loaded snapshot itself (fit rows of classes that appear later are derived from it)
	nodeIdx  map[string]int        // node name -> list index of the loaded snapshot (Filter's bit test)
	fitRows  int                   // fit classes the device holds rows for (len(reps) at the last bs_fit_load)
	groups   int                   // groups of the last bs_groups_load (sizes admit / ready)
	ranks    int                   // ranks the context was sharded over (bs_shard_set; 0 / 1 = a single-rank context)
}

func newGPUCore(device int, scalars []corev1.ResourceName) (*gpuCore, error) {
	cfg := C.bs_config{abi_version: C.BS_ABI_VERSION, device: C.int32_t(device),
		scalar_lanes: C.uint32_t(len(scalars)), eph_gate: 1}
	var ctx *C.bs_ctx
	if rc := C.bs_create(&cfg, &ctx); rc != C.BS_OK {
		return nil, fmt.Errorf("bs_create: %s", C.GoString(C.bs_strerror(rc)))
	}
	return &gpuCore{ctx: ctx, scalars: scalars, classes: map[string]uint32{}, interned: map[string]uint64{"": 0}}, nil
}

func (g *gpuCore) close() { C.bs_destroy(g.ctx) }

func (g *gpuCore) check(where string, rc C.int) error {
	if rc == C.BS_OK {
		return nil
	}
	return fmt.Errorf("%s: %s (%s)", where, C.GoString(C.bs_strerror(rc)), C.GoString(C.bs_last_error(g.ctx)))
}

func (g *gpuCore) intern64(s string) uint64 {
	if v, ok := g.interned[s]; ok {
		return v
	}
	v := uint64(len(g.interned))
	g.interned[s] = v
	return v
}

// classOf: pods whose node selector, required node affinity and tolerations are equal share a fit class —
// checkFit (core.go:741-759) looks at nothing else of the pod.
func (g *gpuCore) classOf(p *corev1.Pod) uint32 {
	var b strings.Builder
	keys := make([]string, 0, len(p.Spec.NodeSelector))
	for k := range p.Spec.NodeSelector {
		keys = append(keys, k)
	}
	sort.Strings(keys)
	for _, k := range keys {
		fmt.Fprintf(&b, "%s=%s;", k, p.Spec.NodeSelector[k])
	}
	if a := p.Spec.Affinity; a != nil && a.NodeAffinity != nil && a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil {
		fmt.Fprintf(&b, "|%v", a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms)
	}
	fmt.Fprintf(&b, "|%v", p.Spec.Tolerations)
	sig := b.String()
	if c, ok := g.classes[sig]; ok {
		return c
	}
	c := uint32(len(g.classes))
	g.classes[sig] = c
	g.reps = append(g.reps, p)
	return c
}

// lanes flattens an upstream nodeinfo.Resource (MilliCPU, Memory, EphemeralStorage, AllowedPodNumber,
// ScalarResources) to L int64 lanes + presence bits (bit s: the Go map has key s).
func (g *gpuCore) lanes(r *nodeinfo.Resource, dst []C.int64_t, stride, i int) (present C.uint32_t) {
	dst[0*stride+i] = C.int64_t(r.MilliCPU)
	dst[1*stride+i] = C.int64_t(r.Memory)
	dst[2*stride+i] = C.int64_t(r.EphemeralStorage)
	dst[3*stride+i] = C.int64_t(r.AllowedPodNumber)
	for s, name := range g.scalars {
		if v, ok := r.ScalarResources[name]; ok {
			dst[(4+s)*stride+i] = C.int64_t(v)
			present |= 1 << uint(s)
		}
	}
	return
}

// loadSnapshot replaces the three SnapshotSharedLister() walks of core.go:437,567,597 — once per scheduling cycle.
func (g *gpuCore) loadSnapshot(infos []*nodeinfo.NodeInfo) error {
	n, L := len(infos), 4+len(g.scalars)
	alloc := make([]C.int64_t, L*n+1)
	req := make([]C.int64_t, L*n+1)
	ap, rp := make([]C.uint32_t, n+1), make([]C.uint32_t, n+1)
	flags := make([]C.uint8_t, n+1)
	words := (n + 31) / 32
	fit := make([]C.uint32_t, len(g.reps)*words+1)
	nodeIdx := make(map[string]int, n)
	for i, info := range infos {
		switch {
		case info == nil: // core.go:606
			flags[i] = C.BS_NODE_NIL
			continue
		case info.Node() == nil: // core.go:610
			flags[i] = C.BS_NODE_NO_NODE
			continue
		}
		if info.Node().Spec.Unschedulable { // core.go:615
			flags[i] |= C.BS_NODE_UNSCHEDULABLE
		}
		if _, err := info.Taints(); err != nil { // core.go:639
			flags[i] |= C.BS_NODE_TAINT_ERR
		}
		nodeIdx[info.Node().Name] = i
		a, r := info.AllocatableResource(), info.RequestedResource()
		if r.AllowedPodNumber == 0 { // core.go:650-653: podCount
			r.AllowedPodNumber = len(info.Pods())
		}
		ap[i] = g.lanes(&a, alloc, n, i)
		rp[i] = g.lanes(&r, req, n, i)
		for c, rep := range g.reps { // checkFit, core.go:741-759, once per class and node (or bs_fit_build on the device)
			if checkFit(rep, info) {
				fit[c*words+i/32] |= 1 << uint(i%32)
			}
		}
	}
	// cgo pointer rule: every array crosses as its OWN argument (bs_nodes_load_flat) — a Go-allocated C.bs_nodes_soa holding
	// these slice pointers, passed by pointer, would be a Go pointer to Go memory that holds Go pointers (cgocheck panic).
	g.mu.Lock()
	defer g.mu.Unlock()
	if err := g.check("bs_nodes_load_flat", C.bs_nodes_load_flat(g.ctx, C.uint32_t(n), &alloc[0], &req[0], &ap[0], &rp[0], &flags[0])); err != nil {
		return err
	}
	g.nodes, g.infos, g.nodeIdx = n, infos, nodeIdx
	if err := g.check("bs_fit_load", C.bs_fit_load(g.ctx, C.uint32_t(len(g.reps)), &fit[0])); err != nil {
		return err
	}
	g.fitRows = len(g.reps)
	return nil
}

// ensureFitRows: classOf may have met a pod template the loaded fit rows do not cover (a new nodeSelector / affinity /
// tolerations signature in the queue or in a group's pod).  bs_batch_run refuses class indices beyond the loaded rows
// (BS_ERR_INVALID), so the rows of every class are derived again from the loaded snapshot before the batch.
// Caller holds g.mu.
func (g *gpuCore) ensureFitRows() error {
	if len(g.reps) == g.fitRows || g.infos == nil {
		return nil
	}
	words := (g.nodes + 31) / 32
	fit := make([]C.uint32_t, len(g.reps)*words+1)
	for i, info := range g.infos {
		if info == nil || info.Node() == nil {
			continue
		}
		for c, rep := range g.reps { // checkFit, core.go:741-759
			if checkFit(rep, info) {
				fit[c*words+i/32] |= 1 << uint(i%32)
			}
		}
	}
	if err := g.check("bs_fit_load", C.bs_fit_load(g.ctx, C.uint32_t(len(g.reps)), &fit[0])); err != nil {
		return err
	}
	g.fitRows = len(g.reps)
	return nil
}

// clusterFits is the 1:1 replacement of the body of compareClusterResourceAndRequire (core.go:595-632):
// same arguments, same bool.
//
//	func (sop *ScheduleOperation) compareClusterResourceAndRequire(pod *corev1.Pod, reqResource *nodeinfo.Resource, percent float32) bool {
//		return sop.gpu.clusterFits(sop.gpu.classOf(pod), reqResource, percent)
//	}
func (g *gpuCore) clusterFits(class uint32, req *nodeinfo.Resource, percent float32) bool {
	L := 4 + len(g.scalars)
	lanes := make([]C.int64_t, L)
	present := g.lanes(req, lanes, 1, 0)
	var fits C.uint8_t
	var firstK C.uint32_t
	g.mu.Lock()
	rc := C.bs_cluster_fits(g.ctx, C.uint32_t(class), C.float(percent), &lanes[0], present, &fits, &firstK)
	g.mu.Unlock()
	return rc == C.BS_OK && fits != 0
}

// cycleCounters: how often the library had to run a batch twice (metrics for the shim's dashboards; both stay at or near zero
// in a steady state).  reruns: BS_BATCH_FILTER_DENY batches settled by fixed-point re-runs (a pod let through on its
// lastPermittedPod entry failed Filter in front of its group's first eligible pod or of the batch's first findMaxPG call).
// guessed / missed: batches launched on the previous cycle's findMaxPG answer, and wrong guesses that were run again.
func (g *gpuCore) cycleCounters() (reruns, guessed, missed uint64) {
	var r, a, m C.uint64_t
	g.mu.Lock()
	defer g.mu.Unlock()
	if C.bs_filter_deny_stats(g.ctx, &r) != C.BS_OK || C.bs_speculation_stats(g.ctx, &a, &m) != C.BS_OK {
		return 0, 0, 0
	}
	return uint64(r), uint64(a), uint64(m)
}
