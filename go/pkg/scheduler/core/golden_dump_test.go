// Package core — golden-vector dump from the REFERENCE's own functions, for pinning the CPU oracle of the MI355X core.
//
// SOURCE ONLY: never compiled or run where this library is developed (no Go toolchain, k8s.io/kubernetes v1.17.5 not
// vendored).  To produce the vectors, copy this file next to pkg/scheduler/core/core.go of tenstack/batch-scheduler
// (it calls unexported functions) and run
//
//	BS_GOLDEN_IN=$BSCHED/tests/golden/go_reference_input.json \
//	BS_GOLDEN_OUT=$BSCHED/tests/golden/go_reference_dump.json go test -run TestGoldenDump ./pkg/scheduler/core/
//
// The input is written by tools/dump_golden_input.py (seeded, small).  tests/test_go_reference_dump.py then checks the
// oracle (oracle/bs_oracle.c) against every entry of the dump; until the dump exists that test is skipped and the rows of
// SURVEY.md 8(c) it covers stay "parity unpinned by the reference".
//
// What is dumped, all through the reference's own code paths:
//
//	find_max_pg         findMaxPG (core.go:701-739) called repeatedly — Go map order is random, so the SET of leaders seen
//	pre_allocated       getPreAllocatedResource (core.go:774-793) for every group at its matched count and at 0
//	single_node         singleNodeResource (core.go:634-670) for every (class, node) at percent 1 and 0.7
//	cluster_fits        compareClusterResourceAndRequire (core.go:595-632) + the index at which its loop exits, re-derived
//	                    with the reference's singleNodeResource / compareResourceAndRequire (the function returns only a bool)
//	left_resource       getLeftResource (core.go:436-475)
//	filter              computeResourceSatisfied (core.go:514-564) for sampled (pod, node) pairs under the dumped leader
//	prefilter_sequence  ScheduleOperation.PreFilter (core.go:88-167) for the pods in queue order, error strings
package core

import (
	"encoding/json"
	"fmt"
	"os"
	"sort"
	"testing"
	"time"

	gochache "github.com/patrickmn/go-cache"
	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
	"k8s.io/apimachinery/pkg/types"
	framework "k8s.io/kubernetes/pkg/scheduler/framework/v1alpha1"
	"k8s.io/kubernetes/pkg/scheduler/listers"
	"k8s.io/kubernetes/pkg/scheduler/nodeinfo"

	pgv1 "github.com/tenstack/batch-scheduler/pkg/apis/podgroup/v1"
	"github.com/tenstack/batch-scheduler/pkg/scheduler/cache"
	"github.com/tenstack/batch-scheduler/pkg/util"
)

type goldenNode struct {
	Name          string  `json:"name"`
	Alloc         []int64 `json:"alloc"`
	AllocKeys     []bool  `json:"alloc_keys"`
	Requested     []int64 `json:"requested"`
	ReqKeys       []bool  `json:"req_keys"`
	PodCount      int     `json:"pod_count"`
	Unschedulable bool    `json:"unschedulable"`
	Fit           []bool  `json:"fit"` // per class
}

type goldenGroup struct {
	Name         string  `json:"name"` // "ns/name"
	MinMember    uint32  `json:"min_member"`
	Scheduled    uint32  `json:"scheduled"`
	Matched      int     `json:"matched"`
	Latch        bool    `json:"latch"`
	HasPod       bool    `json:"has_pod"`
	Cls          int     `json:"cls"`
	MinResources []int64 `json:"min_resources"` // nil: Spec.MinResources == nil
	MinResKeys   []bool  `json:"min_res_keys"`
}

type goldenPod struct {
	UID     string  `json:"uid"`
	Group   string  `json:"group"` // group NAME without namespace, "" = no label
	Req     []int64 `json:"req"`
	ReqKeys []bool  `json:"req_keys"`
	Cls     int     `json:"cls"`
}

type goldenQuery struct {
	Cls     int     `json:"cls"`
	Percent float32 `json:"percent"`
	Req     []int64 `json:"req"`
	ReqKeys []bool  `json:"req_keys"`
}

type goldenInput struct {
	Lanes       []string      `json:"lanes"`
	Classes     int           `json:"classes"`
	Nodes       []goldenNode  `json:"nodes"`
	Groups      []goldenGroup `json:"groups"`
	Pods        []goldenPod   `json:"pods"`
	Queries     []goldenQuery `json:"queries"`
	FilterPairs [][2]int      `json:"filter_pairs"`
}

type gLister struct {
	infos  []*nodeinfo.NodeInfo
	byName map[string]*nodeinfo.NodeInfo
}

func (f gLister) List() ([]*nodeinfo.NodeInfo, error)                    { return f.infos, nil }
func (f gLister) HavePodsWithAffinityList() ([]*nodeinfo.NodeInfo, error) { return nil, nil }
func (f gLister) Get(name string) (*nodeinfo.NodeInfo, error) {
	if i, ok := f.byName[name]; ok {
		return i, nil
	}
	return nil, fmt.Errorf("nodeinfo not found for node name %q", name)
}
func (f gLister) Pods() listers.PodLister           { return nil }
func (f gLister) NodeInfos() listers.NodeInfoLister { return f }

type gHandle struct {
	framework.FrameworkHandle
	l gLister
}

func (h gHandle) SnapshotSharedLister() listers.SharedLister { return h.l }

func gq(lane string, v int64) resource.Quantity {
	if lane == "cpu" {
		return *resource.NewMilliQuantity(v, resource.DecimalSI)
	}
	return *resource.NewQuantity(v, resource.BinarySI)
}

func gList(lanes []string, v []int64, keys []bool, skipPods bool) corev1.ResourceList {
	rl := corev1.ResourceList{}
	for l, name := range lanes {
		if l == 3 && skipPods {
			continue
		}
		if l < 4 || keys[l-4] {
			rl[corev1.ResourceName(name)] = gq(name, v[l])
		}
	}
	return rl
}

// class c's representative pod selects nodes labelled fit-c=1: checkFit (core.go:741-759) then reproduces fit[c][node]
func gRepPod(cls int, name, ns, group string, uid string, req corev1.ResourceList) *corev1.Pod {
	p := &corev1.Pod{ObjectMeta: metav1.ObjectMeta{Name: name, Namespace: ns, UID: types.UID(uid)},
		Spec: corev1.PodSpec{NodeSelector: map[string]string{fmt.Sprintf("fit-%d", cls): "1"},
			Containers: []corev1.Container{{Resources: corev1.ResourceRequirements{Requests: req}}}}}
	if group != "" {
		p.Labels = map[string]string{util.PodGroupLabel: group}
	}
	return p
}

func gLanes(in *goldenInput, r *nodeinfo.Resource) ([]int64, []bool) {
	out := []int64{r.MilliCPU, r.Memory, r.EphemeralStorage, int64(r.AllowedPodNumber)}
	keys := []bool{}
	for _, name := range in.Lanes[4:] {
		v, ok := r.ScalarResources[corev1.ResourceName(name)]
		out = append(out, v)
		keys = append(keys, ok)
	}
	return out, keys
}

func gResource(in *goldenInput, v []int64, keys []bool) *nodeinfo.Resource {
	r := &nodeinfo.Resource{MilliCPU: v[0], Memory: v[1], EphemeralStorage: v[2], AllowedPodNumber: int(v[3])}
	for l := 4; l < len(in.Lanes); l++ {
		if keys[l-4] {
			r.SetScalar(corev1.ResourceName(in.Lanes[l]), v[l])
		}
	}
	return r
}

func TestGoldenDump(t *testing.T) {
	inPath, outPath := os.Getenv("BS_GOLDEN_IN"), os.Getenv("BS_GOLDEN_OUT")
	if inPath == "" || outPath == "" {
		t.Skip("BS_GOLDEN_IN / BS_GOLDEN_OUT not set")
	}
	raw, err := os.ReadFile(inPath)
	if err != nil {
		t.Fatal(err)
	}
	var in goldenInput
	if err := json.Unmarshal(raw, &in); err != nil {
		t.Fatal(err)
	}
	// ---- node snapshot
	lister := gLister{byName: map[string]*nodeinfo.NodeInfo{}}
	for _, n := range in.Nodes {
		labels := map[string]string{}
		for c, ok := range n.Fit {
			if ok {
				labels[fmt.Sprintf("fit-%d", c)] = "1"
			}
		}
		alloc := gList(in.Lanes, n.Alloc, n.AllocKeys, false)
		node := &corev1.Node{ObjectMeta: metav1.ObjectMeta{Name: n.Name, Labels: labels},
			Spec:   corev1.NodeSpec{Unschedulable: n.Unschedulable},
			Status: corev1.NodeStatus{Capacity: alloc, Allocatable: alloc}}
		info := nodeinfo.NewNodeInfo()
		info.SetNode(node)
		// pod_count resident pods; the first carries the node's requested totals (AddPod sums container Requests)
		for k := 0; k < n.PodCount; k++ {
			rl := corev1.ResourceList{}
			if k == 0 {
				rl = gList(in.Lanes, n.Requested, n.ReqKeys, true)
			}
			info.AddPod(&corev1.Pod{ObjectMeta: metav1.ObjectMeta{Name: fmt.Sprintf("%s-res-%d", n.Name, k), UID: types.UID(fmt.Sprintf("%s-res-%d", n.Name, k))},
				Spec: corev1.PodSpec{Containers: []corev1.Container{{Resources: corev1.ResourceRequirements{Requests: rl}}}}})
		}
		lister.infos = append(lister.infos, info)
		lister.byName[n.Name] = info
	}
	// ---- PodGroup cache
	pgCache := cache.NewPGStatusCache()
	reps := make([]*corev1.Pod, in.Classes)
	for c := range reps {
		reps[c] = gRepPod(c, fmt.Sprintf("rep-%d", c), "ns", "", fmt.Sprintf("rep-%d", c), corev1.ResourceList{})
	}
	for _, g := range in.Groups {
		pg := &pgv1.PodGroup{ObjectMeta: metav1.ObjectMeta{Name: g.Name[len("ns/"):], Namespace: "ns"},
			Spec: pgv1.PodGroupSpec{MinMember: g.MinMember}, Status: pgv1.PodGroupStatus{Scheduled: g.Scheduled}}
		if g.MinResources != nil {
			rl := gList(in.Lanes, g.MinResources, g.MinResKeys, false)
			pg.Spec.MinResources = &rl
		}
		pgs := &cache.PodGroupMatchStatus{PodGroup: pg, MatchedPodNodes: gochache.New(time.Hour, time.Hour), PodNameUIDs: gochache.New(time.Hour, time.Hour),
			Scheduled: g.Latch}
		for k := 0; k < g.Matched; k++ {
			pgs.MatchedPodNodes.Set(fmt.Sprintf("%s-m%d", g.Name, k), "node", time.Hour)
		}
		if g.HasPod {
			pgs.Pod = reps[g.Cls]
		}
		pgCache.PGStatusMap[g.Name] = pgs
	}
	maxSche := time.Minute
	sop := &ScheduleOperation{frameworkHandler: gHandle{l: lister}, podGroupStatusCache: pgCache, maxScheTime: &maxSche,
		lastDeniedPG: gochache.New(30*time.Second, 3*time.Second), lastPermittedPod: gochache.New(3*time.Second, 3*time.Second)}
	out := map[string]interface{}{}

	// ---- findMaxPG: the set of answers over many map iterations
	seen := map[string]bool{}
	for k := 0; k < 200; k++ {
		name, _, _ := findMaxPG(pgCache)
		seen[name] = true
	}
	leaders := []string{}
	for n := range seen {
		leaders = append(leaders, n)
	}
	sort.Strings(leaders)
	out["find_max_pg"] = map[string]interface{}{"leaders_seen": leaders}

	// ---- getPreAllocatedResource
	pre := []interface{}{}
	for gi, g := range in.Groups {
		for _, m := range []int{g.Matched, 0} {
			r := getPreAllocatedResource(pgCache.PGStatusMap[g.Name], m)
			lanes, keys := gLanes(&in, &r)
			pre = append(pre, map[string]interface{}{"group": gi, "matched": m, "lanes": lanes, "keys": keys})
		}
	}
	out["pre_allocated"] = pre

	// ---- singleNodeResource
	single := []interface{}{}
	for c := 0; c < in.Classes; c++ {
		for k, info := range lister.infos {
			for _, pct := range []float32{1, 0.7} {
				lanes, keys := gLanes(&in, singleNodeResource(info, reps[c], pct))
				single = append(single, map[string]interface{}{"cls": c, "node": k, "percent": pct, "lanes": lanes, "keys": keys})
			}
		}
	}
	out["single_node"] = single

	// ---- compareClusterResourceAndRequire + the exit index of its loop
	fits := []interface{}{}
	for qi, q := range in.Queries {
		req := gResource(&in, q.Req, q.ReqKeys)
		ok := sop.compareClusterResourceAndRequire(reps[q.Cls], req, q.Percent)
		firstK := -1
		var left nodeinfo.Resource
		for k, info := range lister.infos { // the loop of core.go:602-631 with the reference's own helpers
			if info == nil || info.Node() == nil || info.Node().Spec.Unschedulable {
				continue
			}
			left.Add(singleNodeResource(info, reps[q.Cls], q.Percent).ResourceList())
			if compareResourceAndRequire(&left, req) {
				firstK = k
				break
			}
		}
		if ok != (firstK >= 0) {
			t.Fatalf("query %d: compareClusterResourceAndRequire says %v, the replayed loop exits at %d", qi, ok, firstK)
		}
		fits = append(fits, map[string]interface{}{"query": qi, "fits": ok, "first_k": firstK})
	}
	out["cluster_fits"] = fits

	// ---- getLeftResource
	left := []interface{}{}
	for k, n := range in.Nodes {
		r := sop.getLeftResource(n.Name)
		if r == nil {
			left = append(left, map[string]interface{}{"node": k, "lanes": nil})
			continue
		}
		lanes, keys := gLanes(&in, r)
		left = append(left, map[string]interface{}{"node": k, "lanes": lanes, "keys": keys})
	}
	out["left_resource"] = left

	// ---- computeResourceSatisfied under the (first) leader seen
	mkPod := func(i int) *corev1.Pod {
		p := in.Pods[i]
		return gRepPod(p.Cls, "pod-"+p.UID, "ns", p.Group, p.UID, gList(in.Lanes, p.Req, p.ReqKeys, true))
	}
	filt := []interface{}{}
	if len(leaders) == 1 && leaders[0] != "" {
		sop.maxFinishedPG = leaders[0]
		sop.maxPGStatus = pgCache.PGStatusMap[leaders[0]]
		for _, pr := range in.FilterPairs {
			p := in.Pods[pr[0]]
			if p.Group == "" {
				continue
			}
			pgs, ok := pgCache.PGStatusMap["ns/"+p.Group]
			if !ok {
				continue
			}
			msg := ""
			if err := sop.computeResourceSatisfied(pgs, mkPod(pr[0]), in.Nodes[pr[1]].Name); err != nil {
				msg = err.Error()
			}
			filt = append(filt, map[string]interface{}{"pod": pr[0], "node": pr[1], "leader": leaders[0], "err": msg})
		}
	}
	out["filter"] = filt

	// ---- PreFilter in queue order (mutates the cache: first-pod capture, MinResources default, deny entries)
	seq := []interface{}{}
	for i := range in.Pods {
		msg := ""
		if err := sop.PreFilter(mkPod(i)); err != nil {
			msg = err.Error()
		}
		seq = append(seq, map[string]interface{}{"pod": i, "err": msg, "leader_after": sop.maxFinishedPG})
	}
	out["prefilter_sequence"] = seq

	enc, err := json.MarshalIndent(out, "", " ")
	if err != nil {
		t.Fatal(err)
	}
	if err := os.WriteFile(outPath, enc, 0o644); err != nil {
		t.Fatal(err)
	}
}
