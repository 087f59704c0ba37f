// Package core — benchmark of the REFERENCE's own Go path on the same seeded snapshot the GPU bench uses.
//
// SOURCE ONLY: this image has no Go toolchain and k8s.io/kubernetes v1.17.5 is not vendored, so this file
// has never been compiled or run here.  To use it: copy it next to pkg/scheduler/core/core.go of
// tenstack/batch-scheduler (it needs the unexported compareClusterResourceAndRequire), dump a snapshot with
//     python tools/dump_snapshot.py cfg3 tail > /tmp/snapshot.json
// and run   BS_SNAPSHOT=/tmp/snapshot.json go test -run xxx -bench PreFilterScan ./pkg/scheduler/core/
// It reports ns per compareClusterResourceAndRequire call; pod x node fit evals/s = nodes / (ns * 1e-9).
package core

import (
	"encoding/json"
	"fmt"
	"os"
	"testing"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
	framework "k8s.io/kubernetes/pkg/scheduler/framework/v1alpha1"
	"k8s.io/kubernetes/pkg/scheduler/listers"
	"k8s.io/kubernetes/pkg/scheduler/nodeinfo"
)

type snapshot struct {
	Lanes     []string  `json:"lanes"`     // resource name per lane: cpu(milli), memory, ephemeral-storage, pods, extended...
	Alloc     [][]int64 `json:"alloc"`     // [lane][node]
	Requested [][]int64 `json:"requested"` // [lane][node]; pods lane = pod count
	ReqKey    [][]bool  `json:"req_key"`   // [scalar][node]: requested map carries the key
	Unsched   []bool    `json:"unschedulable"`
	Requests  [][]int64 `json:"requests"`  // [query][lane]: the request vectors the GPU scan evaluated
	Percent   float32   `json:"percent"`
}

type fakeLister struct{ infos []*nodeinfo.NodeInfo }

func (f fakeLister) List() ([]*nodeinfo.NodeInfo, error)                       { return f.infos, nil }
func (f fakeLister) HavePodsWithAffinityList() ([]*nodeinfo.NodeInfo, error)    { return nil, nil }
func (f fakeLister) Get(name string) (*nodeinfo.NodeInfo, error)               { return nil, fmt.Errorf("unused") }
func (f fakeLister) Pods() listers.PodLister                                   { return nil }
func (f fakeLister) NodeInfos() listers.NodeInfoLister                         { return f }

type fakeHandle struct {
	framework.FrameworkHandle
	l fakeLister
}

func (h fakeHandle) SnapshotSharedLister() listers.SharedLister { return h.l }

func quantity(lane string, v int64) resource.Quantity {
	if lane == "cpu" {
		return *resource.NewMilliQuantity(v, resource.DecimalSI)
	}
	return *resource.NewQuantity(v, resource.BinarySI)
}

func loadSnapshot(tb testing.TB) (*ScheduleOperation, *snapshot) {
	path := os.Getenv("BS_SNAPSHOT")
	if path == "" {
		tb.Skip("BS_SNAPSHOT not set")
	}
	raw, err := os.ReadFile(path)
	if err != nil {
		tb.Fatal(err)
	}
	var s snapshot
	if err := json.Unmarshal(raw, &s); err != nil {
		tb.Fatal(err)
	}
	n := len(s.Alloc[0])
	infos := make([]*nodeinfo.NodeInfo, n)
	for i := 0; i < n; i++ {
		alloc := corev1.ResourceList{}
		for l, name := range s.Lanes {
			if s.Alloc[l][i] != 0 || l < 4 {
				alloc[corev1.ResourceName(name)] = quantity(name, s.Alloc[l][i])
			}
		}
		node := &corev1.Node{ObjectMeta: metav1.ObjectMeta{Name: fmt.Sprintf("n%d", i)},
			Spec:   corev1.NodeSpec{Unschedulable: s.Unsched[i]},
			Status: corev1.NodeStatus{Capacity: alloc, Allocatable: alloc}}
		info := nodeinfo.NewNodeInfo()
		info.SetNode(node)
		// one resident pod carrying the node's requested totals (AddPod sums container Requests)
		reqs := corev1.ResourceList{}
		for l, name := range s.Lanes {
			if l == 3 {
				continue
			}
			if l < 4 || s.ReqKey[l-4][i] {
				reqs[corev1.ResourceName(name)] = quantity(name, s.Requested[l][i])
			}
		}
		info.AddPod(&corev1.Pod{Spec: corev1.PodSpec{Containers: []corev1.Container{{Resources: corev1.ResourceRequirements{Requests: reqs}}}}})
		infos[i] = info
	}
	sop := &ScheduleOperation{frameworkHandler: fakeHandle{l: fakeLister{infos: infos}}}
	return sop, &s
}

// BenchmarkPreFilterScan times the reference's hot loop (core.go:595-632) for every request vector.
func BenchmarkPreFilterScan(b *testing.B) {
	sop, s := loadSnapshot(b)
	rep := &corev1.Pod{}
	reqs := make([]*nodeinfo.Resource, len(s.Requests))
	for q, lanes := range s.Requests {
		r := &nodeinfo.Resource{MilliCPU: lanes[0], Memory: lanes[1], EphemeralStorage: lanes[2], AllowedPodNumber: int(lanes[3])}
		for l := 4; l < len(lanes); l++ {
			r.SetScalar(corev1.ResourceName(s.Lanes[l]), lanes[l])
		}
		reqs[q] = r
	}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		sop.compareClusterResourceAndRequire(rep, reqs[i%len(reqs)], s.Percent)
	}
}
