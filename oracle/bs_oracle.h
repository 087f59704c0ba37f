/*
 * bs_oracle.h — CPU oracle: a plain-C restatement of the reference's gang-feasibility path
 * (/root/reference/pkg/scheduler/core/core.go).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it, and only as the
 * checker / the timed CPU baseline.  The product (libbsched.so) never links or calls it.
 *
 * Parity pin status: the reference cannot be built here (no Go toolchain, k8s.io/kubernetes
 * v1.17.5 not vendored), so this is a restatement.  It is pinned against the reference's only
 * result-fixing test (core_test.go:27-115, three vectors) and the float32 known-answer table
 * derived on x86-64 SSE (tests/golden/).  Everything else — prefix early exit, findMaxPG,
 * getPreAllocatedResource, computeResourceSatisfied, Permit quorum — is PARITY UNPINNED by the
 * reference itself; it is cross-checked against an independent dict-based Python restatement
 * (oracle/naive_ref.py) and by line-by-line review against core.go.
 *
 * Types shared with the product are the POD structs of include/bsched.h (types only).
 */
#ifndef BS_ORACLE_H
#define BS_ORACLE_H

#include "../include/bsched.h"

#ifdef __cplusplus
extern "C" {
#endif

/* upstream nodeinfo.Resource flattened (k8s.io/kubernetes v1.17.5 pkg/scheduler/nodeinfo,
 * not vendored): v[0..3] fixed fields, v[4+s] = ScalarResources[s], present bit s = key exists */
typedef struct orc_resource {
  int64_t v[BS_MAX_LANES];
  uint32_t present;
} orc_resource;

typedef struct orc_snapshot {
  bs_nodes_soa nodes;
  const uint32_t* fit;   /* [n_classes][ceil(n/32)] */
  uint32_t n_classes;
  uint32_t S;            /* scalar lanes */
  uint32_t eph_gate;     /* LocalStorageCapacityIsolation */
} orc_snapshot;

/* int64(float32(a) * pct), core.go:656-659,667 (amd64: CVTSQ2SS, MULSS, CVTTSS2SQ) */
int64_t orc_scale(int64_t a, float pct);

/* Resource.Add(ResourceList) of upstream nodeinfo (call sites core.go:159,527,552,589,621,786) */
void orc_resource_add(orc_resource* r, const orc_resource* rl, uint32_t S, uint32_t eph_gate);

/* singleNodeResource, core.go:634-670 (checkFit core.go:741-759 arrives as the fit bit) */
void orc_single_node_resource(const orc_snapshot* snap, uint32_t cls, uint32_t node, float pct,
                              orc_resource* left);
/* compareResourceAndRequire, core.go:672-699 */
int orc_compare_resource_and_require(const orc_resource* left, const orc_resource* req, uint32_t S);
/* compareClusterResourceAndRequire, core.go:595-632.  *first_k = count-1 at the early exit or
 * BS_K_NONE; *iters += loop iterations executed (core.go:604). */
int orc_compare_cluster(const orc_snapshot* snap, uint32_t cls, const orc_resource* req, float pct,
                        uint32_t* first_k, uint64_t* iters);
/* computeClusterResource, core.go:566-593 */
void orc_compute_cluster_resource(const orc_snapshot* snap, uint32_t cls, orc_resource* total,
                                  uint64_t* iters);
/* getLeftResource, core.go:436-475; returns 0 when the reference returns nil */
int orc_get_left_resource(const orc_snapshot* snap, uint32_t node, orc_resource* left);
/* running sums of compareClusterResourceAndRequire for every non-skipped node (no early exit) */
uint32_t orc_scan_prefix(const orc_snapshot* snap, uint32_t cls, float pct, int64_t* prefix,
                         uint32_t* present, uint32_t* node_index);
/* singleNodeResource for all nodes: left[L][n], present[n] */
void orc_node_left(const orc_snapshot* snap, uint32_t cls, float pct, int64_t* left,
                   uint32_t* present);

/* findMaxPG, core.go:701-739, over groups in array order.  Returns leader index or -1;
 * *panic = 1 when the uint32 division by zero at :716-717 would fire. */
int32_t orc_find_max_pg(const bs_groups_soa* groups, uint32_t* max_finished, uint8_t* panic);
/* getPreAllocatedResource, core.go:774-793 */
void orc_get_pre_allocated(const bs_groups_soa* groups, uint32_t g, int64_t matched, uint32_t S,
                           uint32_t eph_gate, orc_resource* out);
/* Permit's quorum predicate, core.go:303 */
int orc_permit_ready(uint32_t matched, uint32_t min_member, uint32_t status_scheduled);

/* ScheduleOperation.PreFilter for pod i (core.go:88-167) on mutable group state.
 * Mutates groups exactly as the reference does (fillOccupiedObj core.go:477-512, deny flag =
 * AddToDenyCache core.go:423-425).  faithful_cost != 0 also executes the decision-irrelevant
 * computeClusterResource scan evaluated as a klog argument at core.go:152. */
typedef struct orc_sop {
  orc_snapshot snap;
  bs_groups_soa groups;     /* mutable */
  int32_t max_finished_pg;  /* sop.maxFinishedPG as an index, -1 == "" (core.go:58,121) */
  int has_max_status;       /* sop.maxPGStatus != nil (core.go:59,122) */
  int faithful_cost;
  uint64_t iters;           /* node-loop iterations executed so far */
} orc_sop;
uint8_t orc_prefilter(orc_sop* sop, const bs_pods_soa* pods, uint32_t i, uint32_t* first_k);
/* ScheduleOperation.Filter's computeResourceSatisfied for (pod i, node), core.go:514-564, with
 * sop.maxFinishedPG / maxPGStatus given by `leader` (-1 = nil). */
uint8_t orc_filter_node(const orc_sop* sop, const bs_pods_soa* pods, uint32_t i, int32_t leader,
                        uint32_t node, uint8_t* fn_code);

/* Sequential replay of one batch: PreFilter for pods 0..p-1 in order, then (stages & FILTER)
 * Filter over all nodes for pods that passed, then tallies.  Mutates sop->groups. */
void orc_batch(orc_sop* sop, const bs_pods_soa* pods, uint32_t stages, const bs_batch_out* out);

/* go-cache v2.1.0 TTL map with a virtual clock (patrickmn/go-cache, not vendored): used by the
 * sequential replay harness. */
typedef struct orc_ttl orc_ttl;
orc_ttl* orc_ttl_new(void);
void orc_ttl_free(orc_ttl* t);
void orc_ttl_set(orc_ttl* t, uint64_t key, uint64_t val, int64_t now_ns, int64_t ttl_ns);
int  orc_ttl_add(orc_ttl* t, uint64_t key, uint64_t val, int64_t now_ns, int64_t ttl_ns); /* 0 ok, -1 exists */
int  orc_ttl_get(const orc_ttl* t, uint64_t key, int64_t now_ns, uint64_t* val);
void orc_ttl_delete(orc_ttl* t, uint64_t key);
uint32_t orc_ttl_count(const orc_ttl* t, int64_t now_ns); /* len(Items()) */

/* checkFit (core.go:741-759) for one (class, node) pair and for all pairs; bs_oracle_fit.c lists the
 * k8s.io/kubernetes v1.17.5 rules it follows (U6).  fit_bits: [c][ceil(n/32)]. */
int  orc_check_fit(const bs_node_labels* nodes, const uint8_t* node_flags, const bs_fit_templates* tpl,
                   uint32_t cls, uint32_t node);
void orc_fit_build(const bs_node_labels* nodes, const uint8_t* node_flags, const bs_fit_templates* tpl,
                   uint32_t* fit_bits);

#ifdef __cplusplus
}
#endif
#endif
