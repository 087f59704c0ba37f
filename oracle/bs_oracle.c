/*
 * bs_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY, see bs_oracle.h).
 *
 * Plain-C restatement of /root/reference/pkg/scheduler/core/core.go.  Every function cites the
 * reference lines it follows and keeps the reference's sequential loop structure (per pod, per
 * node, early exit) so that it can also be timed as the CPU baseline ("port").
 *
 * Semantics recalled from un-vendored upstream code (k8s.io/kubernetes v1.17.5,
 * pkg/scheduler/nodeinfo; go.mod:102) — isolated here, each a parity risk if mis-recalled:
 *   U1 Resource.Add(rl): cpu += MilliValue, memory += Value, pods += int(Value),
 *      ephemeral-storage += Value only when feature gate LocalStorageCapacityIsolation is on,
 *      any other name is added to ScalarResources (key created even for a zero quantity) iff
 *      IsScalarResourceName, else dropped.  The Go shim maps names to lanes, so "dropped" names
 *      never reach this file.
 *   U2 Resource.ResourceList() always emits cpu, memory, pods, ephemeral-storage plus every
 *      ScalarResources key, and round-trips int64 losslessly through Quantity.
 *   U3 AllocatableResource()/RequestedResource() return copies; requestedResource.
 *      AllowedPodNumber is never incremented by AddPod, so podCount falls back to len(Pods())
 *      (core.go:650-653) — resolved by the shim into the requested pods lane.
 *   U4 Go on amd64: float32(int64) is CVTSQ2SS (round to nearest even), float32*float32 is
 *      MULSS, int64(float32) is CVTTSS2SQ (truncate; out of range -> 0x8000000000000000).
 *   U5 Go signed integer overflow wraps (two's complement).
 */
#include "bs_oracle.h"

#include <stdlib.h>
#include <string.h>

#define LANES(S) (BS_FIXED_LANES + (S))

static inline int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t wrap_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

/* core.go:656-659,667: int64(float32(a) * percent).  U4. */
int64_t orc_scale(int64_t a, float pct) {
  volatile float f = (float)a;       /* CVTSQ2SS; volatile forbids excess precision / folding */
  volatile float m = f * pct;        /* MULSS, single rounding */
  float r = m;
  if (!(r < 9223372036854775808.0f) || r < -9223372036854775808.0f)
    return INT64_MIN;                /* CVTTSS2SQ integer-indefinite (also NaN) */
  return (int64_t)r;                 /* CVTTSS2SQ: truncate toward zero */
}

/* U1: Resource.Add(ResourceList).  `rl` stands for a ResourceList; for lists produced by
 * Resource.ResourceList() all four fixed keys exist (U2), absent keys contribute 0 either way. */
void orc_resource_add(orc_resource* r, const orc_resource* rl, uint32_t S, uint32_t eph_gate) {
  r->v[BS_LANE_CPU] = wrap_add(r->v[BS_LANE_CPU], rl->v[BS_LANE_CPU]);
  r->v[BS_LANE_MEM] = wrap_add(r->v[BS_LANE_MEM], rl->v[BS_LANE_MEM]);
  r->v[BS_LANE_PODS] = wrap_add(r->v[BS_LANE_PODS], rl->v[BS_LANE_PODS]);
  if (eph_gate) r->v[BS_LANE_EPH] = wrap_add(r->v[BS_LANE_EPH], rl->v[BS_LANE_EPH]);
  for (uint32_t s = 0; s < S; ++s) {
    if (rl->present & (1u << s)) {   /* AddScalar creates the key */
      r->v[4 + s] = wrap_add(r->v[4 + s], rl->v[4 + s]);
      r->present |= 1u << s;
    }
  }
}

static inline int fit_bit(const orc_snapshot* snap, uint32_t cls, uint32_t node) {
  uint32_t words = (snap->nodes.n + 31u) / 32u;
  return (snap->fit[(size_t)cls * words + (node >> 5)] >> (node & 31u)) & 1u;
}

/* core.go:634-670 */
void orc_single_node_resource(const orc_snapshot* snap, uint32_t cls, uint32_t node, float pct,
                              orc_resource* left) {
  const bs_nodes_soa* nd = &snap->nodes;
  const size_t n = nd->n;
  memset(left, 0, sizeof(*left));                   /* :635-637 empty map */
  if (nd->flags[node] & BS_NODE_TAINT_ERR) return;  /* :639-641 */
  if (!fit_bit(snap, cls, node)) return;            /* :642-645 checkFit :741-759 */
  /* :647-653 allocatable / requested / podCount (U3: pods lane of requested = podCount) */
  left->v[BS_LANE_PODS] = wrap_sub(orc_scale(nd->allocatable[BS_LANE_PODS * n + node], pct),
                                   nd->requested[BS_LANE_PODS * n + node]);   /* :656 */
  left->v[BS_LANE_CPU] = wrap_sub(orc_scale(nd->allocatable[BS_LANE_CPU * n + node], pct),
                                  nd->requested[BS_LANE_CPU * n + node]);     /* :657 */
  left->v[BS_LANE_MEM] = wrap_sub(orc_scale(nd->allocatable[BS_LANE_MEM * n + node], pct),
                                  nd->requested[BS_LANE_MEM * n + node]);     /* :658 */
  left->v[BS_LANE_EPH] = wrap_sub(orc_scale(nd->allocatable[BS_LANE_EPH * n + node], pct),
                                  nd->requested[BS_LANE_EPH * n + node]);     /* :659 */
  for (uint32_t s = 0; s < snap->S; ++s) {          /* :662-668 */
    if (!(nd->allocatable_present[node] & (1u << s))) continue;  /* range over allocatable */
    if (!(nd->requested_present[node] & (1u << s))) continue;    /* :663-666 */
    left->v[4 + s] = wrap_sub(orc_scale(nd->allocatable[(4 + s) * n + node], pct),
                              nd->requested[(4 + s) * n + node]);             /* :667 */
    left->present |= 1u << s;
  }
}

/* core.go:672-699 */
int orc_compare_resource_and_require(const orc_resource* left, const orc_resource* req, uint32_t S) {
  if (left->v[BS_LANE_MEM] < req->v[BS_LANE_MEM]) return 0;    /* :673 */
  if (left->v[BS_LANE_CPU] < req->v[BS_LANE_CPU]) return 0;    /* :676 */
  if (left->v[BS_LANE_EPH] < req->v[BS_LANE_EPH]) return 0;    /* :679 */
  if (left->v[BS_LANE_PODS] < req->v[BS_LANE_PODS]) return 0;  /* :683 */
  for (uint32_t s = 0; s < S; ++s) {                           /* :686 range req.ScalarResources */
    if (!(req->present & (1u << s))) continue;
    int64_t v1 = req->v[4 + s];
    if (!(left->present & (1u << s))) {                        /* :688-692 */
      if (v1 != 0) return 0;
      continue;
    }
    if (v1 > left->v[4 + s]) return 0;                         /* :694 */
  }
  return 1;
}

/* core.go:595-632 */
int orc_compare_cluster(const orc_snapshot* snap, uint32_t cls, const orc_resource* req, float pct,
                        uint32_t* first_k, uint64_t* iters) {
  orc_resource left_resources;                       /* :602 */
  memset(&left_resources, 0, sizeof(left_resources));
  uint32_t count = 0;                                /* :603 */
  if (first_k) *first_k = BS_K_NONE;
  for (uint32_t k = 0; k < snap->nodes.n; ++k) {     /* :604 */
    count++;                                         /* :605 */
    if (iters) (*iters)++;
    if (snap->nodes.flags[k] & BS_NODE_SKIP_MASK) continue;   /* :606-617 */
    orc_resource left;
    orc_single_node_resource(snap, cls, k, pct, &left);       /* :619 */
    orc_resource_add(&left_resources, &left, snap->S, snap->eph_gate); /* :621 */
    if (orc_compare_resource_and_require(&left_resources, req, snap->S)) {   /* :623 */
      if (first_k) *first_k = count - 1;
      return 1;                                      /* :626 */
    }
  }
  return 0;                                          /* :631 */
}

/* core.go:566-593 */
void orc_compute_cluster_resource(const orc_snapshot* snap, uint32_t cls, orc_resource* total,
                                  uint64_t* iters) {
  memset(total, 0, sizeof(*total));                  /* :572 */
  for (uint32_t k = 0; k < snap->nodes.n; ++k) {     /* :573 */
    if (iters) (*iters)++;
    if (snap->nodes.flags[k] & BS_NODE_SKIP_MASK) continue;   /* :574-585 */
    orc_resource left;
    orc_single_node_resource(snap, cls, k, 1.0f, &left);      /* :587 */
    orc_resource_add(total, &left, snap->S, snap->eph_gate);  /* :589 */
  }
}

/* core.go:436-475.  Scalars: leftResource.ScalarResources is nil, Clone() of a nil map is nil,
 * so the loop at :466-472 never runs — left has no scalar keys. */
int orc_get_left_resource(const orc_snapshot* snap, uint32_t node, orc_resource* left) {
  const bs_nodes_soa* nd = &snap->nodes;
  const size_t n = nd->n;
  if (node >= nd->n) return 0;                                       /* :442-445 Get() error */
  if (nd->flags[node] & (BS_NODE_NIL | BS_NODE_NO_NODE)) return 0;   /* :443-449 */
  memset(left, 0, sizeof(*left));
  left->v[BS_LANE_CPU] = wrap_sub(nd->allocatable[BS_LANE_CPU * n + node], nd->requested[BS_LANE_CPU * n + node]);    /* :460 */
  left->v[BS_LANE_PODS] = wrap_sub(nd->allocatable[BS_LANE_PODS * n + node], nd->requested[BS_LANE_PODS * n + node]); /* :461 */
  left->v[BS_LANE_MEM] = wrap_sub(nd->allocatable[BS_LANE_MEM * n + node], nd->requested[BS_LANE_MEM * n + node]);    /* :462 */
  left->v[BS_LANE_EPH] = wrap_sub(nd->allocatable[BS_LANE_EPH * n + node], nd->requested[BS_LANE_EPH * n + node]);    /* :463 */
  return 1;
}

uint32_t orc_scan_prefix(const orc_snapshot* snap, uint32_t cls, float pct, int64_t* prefix,
                         uint32_t* present, uint32_t* node_index) {
  const uint32_t L = LANES(snap->S);
  const size_t n = snap->nodes.n;
  orc_resource sum;
  memset(&sum, 0, sizeof(sum));
  uint32_t rows = 0;
  for (uint32_t k = 0; k < snap->nodes.n; ++k) {
    if (snap->nodes.flags[k] & BS_NODE_SKIP_MASK) continue;
    orc_resource left;
    orc_single_node_resource(snap, cls, k, pct, &left);
    orc_resource_add(&sum, &left, snap->S, snap->eph_gate);
    for (uint32_t j = 0; j < L; ++j) prefix[j * n + rows] = sum.v[j];
    present[rows] = sum.present;
    node_index[rows] = k;
    rows++;
  }
  return rows;
}

void orc_node_left(const orc_snapshot* snap, uint32_t cls, float pct, int64_t* left_out,
                   uint32_t* present) {
  const uint32_t L = LANES(snap->S);
  const size_t n = snap->nodes.n;
  for (uint32_t k = 0; k < snap->nodes.n; ++k) {
    orc_resource left;
    orc_single_node_resource(snap, cls, k, pct, &left);
    for (uint32_t j = 0; j < L; ++j) left_out[j * n + k] = left.v[j];
    present[k] = left.present;
  }
}

/* core.go:701-739.  Go ranges over a map (random order); here: array order. */
int32_t orc_find_max_pg(const bs_groups_soa* gr, uint32_t* max_finished_out, uint8_t* panic) {
  int32_t max_pg = -1;           /* maxFinishedPG "" / maxPGStatus nil */
  uint32_t max_finished = 0;
  if (panic) *panic = 0;
  for (uint32_t g = 0; g < gr->g; ++g) {                       /* :703 */
    uint32_t finished = 0;                                      /* :705 */
    if (gr->flags[g] & BS_GROUP_SCHEDULED_LATCH) continue;      /* :706-708 */
    if (!(gr->flags[g] & BS_GROUP_HAS_POD)) continue;           /* :709-711 */
    uint32_t mm = gr->min_member[g], sc = gr->status_scheduled[g];
    if ((uint32_t)(mm - sc) == 0u) {                            /* :712-714 uint32: "<= 0" is "== 0" */
      finished = 0;
    } else {
      if (mm == 0u) {                                           /* :716-717 integer divide by zero */
        if (panic) *panic = 1;
        if (max_finished_out) *max_finished_out = max_finished;
        return -1;
      }
      finished = (uint32_t)((uint32_t)(gr->matched[g] + sc) * 1000u) / mm;   /* :716-717 uint32 wrap */
    }
    if (finished > max_finished) {                              /* :721-724 */
      max_finished = finished;
      max_pg = (int32_t)g;
    } else if (finished == max_finished) {                      /* :725 */
      if (max_pg < 0 ||
          (gr->status_scheduled[max_pg] >= gr->min_member[max_pg] &&
           gr->status_scheduled[g] == 0u)) {                    /* :729-731 (|| binds looser than &&) */
        max_finished = finished;
        max_pg = (int32_t)g;
      }
    }
  }
  if (max_finished_out) *max_finished_out = max_finished;
  return max_pg;
}

static void group_min_resources(const bs_groups_soa* gr, uint32_t g, uint32_t S, orc_resource* out) {
  memset(out, 0, sizeof(*out));
  for (uint32_t j = 0; j < LANES(S); ++j) out->v[j] = gr->min_resources[(size_t)j * gr->g + g];
  out->present = gr->min_resources_present[g];
}

/* core.go:774-793.  The reference adds MinResources `notFinished` times in a loop (:784-788);
 * repeated wrapping addition == wrapping multiplication (U5), used here so that a huge
 * MinMember cannot stall the oracle. */
void orc_get_pre_allocated(const bs_groups_soa* gr, uint32_t g, int64_t matched, uint32_t S,
                           uint32_t eph_gate, orc_resource* out) {
  memset(out, 0, sizeof(*out));
  int64_t not_finished = 0;                                         /* :776 */
  int64_t scheduled = (int64_t)gr->status_scheduled[g];             /* :777 */
  if (matched != 0) not_finished = (int64_t)gr->min_member[g] - matched;   /* :778-779 */
  else not_finished = (int64_t)gr->min_member[g] - scheduled;              /* :780-783 */
  if (not_finished > 0 && (gr->flags[g] & BS_GROUP_HAS_MINRES)) {   /* :784-788 */
    orc_resource mr, times;
    group_min_resources(gr, g, S, &mr);
    memset(&times, 0, sizeof(times));
    for (uint32_t j = 0; j < LANES(S); ++j) times.v[j] = wrap_mul(mr.v[j], not_finished);
    times.present = mr.present;
    orc_resource_add(out, &times, S, eph_gate);
  }
  if (out->v[BS_LANE_PODS] == 0)                                    /* :789-791 */
    out->v[BS_LANE_PODS] = (int64_t)gr->min_member[g] + 1;
}

/* core.go:303: uint32(len(Items())) >= MinMember - Status.Scheduled, all uint32 (wraps) */
int orc_permit_ready(uint32_t matched, uint32_t min_member, uint32_t status_scheduled) {
  return matched >= (uint32_t)(min_member - status_scheduled);
}

static void pod_require(const bs_pods_soa* pods, uint32_t i, uint32_t S, uint32_t eph_gate,
                        orc_resource* out) {
  /* getPodResourceRequire core.go:761-772: a fresh Resource built by Add — the shim hands over
   * its lanes; re-applying Add to a zero Resource reproduces the feature-gate rule (U1). */
  orc_resource raw;
  memset(&raw, 0, sizeof(raw));
  for (uint32_t j = 0; j < LANES(S); ++j) raw.v[j] = pods->req[(size_t)j * pods->p + i];
  raw.present = pods->req_present[i];
  memset(out, 0, sizeof(*out));
  orc_resource_add(out, &raw, S, eph_gate);
}

/* core.go:477-512 on flattened state; returns 0 ok, 1 error */
static int fill_occupied_obj(orc_sop* sop, const bs_pods_soa* pods, uint32_t i, uint32_t g) {
  bs_groups_soa* gr = &sop->groups;
  const uint32_t S = sop->snap.S;
  if (!(gr->flags[g] & BS_GROUP_HAS_POD)) {                 /* :486-488 */
    gr->flags[g] |= BS_GROUP_HAS_POD;
    gr->cls[g] = pods->cls[i];
  }
  if (!(gr->flags[g] & BS_GROUP_HAS_MINRES)) {              /* :489-493 */
    orc_resource r;
    pod_require(pods, i, S, sop->snap.eph_gate, &r);
    for (uint32_t j = 0; j < LANES(S); ++j) gr->min_resources[(size_t)j * gr->g + g] = r.v[j];
    gr->min_resources_present[g] = r.present;
    gr->flags[g] |= BS_GROUP_HAS_MINRES;
  }
  uint64_t refs = pods->owner[i];                           /* :482-485 sorted+joined, interned */
  if (gr->occupied_by[g] == 0) {                            /* :494 */
    if (refs != 0) gr->occupied_by[g] = refs;               /* :496-500 */
    return 0;                                               /* :501 */
  }
  if (refs == 0) return 1;                                  /* :504-506 */
  if (refs != gr->occupied_by[g]) return 1;                 /* :507-510 */
  return 0;
}

/* core.go:88-167 */
uint8_t orc_prefilter(orc_sop* sop, const bs_pods_soa* pods, uint32_t i, uint32_t* first_k) {
  bs_groups_soa* gr = &sop->groups;
  const uint32_t S = sop->snap.S, gate = sop->snap.eph_gate;
  uint32_t fk = BS_K_NOT_SCANNED;
  if (first_k) *first_k = fk;
  int32_t gi = pods->group[i];
  if (gi == BS_POD_NOT_GROUPED) return BS_PF_PASS_NOT_GROUPED;                 /* :89-92 */
  if (pods->flags[i] & BS_POD_LAST_PERMITTED) return BS_PF_PASS_LAST_PERMITTED; /* :95-98 */
  if (gi < 0 || (uint32_t)gi >= gr->g) return BS_PF_ERR_PG_NOT_FOUND;          /* :100-103 */
  const uint32_t g = (uint32_t)gi;
  if (gr->flags[g] & BS_GROUP_DENIED) return BS_PF_ERR_DENIED;                 /* :105-110 */
  if (fill_occupied_obj(sop, pods, i, g)) return BS_PF_ERR_OCCUPIED;           /* :113-115 */

  uint8_t panic = 0;
  int32_t leader = orc_find_max_pg(gr, NULL, &panic);                          /* :118-123 */
  if (panic) return BS_PF_PANIC_DIV0;
  sop->max_finished_pg = leader;                                               /* :121 */
  sop->has_max_status = leader >= 0;                                           /* :122 */
  if (leader < 0) return BS_PF_PASS_NO_MAX;                                    /* :127-130 */

  int64_t matched = (int64_t)gr->matched[leader];                              /* :132-135 */
  if (matched == 0) {                                                          /* :136 */
    /* maxPGStatus = pgs; maxFinishedPG = fullName (locals only, :137-138) */
    orc_resource pre;
    orc_get_pre_allocated(gr, g, matched, S, gate, &pre);                      /* :139 */
    int ok = orc_compare_cluster(&sop->snap, gr->cls[g], &pre, 1.0f, &fk, &sop->iters); /* :140 */
    if (first_k) *first_k = fk;
    if (!ok) {
      gr->flags[g] |= BS_GROUP_DENIED;                                         /* :142 */
      return BS_PF_REJECT_FIRST;                                               /* :143 */
    }
    return BS_PF_PASS_FIRST_FITS;                                              /* :146 */
  }
  if (sop->max_finished_pg == gi) {                                            /* :150 */
    if (sop->faithful_cost) {                                                  /* :152: klog argument is evaluated */
      orc_resource tot;
      orc_compute_cluster_resource(&sop->snap, gr->cls[leader], &tot, &sop->iters);
    }
    return BS_PF_PASS_IS_MAX;                                                  /* :154 */
  }
  orc_resource pre, cur;
  orc_get_pre_allocated(gr, (uint32_t)leader, matched, S, gate, &pre);         /* :157 */
  pod_require(pods, i, S, gate, &cur);                                         /* :158 */
  orc_resource_add(&pre, &cur, S, gate);                                       /* :159 */
  int ok = orc_compare_cluster(&sop->snap, gr->cls[leader], &pre, 0.7f, &fk, &sop->iters); /* :161 */
  if (first_k) *first_k = fk;
  if (!ok) {
    gr->flags[g] |= BS_GROUP_DENIED;                                           /* :163 */
    return BS_PF_REJECT_RESERVE;                                               /* :164 */
  }
  return BS_PF_PASS_RESERVE_FITS;                                              /* :166 */
}

/* core.go:170-191 + :514-564 for one node */
uint8_t orc_filter_node(const orc_sop* sop, const bs_pods_soa* pods, uint32_t i, int32_t leader,
                        uint32_t node, uint8_t* fn_code) {
  const bs_groups_soa* gr = &sop->groups;
  const uint32_t S = sop->snap.S, gate = sop->snap.eph_gate;
  if (fn_code) *fn_code = BS_FN_PASS_CASE2;
  int32_t gi = pods->group[i];
  if (gi == BS_POD_NOT_GROUPED) return BS_FL_PASS_NOT_GROUPED;          /* :171-174 */
  if (gi < 0 || (uint32_t)gi >= gr->g) return BS_FL_ERR_PG_NOT_FOUND;   /* :177-180 */
  /* computeResourceSatisfied */
  if (leader < 0) return BS_FL_PANIC_NIL_MAX;                           /* :525 nil deref (Q11) */
  int have_max_single = 0;
  orc_resource max_single;
  memset(&max_single, 0, sizeof(max_single));
  if (gr->flags[leader] & BS_GROUP_HAS_MINRES) {                        /* :525-528 */
    orc_resource mr;
    group_min_resources(gr, (uint32_t)leader, S, &mr);
    orc_resource_add(&max_single, &mr, S, gate);
    have_max_single = 1;
  }
  if (leader == gi) return BS_FL_PASS_IS_MAX;                           /* :531-535 */
  if (!have_max_single) return BS_FL_PASS_NO_MINRES;                    /* :542-544 */
  orc_resource left;
  if (!orc_get_left_resource(&sop->snap, node, &left)) {                /* :545-548 */
    if (fn_code) *fn_code = BS_FN_ERR_SNAPSHOT;
    return BS_FL_EVALUATED;
  }
  orc_resource cur;
  pod_require(pods, i, S, gate, &cur);                                  /* :551 */
  orc_resource_add(&cur, &max_single, S, gate);                         /* :552 */
  if (orc_compare_resource_and_require(&left, &cur, S)) {               /* :553-555 */
    if (fn_code) *fn_code = BS_FN_PASS_CASE2;
    return BS_FL_EVALUATED;
  }
  if (!orc_compare_resource_and_require(&left, &max_single, S)) {       /* :558-561 */
    if (fn_code) *fn_code = BS_FN_PASS_CASE3;
    return BS_FL_EVALUATED;
  }
  if (fn_code) *fn_code = BS_FN_ERR_NOT_ENOUGH;                         /* :562-563 */
  return BS_FL_EVALUATED;
}

void orc_batch(orc_sop* sop, const bs_pods_soa* pods, uint32_t stages, const bs_batch_out* out) {
  const uint32_t P = pods->p, G = sop->groups.g, N = sop->snap.nodes.n;
  const uint32_t W = (N + 63u) / 64u;
  uint32_t* admit = (uint32_t*)calloc(G ? G : 1, sizeof(uint32_t));
  if (out->fl_bitmap && (stages & BS_STAGE_FILTER)) memset(out->fl_bitmap, 0, (size_t)W * P * sizeof(uint64_t));
  for (uint32_t i = 0; i < P; ++i) {
    uint32_t fk = BS_K_NOT_SCANNED;
    uint8_t code = BS_PF_PASS_NOT_GROUPED;
    if (stages & BS_STAGE_PREFILTER) code = orc_prefilter(sop, pods, i, &fk);
    /* sop.maxFinishedPG as this call left it (stale when the call returned before core.go:120) */
    int32_t leader = sop->has_max_status ? sop->max_finished_pg : -1;
    if (out->pf_code) out->pf_code[i] = code;
    if (out->pf_first_k) out->pf_first_k[i] = fk;
    if (out->pf_leader) out->pf_leader[i] = leader;
    uint32_t feasible = 0;
    uint8_t fl = BS_FL_NOT_RUN;
    if ((stages & BS_STAGE_FILTER) && BS_PF_IS_PASS(code)) {
      fl = BS_FL_PASS_NOT_GROUPED;
      for (uint32_t k = 0; k < N; ++k) {
        uint8_t fn = 0;
        fl = orc_filter_node(sop, pods, i, leader, k, &fn);
        int pass = (fl < 16u) && (fl != BS_FL_EVALUATED || fn < 16u);
        if (pass) {
          feasible++;
          if (out->fl_bitmap) out->fl_bitmap[(size_t)(k >> 6) * P + i] |= 1ull << (k & 63u);
        }
      }
      if (N == 0) fl = orc_filter_node(sop, pods, i, leader, 0, NULL);
    }
    if (out->fl_code) out->fl_code[i] = fl;
    if (out->fl_feasible) out->fl_feasible[i] = feasible;
    int32_t gi = pods->group[i];
    /* core.go:183-185: Filter returned an error on some node (computeResourceSatisfied: neither case 2 nor case 3) ->
     * AddToDenyCache(fullName).  Every later pod of the group that gets to :105 is turned away there. */
    if ((stages & BS_BATCH_FILTER_DENY) && fl == BS_FL_EVALUATED && feasible < N && gi >= 0 && (uint32_t)gi < G)
      sop->groups.flags[gi] |= BS_GROUP_DENIED;
    if (gi >= 0 && (uint32_t)gi < G && BS_PF_IS_PASS(code) &&
        (!(stages & BS_STAGE_FILTER) || feasible > 0))
      admit[gi]++;
  }
  for (uint32_t g = 0; g < G; ++g) {
    if (out->group_admit) out->group_admit[g] = admit[g];
    if (out->group_ready)
      out->group_ready[g] = (uint8_t)orc_permit_ready(sop->groups.matched[g] + admit[g],
                                                      sop->groups.min_member[g],
                                                      sop->groups.status_scheduled[g]);
  }
  free(admit);
}

/* ---- go-cache v2.1.0 semantics (patrickmn/go-cache cache.go, not vendored):
 *   Set: overwrite, expiration = now + ttl (ttl > 0).   Add: error if a non-expired item exists.
 *   Get: found iff item exists and !(expiration > 0 && now > expiration).   Items(): unexpired. */
typedef struct ttl_item { uint64_t key, val; int64_t exp; int used; } ttl_item;
struct orc_ttl { ttl_item* items; uint32_t cap, count; };

orc_ttl* orc_ttl_new(void) {
  orc_ttl* t = (orc_ttl*)calloc(1, sizeof(orc_ttl));
  t->cap = 64;
  t->items = (ttl_item*)calloc(t->cap, sizeof(ttl_item));
  return t;
}
void orc_ttl_free(orc_ttl* t) { if (t) { free(t->items); free(t); } }
static uint64_t ttl_hash(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
static ttl_item* ttl_find(const orc_ttl* t, uint64_t key) {
  for (uint32_t h = (uint32_t)(ttl_hash(key) & (t->cap - 1)), probes = 0; probes < t->cap; ++probes, h = (h + 1) & (t->cap - 1)) {
    ttl_item* it = &t->items[h];
    if (it->used == 0) return NULL;
    if (it->used == 1 && it->key == key) return it;
  }
  return NULL;
}
static void ttl_insert_raw(orc_ttl* t, uint64_t key, uint64_t val, int64_t exp) {
  for (uint32_t h = (uint32_t)(ttl_hash(key) & (t->cap - 1));; h = (h + 1) & (t->cap - 1)) {
    ttl_item* it = &t->items[h];
    if (it->used != 1) { it->key = key; it->val = val; it->exp = exp; it->used = 1; t->count++; return; }
  }
}
static void ttl_grow(orc_ttl* t) {
  ttl_item* old = t->items; uint32_t oc = t->cap;
  t->cap *= 2; t->count = 0;
  t->items = (ttl_item*)calloc(t->cap, sizeof(ttl_item));
  for (uint32_t i = 0; i < oc; ++i) if (old[i].used == 1) ttl_insert_raw(t, old[i].key, old[i].val, old[i].exp);
  free(old);
}
void orc_ttl_set(orc_ttl* t, uint64_t key, uint64_t val, int64_t now_ns, int64_t ttl_ns) {
  int64_t exp = ttl_ns > 0 ? now_ns + ttl_ns : 0;
  ttl_item* it = ttl_find(t, key);
  if (it) { it->val = val; it->exp = exp; return; }
  if ((t->count + 1) * 2 > t->cap) ttl_grow(t);
  ttl_insert_raw(t, key, val, exp);
}
int orc_ttl_get(const orc_ttl* t, uint64_t key, int64_t now_ns, uint64_t* val) {
  ttl_item* it = ttl_find(t, key);
  if (!it) return 0;
  if (it->exp > 0 && now_ns > it->exp) return 0;
  if (val) *val = it->val;
  return 1;
}
int orc_ttl_add(orc_ttl* t, uint64_t key, uint64_t val, int64_t now_ns, int64_t ttl_ns) {
  if (orc_ttl_get(t, key, now_ns, NULL)) return -1;
  orc_ttl_set(t, key, val, now_ns, ttl_ns);
  return 0;
}
void orc_ttl_delete(orc_ttl* t, uint64_t key) {
  ttl_item* it = ttl_find(t, key);
  if (it) { it->used = 2; t->count--; }   /* tombstone */
}
uint32_t orc_ttl_count(const orc_ttl* t, int64_t now_ns) {
  uint32_t c = 0;
  for (uint32_t i = 0; i < t->cap; ++i)
    if (t->items[i].used == 1 && !(t->items[i].exp > 0 && now_ns > t->items[i].exp)) c++;
  return c;
}
