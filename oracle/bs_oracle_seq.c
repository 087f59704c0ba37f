/*
 * bs_oracle_seq.c — CPU oracle, sequential scheduling pass (TEST INFRASTRUCTURE ONLY, see bs_oracle.h).
 *
 * One pass of the reference over the pending queue, pod by pod, the way upstream's scheduleOne drives the plugin:
 *   PreFilter            core.go:88-167    (orc_prefilter: deny entries, first-pod capture, findMaxPG and the node scan)
 *   [Filter              core.go:170-191,  :514-564 — only when the FILTER stage is on; the shipped config leaves it off.  It gates the
 *                        node choice; with BS_BATCH_FILTER_DENY its TTL writes happen as well (:183-185 deny entry, :188 lastPermittedPod),
 *                        under the offer rule "every node"]
 *   node choice + assume upstream (NodeResourcesFit / priorities / cache.AssumePod -> NodeInfo.AddPod), NOT plugin code, not
 *                        vendored: restated as the rule host/bs_drain.cpp states — FIRST FIT in list order over nodes without
 *                        a BS_NODE_* flag whose checkFit bit is set for the pod's class and that hold the request (lane j in
 *                        {cpu, mem, eph} binds when request > 0; pods lane: requested + 1 <= allocatable; a requested scalar
 *                        needs the allocatable key); assume = requested += request, pods lane + 1.  Stated, unpinned.
 *   Permit               core.go:268-309   matched + 1 (:290), quorum :303, latch :305
 *   release + PostBind   batchscheduler.go:254-344, core.go:312-362: when the quorum turns true StartBatchSchedule allows EVERY entry of
 *                        MatchedPodNodes (:292,:301-333) — the pods this pass placed and the ones that were already waiting when it
 *                        began (groups.matched on entry) —, deletes each entry (:326) and PostBind counts each into Status.Scheduled
 *                        (core.go:327); the phase turns Scheduled when Status.Scheduled >= MinMember (core.go:329-330), after which
 *                        StartBatchSchedule releases nobody any more (batchscheduler.go:258-261: BS_GROUP_PHASE_CLOSED): a late
 *                        member of such a gang is assumed, counted by Permit and waits for its Permit timeout.
 *                        Pinned against the object-level replay oracle/naive_seq.py (real TTL maps, start_batch, postbind):
 *                        tests/test_seq_oracle_pin.py.
 * It records, per released gang, when its FIRST pod entered PreFilter and when the quorum turned true — SURVEY 8(d)(2)'s
 * gang-admit latency of the sequential path — and is the timed CPU baseline beside the batched drain (bench.py).
 * Pods that pass PreFilter but find no node stay pending and keep nothing; pods of a gang that never reaches its quorum keep
 * what they assumed (that is what the reference does until the Permit timeout).
 */
#define _POSIX_C_SOURCE 199309L
#include "bs_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct orc_seq_io {
  orc_sop* sop;              /* snapshot + group state; snap.nodes.requested / requested_present and the group counters are UPDATED */
  const bs_pods_soa* pods;   /* the queue */
  uint32_t stages;           /* BS_STAGE_FILTER: the plugin's Filter gates the node choice */
  uint8_t* pf_code;          /* [p] */
  int32_t* pod_node;         /* [p] node of every RELEASED pod, -1 otherwise */
  uint32_t cap;
  uint32_t* released_group;  /* [cap] in release order */
  uint32_t* released_pods;   /* [cap] */
  int64_t* first_ns;         /* [cap] first pod of the gang entered PreFilter (since the pass began) */
  int64_t* ready_ns;         /* [cap] quorum turned true */
  uint32_t n_released;
  int64_t total_ns;
  uint32_t* pf_first_k;      /* [p] optional: first_k of the pod's node scan (as bs_batch_out.pf_first_k)                 */
  int32_t* pf_leader;        /* [p] optional: sop.maxFinishedPG as the pod's PreFilter left it (-1 none)                  */
  uint8_t* last_permitted;   /* [p] optional, with BS_STAGE_FILTER | BS_BATCH_FILTER_DENY: 1 = a Filter call of the pod passed, i.e. the pass
                              * left a lastPermittedPod entry for it (core.go:188) — its next PreFilter within 2 s returns at :95-98 */
  int64_t pick_ns;           /* out: of total_ns, the time spent in the node-choice loop (UPSTREAM's work, not the plugin's; with the FILTER stage it
                              * contains the plugin's Filter calls on the nodes tried) */
} orc_seq_io;

static int64_t mono_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

static int holds(const orc_snapshot* s, uint32_t k, const int64_t* req, uint32_t pres) {
  const bs_nodes_soa* nd = &s->nodes;
  const uint32_t N = nd->n;
  for (uint32_t j = 0; j < 3; ++j) {
    const int64_t left = (int64_t)((uint64_t)nd->allocatable[(size_t)j * N + k] - (uint64_t)nd->requested[(size_t)j * N + k]);
    if (req[j] > 0 && req[j] > left) return 0;
  }
  if (nd->requested[(size_t)3 * N + k] + 1 > nd->allocatable[(size_t)3 * N + k]) return 0;
  for (uint32_t sc = 0; sc < s->S; ++sc) {
    if (!((pres >> sc) & 1u) || req[4 + sc] <= 0) continue;
    if (!((nd->allocatable_present[k] >> sc) & 1u)) return 0;
    const int64_t rq = ((nd->requested_present[k] >> sc) & 1u) ? nd->requested[(size_t)(4 + sc) * N + k] : 0;
    if (req[4 + sc] > nd->allocatable[(size_t)(4 + sc) * N + k] - rq) return 0;
  }
  return 1;
}

void orc_seq_replay(orc_seq_io* io) {
  orc_sop* sop = io->sop;
  const orc_snapshot* s = &sop->snap;
  const bs_pods_soa* pods = io->pods;
  const uint32_t P = pods->p, N = s->nodes.n, G = sop->groups.g, L = 4 + s->S, fw = (N + 31) / 32;
  int64_t* requested = (int64_t*)s->nodes.requested;            /* mutable by contract */
  uint32_t* rpres = (uint32_t*)s->nodes.requested_present;
  int64_t* t_first = (int64_t*)malloc(sizeof(int64_t) * (G ? G : 1));
  uint32_t* slot_of = (uint32_t*)malloc(sizeof(uint32_t) * (G ? G : 1));   /* release record of a latched group */
  /* waiting pods per group as linked lists through next_wait */
  int32_t* head = (int32_t*)malloc(sizeof(int32_t) * (G ? G : 1));
  int32_t* next_wait = (int32_t*)malloc(sizeof(int32_t) * (P ? P : 1));
  int32_t* assumed_on = (int32_t*)malloc(sizeof(int32_t) * (P ? P : 1));
  for (uint32_t g = 0; g < G; ++g) { t_first[g] = -1; head[g] = -1; slot_of[g] = 0xFFFFFFFFu; }
  for (uint32_t i = 0; i < P; ++i) io->pod_node[i] = -1;
  io->n_released = 0;
  io->pick_ns = 0;
  const int64_t t0 = mono_ns();
  for (uint32_t i = 0; i < P; ++i) {
    const int32_t gi = pods->group[i];
    const int grouped = gi >= 0 && (uint32_t)gi < G;
    if (grouped && t_first[gi] < 0) t_first[gi] = mono_ns() - t0;
    uint32_t fk;
    const uint8_t code = orc_prefilter(sop, pods, i, &fk);
    io->pf_code[i] = code;
    const int32_t leader = sop->has_max_status ? sop->max_finished_pg : -1;
    if (io->pf_first_k) io->pf_first_k[i] = fk;
    if (io->pf_leader) io->pf_leader[i] = leader;
    if (!BS_PF_IS_PASS(code)) continue;
    if (gi != BS_POD_NOT_GROUPED && !grouped) continue;         /* a labelled pod whose group is unknown only gets here on its lastPermittedPod
                                                                 * entry (core.go:95-98); Permit answers "can not found pod group" (core.go:275-278)
                                                                 * -> Unschedulable (batchscheduler.go:192-193) and the framework forgets the
                                                                 * assumed pod: it holds nothing and is not released */
    if (io->last_permitted) io->last_permitted[i] = 0;
    if ((io->stages & BS_STAGE_FILTER) && (io->stages & BS_BATCH_FILTER_DENY) && grouped) {
      /* Filter (core.go:170-191) with its TTL writes.  Which nodes the framework offers to Filter is upstream's business
       * (percentageOfNodesToScore, parallel fan-out); the rule is the batch form's (bs_batch_run, BS_BATCH_FILTER_DENY): EVERY node of
       * the list, with the node requests as the pods before this one left them.  A failing call deny-lists the group (:183-185 ->
       * :105-110 for the gang's later pods), a passing one enters lastPermittedPod (:188).  The pod itself goes on to the node choice. */
      int failed = 0, passed = 0;
      uint8_t fl = 0;
      for (uint32_t k = 0; k < N; ++k) {
        uint8_t fn = 0;
        fl = orc_filter_node(sop, pods, i, leader, k, &fn);
        if ((fl < 16u) && (fl != BS_FL_EVALUATED || fn < 16u)) passed = 1; else failed = 1;
      }
      if (N && fl == BS_FL_EVALUATED && failed) sop->groups.flags[gi] |= BS_GROUP_DENIED;
      if (passed && io->last_permitted) io->last_permitted[i] = 1;
    }
    int64_t req[BS_MAX_LANES];
    for (uint32_t j = 0; j < L; ++j) req[j] = pods->req[(size_t)j * P + i];
    const uint32_t pres = pods->req_present[i], cls = pods->cls[i];
    int32_t at = -1;
    const int64_t t_pick = mono_ns();
    for (uint32_t k = 0; k < N && at < 0; ++k) {
      if (s->nodes.flags[k]) continue;
      if (cls >= s->n_classes || !((s->fit[(size_t)cls * fw + (k >> 5)] >> (k & 31u)) & 1u)) continue;
      if (io->stages & BS_STAGE_FILTER) {
        uint8_t fn = 0;
        const uint8_t fl = orc_filter_node(sop, pods, i, leader, k, &fn);
        if (!((fl < 16u) && (fl != BS_FL_EVALUATED || fn < 16u))) continue;
      }
      if (!holds(s, k, req, pres)) continue;
      at = (int32_t)k;
    }
    io->pick_ns += mono_ns() - t_pick;
    if (at < 0) continue;                                       /* unschedulable this pass: holds nothing */
    for (uint32_t j = 0; j < 3; ++j) requested[(size_t)j * N + at] += req[j];
    requested[(size_t)3 * N + at] += 1;
    for (uint32_t sc = 0; sc < s->S; ++sc)
      if ((pres >> sc) & 1u) {
        if (!((rpres[at] >> sc) & 1u)) requested[(size_t)(4 + sc) * N + at] = 0;
        requested[(size_t)(4 + sc) * N + at] += req[4 + sc];
        rpres[at] |= 1u << sc;
      }
    assumed_on[i] = at;
    if (!grouped) { io->pod_node[i] = at; continue; }           /* no label, core.go:269-272: Permit lets it through at once */
    bs_groups_soa* gr = &sop->groups;
    gr->matched[gi] += 1;                                       /* :290 MatchedPodNodes.Set */
    next_wait[i] = head[gi];
    head[gi] = (int32_t)i;
    if (!orc_permit_ready(gr->matched[gi], gr->min_member[gi], gr->status_scheduled[gi])) continue;   /* :303 */
    gr->flags[gi] |= BS_GROUP_SCHEDULED_LATCH;                  /* :305 */
    /* sendStartScheduleSignal -> StartBatchSchedule, batchscheduler.go:254-344 */
    if (gr->flags[gi] & BS_GROUP_PHASE_CLOSED) continue;        /* :258-261 phase is neither PreScheduling nor Scheduling: nobody is
                                                                 * released, the pod waits on (a late member of a gang in phase Scheduled) */
    const uint32_t k = gr->matched[gi];                         /* :292,:301 EVERY entry of MatchedPodNodes: the pods of this pass and the
                                                                 * ones that were waiting when it began (they have no queue index here) */
    for (int32_t w = head[gi]; w >= 0; w = next_wait[w]) io->pod_node[w] = assumed_on[w];   /* :324 Allow */
    head[gi] = -1;
    gr->matched[gi] = 0;                                        /* :326 pendingPods.Delete(uid), every entry */
    gr->status_scheduled[gi] += k;                              /* PostBind once per released pod, core.go:327 (uint32) */
    if (gr->status_scheduled[gi] >= gr->min_member[gi]) gr->flags[gi] |= BS_GROUP_PHASE_CLOSED;   /* core.go:329-330 phase Scheduled
                                                                 * (else Scheduling, :331-336: still open) */
    if (slot_of[gi] == 0xFFFFFFFFu) {
      if (io->n_released < io->cap) {
        io->released_group[io->n_released] = (uint32_t)gi;
        io->released_pods[io->n_released] = k;
        io->first_ns[io->n_released] = t_first[gi];
        io->ready_ns[io->n_released] = mono_ns() - t0;
        slot_of[gi] = io->n_released;
      }
      io->n_released++;
    } else {
      io->released_pods[slot_of[gi]] += k;                      /* a second release of the same gang (only when Status.Scheduled stays below MinMember: uint32 wrap) */
    }
  }
  io->total_ns = mono_ns() - t0;
  free(t_first); free(slot_of); free(head); free(next_wait); free(assumed_on);
}


/* ---------------------------------------------------------------------------------------------------------------------
 * All host cores as the CPU baseline of bench.py: n independent batches — disjoint pod subsets, WHOLE groups each (exact
 * whenever no pod's decision depends on a pod of another group: the steady state), each with its own orc_sop — on n
 * threads.  Returns the wall time in nanoseconds from the first thread's creation to the last join.  (Forked Python workers
 * measured 0.25 s for what 256 threads do in a few tens of milliseconds: process start-up and pickling, not arithmetic.)
 * ------------------------------------------------------------------------------------------------------------------- */
#include <pthread.h>

typedef struct {
  orc_sop* sop;
  const bs_pods_soa* pods;
  uint32_t stages;
  const bs_batch_out* out;
} orc_job;

static void* orc_job_run(void* arg) {
  orc_job* j = (orc_job*)arg;
  if (j->pods->p) orc_batch(j->sop, j->pods, j->stages, j->out);
  return NULL;
}

int64_t orc_batch_threads(orc_job* jobs, uint32_t n) {
  pthread_t* th = (pthread_t*)calloc(n ? n : 1, sizeof(pthread_t));
  unsigned char* started = (unsigned char*)calloc(n ? n : 1, 1);
  if (!th || !started) { free(th); free(started); return -1; }
  const int64_t t0 = mono_ns();
  for (uint32_t i = 0; i < n; ++i) {
    if (pthread_create(&th[i], NULL, orc_job_run, &jobs[i]) == 0) started[i] = 1;
    else orc_job_run(&jobs[i]);                                   /* no thread to be had: run it here */
  }
  for (uint32_t i = 0; i < n; ++i)
    if (started[i]) pthread_join(th[i], NULL);
  const int64_t dt = mono_ns() - t0;
  free(th);
  free(started);
  return dt;
}
