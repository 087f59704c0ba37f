/*
 * bsched.h — C ABI of the MI355X gang-feasibility core.
 *
 * This is the drop-in boundary for ONE hot path of tenstack/batch-scheduler: the
 * PreFilter / Filter / Permit resource-fit arithmetic of
 * /root/reference/pkg/scheduler/core/core.go.  The reference has no FFI today
 * (CGO_ENABLED=0, Makefile:28); the Go plugin keeps its framework surface
 * (batchscheduler.go:102 PreFilter, :151 Filter, :165 Permit, :214 Less) and binds the
 * entry points below through cgo (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - Only flat arrays of scalars cross the boundary: no strings, no pointers to
 *    pointers.  Strings the reference compares (group full names, OwnerReferences UIDs,
 *    resource names, node names) are interned to integers by the Go shim.
 *  - A "resource vector" (upstream nodeinfo.Resource) is L = 4 + S int64 lanes:
 *      lane 0 MilliCPU, 1 Memory, 2 EphemeralStorage, 3 AllowedPodNumber,
 *      lane 4+s = ScalarResources[name_s]  with a presence bit s (Go map key exists).
 *    S (scalar lanes) is fixed per context (bs_config.scalar_lanes <= BS_MAX_SCALARS).
 *  - 2-D arrays are lane-major SoA: x[lane * count + index].
 *  - The caller owns every buffer it passes; the library copies during the call and
 *    keeps no caller pointer after return.  That is one half of the cgo pointer rules; the other
 *    half — a Go pointer passed to C may not point at Go memory holding Go pointers — rules out
 *    passing a Go-allocated struct of slice pointers BY POINTER: Go callers use the *_flat entry
 *    points (every array its own argument; see "flat-argument forms" below), C / C++ / ctypes
 *    callers may use either form.  The library owns device memory, its HIP stream and events.
 *  - Every function returns BS_OK (0) or a negative bs_status; nothing throws or aborts
 *    across the ABI.  Reference panics (uint32 divide by zero core.go:716-717, nil
 *    maxPGStatus core.go:525) are reported as decision codes, not crashes.
 *  - Mutating calls on one bs_ctx must be serialised by the caller.
 */
#ifndef BSCHED_H
#define BSCHED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BS_ABI_VERSION 7u

enum {
  BS_LANE_CPU = 0,       /* Resource.MilliCPU          */
  BS_LANE_MEM = 1,       /* Resource.Memory            */
  BS_LANE_EPH = 2,       /* Resource.EphemeralStorage  */
  BS_LANE_PODS = 3,      /* Resource.AllowedPodNumber  */
  BS_FIXED_LANES = 4,
  BS_MAX_SCALARS = 12,
  BS_MAX_LANES = 16
};

typedef enum bs_status {
  BS_OK = 0,
  BS_ERR_INVALID = -1,   /* bad argument / inconsistent sizes                         */
  BS_ERR_NO_DEVICE = -2, /* no gfx950 device, or HIP runtime failure at create time   */
  BS_ERR_HIP = -3,       /* a HIP call failed (bs_last_error has the text)            */
  BS_ERR_STATE = -4,     /* call order: nodes / fit / groups / pods not loaded yet    */
  BS_ERR_CAPACITY = -5,  /* exceeds configured or addressable capacity                */
  BS_ERR_NOMEM = -6,
  BS_ERR_COMM = -7,      /* RCCL failure                                              */
  BS_ERR_RETRY = -8      /* (ABI v7) the last batch's results are void for a reason the library has already repaired (the class / pair id
                          * space of a queue patch overflowed and the queue was re-derived; an in-launch hand-over timed out and the context
                          * went over to separate launches): nothing is wrong with the caller's state — run the batch again.  Distinct from
                          * BS_ERR_STATE, which bs_batch_map answers for a VALID batch that merely wrote no host results               */
} bs_status;

/* ---- node flags (per NodeInfo in list order) -------------------------------------- */
#define BS_NODE_NIL            0x01u /* list entry is nil           core.go:606 (skip); Filter: core.go:447 -> error */
#define BS_NODE_NO_NODE        0x02u /* info.Node() == nil          core.go:610 (skip); Filter: Get() fails core.go:443 */
#define BS_NODE_UNSCHEDULABLE  0x04u /* Spec.Unschedulable          core.go:615 (skip); NOT consulted by Filter */
#define BS_NODE_TAINT_ERR      0x08u /* info.Taints() returned err  core.go:639 (adds zeros, still compared) */
#define BS_NODE_SKIP_MASK      0x07u

/* ---- group flags (cache.PodGroupMatchStatus, cache.go:52-67) ---------------------- */
#define BS_GROUP_SCHEDULED_LATCH 0x01u /* pgs.Scheduled (set core.go:305, never cleared) */
#define BS_GROUP_HAS_POD         0x02u /* pgs.Pod != nil (first pod seen, core.go:486-488) */
#define BS_GROUP_HAS_MINRES      0x04u /* Spec.MinResources != nil (core.go:489-493)      */
#define BS_GROUP_DENIED          0x08u /* live entry in lastDeniedPG (core.go:105-110)    */
#define BS_GROUP_PHASE_CLOSED    0x10u /* Status.Phase is none of Pending / PreScheduling / Scheduling: StartBatchSchedule returns without
                                        * releasing anybody (batchscheduler.go:258-261).  Read and written by bs_seq_run only (set when PostBind
                                        * turns the phase to Scheduled, core.go:329-330); the batch entry points ignore it */

/* ---- pod flags / group sentinels -------------------------------------------------- */
#define BS_POD_LAST_PERMITTED 0x01u   /* live entry in lastPermittedPod (core.go:95-98) */
#define BS_POD_NOT_GROUPED   (-1)     /* no PodGroupLabel (util/k8s.go:62-70)           */
#define BS_POD_GROUP_MISSING (-2)     /* label set but podGroupStatusCache.Get == nil   */

/* ---- PreFilter decision codes (which `return` of core.go:88-167 fired) ------------ */
#define BS_PF_PASS_NOT_GROUPED    0u  /* core.go:89-92   */
#define BS_PF_PASS_LAST_PERMITTED 1u  /* core.go:95-98   */
#define BS_PF_PASS_NO_MAX         2u  /* core.go:127-130 */
#define BS_PF_PASS_FIRST_FITS     3u  /* core.go:136-146 */
#define BS_PF_PASS_IS_MAX         4u  /* core.go:150-155 */
#define BS_PF_PASS_RESERVE_FITS   5u  /* core.go:157-166 */
#define BS_PF_ERR_PG_NOT_FOUND   16u  /* core.go:100-103 */
#define BS_PF_ERR_DENIED         17u  /* core.go:105-110 */
#define BS_PF_ERR_OCCUPIED       18u  /* core.go:113-115 <- :503-510 */
#define BS_PF_REJECT_FIRST       19u  /* core.go:140-144 (+AddToDenyCache) */
#define BS_PF_REJECT_RESERVE     20u  /* core.go:161-165 (+AddToDenyCache) */
#define BS_PF_PANIC_DIV0         32u  /* findMaxPG uint32 divide by zero, core.go:716-717 */
#define BS_PF_IS_PASS(code) ((code) < 16u)

/* ---- Filter per-pod codes (core.go:170-191, :514-564) ----------------------------- */
#define BS_FL_PASS_NOT_GROUPED    0u  /* core.go:171-174: every node passes */
#define BS_FL_PASS_IS_MAX         1u  /* case 1, core.go:531-535            */
#define BS_FL_PASS_NO_MINRES      2u  /* maxSingleRequired == nil, :542-544 */
#define BS_FL_EVALUATED           3u  /* per node: case 2 / case 3 / ErrorResourceNotEnough */
#define BS_FL_ERR_PG_NOT_FOUND   16u  /* core.go:177-180 */
#define BS_FL_PANIC_NIL_MAX      32u  /* sop.maxPGStatus == nil deref, core.go:525 */
#define BS_FL_NOT_RUN            64u  /* PreFilter did not pass: framework never calls Filter */

/* Filter per-(pod,node) codes, bs_filter_one */
#define BS_FN_PASS_CASE2          0u  /* left >= pod + maxSingle, core.go:551-555 */
#define BS_FN_PASS_CASE3          1u  /* node cannot hold a leader member, :557-561 */
#define BS_FN_ERR_NOT_ENOUGH     16u  /* ErrorResourceNotEnough, :562-563 */
#define BS_FN_ERR_SNAPSHOT       17u  /* "SnapShot not initialized", :545-548 */

#define BS_K_NONE        0xFFFFFFFFu  /* scan ran over every node, no prefix satisfied */
#define BS_K_NOT_SCANNED 0xFFFFFFFEu  /* decision taken without a node scan            */

typedef struct bs_ctx bs_ctx;

typedef struct bs_config {
  uint32_t abi_version;   /* BS_ABI_VERSION */
  int32_t  device;        /* HIP device ordinal */
  uint32_t scalar_lanes;  /* S */
  uint32_t eph_gate;      /* upstream feature gate LocalStorageCapacityIsolation (default 1):
                             Resource.Add counts ephemeral-storage only when on */
  uint32_t enable_timing; /* 0 off; 1 hipEvents around the scan and filter kernels of every 8th batch;
                             2 around every kernel group of every batch (diagnostic: the table build and
                             the scan / Filter evaluation then run as separate launches) */
  uint32_t reserved[3];
} bs_config;

/* Node snapshot, in SnapshotSharedLister().NodeInfos().List() order (core.go:597). */
typedef struct bs_nodes_soa {
  uint32_t n;
  const int64_t*  allocatable;        /* [L][n] info.AllocatableResource()                   */
  const int64_t*  requested;          /* [L][n] info.RequestedResource(); pods lane = podCount
                                         exactly as core.go:650-653 / :455-458 resolve it      */
  const uint32_t* allocatable_present;/* [n] bit s: allocatable.ScalarResources has key s     */
  const uint32_t* requested_present;  /* [n] bit s: requested.ScalarResources has key s       */
  const uint8_t*  flags;              /* [n] BS_NODE_*                                        */
} bs_nodes_soa;

/* PodGroup cache state, in the iteration order findMaxPG is to use (core.go:703; Go map
 * order is random, so the caller fixes one and parity is defined given that order). */
typedef struct bs_groups_soa {
  uint32_t g;
  uint32_t* min_member;            /* [g] Spec.MinMember        (types.go:79-101)  */
  uint32_t* status_scheduled;      /* [g] Status.Scheduled      (types.go:104-130) */
  uint32_t* matched;               /* [g] len(MatchedPodNodes.Items()) (cache.go:57) */
  uint8_t*  flags;                 /* [g] BS_GROUP_*                                */
  uint32_t* cls;                   /* [g] fit class of pgs.Pod (valid iff HAS_POD)  */
  int64_t*  min_resources;         /* [L][g] Spec.MinResources (valid iff HAS_MINRES) */
  uint32_t* min_resources_present; /* [g] scalar keys in MinResources               */
  uint64_t* occupied_by;           /* [g] interned Status.OccupiedBy, 0 == ""       */
} bs_groups_soa;

/* Pending pods in scheduling-queue order. */
typedef struct bs_pods_soa {
  uint32_t p;
  const int32_t*  group;       /* [p] group index, BS_POD_NOT_GROUPED or BS_POD_GROUP_MISSING */
  const int64_t*  req;         /* [L][p] getPodResourceRequire(pod), core.go:761-772          */
  const uint32_t* req_present; /* [p] scalar keys in that Resource                            */
  const uint32_t* cls;         /* [p] fit class the pod defines if it becomes pgs.Pod         */
  const uint64_t* owner;       /* [p] interned sorted+joined OwnerReferences UIDs, 0 = none   */
  const uint8_t*  flags;       /* [p] BS_POD_*                                                */
} bs_pods_soa;

/* ---- fit-mask builder: checkFit (core.go:741-759) for every (pod-template class, node) ----------
 * checkFit = predicates.PodMatchNodeSelector && predicates.PodToleratesNodeTaints of
 * k8s.io/kubernetes v1.17.5 (go.mod:102; not vendored).  Strings stay on the caller's side: every
 * string crosses as an interned id (equal strings <=> equal ids, id 0 <=> the empty string), integers
 * that upstream parses with strconv.ParseInt(s, 10, 64) cross as (value, ok). */
#define BS_EFFECT_NONE               0u /* "" (tolerations only: matches every effect) */
#define BS_EFFECT_NO_SCHEDULE        1u
#define BS_EFFECT_PREFER_NO_SCHEDULE 2u /* ignored by PodToleratesNodeTaints' filter */
#define BS_EFFECT_NO_EXECUTE         3u /* any other effect string: a distinct code >= 4 */

#define BS_TOL_OP_DEFAULT 0u /* "" behaves as Equal */
#define BS_TOL_OP_EQUAL   1u
#define BS_TOL_OP_EXISTS  2u /* any other operator string: >= 3, tolerates nothing */

#define BS_OP_IN             0u /* v1.NodeSelectorOperator */
#define BS_OP_NOT_IN         1u
#define BS_OP_EXISTS         2u
#define BS_OP_DOES_NOT_EXIST 3u
#define BS_OP_GT             4u
#define BS_OP_LT             5u
#define BS_OP_INVALID     0x80u /* OR-ed in by the caller: validateLabelKey / validateLabelValue fails
                                   or the operator string is unknown (the selector conversion errors
                                   and the whole term matches nothing); value-count errors are found
                                   by the library from val_off */

#define BS_TPL_HAS_REQUIRED     0x1u /* Affinity.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil */
#define BS_TPL_SELECTOR_INVALID 0x2u /* a Spec.NodeSelector key/value fails label validation:
                                        labels.SelectorFromSet returns the empty selector (matches all) */

typedef struct bs_node_labels {
  uint32_t n;                    /* must equal the loaded snapshot's n; list order                 */
  const uint32_t* name;          /* [n] metadata.name (for matchFields)                            */
  const uint32_t* label_off;     /* [n+1] CSR into label_*: node.Labels (keys unique per node)     */
  const uint32_t* label_key;
  const uint32_t* label_val;
  const int64_t*  label_int;     /* ParseInt(value, 10, 64)                                        */
  const uint8_t*  label_int_ok;  /* 1 iff it parsed                                                */
  const uint32_t* taint_off;     /* [n+1] CSR into taint_*: node.Spec.Taints                       */
  const uint32_t* taint_key;
  const uint32_t* taint_val;
  const uint8_t*  taint_effect;  /* BS_EFFECT_*                                                    */
} bs_node_labels;

typedef struct bs_requirements { /* a table of v1.NodeSelectorRequirement                          */
  uint32_t count;
  const uint32_t* key;           /* [count]                                                        */
  const uint8_t*  op;            /* [count] BS_OP_* (| BS_OP_INVALID)                              */
  const uint32_t* val_off;       /* [count+1] CSR into val*                                        */
  const uint32_t* val;
  const int64_t*  val_int;
  const uint8_t*  val_int_ok;
} bs_requirements;

typedef struct bs_fit_templates { /* one entry per distinct pod template (fit class)               */
  uint32_t c;
  uint32_t field_name_key;       /* interned "metadata.name"                                       */
  const uint8_t*  flags;         /* [c] BS_TPL_*                                                   */
  const uint32_t* sel_off;       /* [c+1] CSR: Spec.NodeSelector pairs                             */
  const uint32_t* sel_key;
  const uint32_t* sel_val;
  const uint32_t* term_off;      /* [c+1] CSR: Required.NodeSelectorTerms (ORed)                   */
  const uint32_t* term_expr_off; /* [terms+1] CSR into exprs:  term.MatchExpressions (ANDed)       */
  const uint32_t* term_field_off;/* [terms+1] CSR into fields: term.MatchFields (ANDed)            */
  bs_requirements exprs;
  bs_requirements fields;
  const uint32_t* tol_off;       /* [c+1] CSR: Spec.Tolerations                                    */
  const uint32_t* tol_key;
  const uint32_t* tol_val;
  const uint8_t*  tol_op;        /* BS_TOL_OP_*                                                    */
  const uint8_t*  tol_effect;    /* BS_EFFECT_*                                                    */
} bs_fit_templates;

/* Outputs of one batch (any pointer may be NULL = not wanted). */
typedef struct bs_batch_out {
  uint8_t*  pf_code;       /* [p] BS_PF_*                                                     */
  uint32_t* pf_first_k;    /* [p] list index of the node whose prefix first satisfied the
                              request (reference `count`-1, core.go:605,623), BS_K_NONE,
                              or BS_K_NOT_SCANNED                                             */
  int32_t*  pf_leader;     /* [p] group index findMaxPG returned for this pod (sop.maxPGStatus,
                              core.go:120-122), -1 none                                       */
  uint8_t*  fl_code;       /* [p] BS_FL_*                                                     */
  uint32_t* fl_feasible;   /* [p] nodes on which Filter returns nil                           */
  uint64_t* fl_bitmap;     /* [ceil(n/64)][p] word-major: bit (node&63) of word node>>6.  OPT-IN: the
                              pods x nodes bitmap is only materialised (one extra streaming kernel +
                              a p*ceil(n/64)*8-byte copy) when this pointer is set; fl_rows below
                              carries the same information ~p/rows times smaller                */
  uint32_t* group_admit;   /* [g] pods of the group that pass PreFilter and (if Filter ran)
                              have >=1 feasible node; summed over ranks when sharded          */
  uint8_t*  group_ready;   /* [g] quorum predicate core.go:303 with matched+admit             */
  /* Filter results by distinct request ("slot rows").  Filter(pod, node) (core.go:170-191) ==
   *   fl_code[pod] == BS_FL_EVALUATED ? bit (node&63) of fl_rows[(node>>6) * fl_rows_cap + fl_slot[pod]]
   *                                   : fl_code[pod] < 16        (every node / no node)
   * so the Go plugin's Filter is a bit test with no cgo crossing.  Pods with equal derived requests share
   * a row: (request class, leader seen) in steady state, (leader run, request class) while first-pod captures
   * or MinResources defaults can still happen in the batch; on the general chain (more than sixteen leader
   * changes in one batch) a row is the pod itself (fl_slot[pod] == pod).
   * Rows no pod of the batch refers to are unspecified.  bs_filter_rows_count never exceeds 2 x the request classes the
   * library knows (<= 2 x (pods at the last bs_pods_load + pods inserted since): classes keep their number between two
   * derivations, see bs_pods_apply). */
  uint32_t* fl_slot;          /* [p] row of pod p; meaningful iff fl_code[p] == BS_FL_EVALUATED        */
  uint64_t* fl_rows;          /* [ceil(n/64)][fl_rows_cap] word-major; rows >= *fl_rows_n untouched     */
  uint32_t* fl_rows_feasible; /* [fl_rows_cap] feasible-node count per row (NULL ok)                    */
  uint32_t  fl_rows_cap;      /* in: capacity in rows (bs_filter_rows_count tells how many are needed)  */
  uint32_t* fl_rows_n;        /* out: rows of this batch (NULL ok); BS_ERR_CAPACITY if > fl_rows_cap    */
} bs_batch_out;

/* bs_batch_run stage bits */
#define BS_STAGE_PREFILTER 0x1u
#define BS_STAGE_FILTER    0x2u
#define BS_STAGE_TALLY     0x4u   /* per-group admit counts + ready bits */
#define BS_STAGE_ALL       0x7u
#define BS_BATCH_COMMIT    0x100u /* persist first-pod capture / occupancy / deny into the ctx
                                     group state (bs_groups_read), as sequential PreFilter would */
#define BS_BATCH_HOST_RESULTS 0x200u /* latency mode: the last launch of the batch also writes every result bs_batch_read
                                     returns (per-pod arrays, admit, ready, Filter rows) straight into pinned host memory;
                                     bs_batch_read then needs no device-to-host copy and no stream wait — it polls a
                                     completion word (bs_batch_map: not even a host-side copy).  Honoured on the steady-state
                                     chain (two launches) and the positional chain (three) of a single-rank context; a no-op (results are copied
                                     as usual) on the general chain.  Costs the batch a few microseconds of PCIe writes, so
                                     throughput runs leave it off. */

#define BS_BATCH_FILTER_DENY 0x400u /* with BS_STAGE_FILTER: the deny entry a FAILING Filter writes (core.go:183-185: computeResourceSatisfied
                                     errors on some node -> AddToDenyCache(group)) is replayed inside the batch, on the device: every
                                     later pod of the group that gets to the deny check (core.go:105-110) returns BS_PF_ERR_DENIED and
                                     is never offered to Filter.  The batch then equals PreFilter(pod) + Filter(pod, node) for every
                                     node, pod by pod in queue order — the reference's behaviour with the Filter extension point
                                     enabled.  Without the flag Filter is a what-if (its verdicts are reported, the entry is not
                                     written).  Single-rank contexts.  See bs_batch_run. */

/* ---- lifecycle ------------------------------------------------------------------- */
uint32_t    bs_abi_version(void);
const char* bs_strerror(int status);
const char* bs_last_error(const bs_ctx* ctx);           /* text of the last BS_ERR_HIP/COMM */
int bs_create(const bs_config* cfg, bs_ctx** out);
int bs_destroy(bs_ctx* ctx);

/* ---- snapshot / state loads (replaces frameworkHandler.SnapshotSharedLister() reads,
 *      core.go:437,567,597, and the cache reads of cache.go:94-102) ------------------ */
int bs_nodes_load(bs_ctx* ctx, const bs_nodes_soa* nodes);
/* fit[c][n] = checkFit(rep pod of class c, node n), core.go:741-759; bit n&31 of word n>>5 */
int bs_fit_load(bs_ctx* ctx, uint32_t n_classes, const uint32_t* fit_bits);
/* Same result computed on the device from labels / taints / templates (replaces the checkFit call of
 * core.go:646 for every (class, node) at once); nodes flagged NIL / NO_NODE / TAINT_ERR get 0.
 * bs_fit_read copies the current masks out ([n_classes][ceil(n/32)] words, caller-sized). */
int bs_fit_build(bs_ctx* ctx, const bs_node_labels* nodes, const bs_fit_templates* templates);
int bs_fit_read(bs_ctx* ctx, uint32_t* fit_bits_out);
int bs_groups_load(bs_ctx* ctx, const bs_groups_soa* groups);
int bs_groups_read(bs_ctx* ctx, bs_groups_soa* groups_out); /* caller-sized arrays, g must match */
/* Per-cycle group changes without a full reload: Permit adds to MatchedPodNodes (core.go:290), PostBind moves
 * pods to Status.Scheduled (core.go:327), the quorum latch (core.go:305) and the deny TTL (core.go:105,424)
 * flip flag bits.  Each delta REPLACES matched / status_scheduled / flags of group `index`;
 * BS_GROUP_HAS_POD and BS_GROUP_HAS_MINRES must keep their loaded value (a first-pod capture changes cls and
 * MinResources too: use bs_groups_load).  A group index may appear only once per call.  Validated as a whole before
 * anything is applied.
 * Like bs_groups_load it re-runs findMaxPG on the device and returns without waiting for the GPU. */
typedef struct bs_group_delta {
  uint32_t index, matched, status_scheduled, flags;
} bs_group_delta;
int bs_groups_apply(bs_ctx* ctx, const bs_group_delta* deltas, uint32_t count);
/* Zero-copy hand-over: points `view` at the library's pinned upload buffer, sized for `p` pods, so that the caller can
 * marshal the queue in place (cast the const away) and pass the SAME struct to bs_pods_load, which then skips its packing
 * copy.  The pointers stay valid until the next bs_pods_map / bs_pods_load on the context; the resident queue and every
 * other entry point are unaffected by a mapping (the staging buffer has its own layout).  bs_pods_load with pointers
 * into the mapped buffer but another `p` than it was mapped for is refused (BS_ERR_INVALID). */
int bs_pods_map(bs_ctx* ctx, uint32_t p, bs_pods_soa* view);
int bs_pods_load(bs_ctx* ctx, const bs_pods_soa* pods);   /* also derives, on the device, what a batch needs from the pods
                                                              alone: request classes (equal (req lanes, req_present) <=> equal
                                                              class; a batch evaluates every distinct derived request once),
                                                              per-group first pod / first owner, (group, class) pairs.
                                                              Returns without waiting for the GPU.  No call order is implied
                                                              between the loads; fit-class indices (pods.cls, groups.cls) are
                                                              checked against the loaded fit classes by bs_batch_run
                                                              (BS_ERR_INVALID).                                          */

/* ---- the pending queue stays resident: per-cycle pod deltas ---------------------------------------
 * The reference calls PreFilter (core.go:88) for every pending pod in every scheduling cycle, on a queue that
 * changes by a few pods between cycles (a released gang leaves: batchscheduler.go:254-344; new pods arrive; the
 * lastPermittedPod TTL, core.go:95,188, flips a flag).  bs_pods_apply patches the queue loaded by bs_pods_load ON
 * THE DEVICE instead of re-uploading and re-hashing it: pods, their request classes, the (group, request class)
 * pairs and the per-group pod minima stay resident.  One call = one delta against the CURRENT queue (p pods):
 *   remove      [n_remove] indices into the current queue, strictly ascending: stable removal
 *   flag_index  [n_flags]  indices into the current queue, strictly ascending, with flag_value[i] = the pod's new
 *               bs_pods_soa.flags byte (an update of a pod that is also removed is ignored)
 *   insert      insert.p new pods (same arrays as bs_pods_load); insert_at[k] = position of inserted pod k in the
 *               NEW queue (p - n_remove + insert.p pods), strictly ascending; NULL = append at the tail.  Retained
 *               pods keep their relative order and fill the positions in between.
 * Validated as a whole before anything is applied (BS_ERR_INVALID leaves the queue untouched).  Results of the next
 * bs_batch_run equal those after bs_pods_load of the new queue, bit for bit; only fl_slot row NUMBERS may differ
 * (rows are named after request classes, and a class whose last pod left keeps its number until the library
 * re-derives the classes — it does so by itself when the id space fills up).  Returns without waiting for the GPU.
 * bs_pods_count / bs_pods_read report the resident queue (bs_pods_read: caller-sized arrays, out->p must match). */
typedef struct bs_pods_delta {
  uint32_t n_remove;
  const uint32_t* remove;
  uint32_t n_flags;
  const uint32_t* flag_index;
  const uint8_t*  flag_value;
  bs_pods_soa     insert;
  const uint32_t* insert_at;
} bs_pods_delta;
int bs_pods_apply(bs_ctx* ctx, const bs_pods_delta* delta);
int bs_pods_count(const bs_ctx* ctx, uint32_t* p_out);
typedef struct bs_pods_out {
  uint32_t p;
  int32_t*  group;
  int64_t*  req;          /* [L][p] */
  uint32_t* req_present;
  uint32_t* cls;
  uint64_t* owner;
  uint8_t*  flags;
} bs_pods_out;
int bs_pods_read(bs_ctx* ctx, const bs_pods_out* out);

/* node churn (BASELINE config 5): stable delete / append / requested-update */
#define BS_DELTA_UPDATE 0u   /* replace node `index` (all lanes, presence, flags, fit column) */
#define BS_DELTA_APPEND 1u   /* append at the end of the list                                 */
#define BS_DELTA_REMOVE 2u   /* stable delete of node `index`                                 */
typedef struct bs_node_delta {
  uint32_t kind, index;
  int64_t  allocatable[BS_MAX_LANES];
  int64_t  requested[BS_MAX_LANES];
  uint32_t allocatable_present, requested_present;
  uint32_t flags;
  uint32_t fit_default;      /* 1: node fits every class except those listed in ...        */
  uint32_t n_fit_exceptions; /* ... fit_exceptions (classes whose bit is !fit_default)      */
  uint32_t fit_exceptions[8];
} bs_node_delta;
int bs_nodes_apply(bs_ctx* ctx, const bs_node_delta* deltas, uint32_t count);
int bs_nodes_count(const bs_ctx* ctx, uint32_t* n_out);
/* The scheduler's assume step for pods the plugin let through (upstream cache.AssumePod -> NodeInfo.AddPod ‡: the node's
 * requested resources grow by the pod's request, its pod count by one) — what makes the next PreFilter (core.go:88) see a
 * smaller cluster.  Node `index` gets a new requested vector (pods lane = podCount, as bs_nodes_soa.requested) and new
 * requested-present bits; allocatable, flags, fit column and list position are unchanged.  Unlike bs_nodes_apply (list
 * surgery, re-upload from the first changed node, a stream wait) this is a device-side scatter: ONE launch, the deltas read
 * from pinned memory, nothing waited for.  A node index may appear only once per call; validated as a whole first. */
typedef struct bs_node_request {
  uint32_t index, requested_present;
  int64_t  requested[BS_MAX_LANES];
} bs_node_request;
int bs_nodes_assume(bs_ctx* ctx, const bs_node_request* reqs, uint32_t count);

/* ---- single queries: 1:1 drop-ins ------------------------------------------------- */
/* compareClusterResourceAndRequire(pod of class `cls`, req, percent), core.go:595-632.
 * *fits = its bool; *first_k as in bs_batch_out.pf_first_k (BS_K_NONE when false). */
int bs_cluster_fits(bs_ctx* ctx, uint32_t cls, float percent, const int64_t* req /*[L]*/,
                    uint32_t req_present, uint8_t* fits, uint32_t* first_k);
/* singleNodeResource for every node (core.go:634-670): left[L][n] and presence[n];
 * flagged-skip nodes are still evaluated (the skip lives in the callers). */
int bs_node_left(bs_ctx* ctx, uint32_t cls, float percent, int64_t* left, uint32_t* present);
/* The running sums compareClusterResourceAndRequire forms (core.go:602,621): one row per
 * non-skipped node in list order.  prefix[L][n], present[n], node_index[n]; *rows = count. */
int bs_scan_prefix(bs_ctx* ctx, uint32_t cls, float percent, int64_t* prefix, uint32_t* present,
                   uint32_t* node_index, uint32_t* rows);
/* computeClusterResource(pod of class cls), core.go:566-593 (log-only in the reference). */
int bs_cluster_total(bs_ctx* ctx, uint32_t cls, int64_t* total /*[L]*/, uint32_t* present);
/* computeResourceSatisfied for one (pod, node) given leader group (core.go:514-564);
 * leader < 0 reports BS_FL_PANIC_NIL_MAX in *fl_code.  *fn_code valid iff *fl_code == BS_FL_EVALUATED. */
int bs_filter_one(bs_ctx* ctx, int32_t pod_group, const int64_t* pod_req, uint32_t pod_req_present,
                  int32_t leader, uint32_t node, uint8_t* fl_code, uint8_t* fn_code);
/* findMaxPG over the loaded group state (core.go:701-739): *leader = index or -1;
 * *finished = maxFinished; returns BS_OK and sets *panic=1 on the divide-by-zero case. */
int bs_find_max_pg(bs_ctx* ctx, int32_t* leader, uint32_t* finished, uint8_t* panic);

/* ---- the batched hot path --------------------------------------------------------- */
/* Decisions equal to calling the reference's PreFilter for pods 0..p-1 in order against the
 * frozen snapshot and group counters, with the in-batch side effects of core.go:113
 * (first-pod capture, MinResources default, occupancy) and :142,:163 (deny cache) replayed
 * in queue order.  Filter is evaluated for pods that passed, with sop.maxPGStatus as that
 * pod's PreFilter left it.  Filter's own TTL writes:
 *   core.go:188 (lastPermittedPod.Add on a passing node) only ever concerns the SAME pod's next PreFilter — no other pod of the
 *     batch can see it; not replayed.
 *   core.go:183-185 (AddToDenyCache when Filter fails on a node) turns every later pod of the group that reaches the deny
 *     check into ERR_DENIED.  Without BS_BATCH_FILTER_DENY Filter is a what-if: the event is visible in the results (fl_code ==
 *     BS_FL_EVALUATED && fl_feasible < nodes), the entry is not written, and the batch is the sequential PreFilter + Filter run up
 *     to and including, per group, the first such pod (the shipped config does not enable Filter:
 *     deploy/scheduler/config/batch_scheduler_config.json).  WITH the flag the entry is replayed inside the batch, on the device,
 *     on every chain (csrc/bs_fdeny.hpp): the results equal PreFilter(pod) followed by Filter(pod, node) on every node, pod by pod
 *     in queue order — every output, the stale leader included (tests/test_gpu_filter_deny.py against the oracle's batch with the
 *     flag; tests/test_batch_vs_sequential.py R1F against the host mirror's calls; tests/test_filter_deny_pass.py pins the oracle
 *     on an independent object-level replay).  One short launch behind the chain does it (two on the general chain); when a pod it
 *     turns away was needed by somebody else (it would have brought a changed findMaxPG result into sop.maxFinishedPG, or been its
 *     group's first-pod capture — only behind a pod let through on its lastPermittedPod entry that failed Filter), the batch is
 *     settled by fixed-point re-runs the first time its results are asked for (bs_batch_sync / read / map; at once for a
 *     committing batch) — bs_filter_deny_stats counts them; after a re-run that left the three-launch chains bs_batch_map
 *     answers BS_ERR_STATE (read it with bs_batch_read).  BS_BATCH_COMMIT persists the entries with PreFilter's own.
 * Asynchronous on the context stream; bs_batch_sync waits. */
int bs_batch_run(bs_ctx* ctx, uint32_t stages);
int bs_batch_sync(bs_ctx* ctx);
int bs_batch_read(bs_ctx* ctx, const bs_batch_out* out);
/* Zero-copy form of bs_batch_read for a batch that ran with BS_BATCH_HOST_RESULTS: waits for the batch's completion word
 * and hands out read-only pointers into the pinned host memory the last launch wrote — no device-to-host copy, no stream
 * wait and no host-side memcpy (bs_batch_read spends most of a latency-mode cycle copying ~40 bytes per pod out of that
 * very memory).  The pointers stay valid, and their contents unchanged, until the next bs_batch_run on this context.
 * BS_ERR_STATE when the last batch did not write host results (general chain, sharded / external-reduce contexts,
 * BS_BATCH_COMMIT, or the flag not set): use bs_batch_read.  Per-pod arrays hold `p` entries, per-group arrays `g`.
 * Filter rows: word-major with a row stride of `fl_rows_stride` rows (bit test as in bs_batch_out, with fl_rows_stride
 * in place of fl_rows_cap); fl_rows is NULL when Filter did not run or the batch has more rows than the pinned window
 * holds (fl_rows_n still tells how many: fetch them with bs_batch_read). */
typedef struct bs_batch_view {
  uint32_t p, g, words;               /* pods, groups, ceil(nodes / 64) of the batch                         */
  const uint8_t*  pf_code;
  const uint32_t* pf_first_k;
  const int32_t*  pf_leader;
  const uint8_t*  fl_code;
  const uint32_t* fl_feasible;
  const uint32_t* fl_slot;
  const uint32_t* group_admit;        /* NULL unless the batch ran BS_STAGE_TALLY                            */
  const uint8_t*  group_ready;
  const uint64_t* fl_rows;            /* [words][fl_rows_stride]                                             */
  const uint32_t* fl_rows_feasible;   /* [fl_rows_n]                                                         */
  uint32_t fl_rows_stride, fl_rows_n;
} bs_batch_view;
int bs_batch_map(bs_ctx* ctx, bs_batch_view* view);
/* Rows (distinct Filter requests) the last loaded pods can produce: sizes fl_rows / fl_rows_feasible. */
int bs_filter_rows_count(bs_ctx* ctx, uint32_t* rows);

/* ---- the sequential scheduling pass on the device -------------------------------------------------------------------
 * The reference decides POD BY POD: upstream's scheduleOne calls PreFilter (core.go:88-167) for the next pod of the queue
 * against the CURRENT cluster, [Filter, core.go:170-191, on every node when the stage is on,] picks a node and assumes the
 * pod on it, then Permit (core.go:268-309) counts it into its gang and, at the quorum of core.go:303, the waiting pods of
 * the gang are released (batchscheduler.go:254-344) and PostBind (core.go:327) counts them into Status.Scheduled — before
 * the next pod's PreFilter runs.  bs_batch_run answers every pod against a FROZEN snapshot (a pre-screen); bs_seq_run is the
 * reference's own order of events, one pod at a time, with everything resident on the device and no host round trip in
 * between: ONE launch walks the resident queue (bs_pods_load / bs_pods_apply) and for every pod runs PreFilter with the
 * node requests and group counters as the pods before it left them (first-pod capture, MinResources default, OccupiedBy,
 * deny entries, findMaxPG, the node scan), the node choice, the assume step, Permit and the release.
 * Node choice is upstream's business, not the plugin's; the rule here is the one host/bs_drain.cpp and the CPU replay
 * (oracle/bs_oracle_seq.c) state: FIRST FIT in list order over nodes without a BS_NODE_* flag whose checkFit bit is set for
 * the pod's class, that pass the plugin's Filter when BS_STAGE_FILTER is on, and that hold the request (lane j in {cpu, mem,
 * eph} binds when the pod asks for it; pods lane: requested + 1 <= allocatable; a requested scalar needs the allocatable
 * key).  Assume: requested += request, pods lane + 1.  A pod that passes PreFilter but finds no node holds nothing; pods of
 * a gang that never reaches its quorum keep what they assumed (as the reference does until the Permit timeout).
 * `stages`: BS_STAGE_PREFILTER (mandatory) [| BS_STAGE_FILTER [| BS_BATCH_FILTER_DENY]].  Single-rank contexts only.
 * With BS_STAGE_FILTER the plugin's Filter gates the node choice (a what-if: no TTL write).  With BS_BATCH_FILTER_DENY as well, Filter's
 * own TTL writes happen inside the pass under the batch form's offer rule — Filter is called on EVERY node of the list, with the node
 * requests as the pods before this one left them: a node whose Filter fails deny-lists the pod's group (core.go:183-185; the gang's
 * later pods are turned away at :105-110, BS_GROUP_DENIED is set in the group state), a node whose Filter passes leaves the pod's
 * lastPermittedPod entry (:188, reported in `last_permitted`).  The pod itself goes on to the node choice among the passing nodes.
 * The release step is the reference's: at the quorum EVERY entry of MatchedPodNodes binds — the pods this pass placed and the
 * groups.matched pods that were already waiting (they have no queue index: they count in released_pods and in Status.Scheduled and get
 * no pod_node) —, matched returns to 0 (batchscheduler.go:292-333, core.go:327), and once Status.Scheduled >= MinMember the phase is
 * Scheduled (BS_GROUP_PHASE_CLOSED is set) and later members of the gang wait without being released (batchscheduler.go:258-261).
 * On return the context's node requests, group counters / flags / MinResources / OccupiedBy ARE the state the pass left
 * (bs_groups_read, bs_nodes_read; later batches and passes start from it); the queue itself is unchanged (remove the released
 * pods with bs_pods_apply).  Results are bit-identical to the reference's sequential pass on the same inputs
 * (tests/test_gpu_seq.py against oracle/bs_oracle_seq.c).  Synchronous: returns when the pass is done. */
typedef struct bs_seq_out {
  uint8_t*  pf_code;         /* [p] BS_PF_* of every pod's PreFilter call (NULL ok)                                    */
  uint32_t* pf_first_k;      /* [p] as bs_batch_out.pf_first_k (NULL ok)                                               */
  int32_t*  pf_leader;       /* [p] sop.maxFinishedPG as the pod's PreFilter call left it, -1 none (NULL ok)           */
  int32_t*  pod_node;        /* [p] node of every RELEASED pod (its gang reached the quorum, or it has no gang), else -1 (NULL ok) */
  uint32_t  cap;             /* capacity of the four per-gang arrays below                                             */
  uint32_t* released_group;  /* [cap] gangs in the order their quorum turned true                                      */
  uint32_t* released_pods;   /* [cap] pods released with each: every MatchedPodNodes entry, the earlier cycles' waiting pods included */
  int64_t*  first_ns;        /* [cap] device clock, ns since the pass began: the gang's first pod entered PreFilter    */
  int64_t*  ready_ns;        /* [cap] ... the quorum of core.go:303 turned true (NULL ok for all four)                 */
  uint32_t  n_released;      /* out: gangs released (may exceed cap: the first cap are recorded)                       */
  int64_t   total_ns;        /* out: device time of the whole pass                                                     */
  /* out, work counters: first-fit searches; PreFilter node scans; rounds of 1024 nodes those scans went through (they stop at
   * the reference's early exit); rounds of up to 16 candidate tiles of 64 nodes the first-fit searches looked at; findMaxPG
   * folds (the fold is only repeated after a capture / Permit / release changed a group's progress).  With table summaries in LDS
   * (clusters of up to 65 536 nodes) a scan "round" is 16 candidate tiles of 64 nodes looked at exactly; a rejected request
   * usually needs none. */
  uint64_t  node_picks, node_scans, scan_rounds, pick_rounds, leader_folds;
  uint64_t  table_builds;    /* out: (fit class, percent) tables whose tile summaries were taken from scratch (LDS cache misses) */
  uint8_t*  last_permitted;  /* [p] BS_STAGE_FILTER | BS_BATCH_FILTER_DENY passes: 1 = a Filter call of the pod passed, i.e. the pass left a
                              * lastPermittedPod entry for it (core.go:188; the 2 s clock stays with the caller); NULL ok (ABI v6)     */
} bs_seq_out;
int bs_seq_run(bs_ctx* ctx, uint32_t stages, bs_seq_out* out);
/* The node requests as the context holds them (after bs_nodes_load / bs_nodes_apply / bs_nodes_assume / bs_seq_run):
 * requested[L][n] lane-major, requested_present[n]; n = bs_nodes_count. */
int bs_nodes_read(bs_ctx* ctx, int64_t* requested, uint32_t* requested_present);

/* ---- batched queue ordering ---------------------------------------- */
/* The permutation that sorts the pending pods the way the scheduling queue does through ScheduleOperation.Compare
 * (core.go:368-411; Less, batchscheduler.go:214): perm_out[k] = index of the pod at queue position k.  Key, ascending:
 * priority DESCENDING; pods without a PodGroup label, then labelled pods of known groups, then labelled pods whose
 * group the lister does not know (for those Compare is false both ways: they go last within their priority); the
 * group's order rank; the pod's queue timestamp; ties keep input order.  `group` uses bs_pods_soa.group's encoding.
 * bs_queue_order_load hands over, per group of the loaded group state, the dense rank of (CreationTimestamp ascending,
 * group name DESCENDING) — equal (timestamp, name) pairs share a rank (the reference compares names, not namespaces). */
int bs_queue_order_load(bs_ctx* ctx, uint32_t g, const uint32_t* order_rank);
int bs_queue_sort(bs_ctx* ctx, uint32_t p, const int32_t* priority, const int32_t* group, const int64_t* queue_ts,
                  uint32_t* perm_out);

/* ---- pod-axis sharding (one process per GPU) -------------------------------------- */
/* Rank `rank` of `nranks` evaluates only the pods it owns.  Whole groups, balanced by pod count: walking the queue, the
 * first pod of every group carries the weight of the group's pods (an ungrouped pod carries 1), and the running weight W is
 * cut into nranks equal shares — a group (or ungrouped pod) whose weight starts at w belongs to rank floor(w * nranks / P).
 * Groups never straddle ranks, so the deny replay stays exact and the per-group admit counters of different ranks are
 * disjoint (one all-reduce(sum) merges them); no rank holds more than P / nranks pods plus one group, whatever the queue
 * order.  Pods of other ranks report pf_code 0xFF (BS_PF_NOT_OWNED).  The whole batch is loaded on every rank.
 * (batch-scheduler_amd/dist.py owner_ranks is the host mirror of the rule.) */
#define BS_PF_NOT_OWNED 0xFFu
int bs_shard_set(bs_ctx* ctx, uint32_t rank, uint32_t nranks);
/* Device address of the per-group admit counters (uint32[g]) so the caller's collective
 * (torch.distributed / RCCL all-reduce, sum) can run in place between the two halves. */
int bs_group_admit_devptr(bs_ctx* ctx, void** dptr, uint32_t* count);
/* Partitioned mode: the caller loads on each rank ONLY the pods that rank owns (whole groups, group
 * indices global, group state replicated) and reduces the admit counters itself.  With `on` the batch
 * stops after the per-group tally (no quorum pass); the caller all-reduces bs_group_admit_devptr and
 * calls bs_batch_finish.  Decisions equal the single-context batch whenever no first-pod capture can
 * occur in the batch (every group already has its pod): then no pod's decision depends on a pod of
 * another group.  Otherwise use bs_shard_set (whole batch on every rank). */
int bs_reduce_external(bs_ctx* ctx, uint32_t on);
/* Partitioned mode, the one thing a rank cannot know from its own pods: where in ITS queue the whole job's first pod that reaches
 * findMaxPG (core.go:118-123) stands.  A pod that returns before that line (lastPermittedPod, no label, unknown group, deny entry,
 * OccupiedBy) leaves sop.maxFinishedPG as the latest reaching pod IN FRONT OF IT left it — on the whole queue, not on the rank's part
 * of it; pf_leader and the Filter result of a BS_POD_LAST_PERMITTED pod hang on that.  `local_index` = number of this rank's pods
 * that stand in front of the job's first reaching pod (the caller partitions the queue, so it has the queue:
 * batch-scheduler_amd/dist.py first_reach_thresholds is the host rule; go/pkg/scheduler/core/bsched_shard.go firstReachThreshold restates it); 0xFFFFFFFF = none.
 * Valid for the loaded queue AND group state (every bs_pods_load / bs_pods_apply / bs_groups_load / bs_groups_apply resets it: a deny entry or
 * an OccupiedBy change moves the first reaching pod — set it again behind them); honoured by the steady-state chain, which is the chain
 * partitioned mode is exact on (no first-pod capture possible).  With it every output of a partitioned batch equals the single
 * context's (tests/test_gpu_multirank.py: plain equality). */
int bs_first_reach_hint(bs_ctx* ctx, uint32_t local_index);
/* The three multi-rank modes (one process per GPU; node / group / fit state replicated on every rank):
 *   replicated   bs_shard_set(rank, nranks) [or bs_comm_init]: the whole queue on every rank, ownership decided on the device.
 *   partitioned  bs_reduce_external(1): each rank holds only the pods of the groups it owns; with bs_comm_init the library
 *                still performs the all-reduce itself, otherwise the caller reduces bs_group_admit_devptr and calls
 *                bs_batch_finish.
 * In both the per-group admit counters of different ranks are disjoint, so all-reduce(sum) == all-reduce(max).
 * Exactness under sharding: decisions (pf_code, pf_first_k, Filter results, admit, ready) of owned pods equal the
 * single-context batch.  pf_leader — the stale shared field sop.maxFinishedPG a pod leaves behind when it returns before
 * core.go:120 — and with it the Filter result of BS_POD_LAST_PERMITTED pods is exact in replicated mode whenever no
 * first-pod capture can occur in the batch, and in partitioned mode with bs_first_reach_hint (round 5; without the hint such a pod
 * sees the leader left by the latest reaching pod OF ITS OWN RANK'S view).  In replicated mode WITH captures the own-rank view
 * remains (tests/test_gpu_multirank.py). */
/* Use caller-owned device memory (uint32[g], e.g. a torch tensor's data_ptr) for the admit counters,
 * so that a framework collective can reduce it in place.  NULL restores the internal buffer. */
int bs_group_admit_bind(bs_ctx* ctx, void* dptr);
/* HIP stream (hipStream_t) the context launches on, for event timing / stream ordering. */
int bs_stream(bs_ctx* ctx, void** stream);
/* Native RCCL path for hosts without torch (the Go shim): unique id is 128 bytes. */
int bs_comm_unique_id(uint8_t id[128]);
int bs_comm_init(bs_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t nranks);
/* Second half after the all-reduce: ready bits from the (reduced) admit counters. */
int bs_batch_finish(bs_ctx* ctx);

/* ---- flat-argument forms (the cgo binding) -----------------------------------------------------------------------
 * cgo's pointer-passing rule: a Go pointer handed to C may not point at Go memory that itself holds Go pointers.  A
 * Go-allocated C.bs_nodes_soa / bs_groups_soa / bs_pods_soa / bs_pods_delta / bs_batch_out / bs_seq_out / bs_node_labels /
 * bs_fit_templates whose fields point at Go slices is exactly that, and `C.bs_nodes_load(ctx, &soa)` panics under the default
 * cgocheck ("cgo argument has Go pointer to Go pointer").  The entry points below take every array as its OWN argument —
 * scalars and direct pointers to pointer-free arrays only — and forward to the struct forms on the C side; semantics, return
 * codes and NULL conventions are those of the struct forms.  The Go shim (go/pkg/scheduler/core) calls only these for the
 * struct-taking entry points; go/c11_client/shim_client.c goes through them as well, so they are compiled and run here.
 * (bs_group_delta / bs_node_delta / bs_node_request arrays hold no pointers and cross as they are; bs_batch_map fills a
 * caller struct with pointers into the LIBRARY's pinned memory — C pointers, allowed.) */
int bs_nodes_load_flat(bs_ctx* ctx, uint32_t n, const int64_t* allocatable, const int64_t* requested, const uint32_t* allocatable_present,
                       const uint32_t* requested_present, const uint8_t* flags);
int bs_groups_load_flat(bs_ctx* ctx, uint32_t g, const uint32_t* min_member, const uint32_t* status_scheduled, const uint32_t* matched,
                        const uint8_t* flags, const uint32_t* cls, const int64_t* min_resources, const uint32_t* min_resources_present,
                        const uint64_t* occupied_by);
int bs_groups_read_flat(bs_ctx* ctx, uint32_t g, uint32_t* min_member, uint32_t* status_scheduled, uint32_t* matched, uint8_t* flags, uint32_t* cls,
                        int64_t* min_resources, uint32_t* min_resources_present, uint64_t* occupied_by);
int bs_pods_load_flat(bs_ctx* ctx, uint32_t p, const int32_t* group, const int64_t* req, const uint32_t* req_present, const uint32_t* cls,
                      const uint64_t* owner, const uint8_t* flags);
int bs_pods_apply_flat(bs_ctx* ctx, uint32_t n_remove, const uint32_t* remove, uint32_t n_flags, const uint32_t* flag_index, const uint8_t* flag_value,
                       uint32_t n_insert, const int32_t* group, const int64_t* req, const uint32_t* req_present, const uint32_t* cls,
                       const uint64_t* owner, const uint8_t* flags, const uint32_t* insert_at);
int bs_pods_read_flat(bs_ctx* ctx, uint32_t p, int32_t* group, int64_t* req, uint32_t* req_present, uint32_t* cls, uint64_t* owner, uint8_t* flags);
int bs_batch_read_flat(bs_ctx* ctx, uint8_t* pf_code, uint32_t* pf_first_k, int32_t* pf_leader, uint8_t* fl_code, uint32_t* fl_feasible,
                       uint64_t* fl_bitmap, uint32_t* group_admit, uint8_t* group_ready, uint32_t* fl_slot, uint64_t* fl_rows,
                       uint32_t* fl_rows_feasible, uint32_t fl_rows_cap, uint32_t* fl_rows_n);
/* bs_seq_run: the five scalar results come back through `scalars_out` = {n_released, total_ns, node_picks, node_scans, scan_rounds,
 * pick_rounds, leader_folds, table_builds} (int64[8], NULL ok) */
int bs_seq_run_flat(bs_ctx* ctx, uint32_t stages, uint8_t* pf_code, uint32_t* pf_first_k, int32_t* pf_leader, int32_t* pod_node, uint32_t cap,
                    uint32_t* released_group, uint32_t* released_pods, int64_t* first_ns, int64_t* ready_ns, int64_t* scalars_out,
                    uint8_t* last_permitted);
/* bs_fit_build: node tables, then the template tables; `ex_*` = bs_fit_templates.exprs, `fd_*` = bs_fit_templates.fields */
int bs_fit_build_flat(bs_ctx* ctx, uint32_t n, const uint32_t* name, const uint32_t* label_off, const uint32_t* label_key, const uint32_t* label_val,
                      const int64_t* label_int, const uint8_t* label_int_ok, const uint32_t* taint_off, const uint32_t* taint_key,
                      const uint32_t* taint_val, const uint8_t* taint_effect,
                      uint32_t c, uint32_t field_name_key, const uint8_t* tpl_flags, const uint32_t* sel_off, const uint32_t* sel_key,
                      const uint32_t* sel_val, const uint32_t* term_off, const uint32_t* term_expr_off, const uint32_t* term_field_off,
                      uint32_t ex_count, const uint32_t* ex_key, const uint8_t* ex_op, const uint32_t* ex_val_off, const uint32_t* ex_val,
                      const int64_t* ex_val_int, const uint8_t* ex_val_int_ok,
                      uint32_t fd_count, const uint32_t* fd_key, const uint8_t* fd_op, const uint32_t* fd_val_off, const uint32_t* fd_val,
                      const int64_t* fd_val_int, const uint8_t* fd_val_int_ok,
                      const uint32_t* tol_off, const uint32_t* tol_key, const uint32_t* tol_val, const uint8_t* tol_op, const uint8_t* tol_effect);

/* ---- measurement ------------------------------------------------------------------ */
#define BS_KERNEL_PREPASS   0u
#define BS_KERNEL_LEADER    1u
#define BS_KERNEL_QUERY     2u   /* steady state: launch A (per-pod decisions + chunk-local table build) */
#define BS_KERNEL_TABLES    3u
#define BS_KERNEL_SCAN      4u   /* node scan (+ Filter evaluation in steady state: launch B) */
#define BS_KERNEL_RESOLVE   5u   /* final codes (+ tally and quorum in steady state: launch C) */
#define BS_KERNEL_FILTER    6u
#define BS_KERNEL_TALLY     7u
#define BS_KERNEL_COUNT     8u
typedef struct bs_timing {
  double   total_ms[BS_KERNEL_COUNT];   /* hipEvent elapsed, summed over launches */
  uint64_t launches[BS_KERNEL_COUNT];
} bs_timing;
int bs_timing_reset(bs_ctx* ctx);
int bs_timing_get(bs_ctx* ctx, bs_timing* out);   /* synchronises the stream */
const char* bs_kernel_name(uint32_t kernel_id);
/* Work counters of the last batch: evals the scan kernel executed (node rows visited x 64-lane
 * tiles), logical pods x nodes, scan queries, tables built. */
typedef struct bs_batch_stats {
  uint64_t scan_queries, scan_rows_executed, scan_evals_executed, tables_built, logical_evals;
  uint64_t filter_evals;            /* logical: pods x nodes                                  */
  uint64_t filter_distinct;         /* distinct Filter requests actually evaluated            */
  uint64_t filter_evals_executed;   /* filter_distinct x nodes                                */
  uint64_t scan_queries_logical;    /* pods that needed a node scan (scan_queries = distinct ones scanned) */
  uint64_t class_mode;              /* 1: the batch worked on request classes, 0: one slot per pod        */
  uint64_t fast_path;               /* 1: the steady-state chain ran (two launches)                          */
  uint64_t launches;                /* kernel launches of the batch                                        */
  uint64_t chain;                   /* 0 general chain, 1 steady-state chain, 2 positional chain              */
  uint64_t filter_lane_blocks;      /* (ABI v7) throughput regime, transposed Filter item: sum over (tile of 64 request slots, 64-node block) of the
                                     * resource lanes the launch compared there (0..4: the item's lane mask; a pair of tiles is compared on the
                                     * union of their masks) ...                                                                              */
  uint64_t filter_tile_blocks;      /* ... and the number of such (tile, block) units: the ratio is the k of DESIGN.md's VALU-issue bound     */
} bs_batch_stats;
int bs_batch_stats_get(bs_ctx* ctx, bs_batch_stats* out);
/* bs_pods_apply calls so far, and how many of them (plus later batches) had to re-derive classes and pairs from the
 * resident queue instead of patching them (id space used up, very large deltas, group count changed). */
int bs_pods_apply_stats(const bs_ctx* ctx, uint64_t* applies, uint64_t* rederives);
/* BS_BATCH_FILTER_DENY batches that had to be run again so far (see bs_batch_run: a pod turned away by a failing Filter's deny
 * entry was needed by somebody else; the batch is then settled by fixed-point iteration when its results are first asked for). */
int bs_filter_deny_stats(const bs_ctx* ctx, uint64_t* reruns);
/* Batches launched on a GUESSED findMaxPG answer, and how many of the guesses were wrong.  After a group patch (bs_groups_apply)
 * findMaxPG runs again on the device; bs_batch_run would have to wait for its answer (which running-sum table the batch uses) before
 * it could launch anything.  When the loaded state is a steady one (every group has its pod and MinResources) and the answer has not
 * landed yet, the chain is launched on the previous cycle's answer instead and checked when the results are first asked for
 * (bs_batch_sync / read / map): a wrong guess re-runs the batch there — results are never taken from a wrong guess.  Never for a
 * committing batch.  BS_NO_SPECULATE=1 turns it off. */
int bs_speculation_stats(const bs_ctx* ctx, uint64_t* launched, uint64_t* missed);

#ifdef __cplusplus
}
#endif
#endif /* BSCHED_H */
