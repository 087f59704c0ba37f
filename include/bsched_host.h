/* bsched_host.h — host-side pieces of SURVEY.md section 8(f)-4 that need no GPU: the PodGroup phase machine and the JSON
 * merge-patch writer the reference's controller and PostBind send status changes with.  Exported by libbsched_host.so
 * (batch-scheduler_amd/host/bs_phase.cpp) next to the C++ mirror of ScheduleOperation (bs_host.cpp, bound in plugin.py).
 *
 * What each entry point replaces in the reference (tenstack/batch-scheduler):
 *   bsh_merge_patch        pkg/util/k8s.go:34-48 CreateMergePatch = json.Marshal(original), json.Marshal(new),
 *                          evanphx/json-patch v4.5.0+incompatible (go.mod:33) jsonpatch.CreateMergePatch — here on the two JSON texts
 *   bsh_pg_status_json     encoding/json of pgv1.PodGroupStatus with its tags (pkg/apis/podgroup/v1/types.go:104-130)
 *   bsh_pg_status_patch    the PATCH bodies of controller.go:212-220,295-300, core.go:346-351, batchscheduler.go:276-284
 *   bsh_pg_sync            PodGroupController.syncHandler, pkg/scheduler/controller/controller.go:179-311 (the phase machine proper)
 *   bsh_pg_enqueue         pgAdded's filter, controller.go:111-130
 *   bsh_pg_permit          the in-memory transition of ScheduleOperation.Permit, core.go:279-281
 *   bsh_pg_post_bind       ScheduleOperation.PostBind's status arithmetic, core.go:325-360
 *   bsh_pg_start_gate      StartBatchSchedule's phase gate and its ScheduleStartTime stamp, batchscheduler.go:258-285
 *   bsh_phase_closed       the phases in which StartBatchSchedule releases nobody (batchscheduler.go:258-261) = BS_GROUP_PHASE_CLOSED of bsched.h
 *   bsh_phase_name / _parse  the PodGroupPhase strings, types.go:28-56
 *
 * The API-server I/O around them (List pods, Get / Patch PodGroup, the rate-limited work queue) stays with the caller: these are the pure
 * functions in between.  Plain C ABI: scalars, caller-owned buffers, no allocation handed out.  Return 0 or a negative bs_status (bsched.h).
 */
#ifndef BSCHED_HOST_H
#define BSCHED_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PodGroupPhase, types.go:28-56.  BSH_PHASE_NONE is the empty string a freshly created PodGroup carries (controller.go:199-200). */
typedef enum bsh_phase {
  BSH_PHASE_NONE = 0,
  BSH_PHASE_PENDING = 1,       /* "Pending"        */
  BSH_PHASE_RUNNING = 2,       /* "Running"        */
  BSH_PHASE_PRESCHEDULING = 3, /* "PreScheduling"  */
  BSH_PHASE_SCHEDULING = 4,    /* "Scheduling"     */
  BSH_PHASE_SCHEDULED = 5,     /* "Scheduled"      */
  BSH_PHASE_UNKNOWN = 6,       /* "Unknown" (declared by the reference, set nowhere) */
  BSH_PHASE_FINISHED = 7,      /* "Finished"       */
  BSH_PHASE_FAILED = 8         /* "Failed"         */
} bsh_phase;

/* v1.PodPhase of a listed pod (controller.go:248-262) */
typedef enum bsh_pod_phase {
  BSH_POD_PENDING = 0, BSH_POD_RUNNING = 1, BSH_POD_SUCCEEDED = 2, BSH_POD_FAILED = 3, BSH_POD_UNKNOWN = 4
} bsh_pod_phase;

/* pgv1.PodGroupStatus, types.go:104-130.  Times are nanoseconds since the Unix epoch, 0 = the zero time (metav1.Time{}). */
typedef struct bsh_pg_status {
  uint32_t phase;              /* bsh_phase */
  uint32_t scheduled, running, succeeded, failed;
  uint64_t occupied_by;        /* interned OccupiedBy string, 0 = "" (carried, never changed here) */
  int64_t schedule_start_ns;
} bsh_pg_status;

/* what syncHandler does besides computing the new status */
#define BSH_SYNC_PATCH_RECOVER 0x01u /* controller.go:211-220: the first PATCH (Scheduled recovered from the listed pods)            */
#define BSH_SYNC_PATCH         0x02u /* :293-303: the PATCH at the end (status_out differs from what the server holds by then)         */
#define BSH_SYNC_CACHE_DELETE  0x04u /* :304-306: the patched phase is Finished / Failed: the cache entry goes                         */
#define BSH_SYNC_NO_REQUEUE    0x08u /* :227-231: quorum scheduled, nothing running, started > 48 h after creation: not enqueued again */
#define BSH_SYNC_LISTED_PODS   0x10u /* the pod list was consulted (:203-209 and / or :237-243): a caller may skip the List otherwise   */

/* One PodGroup's controller-side state that outlives a sync: the Succeed / Failed uid sets of its cache entry
 * (cache.go:52-67 PodGroupMatchStatus.Succeed / .Failed, filled at controller.go:252-255 and never emptied). */
typedef struct bsh_pg bsh_pg;
bsh_pg* bsh_pg_new(void);
void bsh_pg_free(bsh_pg* pg);
uint32_t bsh_pg_succeeded(const bsh_pg* pg);
uint32_t bsh_pg_failed(const bsh_pg* pg);

/* syncHandler (controller.go:179-311) for one PodGroup: `in` = the status the lister returned, pods = the group's pods as the List of
 * :205 / :245 would return them (uid, bsh_pod_phase).  recovered (nullable) = the status after the first PATCH (:211-220), valid when
 * BSH_SYNC_PATCH_RECOVER is set — the object the final comparison (:293) is made against; out = pgCopy.Status at :293.  actions = BSH_SYNC_*. */
int bsh_pg_sync(bsh_pg* pg, uint32_t min_member, int64_t creation_ns, const bsh_pg_status* in, const uint64_t* pod_uids, const uint8_t* pod_phases,
                uint32_t npods, bsh_pg_status* recovered, bsh_pg_status* out, uint32_t* actions);

/* pgAdded / pgUpdated (controller.go:111-130): 1 when the informer event puts the group on the work queue — not for a Finished / Failed group, and
 * not for one whose quorum was scheduled, with nothing running, more than 48 h after its creation (its pods may have been collected). */
int bsh_pg_enqueue(uint32_t min_member, int64_t creation_ns, const bsh_pg_status* st);
/* Permit's in-memory transition Pending -> PreScheduling (core.go:279-281); every other phase is left alone. */
uint32_t bsh_pg_permit(uint32_t phase);
/* PostBind (core.go:325-360): Scheduled++, phase Scheduled at the quorum else Scheduling (and ScheduleStartTime = now when it was zero);
 * *patch = 1 when the phase changed (:340: only then is a PATCH sent; the counter moves in memory either way, :359). */
int bsh_pg_post_bind(uint32_t min_member, const bsh_pg_status* in, int64_t now_ns, bsh_pg_status* out, uint8_t* patch);
/* StartBatchSchedule's gate (batchscheduler.go:258-285): *release = 0 when the phase is neither PreScheduling nor Scheduling;
 * *stamp = 1 when Status.Scheduled >= MinMember: ScheduleStartTime is patched to now before anybody is allowed. */
int bsh_pg_start_gate(uint32_t min_member, const bsh_pg_status* in, uint8_t* release, uint8_t* stamp);
/* 1 for the phases batchscheduler.go:258-261 returns on: the value of BS_GROUP_PHASE_CLOSED for the group. */
int bsh_phase_closed(uint32_t phase);
const char* bsh_phase_name(uint32_t phase);           /* "" for BSH_PHASE_NONE and for values outside the enum */
int bsh_phase_parse(const char* name);                /* -1: not a PodGroupPhase */

/* JSON merge patch between two JSON OBJECT texts, as jsonpatch.CreateMergePatch builds it (evanphx/json-patch v4.5.0 merge.go getDiff):
 * keys of `modified` that are new, of another JSON type, or of a different value go in whole (objects recurse and are dropped when their
 * diff is empty, arrays are replaced whole), keys only `original` has go in as null; the result is marshalled the way encoding/json
 * marshals a map: keys sorted, no white space, numbers as float64, <, >, & and U+2028 / U+2029 escaped.  Writes a NUL-terminated text into
 * out[cap]; *need = bytes needed including the NUL (BS_ERR_CAPACITY when cap is too small; out may be NULL to ask).  BS_ERR_INVALID: a text
 * that is not one JSON object. */
int bsh_merge_patch(const char* original, const char* modified, char* out, size_t cap, size_t* need);
/* json.Marshal of a PodGroupStatus (types.go:104-130: "phase", "occupiedBy" omitted when empty, "scheduled", "running", "succeeded", "failed",
 * "scheduleStartTime" as RFC 3339 UTC seconds or null for the zero time).  occupied_by (nullable) = the OccupiedBy string. */
int bsh_pg_status_json(const bsh_pg_status* st, const char* occupied_by, char* out, size_t cap, size_t* need);
/* {"status":{...}} merge patch that takes a PodGroup from status `from` to status `to` ("{}" when nothing differs). */
int bsh_pg_status_patch(const bsh_pg_status* from, const bsh_pg_status* to, const char* occupied_by, char* out, size_t cap, size_t* need);

#ifdef __cplusplus
}
#endif
#endif
