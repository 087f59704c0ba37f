"""Python face of host/bs_phase.cpp (include/bsched_host.h): the PodGroup phase machine of the reference's controller
(pkg/scheduler/controller/controller.go:179-311 syncHandler, :111-130 pgAdded), the in-memory transitions of Permit / PostBind /
StartBatchSchedule (core.go:279-281, :325-360, batchscheduler.go:258-285) and CreateMergePatch (pkg/util/k8s.go:34-48).  Same names and argument
meaning as the reference; the API-server I/O between the calls is the caller's.  Host only — nothing here touches the GPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace

from . import capi, plugin

PHASES = ["", "Pending", "Running", "PreScheduling", "Scheduling", "Scheduled", "Unknown", "Finished", "Failed"]      # types.go:28-56 (bsh_phase)
POD_PHASES = {"Pending": 0, "Running": 1, "Succeeded": 2, "Failed": 3, "Unknown": 4}                                  # v1.PodPhase (bsh_pod_phase)
SYNC_PATCH_RECOVER, SYNC_PATCH, SYNC_CACHE_DELETE, SYNC_NO_REQUEUE, SYNC_LISTED_PODS = 1, 2, 4, 8, 16
HOST_PHASE_SYMBOLS = ["bsh_pg_new", "bsh_pg_free", "bsh_pg_succeeded", "bsh_pg_failed", "bsh_pg_sync", "bsh_pg_enqueue", "bsh_pg_permit", "bsh_pg_post_bind",
                      "bsh_pg_start_gate", "bsh_phase_closed", "bsh_phase_name", "bsh_phase_parse", "bsh_merge_patch", "bsh_pg_status_json", "bsh_pg_status_patch"]


class _Status(C.Structure):
    _fields_ = [("phase", C.c_uint32), ("scheduled", C.c_uint32), ("running", C.c_uint32), ("succeeded", C.c_uint32), ("failed", C.c_uint32),
                ("occupied_by", C.c_uint64), ("schedule_start_ns", C.c_int64)]


@dataclass
class PodGroupStatus:
    """pgv1.PodGroupStatus (types.go:104-130); schedule_start_ns = 0 is the zero time."""
    phase: str = ""
    scheduled: int = 0
    running: int = 0
    succeeded: int = 0
    failed: int = 0
    schedule_start_ns: int = 0
    occupied_by: str = ""

    def _c(self) -> _Status:
        return _Status(PHASES.index(self.phase), self.scheduled & 0xFFFFFFFF, self.running & 0xFFFFFFFF, self.succeeded & 0xFFFFFFFF, self.failed & 0xFFFFFFFF,
                       1 if self.occupied_by else 0, self.schedule_start_ns)

    def _from(self, c: _Status) -> "PodGroupStatus":
        return replace(self, phase=PHASES[c.phase], scheduled=c.scheduled, running=c.running, succeeded=c.succeeded, failed=c.failed, schedule_start_ns=c.schedule_start_ns)


_ready = False


def _lib():
    global _ready
    L = plugin.load_host_library()
    if not _ready:
        vp, u64, i64, u32, u8, sz = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_uint8, C.c_size_t
        P = C.POINTER
        L.bsh_pg_new.restype = vp
        L.bsh_pg_free.argtypes = [vp]
        L.bsh_pg_succeeded.restype = u32
        L.bsh_pg_succeeded.argtypes = [vp]
        L.bsh_pg_failed.restype = u32
        L.bsh_pg_failed.argtypes = [vp]
        L.bsh_pg_sync.argtypes = [vp, u32, i64, P(_Status), P(u64), P(u8), u32, P(_Status), P(_Status), P(u32)]
        L.bsh_pg_enqueue.argtypes = [u32, i64, P(_Status)]
        L.bsh_pg_permit.restype = u32
        L.bsh_pg_permit.argtypes = [u32]
        L.bsh_pg_post_bind.argtypes = [u32, P(_Status), i64, P(_Status), P(u8)]
        L.bsh_pg_start_gate.argtypes = [u32, P(_Status), P(u8), P(u8)]
        L.bsh_phase_closed.argtypes = [u32]
        L.bsh_phase_name.restype = C.c_char_p
        L.bsh_phase_name.argtypes = [u32]
        L.bsh_phase_parse.argtypes = [C.c_char_p]
        L.bsh_merge_patch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz, P(sz)]
        L.bsh_pg_status_json.argtypes = [P(_Status), C.c_char_p, C.c_char_p, sz, P(sz)]
        L.bsh_pg_status_patch.argtypes = [P(_Status), P(_Status), C.c_char_p, C.c_char_p, sz, P(sz)]
        _ready = True
    return L


def _text(call) -> str:
    """two-call protocol of the text-producing entry points: ask for the size, then fetch"""
    need = C.c_size_t(0)
    rc = call(None, 0, C.byref(need))
    if rc != 0:
        raise capi.BsError(rc, "bsh text call")
    buf = C.create_string_buffer(need.value)
    rc = call(buf, need.value, C.byref(need))
    if rc != 0:
        raise capi.BsError(rc, "bsh text call")
    return buf.value.decode("utf-8")


def create_merge_patch(original: str, new: str) -> str:
    """util.CreateMergePatch (pkg/util/k8s.go:34-48) on the two marshalled objects."""
    a, b = original.encode("utf-8"), new.encode("utf-8")
    return _text(lambda out, cap, need: _lib().bsh_merge_patch(a, b, out, cap, need))


def status_json(st: PodGroupStatus) -> str:
    c = st._c()
    occ = st.occupied_by.encode("utf-8") if st.occupied_by else None
    return _text(lambda out, cap, need: _lib().bsh_pg_status_json(C.byref(c), occ, out, cap, need))


def status_patch(frm: PodGroupStatus, to: PodGroupStatus) -> str:
    a, b = frm._c(), to._c()
    occ = to.occupied_by.encode("utf-8") if to.occupied_by else None
    return _text(lambda out, cap, need: _lib().bsh_pg_status_patch(C.byref(a), C.byref(b), occ, out, cap, need))


def phase_closed(phase: str) -> bool:
    """StartBatchSchedule releases nobody in this phase (batchscheduler.go:258-261): the group's BS_GROUP_PHASE_CLOSED bit."""
    return bool(_lib().bsh_phase_closed(PHASES.index(phase)))


def permit_phase(phase: str) -> str:
    return PHASES[_lib().bsh_pg_permit(PHASES.index(phase))]


def post_bind(min_member: int, st: PodGroupStatus, now_ns: int):
    """PostBind's status arithmetic (core.go:325-360) -> (pgCopy.Status, a PATCH is sent)"""
    c, out, patch = st._c(), _Status(), C.c_uint8(0)
    rc = _lib().bsh_pg_post_bind(min_member, C.byref(c), now_ns, C.byref(out), C.byref(patch))
    if rc != 0:
        raise capi.BsError(rc, "bsh_pg_post_bind")
    return st._from(out), bool(patch.value)


def start_gate(min_member: int, st: PodGroupStatus):
    """StartBatchSchedule's gate (batchscheduler.go:258-285) -> (pods may be allowed, ScheduleStartTime is stamped first)"""
    c, rel, stamp = st._c(), C.c_uint8(0), C.c_uint8(0)
    _lib().bsh_pg_start_gate(min_member, C.byref(c), C.byref(rel), C.byref(stamp))
    return bool(rel.value), bool(stamp.value)


def enqueue(min_member: int, creation_ns: int, st: PodGroupStatus) -> bool:
    """pgAdded (controller.go:111-130)"""
    c = st._c()
    return bool(_lib().bsh_pg_enqueue(min_member, creation_ns, C.byref(c)))


class PodGroupController:
    """One PodGroup's controller-side state: the Succeed / Failed uid sets of its cache entry (they outlive a sync)."""

    def __init__(self):
        self._h = _lib().bsh_pg_new()

    def close(self):
        if self._h:
            _lib().bsh_pg_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def counts(self):
        return int(_lib().bsh_pg_succeeded(self._h)), int(_lib().bsh_pg_failed(self._h))

    def sync_handler(self, min_member: int, creation_ns: int, st: PodGroupStatus, pods):
        """syncHandler (controller.go:179-311).  pods: [(uid, "Pending" | "Running" | "Succeeded" | "Failed" | "Unknown")] as the List would return them.
        -> (status after the first PATCH or None, pgCopy.Status at the end, actions = SYNC_* bits)"""
        n = len(pods)
        uids = (C.c_uint64 * max(n, 1))(*[int(u) for u, _ in pods])
        phs = (C.c_uint8 * max(n, 1))(*[POD_PHASES[p] for _, p in pods])
        c, rec, out, act = st._c(), _Status(), _Status(), C.c_uint32(0)
        rc = _lib().bsh_pg_sync(self._h, min_member, creation_ns, C.byref(c), uids, phs, n, C.byref(rec), C.byref(out), C.byref(act))
        if rc != 0:
            raise capi.BsError(rc, "bsh_pg_sync")
        return (st._from(rec) if act.value & SYNC_PATCH_RECOVER else None), st._from(out), int(act.value)
