"""Python face of the C++ host-side mirror (libbsched_host.so) of the reference's ScheduleOperation.

Same entry points and argument meaning as pkg/scheduler/core/core.go (PreFilter / Filter / Permit /
PostBind / Compare) and the batch release of batchscheduler.go:254-344, one pod at a time, TTL caches
on a virtual clock.  All node arithmetic goes through the HIP library (bs_find_max_pg,
bs_cluster_fits, bs_filter_one): no GPU, no decisions.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import capi, soa

SECOND = 1_000_000_000
PERMIT_READY, PERMIT_WAITING, PERMIT_NOT_MATCHED, PERMIT_NOT_FOUND = 0, 1, 2, 3

HOST_SYMBOLS = ["bsh_create", "bsh_destroy", "bsh_set_time", "bsh_time", "bsh_gpu_calls", "bsh_add_group", "bsh_prefilter", "bsh_filter",
                "bsh_permit", "bsh_postbind", "bsh_less", "bsh_start_batch", "bsh_sync", "bsh_group_matched", "bsh_group_status_scheduled",
                "bsh_group_flags", "bsh_group_denied", "bsh_ttl_new", "bsh_ttl_free", "bsh_ttl_set", "bsh_ttl_add", "bsh_ttl_get",
                "bsh_ttl_delete", "bsh_ttl_count", "bsh_drain"]

_hlib = None


def load_host_library():
    global _hlib
    if _hlib is None:
        path = _build.HOST_LIB_PATH
        if not os.path.exists(path):
            raise capi.BsError(-2, "load_host_library", f"{path} not built")
        capi.load_library()          # libbsched.so first (RTLD_GLOBAL not needed: DT_NEEDED + rpath $ORIGIN)
        L = C.CDLL(path)
        vp, u64, i64, u32, i32, u8 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_int32, C.c_uint8
        P = C.POINTER
        L.bsh_create.restype = vp
        L.bsh_create.argtypes = [vp, u32, i64]
        L.bsh_destroy.argtypes = [vp]
        L.bsh_set_time.argtypes = [vp, i64]
        L.bsh_time.restype = i64
        L.bsh_time.argtypes = [vp]
        L.bsh_gpu_calls.restype = u64
        L.bsh_gpu_calls.argtypes = [vp]
        L.bsh_add_group.restype = i32
        L.bsh_add_group.argtypes = [vp, u32, u32, i64, i64, u64, P(i64), u32]
        L.bsh_prefilter.argtypes = [vp, u64, u64, i32, P(i64), u32, u32, u64, P(u32)]
        L.bsh_filter.argtypes = [vp, u64, i32, P(i64), u32, u32, P(u8), P(u8)]
        L.bsh_permit.argtypes = [vp, u64, u64, i32, u32, P(u8)]
        L.bsh_postbind.argtypes = [vp, i32]
        L.bsh_sync.argtypes = [vp]
        L.bsh_less.argtypes = [vp, i32, i32, i64, i32, i32, i64]
        L.bsh_start_batch.restype = u32
        L.bsh_start_batch.argtypes = [vp, i32, P(u64), P(u32), u32]
        for n in ("bsh_group_matched", "bsh_group_status_scheduled", "bsh_group_flags"):
            getattr(L, n).restype = u32
            getattr(L, n).argtypes = [vp, i32]
        L.bsh_group_denied.argtypes = [vp, i32]
        L.bsh_ttl_new.restype = vp
        L.bsh_ttl_free.argtypes = [vp]
        L.bsh_ttl_set.argtypes = [vp, u64, u64, i64, i64]
        L.bsh_ttl_add.argtypes = [vp, u64, u64, i64, i64]
        L.bsh_ttl_get.argtypes = [vp, u64, i64, P(u64)]
        L.bsh_ttl_delete.argtypes = [vp, u64]
        L.bsh_ttl_count.restype = u32
        L.bsh_ttl_count.argtypes = [vp, i64]
        L.bsh_drain.argtypes = [P(DrainIO)]
        _hlib = L
    return _hlib


class DrainIO(C.Structure):
    """bsh_drain_io of host/bs_drain.cpp"""
    _fields_ = [("ctx", C.c_void_p), ("lanes", C.c_uint32),
                ("n", C.c_uint32), ("allocatable", C.POINTER(C.c_int64)), ("requested", C.POINTER(C.c_int64)),
                ("allocatable_present", C.POINTER(C.c_uint32)), ("requested_present", C.POINTER(C.c_uint32)), ("node_flags", C.POINTER(C.c_uint8)),
                ("fit_bits", C.POINTER(C.c_uint32)), ("n_classes", C.c_uint32),
                ("g", C.c_uint32), ("min_member", C.POINTER(C.c_uint32)), ("status_scheduled", C.POINTER(C.c_uint32)), ("matched", C.POINTER(C.c_uint32)),
                ("group_flags", C.POINTER(C.c_uint8)),
                ("pods", soa.PodsStruct), ("stages", C.c_uint32), ("max_cycles", C.c_uint32),
                ("cap", C.c_uint32), ("admitted_group", C.POINTER(C.c_uint32)), ("admitted_pods", C.POINTER(C.c_uint32)),
                ("admitted_ns", C.POINTER(C.c_int64)), ("cycle_ns", C.POINTER(C.c_int64)), ("pod_node", C.POINTER(C.c_int32)),
                ("n_admitted", C.c_uint32), ("n_cycles", C.c_uint32), ("n_stuck", C.c_uint32), ("pods_left", C.c_uint32), ("total_ns", C.c_int64)]


def drain(ctx: capi.Context, nodes: soa.Nodes, fit: soa.FitMasks, groups: soa.Groups, pods: soa.Pods, stages: int, max_cycles: int = 0) -> dict:
    """The batched scheduling cycle until no gang is ready (host/bs_drain.cpp): score the queue, release the first ready gang,
    assume its pods (first fit), patch nodes / groups / queue on the device, score again.  `ctx` must hold exactly this state;
    nodes.requested / requested_present and the group counters are updated IN PLACE (pass copies to keep the originals)."""
    lib = load_host_library()
    u32p, i64p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    cap = max(groups.g, 1)
    adm_g, adm_p = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    adm_ns, cyc_ns = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
    pod_node = np.full(max(pods.p, 1), -1, np.int32)
    bits = np.ascontiguousarray(fit.bits, np.uint32)
    io = DrainIO()
    io.ctx, io.lanes = ctx._h, ctx.L
    io.n = nodes.n
    io.allocatable, io.requested = nodes.allocatable.ctypes.data_as(i64p), nodes.requested.ctypes.data_as(i64p)
    io.allocatable_present, io.requested_present = nodes.allocatable_present.ctypes.data_as(u32p), nodes.requested_present.ctypes.data_as(u32p)
    io.node_flags = nodes.flags.ctypes.data_as(u8p)
    io.fit_bits, io.n_classes = bits.ctypes.data_as(u32p), fit.n_classes
    io.g = groups.g
    io.min_member, io.status_scheduled = groups.min_member.ctypes.data_as(u32p), groups.status_scheduled.ctypes.data_as(u32p)
    io.matched, io.group_flags = groups.matched.ctypes.data_as(u32p), groups.flags.ctypes.data_as(u8p)
    io.pods = pods.as_struct()
    io.stages, io.max_cycles, io.cap = stages, max_cycles, cap
    io.admitted_group, io.admitted_pods = adm_g.ctypes.data_as(u32p), adm_p.ctypes.data_as(u32p)
    io.admitted_ns, io.cycle_ns = adm_ns.ctypes.data_as(i64p), cyc_ns.ctypes.data_as(i64p)
    io.pod_node = pod_node.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib.bsh_drain(C.byref(io))
    if rc != 0:
        raise capi.BsError(rc, "bsh_drain", ctx._lib.bs_last_error(ctx._h).decode())
    k = min(int(io.n_admitted), cap)
    ctx.p = int(io.pods_left)
    return dict(admitted_group=adm_g[:k].copy(), admitted_pods=adm_p[:k].copy(), admitted_ns=adm_ns[:k].copy(), cycle_ns=cyc_ns[:k].copy(),
                pod_node=pod_node[: pods.p].copy(), n_admitted=int(io.n_admitted), n_cycles=int(io.n_cycles), n_stuck=int(io.n_stuck),
                pods_left=int(io.pods_left), total_ns=int(io.total_ns))


class ScheduleOperation:
    """core.go ScheduleOperation over one capi.Context (which holds the node snapshot)."""

    def __init__(self, ctx: capi.Context, max_schedule_time_s: float = 60.0):
        self._lib = load_host_library()
        self.ctx = ctx
        self.L = ctx.L
        self._h = C.c_void_p(self._lib.bsh_create(ctx._h, ctx.S, int(max_schedule_time_s * SECOND)))
        if not self._h:
            raise capi.BsError(-1, "bsh_create")

    def close(self):
        if self._h:
            self._lib.bsh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _lanes(self, req):
        a = np.zeros(soa.MAX_LANES, np.int64)
        if req is not None:
            a[: len(req)] = req
        return a, a.ctypes.data_as(C.POINTER(C.c_int64))

    def set_time(self, seconds: float):
        self._lib.bsh_set_time(self._h, int(round(seconds * SECOND)))

    @property
    def gpu_calls(self) -> int:
        return int(self._lib.bsh_gpu_calls(self._h))

    def add_group(self, min_member: int, status_scheduled: int = 0, max_schedule_time_s: float | None = None, creation_ts: int = 0,
                  name_rank: int = 0, min_resources=None, min_resources_present: int = 0) -> int:
        keep, ptr = self._lanes(min_resources)
        null = C.POINTER(C.c_int64)()
        mst = -1 if max_schedule_time_s is None else int(max_schedule_time_s * SECOND)
        return int(self._lib.bsh_add_group(self._h, min_member, status_scheduled, mst, creation_ts, name_rank,
                                           ptr if min_resources is not None else null, min_resources_present))

    def PreFilter(self, uid: int, name: int, group: int, req, present: int = 0, cls: int = 0, owner: int = 0):
        keep, ptr = self._lanes(req)
        fk = C.c_uint32(0)
        code = self._lib.bsh_prefilter(self._h, uid, name, group, ptr, present, cls, owner, C.byref(fk))
        if code < 0:
            raise capi.BsError(code, "bsh_prefilter")
        return int(code), int(fk.value)

    def Filter(self, uid: int, group: int, req, present: int, node: int):
        keep, ptr = self._lanes(req)
        fl, fn = C.c_uint8(0), C.c_uint8(0)
        rc = self._lib.bsh_filter(self._h, uid, group, ptr, present, node, C.byref(fl), C.byref(fn))
        if rc != 0:
            raise capi.BsError(rc, "bsh_filter")
        return int(fl.value), int(fn.value)

    def Permit(self, uid: int, name: int, group: int, node: int):
        ready = C.c_uint8(0)
        code = self._lib.bsh_permit(self._h, uid, name, group, node, C.byref(ready))
        return bool(ready.value), int(code)

    def sync(self):
        """push the PodGroup cache to the device (bs_groups_load) — the state a following bs_batch_run starts from"""
        rc = self._lib.bsh_sync(self._h)
        if rc != 0:
            raise capi.BsError(rc, "bsh_sync")

    def PostBind(self, group: int):
        self._lib.bsh_postbind(self._h, group)

    def Less(self, a, b) -> bool:
        """a, b = (group, priority, queue_timestamp) — batchscheduler.go:214 -> core.go:368"""
        return bool(self._lib.bsh_less(self._h, a[0], a[1], a[2], b[0], b[1], b[2]))

    def StartBatchSchedule(self, group: int, cap: int | None = None):
        """batchscheduler.go:254-344: releases the whole gang or nothing.  Buffers are sized from the group's
        matched count; an explicit `cap` that is too small releases nothing and raises."""
        if cap is None:
            cap = max(1, int(self._lib.bsh_group_matched(self._h, group))) if 0 <= group else 1
        uids = (C.c_uint64 * cap)()
        nodes = (C.c_uint32 * cap)()
        n = int(self._lib.bsh_start_batch(self._h, group, uids, nodes, cap))
        if n == 0xFFFFFFFF:
            raise ValueError(f"StartBatchSchedule: gang of group {group} does not fit cap={cap}; nothing was released")
        return [(int(uids[i]), int(nodes[i])) for i in range(n)]

    def group_state(self, g: int) -> dict:
        f = int(self._lib.bsh_group_flags(self._h, g))
        return dict(matched=int(self._lib.bsh_group_matched(self._h, g)), status_scheduled=int(self._lib.bsh_group_status_scheduled(self._h, g)),
                    scheduled_latch=bool(f & 1), has_pod=bool(f & 2), has_minres=bool(f & 4), phase=f >> 8,
                    denied=bool(self._lib.bsh_group_denied(self._h, g)))
