"""ctypes binding of the C ABI (include/bsched.h) implemented by libbsched.so (HIP, gfx950).

This is what a host language does at the boundary: hand flat SoA buffers in, get decision arrays
out.  No torch types cross it.  There is deliberately NO fallback: a missing library, a missing
GPU or a failing HIP call raises BsError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import fitspec, soa

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.environ.get("BS_LIB_DIR") or HERE, "libbsched.so")      # (BS_LIB_DIR: see build.py)

KERNEL_COUNT = 8
KERNEL_PREPASS, KERNEL_LEADER, KERNEL_QUERY, KERNEL_TABLES, KERNEL_SCAN, KERNEL_RESOLVE, KERNEL_FILTER, KERNEL_TALLY = range(8)

# every symbol include/bsched.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "bs_abi_version", "bs_strerror", "bs_last_error", "bs_create", "bs_destroy",
    "bs_nodes_load", "bs_fit_load", "bs_fit_build", "bs_fit_read", "bs_groups_load", "bs_groups_read", "bs_groups_apply", "bs_pods_map", "bs_pods_load",
    "bs_pods_apply", "bs_pods_count", "bs_pods_read", "bs_pods_apply_stats", "bs_filter_deny_stats", "bs_speculation_stats",
    "bs_nodes_apply", "bs_nodes_count", "bs_nodes_assume",
    "bs_cluster_fits", "bs_node_left", "bs_scan_prefix", "bs_cluster_total", "bs_filter_one", "bs_find_max_pg",
    "bs_batch_run", "bs_batch_sync", "bs_batch_read", "bs_batch_map", "bs_filter_rows_count", "bs_queue_order_load", "bs_queue_sort",
    "bs_shard_set", "bs_reduce_external", "bs_group_admit_devptr", "bs_group_admit_bind", "bs_stream", "bs_comm_unique_id", "bs_comm_init", "bs_batch_finish",
    "bs_timing_reset", "bs_timing_get", "bs_kernel_name", "bs_batch_stats_get",
    "bs_seq_run", "bs_nodes_read", "bs_first_reach_hint",
    "bs_nodes_load_flat", "bs_groups_load_flat", "bs_groups_read_flat", "bs_pods_load_flat", "bs_pods_apply_flat", "bs_pods_read_flat",
    "bs_batch_read_flat", "bs_seq_run_flat", "bs_fit_build_flat",
]


class BsError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("scalar_lanes", C.c_uint32),
                ("eph_gate", C.c_uint32), ("enable_timing", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class NodeDelta(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32),
                ("allocatable", C.c_int64 * soa.MAX_LANES), ("requested", C.c_int64 * soa.MAX_LANES),
                ("allocatable_present", C.c_uint32), ("requested_present", C.c_uint32), ("flags", C.c_uint32),
                ("fit_default", C.c_uint32), ("n_fit_exceptions", C.c_uint32), ("fit_exceptions", C.c_uint32 * 8)]


class NodeRequest(C.Structure):
    _fields_ = [("index", C.c_uint32), ("requested_present", C.c_uint32), ("requested", C.c_int64 * soa.MAX_LANES)]


DELTA_UPDATE, DELTA_APPEND, DELTA_REMOVE = 0, 1, 2


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double * KERNEL_COUNT), ("launches", C.c_uint64 * KERNEL_COUNT)]


class BatchStats(C.Structure):
    _fields_ = [("scan_queries", C.c_uint64), ("scan_rows_executed", C.c_uint64), ("scan_evals_executed", C.c_uint64),
                ("tables_built", C.c_uint64), ("logical_evals", C.c_uint64), ("filter_evals", C.c_uint64),
                ("filter_distinct", C.c_uint64), ("filter_evals_executed", C.c_uint64),
                ("scan_queries_logical", C.c_uint64), ("class_mode", C.c_uint64), ("fast_path", C.c_uint64), ("launches", C.c_uint64),
                ("chain", C.c_uint64), ("filter_lane_blocks", C.c_uint64), ("filter_tile_blocks", C.c_uint64)]


class SeqOut(C.Structure):
    """bs_seq_out"""
    _fields_ = [("pf_code", C.POINTER(C.c_uint8)), ("pf_first_k", C.POINTER(C.c_uint32)), ("pf_leader", C.POINTER(C.c_int32)),
                ("pod_node", C.POINTER(C.c_int32)), ("cap", C.c_uint32), ("released_group", C.POINTER(C.c_uint32)),
                ("released_pods", C.POINTER(C.c_uint32)), ("first_ns", C.POINTER(C.c_int64)), ("ready_ns", C.POINTER(C.c_int64)),
                ("n_released", C.c_uint32), ("total_ns", C.c_int64), ("node_picks", C.c_uint64), ("node_scans", C.c_uint64),
                ("scan_rounds", C.c_uint64), ("pick_rounds", C.c_uint64), ("leader_folds", C.c_uint64), ("table_builds", C.c_uint64),
                ("last_permitted", C.POINTER(C.c_uint8))]


_lib = None


def load_library(path: str | None = None):
    """dlopen libbsched.so and declare prototypes.  Raises if the HIP library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise BsError(-2, "load_library", f"{p} not built: run `python __graft_entry__.py build` (hipcc, gfx950)")
    L = C.CDLL(p)
    vp, u32, i32, u8 = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint8
    P = C.POINTER
    L.bs_abi_version.restype = u32
    L.bs_strerror.restype = C.c_char_p
    L.bs_strerror.argtypes = [C.c_int]
    L.bs_last_error.restype = C.c_char_p
    L.bs_last_error.argtypes = [vp]
    L.bs_kernel_name.restype = C.c_char_p
    L.bs_kernel_name.argtypes = [u32]
    L.bs_create.argtypes = [P(Config), P(vp)]
    L.bs_destroy.argtypes = [vp]
    L.bs_nodes_load.argtypes = [vp, P(soa.NodesStruct)]
    L.bs_fit_load.argtypes = [vp, u32, P(u32)]
    L.bs_fit_build.argtypes = [vp, P(fitspec.NodeLabelsStruct), P(fitspec.FitTemplatesStruct)]
    L.bs_fit_read.argtypes = [vp, P(u32)]
    L.bs_groups_load.argtypes = [vp, P(soa.GroupsStruct)]
    L.bs_groups_read.argtypes = [vp, P(soa.GroupsStruct)]
    L.bs_groups_apply.argtypes = [vp, P(soa.GroupDelta), u32]
    L.bs_filter_rows_count.argtypes = [vp, P(u32)]
    L.bs_pods_load.argtypes = [vp, P(soa.PodsStruct)]
    L.bs_pods_map.argtypes = [vp, u32, P(soa.PodsStruct)]
    L.bs_pods_apply.argtypes = [vp, P(soa.PodsDeltaStruct)]
    L.bs_pods_count.argtypes = [vp, P(u32)]
    L.bs_pods_apply_stats.argtypes = [vp, P(C.c_uint64), P(C.c_uint64)]
    L.bs_filter_deny_stats.argtypes = [vp, P(C.c_uint64)]
    L.bs_speculation_stats.argtypes = [vp, P(C.c_uint64), P(C.c_uint64)]
    L.bs_pods_read.argtypes = [vp, P(soa.PodsOutStruct)]
    L.bs_queue_order_load.argtypes = [vp, u32, P(u32)]
    L.bs_queue_sort.argtypes = [vp, u32, P(i32), P(i32), P(C.c_int64), P(u32)]
    L.bs_nodes_apply.argtypes = [vp, P(NodeDelta), u32]
    L.bs_nodes_count.argtypes = [vp, P(u32)]
    L.bs_nodes_assume.argtypes = [vp, P(NodeRequest), u32]
    L.bs_cluster_fits.argtypes = [vp, u32, C.c_float, P(C.c_int64), u32, P(u8), P(u32)]
    L.bs_node_left.argtypes = [vp, u32, C.c_float, P(C.c_int64), P(u32)]
    L.bs_scan_prefix.argtypes = [vp, u32, C.c_float, P(C.c_int64), P(u32), P(u32), P(u32)]
    L.bs_cluster_total.argtypes = [vp, u32, P(C.c_int64), P(u32)]
    L.bs_filter_one.argtypes = [vp, i32, P(C.c_int64), u32, i32, u32, P(u8), P(u8)]
    L.bs_find_max_pg.argtypes = [vp, P(i32), P(u32), P(u8)]
    L.bs_batch_run.argtypes = [vp, u32]
    L.bs_batch_sync.argtypes = [vp]
    L.bs_batch_read.argtypes = [vp, P(soa.BatchOutStruct)]
    L.bs_batch_map.argtypes = [vp, P(soa.BatchViewStruct)]
    L.bs_shard_set.argtypes = [vp, u32, u32]
    L.bs_group_admit_devptr.argtypes = [vp, P(vp), P(u32)]
    L.bs_reduce_external.argtypes = [vp, u32]
    L.bs_first_reach_hint.argtypes = [vp, u32]
    L.bs_group_admit_bind.argtypes = [vp, vp]
    L.bs_stream.argtypes = [vp, P(vp)]
    L.bs_comm_unique_id.argtypes = [P(u8)]
    L.bs_comm_init.argtypes = [vp, P(u8), u32, u32]
    L.bs_batch_finish.argtypes = [vp]
    L.bs_timing_reset.argtypes = [vp]
    L.bs_timing_get.argtypes = [vp, P(Timing)]
    L.bs_batch_stats_get.argtypes = [vp, P(BatchStats)]
    L.bs_seq_run.argtypes = [vp, u32, P(SeqOut)]
    L.bs_nodes_read.argtypes = [vp, P(C.c_int64), P(u32)]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    if path is None:
        _lib = L
    return L


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Context:
    """One bs_ctx: a HIP device, its stream and the resident snapshot / group / pod state."""

    def __init__(self, scalar_lanes: int = 0, eph_gate: int = 1, device: int = 0, enable_timing: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.S = scalar_lanes
        self.L = soa.FIXED_LANES + scalar_lanes
        cfg = Config(self._lib.bs_abi_version(), device, scalar_lanes, eph_gate, enable_timing)
        rc = self._lib.bs_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise BsError(rc, "bs_create", self._lib.bs_strerror(rc).decode())
        self.n = self.g = self.p = 0

    # -- plumbing
    def _chk(self, rc: int, where: str):
        if rc != 0:
            raise BsError(rc, where, self._lib.bs_strerror(rc).decode() + ": " + self._lib.bs_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._lib.bs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- loads
    def load_nodes(self, nodes: soa.Nodes, fit: soa.FitMasks | None = None):
        assert nodes.lanes == self.L, f"context has {self.L} lanes, nodes have {nodes.lanes}"
        st = nodes.as_struct()
        self._chk(self._lib.bs_nodes_load(self._h, C.byref(st)), "bs_nodes_load")
        self.n = nodes.n
        if fit is not None:
            self.load_fit(fit)

    def load_fit(self, fit: soa.FitMasks):
        assert fit.n == self.n
        bits = np.ascontiguousarray(fit.bits, dtype=np.uint32)
        if bits.size == 0:
            bits = np.zeros((fit.n_classes, 1), np.uint32)
        self._chk(self._lib.bs_fit_load(self._h, fit.n_classes, _u32p(bits)), "bs_fit_load")
        self.n_classes = fit.n_classes

    def build_fit(self, node_labels: "fitspec.NodeLabels", templates: "fitspec.FitTemplates"):
        """checkFit (core.go:741-759) for every (class, node) on the device; replaces load_fit."""
        assert node_labels.n == self.n
        ns, ts = node_labels.as_struct(), templates.as_struct()
        self._chk(self._lib.bs_fit_build(self._h, C.byref(ns), C.byref(ts)), "bs_fit_build")
        self.n_classes = templates.c

    def read_fit(self) -> soa.FitMasks:
        words = (self.n + 31) // 32
        bits = np.zeros(self.n_classes * words + 1, np.uint32)        # +1: never hand out a NULL pointer
        self._chk(self._lib.bs_fit_read(self._h, _u32p(bits)), "bs_fit_read")
        return soa.FitMasks(bits[:-1].reshape(self.n_classes, words).copy(), self.n)

    def load_groups(self, groups: soa.Groups):
        assert groups.min_resources.shape[0] == self.L
        st = groups.as_struct()
        self._chk(self._lib.bs_groups_load(self._h, C.byref(st)), "bs_groups_load")
        self.g = groups.g

    def read_groups(self) -> soa.Groups:
        out = soa.Groups.empty(self.g, self.L)
        st = out.as_struct()
        self._chk(self._lib.bs_groups_read(self._h, C.byref(st)), "bs_groups_read")
        return out

    def apply_group_deltas(self, deltas):
        """deltas: iterable of (index, matched, status_scheduled, flags) — bs_groups_apply."""
        deltas = list(deltas)
        arr = (soa.GroupDelta * max(len(deltas), 1))(*[soa.GroupDelta(*map(int, d)) for d in deltas])
        self._chk(self._lib.bs_groups_apply(self._h, arr, len(deltas)), "bs_groups_apply")

    def filter_rows_count(self) -> int:
        n = C.c_uint32(0)
        self._chk(self._lib.bs_filter_rows_count(self._h, C.byref(n)), "bs_filter_rows_count")
        return int(n.value)

    def load_queue_order(self, order_rank):
        """per group: dense rank of (CreationTimestamp ascending, group name descending) — bs_queue_order_load"""
        r = np.ascontiguousarray(order_rank, dtype=np.uint32)
        self._chk(self._lib.bs_queue_order_load(self._h, len(r), _u32p(r if len(r) else np.zeros(1, np.uint32))), "bs_queue_order_load")

    def queue_sort(self, priority, group, queue_ts) -> np.ndarray:
        """the queue order ScheduleOperation.Compare (core.go:368-411) defines: perm[k] = pod at queue position k"""
        pr, gr, ts = (np.ascontiguousarray(priority, np.int32), np.ascontiguousarray(group, np.int32), np.ascontiguousarray(queue_ts, np.int64))
        perm = np.zeros(max(len(pr), 1), np.uint32)
        self._chk(self._lib.bs_queue_sort(self._h, len(pr), pr.ctypes.data_as(C.POINTER(C.c_int32)), gr.ctypes.data_as(C.POINTER(C.c_int32)),
                                          _i64p(ts), _u32p(perm)), "bs_queue_sort")
        return perm[: len(pr)]

    def map_pods(self, p: int) -> soa.Pods:
        """Zero-copy hand-over (bs_pods_map): a Pods whose arrays ARE the library's pinned upload buffer.  Fill them in
        place and pass the object to load_pods — no packing copy.  Valid until the next map_pods / load_pods."""
        st = soa.PodsStruct()
        self._chk(self._lib.bs_pods_map(self._h, p, C.byref(st)), "bs_pods_map")

        def view(ptr, ctype, dtype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype * n)).contents).view(dtype).reshape(shape)
        pods = soa.Pods.__new__(soa.Pods)
        pods.group = view(st.group, C.c_int32, np.int32, (p,))
        pods.req = view(st.req, C.c_int64, np.int64, (self.L, p))
        pods.req_present = view(st.req_present, C.c_uint32, np.uint32, (p,))
        pods.cls = view(st.cls, C.c_uint32, np.uint32, (p,))
        pods.owner = view(st.owner, C.c_uint64, np.uint64, (p,))
        pods.flags = view(st.flags, C.c_uint8, np.uint8, (p,))
        return pods

    def apply_group_deltas_raw(self, arr, n: int):
        """bs_groups_apply with a prebuilt (GroupDelta * n) array (no per-call marshalling)"""
        self._chk(self._lib.bs_groups_apply(self._h, arr, n), "bs_groups_apply")

    def load_pods(self, pods: soa.Pods):
        assert pods.req.shape[0] == self.L
        st = pods.as_struct()
        self._chk(self._lib.bs_pods_load(self._h, C.byref(st)), "bs_pods_load")
        self.p = pods.p

    def apply_pods(self, remove=(), flag_index=(), flag_value=(), insert: soa.Pods | None = None, insert_at=None):
        """bs_pods_apply: patch the resident queue on the device (stable removals, flag updates, insertions)."""
        rem = np.ascontiguousarray(remove, np.uint32)
        fi = np.ascontiguousarray(flag_index, np.uint32)
        fv = np.ascontiguousarray(flag_value, np.uint8)
        assert fi.shape == fv.shape
        d = soa.PodsDeltaStruct()
        d.n_remove, d.remove = len(rem), _u32p(rem if len(rem) else np.zeros(1, np.uint32))
        d.n_flags, d.flag_index = len(fi), _u32p(fi if len(fi) else np.zeros(1, np.uint32))
        d.flag_value = (fv if len(fv) else np.zeros(1, np.uint8)).ctypes.data_as(C.POINTER(C.c_uint8))
        keep = [rem, fi, fv]
        if insert is not None and insert.p:
            assert insert.req.shape[0] == self.L
            d.insert = insert.as_struct()
            if insert_at is not None:
                at = np.ascontiguousarray(insert_at, np.uint32)
                assert len(at) == insert.p
                keep.append(at)
                d.insert_at = _u32p(at)
        self._chk(self._lib.bs_pods_apply(self._h, C.byref(d)), "bs_pods_apply")
        self.p = self.pods_count()

    def apply_pods_raw(self, delta: "soa.PodsDeltaStruct"):
        """bs_pods_apply with a prebuilt struct (no per-call marshalling); the caller tracks p"""
        self._chk(self._lib.bs_pods_apply(self._h, C.byref(delta)), "bs_pods_apply")

    def apply_stats(self) -> tuple[int, int]:
        a, r = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._lib.bs_pods_apply_stats(self._h, C.byref(a), C.byref(r)), "bs_pods_apply_stats")
        return int(a.value), int(r.value)

    def speculation_stats(self) -> tuple[int, int]:
        """(batches launched on a guessed findMaxPG answer, wrong guesses that were re-run) — bs_speculation_stats"""
        a, m = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._lib.bs_speculation_stats(self._h, C.byref(a), C.byref(m)), "bs_speculation_stats")
        return int(a.value), int(m.value)

    def filter_deny_reruns(self) -> int:
        """BS_BATCH_FILTER_DENY batches that had to be run again so far (fixed-point iteration, bs_filter_deny_stats)"""
        r = C.c_uint64(0)
        self._chk(self._lib.bs_filter_deny_stats(self._h, C.byref(r)), "bs_filter_deny_stats")
        return int(r.value)

    def pods_count(self) -> int:
        n = C.c_uint32(0)
        self._chk(self._lib.bs_pods_count(self._h, C.byref(n)), "bs_pods_count")
        return int(n.value)

    def read_pods(self) -> soa.Pods:
        """the resident queue (bs_pods_read)"""
        p = self.pods_count()
        out = soa.Pods.empty(max(p, 1), self.L)
        st = soa.PodsOutStruct(p, *[getattr(out.as_struct(), k) for k in ("group", "req", "req_present", "cls", "owner", "flags")])
        if p:
            # the [L][p] lane stride must be p: read into exactly-sized arrays
            out = soa.Pods.empty(p, self.L)
            s2 = out.as_struct()
            st = soa.PodsOutStruct(p, s2.group, s2.req, s2.req_present, s2.cls, s2.owner, s2.flags)
        self._chk(self._lib.bs_pods_read(self._h, C.byref(st)), "bs_pods_read")
        return out if p else soa.Pods.empty(0, self.L)

    def apply_node_deltas(self, deltas: list):
        arr = (NodeDelta * len(deltas))(*deltas)
        self._chk(self._lib.bs_nodes_apply(self._h, arr, len(deltas)), "bs_nodes_apply")
        n = C.c_uint32(0)
        self._chk(self._lib.bs_nodes_count(self._h, C.byref(n)), "bs_nodes_count")
        self.n = int(n.value)

    def assume_nodes(self, reqs):
        """bs_nodes_assume: reqs = iterable of (node index, requested lanes, requested_present)"""
        reqs = list(reqs)
        arr = (NodeRequest * max(len(reqs), 1))()
        for k, (idx, lanes, pres) in enumerate(reqs):
            arr[k].index, arr[k].requested_present = int(idx), int(pres)
            for j, v in enumerate(lanes):
                arr[k].requested[j] = int(v)
        self._chk(self._lib.bs_nodes_assume(self._h, arr, len(reqs)), "bs_nodes_assume")

    # -- single queries
    def cluster_fits(self, cls: int, pct: float, req, present: int = 0):
        r = np.zeros(soa.MAX_LANES, np.int64)
        r[: len(req)] = req
        fits, fk = C.c_uint8(0), C.c_uint32(0)
        self._chk(self._lib.bs_cluster_fits(self._h, cls, float(np.float32(pct)), _i64p(r), present, C.byref(fits), C.byref(fk)),
                  "bs_cluster_fits")
        return bool(fits.value), int(fk.value)

    def node_left(self, cls: int, pct: float):
        left = np.zeros((self.L, self.n), np.int64)
        present = np.zeros(max(self.n, 1), np.uint32)
        self._chk(self._lib.bs_node_left(self._h, cls, float(np.float32(pct)), _i64p(left), _u32p(present)), "bs_node_left")
        return left, present[: self.n]

    def scan_prefix(self, cls: int, pct: float):
        n = max(self.n, 1)
        prefix = np.zeros((self.L, n), np.int64)
        present, idx = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        rows = C.c_uint32(0)
        self._chk(self._lib.bs_scan_prefix(self._h, cls, float(np.float32(pct)), _i64p(prefix), _u32p(present), _u32p(idx), C.byref(rows)),
                  "bs_scan_prefix")
        r = int(rows.value)
        return prefix[:, :r].copy(), present[:r].copy(), idx[:r].copy()

    def cluster_total(self, cls: int):
        tot = np.zeros(soa.MAX_LANES, np.int64)
        pr = C.c_uint32(0)
        self._chk(self._lib.bs_cluster_total(self._h, cls, _i64p(tot), C.byref(pr)), "bs_cluster_total")
        return tot[: self.L].tolist(), int(pr.value)

    def filter_one(self, pod_group: int, req, present: int, leader: int, node: int):
        r = np.zeros(soa.MAX_LANES, np.int64)
        r[: len(req)] = req
        fl, fn = C.c_uint8(0), C.c_uint8(0)
        self._chk(self._lib.bs_filter_one(self._h, pod_group, _i64p(r), present, leader, node, C.byref(fl), C.byref(fn)), "bs_filter_one")
        return int(fl.value), int(fn.value)

    def find_max_pg(self):
        leader, fin, pan = C.c_int32(0), C.c_uint32(0), C.c_uint8(0)
        self._chk(self._lib.bs_find_max_pg(self._h, C.byref(leader), C.byref(fin), C.byref(pan)), "bs_find_max_pg")
        return int(leader.value), bool(pan.value)

    # -- batch
    def run(self, stages: int = soa.STAGE_ALL):
        self._chk(self._lib.bs_batch_run(self._h, stages), "bs_batch_run")

    def sync(self):
        self._chk(self._lib.bs_batch_sync(self._h), "bs_batch_sync")

    def finish(self):
        self._chk(self._lib.bs_batch_finish(self._h), "bs_batch_finish")

    def read(self, bitmap: bool = True, out: soa.BatchOut | None = None, rows: bool | None = None) -> soa.BatchOut:
        """Copy the results of the last batch to the host (into `out` when given: no allocation).
        rows: also the Filter slot rows (fl_rows / fl_slot; default: whenever the bitmap is asked for);
        bitmap: the expanded pods x nodes bitmap (opt-in on the library side)."""
        if rows is None:
            rows = bitmap
        if out is None:
            out = soa.BatchOut.alloc(self.p, self.g, self.n, bitmap=bitmap, rows_cap=max(self.filter_rows_count(), 1) if rows else 0)
        st = out.as_struct()
        self._chk(self._lib.bs_batch_read(self._h, C.byref(st)), "bs_batch_read")
        return out

    def map_raw(self, view: soa.BatchViewStruct | None = None) -> soa.BatchViewStruct:
        """bs_batch_map: wait for the latency-mode batch and return the pointer view (no copy of any kind)."""
        if view is None:
            view = soa.BatchViewStruct()
        self._chk(self._lib.bs_batch_map(self._h, C.byref(view)), "bs_batch_map")
        return view

    def map_results(self) -> dict:
        """bs_batch_map as numpy views over the pinned result memory (valid until the next run)."""
        v = self.map_raw()
        arr = np.ctypeslib.as_array

        def a(ptr, n):
            return arr(ptr, shape=(n,)) if n and ptr else None
        out = {"p": v.p, "g": v.g, "pf_code": a(v.pf_code, v.p), "pf_first_k": a(v.pf_first_k, v.p), "pf_leader": a(v.pf_leader, v.p),
               "fl_code": a(v.fl_code, v.p), "fl_feasible": a(v.fl_feasible, v.p), "fl_slot": a(v.fl_slot, v.p),
               "group_admit": a(v.group_admit, v.g), "group_ready": a(v.group_ready, v.g), "fl_rows_n": v.fl_rows_n, "fl_rows": None, "fl_rows_feasible": None}
        if v.fl_rows and v.fl_rows_n:
            out["fl_rows"] = arr(v.fl_rows, shape=(v.words, v.fl_rows_stride))[:, : v.fl_rows_n]
            out["fl_rows_feasible"] = arr(v.fl_rows_feasible, shape=(v.fl_rows_n,))
        return out

    def batch(self, stages: int = soa.STAGE_ALL, bitmap: bool = True, rows: bool | None = None) -> soa.BatchOut:
        self.run(stages)
        return self.read(bitmap=bitmap, rows=rows)

    # -- the sequential pass
    def seq_run(self, stages: int = soa.STAGE_PREFILTER, cap: int | None = None) -> dict:
        """bs_seq_run: the reference's pod-by-pod cycle (PreFilter -> node choice -> assume -> Permit -> release) over the
        resident queue, on the device.  The context's node requests and group state are what the pass left."""
        p, cap = self.pods_count(), max(self.g if cap is None else cap, 1)
        n = max(p, 1)
        pf, fk = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        ld, node = np.zeros(n, np.int32), np.full(n, -1, np.int32)
        rg, rp = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        t0, t1 = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        lp = np.zeros(n, np.uint8)
        o = SeqOut(pf.ctypes.data_as(C.POINTER(C.c_uint8)), _u32p(fk), ld.ctypes.data_as(C.POINTER(C.c_int32)), node.ctypes.data_as(C.POINTER(C.c_int32)),
                   cap, _u32p(rg), _u32p(rp), _i64p(t0), _i64p(t1), 0, 0, 0, 0, 0, 0, 0, 0, lp.ctypes.data_as(C.POINTER(C.c_uint8)))
        self._chk(self._lib.bs_seq_run(self._h, stages, C.byref(o)), "bs_seq_run")
        k = min(int(o.n_released), cap)
        return dict(pf_code=pf[:p], pf_first_k=fk[:p], pf_leader=ld[:p], pod_node=node[:p], last_permitted=lp[:p], released_group=rg[:k], released_pods=rp[:k],
                    first_ns=t0[:k], ready_ns=t1[:k], n_released=int(o.n_released), total_ns=int(o.total_ns), node_picks=int(o.node_picks),
                    node_scans=int(o.node_scans), scan_rounds=int(o.scan_rounds), pick_rounds=int(o.pick_rounds), leader_folds=int(o.leader_folds), table_builds=int(o.table_builds))

    def read_node_requests(self):
        """bs_nodes_read: (requested [L][n], requested_present [n]) as the context holds them"""
        req = np.zeros((self.L, max(self.n, 1)), np.int64) if self.n == 0 else np.zeros((self.L, self.n), np.int64)
        pres = np.zeros(max(self.n, 1), np.uint32)
        self._chk(self._lib.bs_nodes_read(self._h, _i64p(req), _u32p(pres)), "bs_nodes_read")
        return req[:, : self.n], pres[: self.n]

    # -- sharding / measurement
    def set_shard(self, rank: int, nranks: int):
        self._chk(self._lib.bs_shard_set(self._h, rank, nranks), "bs_shard_set")

    def reduce_external(self, on: bool = True):
        self._chk(self._lib.bs_reduce_external(self._h, 1 if on else 0), "bs_reduce_external")

    def first_reach_hint(self, local_index: int):
        """bs_first_reach_hint: partitioned mode, this rank's pods in front of the job's first pod that reaches findMaxPG"""
        self._chk(self._lib.bs_first_reach_hint(self._h, int(local_index) & 0xFFFFFFFF), "bs_first_reach_hint")

    def admit_devptr(self):
        p, n = C.c_void_p(), C.c_uint32(0)
        self._chk(self._lib.bs_group_admit_devptr(self._h, C.byref(p), C.byref(n)), "bs_group_admit_devptr")
        return int(p.value or 0), int(n.value)

    def bind_admit(self, dptr: int | None):
        self._chk(self._lib.bs_group_admit_bind(self._h, C.c_void_p(dptr or 0)), "bs_group_admit_bind")

    def stream(self) -> int:
        p = C.c_void_p()
        self._chk(self._lib.bs_stream(self._h, C.byref(p)), "bs_stream")
        return int(p.value or 0)

    def comm_init(self, uid: bytes, rank: int, nranks: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self._lib.bs_comm_init(self._h, buf, rank, nranks), "bs_comm_init")

    def timing_reset(self):
        self._chk(self._lib.bs_timing_reset(self._h), "bs_timing_reset")

    def timing(self) -> dict:
        t = Timing()
        self._chk(self._lib.bs_timing_get(self._h, C.byref(t)), "bs_timing_get")
        return {self._lib.bs_kernel_name(i).decode(): (float(t.total_ms[i]), int(t.launches[i])) for i in range(KERNEL_COUNT)}

    def stats_arm(self):
        s = BatchStats()
        self._chk(self._lib.bs_batch_stats_get(self._h, C.byref(s)), "bs_batch_stats_get")

    def stats(self, stages: int = soa.STAGE_ALL) -> dict:
        """Work counters of one instrumented batch (arm, run, read)."""
        self.stats_arm()
        self.run(stages)
        self.sync()
        return self.stats_read()

    def stats_read(self) -> dict:
        s = BatchStats()
        self._chk(self._lib.bs_batch_stats_get(self._h, C.byref(s)), "bs_batch_stats_get")
        return {k: int(getattr(s, k)) for k, _ in BatchStats._fields_}


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = (C.c_uint8 * 128)()
    rc = lib.bs_comm_unique_id(buf)
    if rc != 0:
        raise BsError(rc, "bs_comm_unique_id")
    return bytes(buf)
