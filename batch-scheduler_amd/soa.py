"""Flat SoA containers that cross the C ABI of include/bsched.h.

Each class owns contiguous numpy arrays laid out exactly as the header documents
(lane-major: ``x[lane, index]``) and can hand out the matching ctypes struct.  They carry the
reference's state flattened to int64 lanes:

* ``Nodes``  — the scheduler's NodeInfo snapshot in list order (core.go:597),
* ``Groups`` — cache.PodGroupMatchStatus + PodGroup spec/status fields (cache.go:52-67,
  types.go:79-130) in the iteration order findMaxPG is to use (core.go:703),
* ``Pods``   — pending pods in queue order with getPodResourceRequire lanes (core.go:761-772).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

FIXED_LANES = 4
MAX_SCALARS = 12
MAX_LANES = 16
LANE_CPU, LANE_MEM, LANE_EPH, LANE_PODS = 0, 1, 2, 3

NODE_NIL, NODE_NO_NODE, NODE_UNSCHEDULABLE, NODE_TAINT_ERR = 0x01, 0x02, 0x04, 0x08
NODE_SKIP_MASK = 0x07
GROUP_SCHEDULED_LATCH, GROUP_HAS_POD, GROUP_HAS_MINRES, GROUP_DENIED, GROUP_PHASE_CLOSED = 0x01, 0x02, 0x04, 0x08, 0x10
POD_LAST_PERMITTED = 0x01
POD_NOT_GROUPED, POD_GROUP_MISSING = -1, -2

PF_PASS_NOT_GROUPED, PF_PASS_LAST_PERMITTED, PF_PASS_NO_MAX = 0, 1, 2
PF_PASS_FIRST_FITS, PF_PASS_IS_MAX, PF_PASS_RESERVE_FITS = 3, 4, 5
PF_ERR_PG_NOT_FOUND, PF_ERR_DENIED, PF_ERR_OCCUPIED = 16, 17, 18
PF_REJECT_FIRST, PF_REJECT_RESERVE, PF_PANIC_DIV0 = 19, 20, 32
FL_PASS_NOT_GROUPED, FL_PASS_IS_MAX, FL_PASS_NO_MINRES, FL_EVALUATED = 0, 1, 2, 3
FL_ERR_PG_NOT_FOUND, FL_PANIC_NIL_MAX, FL_NOT_RUN = 16, 32, 64
FN_PASS_CASE2, FN_PASS_CASE3, FN_ERR_NOT_ENOUGH, FN_ERR_SNAPSHOT = 0, 1, 16, 17
K_NONE, K_NOT_SCANNED = 0xFFFFFFFF, 0xFFFFFFFE
STAGE_PREFILTER, STAGE_FILTER, STAGE_TALLY, STAGE_ALL, BATCH_COMMIT, BATCH_HOST_RESULTS = 1, 2, 4, 7, 0x100, 0x200
BATCH_FILTER_DENY = 0x400        # Filter's deny entry (core.go:183-185) replayed inside the batch

PF_NAMES = {0: "PASS_NOT_GROUPED", 1: "PASS_LAST_PERMITTED", 2: "PASS_NO_MAX", 3: "PASS_FIRST_FITS",
            4: "PASS_IS_MAX", 5: "PASS_RESERVE_FITS", 16: "ERR_PG_NOT_FOUND", 17: "ERR_DENIED",
            18: "ERR_OCCUPIED", 19: "REJECT_FIRST", 20: "REJECT_RESERVE", 32: "PANIC_DIV0"}


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _arr(x, dtype, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


class NodesStruct(C.Structure):
    _fields_ = [("n", C.c_uint32),
                ("allocatable", C.POINTER(C.c_int64)),
                ("requested", C.POINTER(C.c_int64)),
                ("allocatable_present", C.POINTER(C.c_uint32)),
                ("requested_present", C.POINTER(C.c_uint32)),
                ("flags", C.POINTER(C.c_uint8))]


class GroupsStruct(C.Structure):
    _fields_ = [("g", C.c_uint32),
                ("min_member", C.POINTER(C.c_uint32)),
                ("status_scheduled", C.POINTER(C.c_uint32)),
                ("matched", C.POINTER(C.c_uint32)),
                ("flags", C.POINTER(C.c_uint8)),
                ("cls", C.POINTER(C.c_uint32)),
                ("min_resources", C.POINTER(C.c_int64)),
                ("min_resources_present", C.POINTER(C.c_uint32)),
                ("occupied_by", C.POINTER(C.c_uint64))]


class PodsStruct(C.Structure):
    _fields_ = [("p", C.c_uint32),
                ("group", C.POINTER(C.c_int32)),
                ("req", C.POINTER(C.c_int64)),
                ("req_present", C.POINTER(C.c_uint32)),
                ("cls", C.POINTER(C.c_uint32)),
                ("owner", C.POINTER(C.c_uint64)),
                ("flags", C.POINTER(C.c_uint8))]


class BatchOutStruct(C.Structure):
    _fields_ = [("pf_code", C.POINTER(C.c_uint8)),
                ("pf_first_k", C.POINTER(C.c_uint32)),
                ("pf_leader", C.POINTER(C.c_int32)),
                ("fl_code", C.POINTER(C.c_uint8)),
                ("fl_feasible", C.POINTER(C.c_uint32)),
                ("fl_bitmap", C.POINTER(C.c_uint64)),
                ("group_admit", C.POINTER(C.c_uint32)),
                ("group_ready", C.POINTER(C.c_uint8)),
                ("fl_slot", C.POINTER(C.c_uint32)),
                ("fl_rows", C.POINTER(C.c_uint64)),
                ("fl_rows_feasible", C.POINTER(C.c_uint32)),
                ("fl_rows_cap", C.c_uint32),
                ("fl_rows_n", C.POINTER(C.c_uint32))]


class BatchViewStruct(C.Structure):
    """bs_batch_view: read-only pointers into the pinned result memory of a latency-mode batch (bs_batch_map)."""
    _fields_ = [("p", C.c_uint32), ("g", C.c_uint32), ("words", C.c_uint32),
                ("pf_code", C.POINTER(C.c_uint8)),
                ("pf_first_k", C.POINTER(C.c_uint32)),
                ("pf_leader", C.POINTER(C.c_int32)),
                ("fl_code", C.POINTER(C.c_uint8)),
                ("fl_feasible", C.POINTER(C.c_uint32)),
                ("fl_slot", C.POINTER(C.c_uint32)),
                ("group_admit", C.POINTER(C.c_uint32)),
                ("group_ready", C.POINTER(C.c_uint8)),
                ("fl_rows", C.POINTER(C.c_uint64)),
                ("fl_rows_feasible", C.POINTER(C.c_uint32)),
                ("fl_rows_stride", C.c_uint32), ("fl_rows_n", C.c_uint32)]


class PodsDeltaStruct(C.Structure):
    """bs_pods_delta: stable removals, flag updates and insertions against the resident queue (bs_pods_apply)."""
    _fields_ = [("n_remove", C.c_uint32), ("remove", C.POINTER(C.c_uint32)),
                ("n_flags", C.c_uint32), ("flag_index", C.POINTER(C.c_uint32)), ("flag_value", C.POINTER(C.c_uint8)),
                ("insert", PodsStruct), ("insert_at", C.POINTER(C.c_uint32))]


class PodsOutStruct(C.Structure):
    _fields_ = [("p", C.c_uint32),
                ("group", C.POINTER(C.c_int32)),
                ("req", C.POINTER(C.c_int64)),
                ("req_present", C.POINTER(C.c_uint32)),
                ("cls", C.POINTER(C.c_uint32)),
                ("owner", C.POINTER(C.c_uint64)),
                ("flags", C.POINTER(C.c_uint8))]


class GroupDelta(C.Structure):
    """bs_group_delta: replaces matched / status_scheduled / flags of group `index` (bs_groups_apply)."""
    _fields_ = [("index", C.c_uint32), ("matched", C.c_uint32), ("status_scheduled", C.c_uint32), ("flags", C.c_uint32)]


@dataclass
class Nodes:
    """NodeInfo snapshot.  allocatable/requested are [L, n] int64."""
    allocatable: np.ndarray
    requested: np.ndarray
    allocatable_present: np.ndarray
    requested_present: np.ndarray
    flags: np.ndarray

    def __post_init__(self):
        self.allocatable = _arr(self.allocatable, np.int64)
        self.requested = _arr(self.requested, np.int64)
        L, n = self.allocatable.shape
        assert self.requested.shape == (L, n)
        self.allocatable_present = _arr(self.allocatable_present, np.uint32, (n,))
        self.requested_present = _arr(self.requested_present, np.uint32, (n,))
        self.flags = _arr(self.flags, np.uint8, (n,))

    @property
    def n(self) -> int:
        return self.allocatable.shape[1]

    @property
    def lanes(self) -> int:
        return self.allocatable.shape[0]

    def as_struct(self) -> NodesStruct:
        return NodesStruct(self.n, _ptr(self.allocatable, C.c_int64), _ptr(self.requested, C.c_int64),
                           _ptr(self.allocatable_present, C.c_uint32), _ptr(self.requested_present, C.c_uint32),
                           _ptr(self.flags, C.c_uint8))

    def copy(self) -> "Nodes":
        return Nodes(self.allocatable.copy(), self.requested.copy(), self.allocatable_present.copy(),
                     self.requested_present.copy(), self.flags.copy())


@dataclass
class FitMasks:
    """fit[c, n] = checkFit(rep pod of class c, node n) (core.go:741-759), packed 32 nodes/word."""
    bits: np.ndarray  # [C, ceil(n/32)] uint32
    n: int

    @staticmethod
    def from_bool(fit: np.ndarray) -> "FitMasks":
        fit = np.asarray(fit, dtype=bool)
        c, n = fit.shape
        words = (n + 31) // 32
        padded = np.zeros((c, words * 32), dtype=bool)
        padded[:, :n] = fit
        w = (padded.reshape(c, words, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2)
        return FitMasks(np.ascontiguousarray(w.astype(np.uint32)).reshape(c, words), n)

    def to_bool(self) -> np.ndarray:
        c, words = self.bits.shape
        b = (self.bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1
        return b.reshape(c, words * 32)[:, : self.n].astype(bool)

    @property
    def n_classes(self) -> int:
        return self.bits.shape[0]


@dataclass
class Groups:
    min_member: np.ndarray
    status_scheduled: np.ndarray
    matched: np.ndarray
    flags: np.ndarray
    cls: np.ndarray
    min_resources: np.ndarray          # [L, g]
    min_resources_present: np.ndarray
    occupied_by: np.ndarray

    def __post_init__(self):
        self.min_member = _arr(self.min_member, np.uint32)
        g = self.min_member.shape[0]
        self.status_scheduled = _arr(self.status_scheduled, np.uint32, (g,))
        self.matched = _arr(self.matched, np.uint32, (g,))
        self.flags = _arr(self.flags, np.uint8, (g,))
        self.cls = _arr(self.cls, np.uint32, (g,))
        self.min_resources = _arr(self.min_resources, np.int64)
        assert self.min_resources.shape[1] == g
        self.min_resources_present = _arr(self.min_resources_present, np.uint32, (g,))
        self.occupied_by = _arr(self.occupied_by, np.uint64, (g,))

    @property
    def g(self) -> int:
        return self.min_member.shape[0]

    def as_struct(self) -> GroupsStruct:
        return GroupsStruct(self.g, _ptr(self.min_member, C.c_uint32), _ptr(self.status_scheduled, C.c_uint32),
                            _ptr(self.matched, C.c_uint32), _ptr(self.flags, C.c_uint8), _ptr(self.cls, C.c_uint32),
                            _ptr(self.min_resources, C.c_int64), _ptr(self.min_resources_present, C.c_uint32),
                            _ptr(self.occupied_by, C.c_uint64))

    def copy(self) -> "Groups":
        return Groups(self.min_member.copy(), self.status_scheduled.copy(), self.matched.copy(), self.flags.copy(),
                      self.cls.copy(), self.min_resources.copy(), self.min_resources_present.copy(),
                      self.occupied_by.copy())

    @staticmethod
    def empty(g: int, lanes: int) -> "Groups":
        return Groups(np.zeros(g, np.uint32), np.zeros(g, np.uint32), np.zeros(g, np.uint32), np.zeros(g, np.uint8),
                      np.zeros(g, np.uint32), np.zeros((lanes, g), np.int64), np.zeros(g, np.uint32),
                      np.zeros(g, np.uint64))

    def state_equal(self, other: "Groups") -> bool:
        """Equality of the observable state (cls/min_resources only where the flag says valid)."""
        if not (np.array_equal(self.flags, other.flags) and np.array_equal(self.occupied_by, other.occupied_by)
                and np.array_equal(self.min_member, other.min_member)
                and np.array_equal(self.status_scheduled, other.status_scheduled)
                and np.array_equal(self.matched, other.matched)):
            return False
        hp = (self.flags & GROUP_HAS_POD) != 0
        hm = (self.flags & GROUP_HAS_MINRES) != 0
        return (np.array_equal(self.cls[hp], other.cls[hp])
                and np.array_equal(self.min_resources[:, hm], other.min_resources[:, hm])
                and np.array_equal(self.min_resources_present[hm], other.min_resources_present[hm]))


@dataclass
class Pods:
    group: np.ndarray
    req: np.ndarray            # [L, p]
    req_present: np.ndarray
    cls: np.ndarray
    owner: np.ndarray
    flags: np.ndarray

    def __post_init__(self):
        self.group = _arr(self.group, np.int32)
        p = self.group.shape[0]
        self.req = _arr(self.req, np.int64)
        assert self.req.shape[1] == p
        self.req_present = _arr(self.req_present, np.uint32, (p,))
        self.cls = _arr(self.cls, np.uint32, (p,))
        self.owner = _arr(self.owner, np.uint64, (p,))
        self.flags = _arr(self.flags, np.uint8, (p,))

    @property
    def p(self) -> int:
        return self.group.shape[0]

    def as_struct(self) -> PodsStruct:
        return PodsStruct(self.p, _ptr(self.group, C.c_int32), _ptr(self.req, C.c_int64),
                          _ptr(self.req_present, C.c_uint32), _ptr(self.cls, C.c_uint32),
                          _ptr(self.owner, C.c_uint64), _ptr(self.flags, C.c_uint8))

    def take(self, idx) -> "Pods":
        """the sub-batch of pods `idx` (queue order preserved when idx is increasing)"""
        idx = np.asarray(idx)
        return Pods(self.group[idx], self.req[:, idx], self.req_present[idx], self.cls[idx], self.owner[idx], self.flags[idx])

    def copy(self) -> "Pods":
        return Pods(self.group.copy(), self.req.copy(), self.req_present.copy(), self.cls.copy(),
                    self.owner.copy(), self.flags.copy())

    @staticmethod
    def empty(p: int, lanes: int) -> "Pods":
        return Pods(np.zeros(p, np.int32), np.zeros((lanes, p), np.int64), np.zeros(p, np.uint32), np.zeros(p, np.uint32),
                    np.zeros(p, np.uint64), np.zeros(p, np.uint8))

    def equal(self, other: "Pods") -> bool:
        return all(np.array_equal(getattr(self, k), getattr(other, k)) for k in ("group", "req", "req_present", "cls", "owner", "flags"))

    def patched(self, remove=(), flag_index=(), flag_value=(), insert: "Pods | None" = None, insert_at=None) -> "Pods":
        """The queue bs_pods_apply leaves behind, computed on the host (the specification of the delta): flag updates and
        stable removals against THIS queue, then the inserted pods at their positions in the NEW queue (None = appended)."""
        fl = self.flags.copy()
        if len(flag_index):
            fl[np.asarray(flag_index, np.int64)] = np.asarray(flag_value, np.uint8)
        keep = np.ones(self.p, bool)
        if len(remove):
            keep[np.asarray(remove, np.int64)] = False
        kept = Pods(self.group[keep], self.req[:, keep], self.req_present[keep], self.cls[keep], self.owner[keep], fl[keep])
        ni = insert.p if insert is not None else 0
        if not ni:
            return kept
        pn = kept.p + ni
        at = np.arange(kept.p, pn) if insert_at is None else np.asarray(insert_at, np.int64)
        is_ins = np.zeros(pn, bool)
        is_ins[at] = True
        out = Pods.empty(pn, self.req.shape[0])
        for k in ("group", "req_present", "cls", "owner", "flags"):
            getattr(out, k)[is_ins] = getattr(insert, k)
            getattr(out, k)[~is_ins] = getattr(kept, k)
        out.req[:, is_ins] = insert.req
        out.req[:, ~is_ins] = kept.req
        return out


@dataclass
class BatchOut:
    """Host-side result arrays of one batch.

    Filter results come as rows of the DISTINCT requests (`fl_rows` [words, rows_cap], `fl_slot` [p]) — the
    pods x nodes bitmap `fl_bitmap` is opt-in (the library materialises it only when asked)."""
    pf_code: np.ndarray
    pf_first_k: np.ndarray
    pf_leader: np.ndarray
    fl_code: np.ndarray
    fl_feasible: np.ndarray
    fl_bitmap: np.ndarray | None
    group_admit: np.ndarray
    group_ready: np.ndarray
    fl_slot: np.ndarray | None = None
    fl_rows: np.ndarray | None = None          # [ceil(n/64), rows_cap] uint64
    fl_rows_feasible: np.ndarray | None = None
    fl_rows_n: np.ndarray | None = None        # [1] uint32, filled by the library
    n: int = 0
    _keep: list = field(default_factory=list, repr=False)

    @staticmethod
    def alloc(p: int, g: int, n: int, bitmap: bool = True, rows_cap: int = 0) -> "BatchOut":
        w = (n + 63) // 64
        out = BatchOut(np.zeros(p, np.uint8), np.zeros(p, np.uint32), np.zeros(p, np.int32), np.zeros(p, np.uint8),
                       np.zeros(p, np.uint32), np.zeros((w, p), np.uint64) if bitmap else None,
                       np.zeros(g, np.uint32), np.zeros(g, np.uint8), n=n)
        if rows_cap:
            out.fl_slot = np.zeros(p, np.uint32)
            out.fl_rows = np.zeros((w, rows_cap), np.uint64)
            out.fl_rows_feasible = np.zeros(rows_cap, np.uint32)
            out.fl_rows_n = np.zeros(1, np.uint32)
        return out

    def as_struct(self) -> BatchOutStruct:
        null64, null32 = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
        rows = self.fl_rows is not None
        return BatchOutStruct(_ptr(self.pf_code, C.c_uint8), _ptr(self.pf_first_k, C.c_uint32),
                              _ptr(self.pf_leader, C.c_int32), _ptr(self.fl_code, C.c_uint8),
                              _ptr(self.fl_feasible, C.c_uint32),
                              _ptr(self.fl_bitmap, C.c_uint64) if self.fl_bitmap is not None else null64,
                              _ptr(self.group_admit, C.c_uint32), _ptr(self.group_ready, C.c_uint8),
                              _ptr(self.fl_slot, C.c_uint32) if rows else null32,
                              _ptr(self.fl_rows, C.c_uint64) if rows else null64,
                              _ptr(self.fl_rows_feasible, C.c_uint32) if rows else null32,
                              self.fl_rows.shape[1] if rows else 0,
                              _ptr(self.fl_rows_n, C.c_uint32) if rows else null32)

    def node_passes(self, pod: int, node: int) -> bool:
        """Filter(pod, node) from the slot rows: the bit test the Go plugin's Filter does (no cgo crossing)."""
        if self.fl_rows is None:
            return bool((int(self.fl_bitmap[node >> 6, pod]) >> (node & 63)) & 1)
        fl = int(self.fl_code[pod])
        if fl != FL_EVALUATED:
            return fl < 16 and node < self.n
        return bool((int(self.fl_rows[node >> 6, int(self.fl_slot[pod])]) >> (node & 63)) & 1)

    def bitmap_from_rows(self) -> np.ndarray:
        """[ceil(n/64), p] bitmap rebuilt on the host from the slot rows (tests: must equal the expanded one)."""
        w, p = self.fl_rows.shape[0], self.fl_code.shape[0]
        out = np.zeros((w, p), np.uint64)
        ev = self.fl_code == FL_EVALUATED
        out[:, ev] = self.fl_rows[:, self.fl_slot[ev]]
        full = np.full(w, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
        if self.n & 63 and w:
            full[-1] = np.uint64((1 << (self.n & 63)) - 1)
        out[:, (~ev) & (self.fl_code < 16)] = full[:, None]
        return out
