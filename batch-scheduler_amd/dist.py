"""Pod-axis sharding helpers (one process per GPU).

The device decides ownership itself (owner_rank_of / k_owner_starts in csrc/bs_kernels.hpp): whole groups — the anchor of a
grouped pod is its group's first pod in queue order, an ungrouped pod is its own anchor — balanced by POD COUNT: walking the
queue, every anchor carries the weight of what hangs on it (its group's pods, or 1) and the running weight is cut into
`nranks` equal shares.  Groups therefore never straddle ranks, the deny-cache replay stays rank-local, the per-group admit
counters of different ranks are disjoint — ONE all-reduce(sum) merges them — and no rank holds more than
its share plus one group, however the queue is ordered (cutting the queue POSITIONS of the anchors, the rule of rounds 1-2,
gave rank 0 nearly everything on a queue that is not gang-sorted).
`owner_ranks` is the host mirror of that rule (tests, load accounting, the partitioned mode's split).
"""
from __future__ import annotations

import numpy as np


def owner_ranks(group: np.ndarray, n_groups: int, nranks: int) -> np.ndarray:
    """rank that evaluates each pod; mirrors the device rule bit for bit."""
    group = np.asarray(group)
    p = len(group)
    idx = np.arange(p, dtype=np.int64)
    ng = max(n_groups, 1)
    first = np.full(ng, p, dtype=np.int64)
    valid = (group >= 0) & (group < n_groups)
    np.minimum.at(first, group[valid], idx[valid])
    count = np.bincount(group[valid], minlength=ng).astype(np.int64)
    weight = np.zeros(p, dtype=np.int64)
    weight[~valid] = 1
    has = first < p
    weight[first[has]] = count[has]
    start = np.cumsum(weight) - weight                   # pods owned before queue position i
    anchor = idx.copy()
    anchor[valid] = first[group[valid]]
    return ((start[anchor] * nranks) // max(p, 1)).astype(np.int64)


def all_reduce_admit(counts, dist_module=None):
    """Sum per-group admit counters over ranks, in place.  `counts` is a torch tensor (int32 view of
    the uint32 counters: two's-complement addition is the same bits) on the backend's device."""
    if dist_module is None:
        import torch.distributed as dist_module
    dist_module.all_reduce(counts)
    return counts


def first_reach_thresholds(pods, groups, ranks: np.ndarray, nranks: int) -> list:
    """Partitioned mode: for every rank the argument of bs_first_reach_hint — how many of ITS pods stand in front of the whole queue's
    first pod that reaches findMaxPG (core.go:118-123).  The reaching rule of a batch in which no first-pod capture can occur (the
    steady state; k_fast_query_tables states the same): labelled with a known group (:100-103), no lastPermittedPod entry (:95-98), no deny
    entry (:105-110), OccupiedBy agrees (:494-511: against the group's entry, or — the group has none yet — against the first pod of the
    group that brings owner references), and findMaxPG does not hit the uint32 division by zero of :716-717 (then nobody reaches).
    `ranks` = owner_ranks(...) of the whole queue; returns 0xFFFFFFFF for every rank when no pod reaches."""
    from . import soa
    p = pods.p
    group = np.asarray(pods.group)
    idx = np.arange(p, dtype=np.int64)
    G = groups.g
    grouped = (group >= 0) & (group < G)
    g = np.where(grouped, group, 0)
    cand = ((groups.flags & soa.GROUP_SCHEDULED_LATCH) == 0) & ((groups.flags & soa.GROUP_HAS_POD) != 0)
    panic = bool(np.any(cand & (groups.min_member == 0) & (groups.status_scheduled != 0)))
    none = [0xFFFFFFFF] * nranks
    if panic or p == 0 or G == 0:
        return none
    notperm = (pods.flags & soa.POD_LAST_PERMITTED) == 0
    denied = (groups.flags[g] & soa.GROUP_DENIED) != 0
    occ0 = groups.occupied_by[g]
    own = np.asarray(pods.owner)
    fo = np.full(G, p, dtype=np.int64)                     # first pod of the group (without a lastPermittedPod entry) that has owner references
    brings = grouped & notperm & (own != 0)
    np.minimum.at(fo, g[brings], idx[brings])
    fog = fo[g]
    need_fo = (occ0 == 0) & (fog < p) & (idx > fog)
    occ_err = np.where(occ0 != 0, (own == 0) | (own != occ0), need_fo & ((own == 0) | (own != own[np.minimum(fog, p - 1)])))
    reach = grouped & notperm & ~denied & ~occ_err
    if not reach.any():
        return none
    first = int(np.argmax(reach))
    return [int(np.count_nonzero(np.asarray(ranks)[:first] == r)) for r in range(nranks)]
