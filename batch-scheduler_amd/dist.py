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
