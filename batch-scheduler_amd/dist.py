"""Pod-axis sharding helpers (one process per GPU).

The device decides ownership itself (k_query): a rank owns every pod of a group whose FIRST pod in
queue order falls into the rank's block of the queue, and ungrouped pods by their own index.  Groups
therefore never straddle ranks, the deny-cache replay stays rank-local, and the per-group admit
counters of different ranks are disjoint — ONE all-reduce(sum) merges them.
`owner_ranks` is the host mirror of that rule (tests, load accounting).
"""
from __future__ import annotations

import numpy as np


def owner_ranks(group: np.ndarray, n_groups: int, nranks: int) -> np.ndarray:
    """rank that evaluates each pod; mirrors k_query's rule bit for bit."""
    p = len(group)
    idx = np.arange(p, dtype=np.int64)
    first = np.full(max(n_groups, 1), p, dtype=np.int64)
    valid = (group >= 0) & (group < n_groups)
    np.minimum.at(first, group[valid], idx[valid])
    anchor = idx.copy()
    anchor[valid] = first[group[valid]]
    return ((anchor * nranks) // max(p, 1)).astype(np.int64)


def all_reduce_admit(counts, dist_module=None):
    """Sum per-group admit counters over ranks, in place.  `counts` is a torch tensor (int32 view of
    the uint32 counters: two's-complement addition is the same bits) on the backend's device."""
    if dist_module is None:
        import torch.distributed as dist_module
    dist_module.all_reduce(counts)
    return counts
