"""CPU: dist.first_reach_thresholds (the argument of bs_first_reach_hint, partitioned mode) against the oracle — the whole queue's first
pod whose PreFilter gets to findMaxPG (core.go:118-123) is the first pod whose stale-leader output is no longer the carried one."""
import importlib

import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects


@pytest.mark.parametrize("seed", range(200))
def test_first_reaching_pod_matches_the_oracle(seed, orc, soa):
    bdist = importlib.import_module("batch-scheduler_amd.dist")
    sc = random_objects(seed, n_groups=int(3 + seed % 5), n_pods=int(10 + seed % 30))
    for pgs in sc["cache"].values():                           # steady state: every group has its pod and its MinResources (no capture possible)
        if pgs.pod is None:
            pgs.pod = nv.Pod(pgs.pod_group.name + "-rep", pgs.pod_group.name, {"cpu": 500}, cls=0)
        if pgs.pod_group.min_resources is None:
            pgs.pod_group.min_resources = nv.pod_resource_require(pgs.pod).ResourceList()
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    out = sop.batch(pods, soa.STAGE_PREFILTER, bitmap=False)
    reached = ~np.isin(out.pf_code, [soa.PF_PASS_NOT_GROUPED, soa.PF_PASS_LAST_PERMITTED, soa.PF_ERR_PG_NOT_FOUND, soa.PF_ERR_DENIED, soa.PF_ERR_OCCUPIED, soa.PF_PANIC_DIV0])
    # a deny entry written INSIDE the batch turns later pods of the group away before findMaxPG — behind a reaching pod, so the FIRST one is unaffected
    first = int(np.argmax(reached)) if reached.any() else None
    for world in (1, 2, 3):
        own = bdist.owner_ranks(pods.group, groups.g, world)
        th = bdist.first_reach_thresholds(pods, groups, own, world)
        if first is None:
            assert th == [0xFFFFFFFF] * world
        else:
            assert th == [int(np.count_nonzero(own[:first] == r)) for r in range(world)], (seed, world, first)
