"""Filter's deny entry (core.go:183-185) inside the batch — CPU pin of the ORACLE.

BS_BATCH_FILTER_DENY makes bs_batch_run replay the deny entry a FAILING Filter writes (include/bsched.h; on the device:
csrc/bs_fdeny.hpp, compared with the C oracle's batch with the same flag by tests/test_gpu_filter_deny.py).  Here, without a GPU,
that oracle batch has to equal an independent, object-level SEQUENTIAL replay of the reference (oracle/naive_ref.py: PreFilter, then
Filter on every node, a failing node deny-lists the group) on random scenes — steady and positional, with denied groups,
OccupiedBy, permitted pods, nil nodes — EVERY scene, the pods let through on lastPermittedPod entries whose Filter fails in front of
their group's first-pod capture / the batch's first findMaxPG call included (round 3 set those aside: its host-side forward pass,
plugin.replay_filter_deny, could not do them; the pass is gone)."""
import copy
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import naive_ref as nv  # noqa: E402
from scenarios import random_objects  # noqa: E402


def sequential_with_filter(sc):
    """the reference's calls in queue order: PreFilter(pod); if it passes, Filter(pod, node) for every node; an error on some node
    -> AddToDenyCache(group) (core.go:183-185).  Object level, no SoA, no batch."""
    sc = copy.deepcopy(sc)
    sop = nv.ScheduleOperation(sc["nodes"], sc["cache"])
    sop.denied = set(sc["denied"])
    sop.permitted = set(sc["permitted"])
    gnames = list(sc["cache"].keys())
    N = len(sc["nodes"])
    codes, flc, feas = [], [], []
    admit = {nm: 0 for nm in gnames}
    for pod in sc["pods"]:
        code, _ = sop.prefilter(pod)
        codes.append(code)
        fl, f = nv.soa.FL_NOT_RUN, 0
        if code < 16:
            fl = nv.soa.FL_PASS_NOT_GROUPED
            failed = False
            for k in range(N):
                fl, fn = sop.filter_node(pod, k)
                ok = fl < 16 and (fl != nv.soa.FL_EVALUATED or fn < 16)
                f += ok
                failed = failed or (fl == nv.soa.FL_EVALUATED and fn >= 16)
            if N == 0:
                fl, _ = sop.filter_node(pod, 0)
            if failed:
                sop.denied.add(pod.group)
        flc.append(fl)
        feas.append(f)
        if pod.group in admit and code < 16 and f > 0:
            admit[pod.group] += 1
    return np.array(codes, np.uint8), np.array(flc, np.uint8), np.array(feas, np.uint32), np.array([admit[nm] for nm in gnames], np.uint32)


@pytest.mark.parametrize("steady", [False, True], ids=["positional", "steady"])
def test_oracle_batch_with_filter_deny_equals_the_sequential_replay_everywhere(steady, bsa, soa, orc):
    """the C oracle's batch writes the entry itself (oracle/bs_oracle.c orc_batch); no scene is set aside"""
    from test_gpu_parity import _force_class_mode
    corner = bites = 0
    for seed in range(7000, 7140):
        sc = random_objects(seed, n_nodes=6 + seed % 40, n_groups=7, n_pods=60, n_scalars=seed % 3, n_classes=3)
        if seed % 4 == 1:
            sc["permitted"] = set()
        nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
        if steady:
            rng = np.random.default_rng(seed)
            _force_class_mode(groups, rng, sc["n_classes"])
            groups.matched[:] = rng.integers(1, 4, groups.g)
            for gi, nm in enumerate(sc["cache"].keys()):
                pgs = sc["cache"][nm]
                pgs.matched = int(groups.matched[gi])
                if pgs.pod is None:
                    pgs.pod = nv.Pod(nm + "-rep", nm, {"cpu": 1}, cls=int(groups.cls[gi]))
                pgs.pod.cls = int(groups.cls[gi])
                lanes = ["cpu", "memory", "ephemeral-storage", "pods"] + sc["names"]
                mr = {}
                for j, key in enumerate(lanes):
                    if j < 4 or (int(groups.min_resources_present[gi]) >> (j - 4)) & 1:
                        mr[key] = int(groups.min_resources[j, gi])
                pgs.pod_group.min_resources = mr
        raw = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL, bitmap=False)
        out = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL | soa.BATCH_FILTER_DENY, bitmap=False)
        codes, flc, feas, admit = sequential_with_filter(sc)
        assert np.array_equal(out.pf_code, codes), f"seed {seed}: pf_code"
        assert np.array_equal(out.fl_code, flc), f"seed {seed}: fl_code"
        assert np.array_equal(out.fl_feasible, feas), f"seed {seed}: fl_feasible"
        assert np.array_equal(out.group_admit, admit), f"seed {seed}: group_admit"
        corner += int(((raw.pf_code == soa.PF_PASS_LAST_PERMITTED) & (raw.fl_code == soa.FL_EVALUATED) & (raw.fl_feasible < nodes.n)).any())
        bites += int(not np.array_equal(raw.pf_code, out.pf_code))
    assert corner >= 20 and bites >= 30, (corner, bites)
