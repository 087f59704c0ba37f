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


def mechanical_pass_and_needed_pods(raw, pods, groups, n_nodes, sop_leader0=-1):
    """What the device does behind a chain (csrc/bs_fdeny.hpp k_fd_apply), restated with numpy loops: per group the first pod whose
    Filter fails on a node; every later pod of the group that reaches the deny check is turned away; and the test for "a pod that is
    turned away was needed by somebody else" (the batch then goes to the fixed-point re-runs): (1) it got to findMaxPG and its
    stale-leader value differs from its predecessor's, (2) it was its group's first eligible pod and the group lacked its pod or
    its MinResources."""
    import copy
    out = copy.copy(raw)
    out.pf_code, out.pf_first_k = raw.pf_code.copy(), raw.pf_first_k.copy()
    out.fl_code, out.fl_feasible = raw.fl_code.copy(), raw.fl_feasible.copy()
    event, first_np, needed = {}, {}, False
    for i in range(pods.p):
        g = int(pods.group[i])
        if 0 <= g < groups.g and not (pods.flags[i] & nv.soa.POD_LAST_PERMITTED):
            first_np.setdefault(g, i)
        if 0 <= g < groups.g and raw.fl_code[i] == nv.soa.FL_EVALUATED and raw.fl_feasible[i] < n_nodes:
            event.setdefault(g, i)
    for i in range(pods.p):
        g, code = int(pods.group[i]), int(raw.pf_code[i])
        at_check = 0 <= g < groups.g and code not in (nv.soa.PF_PASS_NOT_GROUPED, nv.soa.PF_PASS_LAST_PERMITTED, nv.soa.PF_ERR_PG_NOT_FOUND)
        if not (at_check and g in event and event[g] < i and code != nv.soa.PF_ERR_DENIED):
            continue
        if code != nv.soa.PF_ERR_OCCUPIED and int(raw.pf_leader[i]) != (int(raw.pf_leader[i - 1]) if i else sop_leader0):
            needed = True
        both = nv.soa.GROUP_HAS_POD | nv.soa.GROUP_HAS_MINRES
        if first_np.get(g) == i and (int(groups.flags[g]) & both) != both:
            needed = True
        out.pf_code[i], out.pf_first_k[i], out.fl_code[i], out.fl_feasible[i] = nv.soa.PF_ERR_DENIED, 0xFFFFFFFE, nv.soa.FL_NOT_RUN, 0
    return out, needed


def test_the_needed_pod_rule_flags_every_scene_the_mechanical_pass_gets_wrong(bsa, soa, orc):
    """CPU pin of the device algorithm's detection rule (the rule was checked this way before the kernel was written): on 800 random
    scenes, whenever the rule does NOT fire, the what-if batch + the mechanical pass equals the oracle's batch with the flag in
    every output (stale leader and first_k included); it fires in a few per cent of the scenes (30 of 800; 7 of those really differ: a
    re-run that was not needed costs one batch, a missed one would cost exactness)."""
    from test_gpu_parity import _force_class_mode
    flagged = wrong_when_flagged = 0
    for steady in (False, True):
        for seed in range(7000, 7400):
            sc = random_objects(seed, n_nodes=6 + seed % 40, n_groups=7, n_pods=60, n_scalars=seed % 3, n_classes=3)
            if seed % 4 == 1:
                sc["permitted"] = set()
            nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
            if steady:
                rng = np.random.default_rng(seed)
                _force_class_mode(groups, rng, sc["n_classes"])
                groups.matched[:] = rng.integers(1, 4, groups.g)
            raw = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL, bitmap=False)
            exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL | soa.BATCH_FILTER_DENY, bitmap=False)
            got, needed = mechanical_pass_and_needed_pods(raw, pods, groups, nodes.n)
            same = all(np.array_equal(getattr(got, n), getattr(exp, n)) for n in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible"))
            if needed:
                flagged += 1
                wrong_when_flagged += int(not same)
            else:
                assert same, f"seed {seed} steady {steady}: the rule did not fire and the mechanical pass is wrong"
    assert 10 <= flagged <= 80, flagged
    assert wrong_when_flagged >= 3, (flagged, wrong_when_flagged)                   # conservative (30 flagged, 7 of them really differ), not idle
