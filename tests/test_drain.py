"""The batched scheduling cycle (host/bs_drain.cpp) against its two CPU statements: the gang-granular reference drain
(tests/drain_ref.py: the same loop over the oracle's batches) and the reference's own pod-by-pod pass (oracle/bs_oracle_seq.c).

What is asserted and why: the GPU drain == the reference drain ALWAYS (same gangs in the same order, same nodes for every
pod, same final node requests and group counters).  The pod-by-pod pass is the reference's semantics; the gang-granular loop
equals it when every gang of the queue is complete and the queue is in Compare order (core.go:368-411 keeps a gang's pods
together) and no leader carries matched pods into the pass: then a gang is decided by its first pod's check (core.go:136-147)
and its other pods pass as the leader's (:150-155) in both.  Outside those conditions (partial gangs hold what they assumed,
reservation checks see a shrinking cluster pod by pod) the sequential pass admits no more gangs than the drain pre-screens —
the README race scene is the reference's own example and both end with exactly one gang through."""
import json
import os

import numpy as np
import pytest

import drain_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def compare_order(pods):
    """queue in Compare order: a gang's pods together, gangs by creation (= index) order, unlabelled pods first"""
    return pods.take(np.argsort(pods.group, kind="stable"))


def complete_gangs(groups, pods):
    """drop the pods of gangs that cannot reach their quorum from the queue alone"""
    cnt = np.bincount(pods.group[pods.group >= 0], minlength=groups.g)
    need = groups.min_member.astype(np.int64) - groups.status_scheduled - groups.matched
    ok = cnt >= need
    keep = (pods.group < 0) | ok[np.clip(pods.group, 0, groups.g - 1)]
    return pods.take(np.nonzero(keep)[0])


def counters_equal(a, b, soa):
    return (np.array_equal(a.matched, b.matched) and np.array_equal(a.status_scheduled, b.status_scheduled)
            and np.array_equal(a.flags & soa.GROUP_SCHEDULED_LATCH, b.flags & soa.GROUP_SCHEDULED_LATCH))


@pytest.mark.parametrize("config", ["tiny", "cfg2"])
@pytest.mark.parametrize("filter_on", [False, True], ids=["prefilter", "prefilter+filter"])
def test_reference_drain_equals_sequential_pass_on_complete_cold_queues(config, filter_on, bsa, soa, orc):
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY | (soa.STAGE_FILTER if filter_on else 0)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, "cold")
    pods = complete_gangs(groups, compare_order(pods))
    d = drain_ref.drain(orc, nodes, fit, groups, pods, st)
    s = orc.seq_replay(nodes, fit, groups, pods, st)
    assert [g for g, _ in d["admitted"]] == s["released_group"].tolist()
    assert [k for _, k in d["admitted"]] == s["released_pods"].tolist()
    assert np.array_equal(d["pod_node"], s["pod_node"])
    assert np.array_equal(d["nodes"].requested, s["nodes"].requested) and np.array_equal(d["nodes"].requested_present, s["nodes"].requested_present)
    assert counters_equal(d["groups"], s["groups"], soa)
    assert len(d["admitted"]) > 0 and np.all(s["ready_ns"] >= s["first_ns"])


def readme_scene(soa):
    scene = json.load(open(os.path.join(GOLD, "readme_race_scene.json")))
    nd = scene["node"]
    alloc = np.array([[nd["allocatable_cpu"]], [64 << 30], [0], [nd["allocatable_pods"]]], np.int64)
    req = np.array([[nd["requested_cpu"]], [0], [0], [nd["pod_count"]]], np.int64)
    nodes = soa.Nodes(alloc, req, [0], [0], [0])
    fit = soa.FitMasks.from_bool(np.ones((1, 1), bool))
    groups = soa.Groups.empty(2, 4)
    groups.min_member[:] = 5
    group = np.array([0] * 5 + [1] * 5, np.int32)                       # Compare order: group1 was created first
    req_p = np.zeros((4, 10), np.int64)
    req_p[0, :] = 1000
    pods = soa.Pods(group, req_p, np.zeros(10, np.uint32), np.zeros(10, np.uint32), np.zeros(10, np.uint64), np.zeros(10, np.uint8))
    return nodes, fit, groups, pods


def test_readme_race_scene_one_gang_through(soa, orc):
    """README.md:78-188 (BASELINE config 1): two gangs of 5 x 1 CPU, one 8-CPU node with 0.9 CPU in use -> group1 5/5, group2 0/5,
    in the pod-by-pod pass and in the gang-granular drain (whose first batch reports BOTH gangs ready: a pre-screen)."""
    nodes, fit, groups, pods = readme_scene(soa)
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    first = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, st, bitmap=False)
    assert first.group_ready.tolist() == [1, 1]
    s = orc.seq_replay(nodes, fit, groups, pods, st)
    d = drain_ref.drain(orc, nodes, fit, groups, pods, st)
    assert s["released_group"].tolist() == [0] and [g for g, _ in d["admitted"]] == [0]
    assert (s["pod_node"] >= 0).tolist() == [True] * 5 + [False] * 5 and np.array_equal(s["pod_node"], d["pod_node"])
    assert s["pf_code"][5] == soa.PF_REJECT_FIRST and np.all(s["pf_code"][6:] == soa.PF_ERR_DENIED)


# ------------------------------------------------------------------------------------------------ GPU
def run_gpu_drain(bsa, nodes, fit, groups, pods, st):
    n2, g2 = nodes.copy(), groups.copy()
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        r = bsa.plugin.drain(ctx, n2, fit, g2, pods, st)
        left = ctx.read_pods()
        back = ctx.read_groups()
    return r, n2, g2, left, back


@pytest.mark.gpu
@pytest.mark.parametrize("config,scenario,ordered", [("tiny", "cold", True), ("tiny", "warm", False), ("tiny", "tail", False), ("tiny", "busy", True),
                                                     ("cfg2", "cold", True), ("cfg2", "tail", False), ("cfg2", "warm", True)])
@pytest.mark.parametrize("mode", ["prefilter", "filter", "latency"])
def test_gpu_drain_equals_reference_drain(config, scenario, ordered, mode, bsa, soa, orc):
    """gang by gang: same release order, same node for every pod, same node requests, group counters and queue at the end —
    on ordered and interleaved queues, complete and partial gangs, with the plugin's Filter gating the node choice, and
    with the results coming home through pinned memory (latency mode)."""
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY | (soa.STAGE_FILTER if mode == "filter" else 0)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    if ordered:
        pods = compare_order(pods)
    d = drain_ref.drain(orc, nodes, fit, groups, pods, st)
    r, n2, g2, left, back = run_gpu_drain(bsa, nodes, fit, groups, pods, st | (soa.BATCH_HOST_RESULTS if mode == "latency" else 0))
    assert r["admitted_group"].tolist() == [g for g, _ in d["admitted"]]
    assert r["admitted_pods"].tolist() == [k for _, k in d["admitted"]]
    assert np.array_equal(r["pod_node"], d["pod_node"])
    assert np.array_equal(n2.requested, d["nodes"].requested) and np.array_equal(n2.requested_present, d["nodes"].requested_present)
    assert counters_equal(g2, d["groups"], soa) and counters_equal(back, d["groups"], soa)
    assert r["n_stuck"] == len(d["stuck"]) and r["pods_left"] == d["pods_left"] == left.p
    assert left.equal(pods.take(np.nonzero(d["pod_node"] < 0)[0])), "the resident queue is what was not released, in order"
    assert np.all(np.diff(r["admitted_ns"]) > 0) and np.all(r["cycle_ns"] > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("config,scenario", [("tiny", "cold"), ("cfg2", "cold"), ("cfg2", "tail")])
def test_gpu_drain_over_subscribed_gangs(config, scenario, bsa, soa, orc):
    """gangs with MORE pending pods than their quorum (MinMember lowered under the queue's pod count): the drain's stated rule is
    all-or-nothing per gang — every passing member is placed or the gang is rolled back and marked stuck — identically in
    host/bs_drain.cpp and in tests/drain_ref.py; the reference's own rule (release at the quorum, the rest stays pending) is
    bs_seq_run's (tests/test_gpu_seq.py)."""
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    pods = compare_order(pods)
    rng = np.random.default_rng(3)
    cut = rng.random(groups.g) < 0.5
    groups.min_member[cut] = np.maximum(groups.min_member[cut].astype(np.int64) - rng.integers(1, 3, int(cut.sum())), 1).astype(np.uint32)
    d = drain_ref.drain(orc, nodes, fit, groups, pods, st)
    r, n2, g2, left, back = run_gpu_drain(bsa, nodes, fit, groups, pods, st)
    assert r["admitted_group"].tolist() == [g for g, _ in d["admitted"]] and r["admitted_pods"].tolist() == [k for _, k in d["admitted"]]
    assert np.array_equal(r["pod_node"], d["pod_node"]) and np.array_equal(n2.requested, d["nodes"].requested)
    assert counters_equal(g2, d["groups"], soa) and counters_equal(back, d["groups"], soa) and r["n_stuck"] == len(d["stuck"])
    assert any(k > int(groups.min_member[g]) for g, k in d["admitted"]), "the scene must release a gang with more pods than its quorum"


@pytest.mark.gpu
def test_gpu_drain_readme_scene_and_sequential_pass(bsa, soa, orc):
    nodes, fit, groups, pods = readme_scene(soa)
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    r, n2, g2, left, _ = run_gpu_drain(bsa, nodes, fit, groups, pods, st)
    assert r["admitted_group"].tolist() == [0] and r["admitted_pods"].tolist() == [5] and left.p == 5 and n2.requested[0, 0] == 5900
    # and on a complete cold queue the GPU drain is the reference's pod-by-pod pass
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    pods = complete_gangs(groups, compare_order(pods))
    s = orc.seq_replay(nodes, fit, groups, pods, st)
    r, n2, g2, left, _ = run_gpu_drain(bsa, nodes, fit, groups, pods, st)
    assert r["admitted_group"].tolist() == s["released_group"].tolist() and np.array_equal(r["pod_node"], s["pod_node"])
    assert np.array_equal(n2.requested, s["nodes"].requested) and counters_equal(g2, s["groups"], soa)
