"""GPU tests of bs_seq_run — the reference's scheduling cycle POD BY POD on the device (csrc/bs_seq.hpp) — against the CPU
restatement of the same pass (oracle/bs_oracle_seq.c: PreFilter core.go:88-167 with the deny / capture / occupancy side
effects, [Filter core.go:170-191, :514-564,] first-fit node choice, assume, Permit core.go:268-309, release + PostBind
core.go:327).  Everything through the C ABI; bit-exact: every pod's code, first_k and stale leader, the node of every released
pod, the gangs in release order, and the node requests and the whole group state the pass leaves behind."""
import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects
from test_drain import compare_order, complete_gangs, readme_scene
from test_gpu_parity import load_ctx

pytestmark = pytest.mark.gpu


def assert_groups_equal(got, exp, soa, where=""):
    assert np.array_equal(got.min_member, exp.min_member), where
    assert np.array_equal(got.status_scheduled, exp.status_scheduled), f"{where}: status_scheduled"
    assert np.array_equal(got.matched, exp.matched), f"{where}: matched"
    assert np.array_equal(got.flags, exp.flags), f"{where}: flags {got.flags} vs {exp.flags}"
    has_pod = (exp.flags & soa.GROUP_HAS_POD) != 0
    assert np.array_equal(got.cls[has_pod], exp.cls[has_pod]), f"{where}: cls"
    has_mr = (exp.flags & soa.GROUP_HAS_MINRES) != 0
    assert np.array_equal(got.min_resources[:, has_mr], exp.min_resources[:, has_mr]), f"{where}: MinResources"
    assert np.array_equal(got.min_resources_present[has_mr], exp.min_resources_present[has_mr]), f"{where}: MinResources keys"
    assert np.array_equal(got.occupied_by, exp.occupied_by), f"{where}: OccupiedBy"


def check_pass(ctx, s, soa, where=""):
    """one bs_seq_run on `ctx` against the oracle's record `s` of the same pass"""
    r = ctx.seq_run(soa.STAGE_PREFILTER | (s["stages"] & (soa.STAGE_FILTER | soa.BATCH_FILTER_DENY)))
    if s["stages"] & soa.BATCH_FILTER_DENY:
        assert np.array_equal(r["last_permitted"], s["last_permitted"]), f"{where}: lastPermittedPod entries the pass leaves (core.go:188)"
    bad = np.nonzero(r["pf_code"] != s["pf_code"])[0]
    assert bad.size == 0, f"{where}: pf_code differs first at pod {bad[0]}: {r['pf_code'][bad[0]]} vs {s['pf_code'][bad[0]]}"
    assert np.array_equal(r["pf_first_k"], s["pf_first_k"]), f"{where}: first_k"
    assert np.array_equal(r["pf_leader"], s["pf_leader"]), f"{where}: stale leader"
    assert np.array_equal(r["pod_node"], s["pod_node"]), f"{where}: pod_node"
    assert r["n_released"] == s["n_released"]
    assert r["released_group"].tolist() == s["released_group"].tolist(), f"{where}: gangs in release order"
    assert r["released_pods"].tolist() == s["released_pods"].tolist(), f"{where}: pods per gang"
    req, pres = ctx.read_node_requests()
    assert np.array_equal(req, s["nodes"].requested), f"{where}: node requests after the pass"
    assert np.array_equal(pres, s["nodes"].requested_present), f"{where}: node request keys after the pass"
    assert_groups_equal(ctx.read_groups(), s["groups"], soa, where)
    if r["n_released"]:
        assert np.all(r["ready_ns"] >= r["first_ns"]) and np.all(np.diff(r["ready_ns"]) >= 0) and r["total_ns"] >= r["ready_ns"][-1]
    return r


def oracle_pass(orc, nodes, fit, groups, pods, stages):
    s = orc.seq_replay(nodes, fit, groups, pods, stages)
    s["stages"] = stages
    return s


def gang_scene(seed, soa, n_nodes=48, n_groups=10, n_pods=140, filter_on=False):
    """random object scene shaped so that gangs really get through: roomy nodes, gangs of 2..6 with their pods in the queue"""
    rng = np.random.default_rng(seed)
    sc = random_objects(seed, n_nodes=n_nodes, n_groups=n_groups, n_pods=n_pods, n_scalars=seed % 3, n_classes=3)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
    if seed % 4 != 3:                                       # most scenes: make room (the raw scenes are 0-110 % utilised)
        nodes.requested[:3] = (nodes.requested[:3] * rng.uniform(0.0, 0.6, nodes.n)).astype(np.int64)
        nodes.requested[3] = np.minimum(nodes.requested[3], np.maximum(nodes.allocatable[3] - rng.integers(1, 30, nodes.n), 0))
    if seed % 3 == 0:
        pods = compare_order(pods)
    if seed % 2:                                            # some gangs in a phase StartBatchSchedule does not release (batchscheduler.go:258-261)
        groups.flags[rng.random(groups.g) < 0.2] |= soa.GROUP_PHASE_CLOSED
    return nodes, fit, groups, pods


@pytest.mark.parametrize("filter_on", [0, 1, 2], ids=["prefilter", "prefilter+filter", "prefilter+filter+ttl-writes"])
@pytest.mark.parametrize("seed", range(4200, 4260))
def test_seq_pass_random_object_scenes(seed, filter_on, bsa, soa, orc):
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY | (soa.STAGE_FILTER if filter_on else 0) | (soa.BATCH_FILTER_DENY if filter_on == 2 else 0)
    nodes, fit, groups, pods = gang_scene(seed, soa)
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        check_pass(ctx, s, soa, f"seed {seed}")


@pytest.mark.parametrize("n_pods", [1, 2, 63, 64, 65, 127, 128, 129])
@pytest.mark.parametrize("seed", [4201, 4202, 4206])
def test_seq_pass_queue_lengths_around_the_staging_window(seed, n_pods, bsa, soa, orc):
    """k_seq_pass stages the input fields of 64 pods at a time in LDS (bs_seq.hpp, kSeqPodWin): queues that end in front of, at and just behind a window
    boundary (the last window is ragged: its tail slots repeat the last pod and are never read)"""
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY | soa.STAGE_FILTER
    nodes, fit, groups, pods = gang_scene(seed, soa)
    pods = pods.take(np.arange(min(n_pods, pods.p)))
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        check_pass(ctx, s, soa, f"seed {seed}, {n_pods} pods")


@pytest.mark.parametrize("seed", range(4300, 4330))
def test_seq_pass_raw_edge_scenes(seed, bsa, soa, orc):
    """the unshaped scenes: nil / unschedulable / taint-error nodes, missing groups, permitted pods, MinMember 0 (the uint32
    division of core.go:716-717), fully scheduled leaders (the tie rule :729-731), negative scalar requests"""
    sc = random_objects(seed)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
    for st in (soa.STAGE_PREFILTER, soa.STAGE_PREFILTER | soa.STAGE_FILTER, soa.STAGE_PREFILTER | soa.STAGE_FILTER | soa.BATCH_FILTER_DENY):
        s = oracle_pass(orc, nodes, fit, groups, pods, st)
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            check_pass(ctx, s, soa, f"seed {seed} stages {st}")


def test_seq_pass_release_kat(bsa, soa, orc):
    """VERDICT r4's reproducer (tests/test_seq_oracle_pin.py::_kat_scene): the quorum releases the two pods that were ALREADY waiting with
    the one that completes it (batchscheduler.go:292-343, core.go:327) -> matched 0, Scheduled 3, phase Scheduled, and the next pod of the
    gang asks for nothing (core.go:136-147 with notFinished 0): PASS_FIRST_FITS, no deny entry"""
    from test_seq_oracle_pin import _kat_scene
    sc = _kat_scene()
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"])
    s = oracle_pass(orc, nodes, fit, groups, pods, soa.STAGE_PREFILTER)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r = check_pass(ctx, s, soa, "release KAT")
        g = ctx.read_groups()
    assert r["pf_code"].tolist() == [soa.PF_PASS_IS_MAX, soa.PF_PASS_FIRST_FITS] and r["released_pods"].tolist() == [3]
    assert int(g.matched[0]) == 0 and int(g.status_scheduled[0]) == 3 and g.flags[0] & soa.GROUP_PHASE_CLOSED and not g.flags[0] & soa.GROUP_DENIED


def test_seq_pass_readme_race_scene(bsa, soa, orc):
    """README.md:78-188 (BASELINE config 1): group1 5/5 through, group2's first pod rejected (:140-144), the rest denied"""
    nodes, fit, groups, pods = readme_scene(soa)
    st = soa.STAGE_PREFILTER
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r = check_pass(ctx, s, soa, "readme")
    assert r["released_group"].tolist() == [0] and r["released_pods"].tolist() == [5]
    assert (r["pod_node"] >= 0).tolist() == [True] * 5 + [False] * 5
    assert r["pf_code"][5] == soa.PF_REJECT_FIRST and np.all(r["pf_code"][6:] == soa.PF_ERR_DENIED)


@pytest.mark.parametrize("config,scenario", [("tiny", "cold"), ("tiny", "warm"), ("tiny", "tail"), ("tiny", "busy"),
                                             ("cfg2", "cold"), ("cfg2", "warm"), ("cfg2", "tail"), ("cfg2", "busy")])
@pytest.mark.parametrize("order", ["as-is", "compare"])
@pytest.mark.parametrize("filter_on", [0, 1, 2], ids=["prefilter", "prefilter+filter", "prefilter+filter+ttl-writes"])
def test_seq_pass_synthetic_configs(config, scenario, order, filter_on, bsa, soa, orc):
    st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if filter_on else 0) | (soa.BATCH_FILTER_DENY if filter_on == 2 else 0)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    if order == "compare":
        pods = compare_order(pods)
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r = check_pass(ctx, s, soa, f"{config}/{scenario}/{order}")
    if scenario == "cold" and order == "compare":
        assert r["n_released"] > 0


@pytest.mark.parametrize("scenario,filter_on", [("tail", 0), ("cold", 0), ("tail", 1), ("tail", 2), ("warm", 2)])
def test_seq_pass_cfg3_full_size(scenario, filter_on, bsa, soa, orc):
    """BASELINE config 3 (10k pods / 2k groups / 5k nodes, 5 lanes), the whole pass against the oracle's"""
    st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if filter_on else 0) | (soa.BATCH_FILTER_DENY if filter_on == 2 else 0)
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", scenario)
    pods = compare_order(pods)
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r = check_pass(ctx, s, soa, f"cfg3/{scenario}")
    if filter_on == 2:                                      # every node offered: some node fails Filter for nearly every pod, the gangs are deny-listed
        assert np.count_nonzero(r["pf_code"] == soa.PF_ERR_DENIED) > 1000
    else:
        assert r["n_released"] > 100


def test_state_after_a_pass_feeds_batches_and_the_next_pass(bsa, soa, orc):
    """bs_seq_run leaves the context in the state the pass produced: a batch on it == the oracle's batch on the oracle's
    post-pass state; released pods leave the queue (bs_pods_apply), new pods arrive, and a second pass == the oracle's."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    pods = compare_order(pods)
    st = soa.STAGE_PREFILTER
    s1 = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r1 = check_pass(ctx, s1, soa, "first pass")
        exp = orc.Sop(orc.Snapshot(s1["nodes"], fit), s1["groups"]).batch(pods, soa.STAGE_ALL, bitmap=False)
        got = ctx.batch(soa.STAGE_ALL, bitmap=False)
        for name in ("pf_code", "pf_first_k", "fl_code", "fl_feasible", "group_admit", "group_ready"):
            assert np.array_equal(getattr(got, name), getattr(exp, name)), f"batch after the pass: {name}"
        gone = np.nonzero(r1["pod_node"] >= 0)[0].astype(np.uint32)
        assert gone.size
        ctx.apply_pods(remove=gone)
        left = pods.take(np.nonzero(r1["pod_node"] < 0)[0])
        assert ctx.read_pods().equal(left)
        s2 = oracle_pass(orc, s1["nodes"], fit, s1["groups"], left, st)
        check_pass(ctx, s2, soa, "second pass")


def test_seq_pass_many_groups_keys_in_global_memory(bsa, soa, orc):
    """more groups than the LDS key window holds (8192): the findMaxPG keys live in global memory"""
    rng = np.random.default_rng(77)
    nodes, fit, _, _, _ = bsa.synth.make("cfg2", "cold")
    G, per = 9000, 2
    groups = soa.Groups.empty(G, nodes.lanes)
    groups.min_member[:] = per
    group = np.repeat(np.arange(G, dtype=np.int32), per)
    req = np.zeros((nodes.lanes, G * per), np.int64)
    req[0] = rng.choice([100, 250, 500], G * per)
    req[1] = rng.choice([1, 2], G * per) * (1 << 28)
    pods = soa.Pods(group, req, np.zeros(G * per, np.uint32), rng.integers(0, fit.n_classes, G * per).astype(np.uint32), np.zeros(G * per, np.uint64),
                    np.zeros(G * per, np.uint8))
    s = oracle_pass(orc, nodes, fit, groups, pods, soa.STAGE_PREFILTER)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        r = check_pass(ctx, s, soa, "9000 groups")
    assert r["n_released"] > 50


@pytest.mark.parametrize("lanes", [9, 16])
def test_seq_pass_many_scalar_lanes(lanes, bsa, soa, orc):
    """S > 4: the kernel variant with a run-time lane count"""
    rng = np.random.default_rng(lanes)
    S = lanes - 4
    N, G, per = 70, 12, 4
    alloc = np.zeros((lanes, N), np.int64)
    alloc[0], alloc[1], alloc[2], alloc[3] = 32000, 64 << 30, 100 << 30, 110
    alloc[4:] = rng.integers(0, 9, (S, N))
    ap = rng.integers(0, 1 << S, N).astype(np.uint32)
    reqd = np.zeros((lanes, N), np.int64)
    reqd[0] = rng.integers(0, 20000, N)
    reqd[1] = rng.integers(0, 30 << 30, N)
    reqd[3] = rng.integers(0, 40, N)
    reqd[4:] = rng.integers(0, 4, (S, N))
    rp = (rng.integers(0, 1 << S, N) & ap).astype(np.uint32)
    nodes = soa.Nodes(alloc, reqd, ap, rp, np.zeros(N, np.uint8))
    fit = soa.FitMasks.from_bool(rng.random((2, N)) < 0.9)
    groups = soa.Groups.empty(G, lanes)
    groups.min_member[:] = per
    P = G * per
    req = np.zeros((lanes, P), np.int64)
    req[0] = rng.choice([500, 1000, 2000], P)
    req[1] = rng.choice([1, 2, 4], P) << 30
    pres = rng.integers(0, 1 << S, P).astype(np.uint32) * (rng.random(P) < 0.5)
    req[4:] = rng.integers(0, 3, (S, P)) * ((pres[None, :] >> np.arange(S)[:, None]) & 1)
    pods = soa.Pods(np.repeat(np.arange(G, dtype=np.int32), per), req, pres.astype(np.uint32), rng.integers(0, 2, P).astype(np.uint32), np.zeros(P, np.uint64),
                    np.zeros(P, np.uint8))
    for st in (soa.STAGE_PREFILTER, soa.STAGE_PREFILTER | soa.STAGE_FILTER):
        s = oracle_pass(orc, nodes, fit, groups, pods, st)
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            check_pass(ctx, s, soa, f"{lanes} lanes")


def test_seq_pass_refusals_and_empty_inputs(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("tiny", "cold")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        with pytest.raises(bsa.BsError) as e:
            ctx.seq_run(soa.STAGE_FILTER)                     # PREFILTER is mandatory
        assert e.value.status == -1
        ctx.set_shard(0, 2)
        with pytest.raises(bsa.BsError) as e:
            ctx.seq_run(soa.STAGE_PREFILTER)                  # a sequential pass does not shard
        assert e.value.status == -4
    empty = pods.take(np.zeros(0, np.int64))
    with load_ctx(bsa, nodes, fit, groups, empty) as ctx:
        r = ctx.seq_run(soa.STAGE_PREFILTER)
        assert r["n_released"] == 0 and r["pf_code"].size == 0
    none = soa.Nodes(np.zeros((nodes.lanes, 0), np.int64), np.zeros((nodes.lanes, 0), np.int64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    fit0 = soa.FitMasks.from_bool(np.ones((fit.n_classes, 0), bool))
    s = oracle_pass(orc, none, fit0, groups, pods, soa.STAGE_PREFILTER)
    with load_ctx(bsa, none, fit0, groups, pods) as ctx:
        check_pass(ctx, s, soa, "no nodes")


@pytest.mark.parametrize("slots", ["0", "1"], ids=["round-scan", "one-slot"])
@pytest.mark.parametrize("seed", range(4200, 4230))
def test_seq_pass_scan_variants(seed, slots, bsa, soa, orc, monkeypatch):
    """the node scan without table summaries (rounds over the node list: what clusters beyond 65 536 nodes get) and with ONE summary
    slot (every change of the table in use rebuilds it): same pass, bit for bit"""
    monkeypatch.setenv("BS_SEQ_CACHE_SLOTS", slots)
    st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if seed % 2 else 0)
    nodes, fit, groups, pods = gang_scene(seed, soa, n_nodes=150)
    s = oracle_pass(orc, nodes, fit, groups, pods, st)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        check_pass(ctx, s, soa, f"seed {seed} slots {slots}")


@pytest.mark.parametrize("slots", ["0", "1"], ids=["round-scan", "one-slot"])
def test_seq_pass_scan_variants_cfg3(slots, bsa, soa, orc, monkeypatch):
    monkeypatch.setenv("BS_SEQ_CACHE_SLOTS", slots)
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", "cold")
    pods = compare_order(pods)
    s = oracle_pass(orc, nodes, fit, groups, pods, soa.STAGE_PREFILTER)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        check_pass(ctx, s, soa, f"cfg3/cold slots {slots}")


def template_scene(soa, seed, negative_at=None, N=200, G=60, per=6):
    """gangs stamped from a few pod templates on a nearly full cluster: the searches of a template walk down the node list until
    nothing fits any more (what the first-fit cursors of k_seq_pass shorten); `negative_at`: an ungrouped pod with a negative scalar
    request in the middle of the queue (its assume step FREES capacity: a node in front of a cursor can fit again)"""
    rng = np.random.default_rng(seed)
    lanes = 5
    alloc = np.zeros((lanes, N), np.int64)
    alloc[0], alloc[1], alloc[2], alloc[3], alloc[4] = 8000, 32 << 30, 200 << 30, 110, 8
    reqd = np.zeros((lanes, N), np.int64)
    reqd[0] = rng.integers(2000, 7500, N)
    reqd[1] = rng.integers(4, 28, N) << 30
    reqd[3] = rng.integers(0, 100, N)
    reqd[4] = rng.integers(5, 9, N)
    ap = np.ones(N, np.uint32)
    rp = (rng.random(N) < 0.8).astype(np.uint32)
    reqd[4] *= rp
    nodes = soa.Nodes(alloc, reqd, ap, rp, np.zeros(N, np.uint8))
    fit = soa.FitMasks.from_bool(rng.random((2, N)) < 0.7)
    groups = soa.Groups.empty(G, lanes)
    groups.min_member[:] = per - 1
    P = G * per
    tpl = rng.integers(0, 4, G)                               # the gang's template
    req = np.zeros((lanes, P), np.int64)
    req[0] = np.repeat(np.array([500, 1000, 2000, 3500])[tpl], per)
    req[1] = np.repeat(np.array([1, 2, 4, 8])[tpl], per) << 30
    want_gpu = np.repeat(tpl % 2 == 1, per)
    req[4] = want_gpu * 1
    pres = want_gpu.astype(np.uint32)
    cls = np.repeat(rng.integers(0, 2, G), per).astype(np.uint32)   # same request under two fit classes: two cursors
    group = np.repeat(np.arange(G, dtype=np.int32), per)
    if negative_at is not None:
        group[negative_at] = soa.POD_NOT_GROUPED
        req[:, negative_at] = 0
        req[4, negative_at] = -3
        pres[negative_at] = 1
    pods = soa.Pods(group, req, pres, cls, np.zeros(P, np.uint64), np.zeros(P, np.uint8))
    return nodes, fit, groups, pods


@pytest.mark.parametrize("cursor", ["on", "off"])
@pytest.mark.parametrize("negative", [False, True], ids=["adds-only", "a-request-frees-capacity"])
@pytest.mark.parametrize("seed", range(4400, 4406))
def test_seq_pass_first_fit_cursors(seed, negative, cursor, bsa, soa, orc, monkeypatch):
    """the per-template first-fit cursors (csrc/bs_seq.hpp, node choice) never change a decision: with them, without them
    (BS_SEQ_NO_CURSOR=1), across a request that frees capacity, and with Filter gating the choice — all == the oracle's pass"""
    if cursor == "off":
        monkeypatch.setenv("BS_SEQ_NO_CURSOR", "1")
    nodes, fit, groups, pods = template_scene(soa, seed, negative_at=(150 + seed % 40) if negative else None)
    for st in (soa.STAGE_PREFILTER, soa.STAGE_PREFILTER | soa.STAGE_FILTER):
        s = oracle_pass(orc, nodes, fit, groups, pods, st)
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            r = check_pass(ctx, s, soa, f"templates seed {seed} stages {st} cursor {cursor}")
        if st == soa.STAGE_PREFILTER and not negative:
            placed = int((s["pod_node"] >= 0).sum())
            assert 0 < placed < pods.p, "the scene is meant to run the cluster full"


def test_seq_pass_cursors_survive_a_queue_patch(bsa, soa, orc):
    """request classes of pods that arrived through bs_pods_apply (ids handed out by the insert wave) key the cursors as well"""
    nodes, fit, groups, pods = template_scene(soa, 4410)
    keep = np.arange(0, pods.p, 2)
    with load_ctx(bsa, nodes, fit, groups, pods.take(keep)) as ctx:
        late = np.arange(1, pods.p, 2)
        ctx.apply_pods(remove=np.zeros(0, np.uint32), insert=pods.take(late))
        queue = pods.take(np.concatenate([keep, late]))
        s = oracle_pass(orc, nodes, fit, groups, queue, soa.STAGE_PREFILTER)
        check_pass(ctx, s, soa, "patched queue")
