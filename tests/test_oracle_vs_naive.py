"""C oracle vs the independent object-level Python restatement on random small scenarios (CPU only)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import naive_ref as nv
from scenarios import naive_batch, random_objects


def _flatten(sc):
    return nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                     denied=sc["denied"], permitted=sc["permitted"])


def _check_batch(seed, orc, soa, **kw):
    sc = random_objects(seed, **kw)
    nodes, fit, groups, pods, gidx = _flatten(sc)
    snap = orc.Snapshot(nodes, fit)
    sop = orc.Sop(snap, groups)
    out = sop.batch(pods, soa.STAGE_ALL)
    ref = naive_batch(sc)          # mutates sc objects
    assert out.pf_code.tolist() == ref["pf_code"], seed
    assert out.pf_first_k.tolist() == ref["pf_first_k"], seed
    assert out.pf_leader.tolist() == ref["pf_leader"], seed
    assert out.fl_code.tolist() == ref["fl_code"], seed
    assert out.fl_feasible.tolist() == ref["fl_feasible"], seed
    for i, row in enumerate(ref["bits"]):
        for k, ok in enumerate(row):
            assert out.node_passes(i, k) == ok, (seed, i, k)
    assert out.group_admit.tolist() == ref["group_admit"], seed
    assert out.group_ready.tolist() == ref["group_ready"], seed
    # post-batch group state: flatten the mutated naive objects and compare
    _, _, groups_after, _, _ = nv.to_soa(sc["nodes"], sc["cache"], [], sc["names"], sc["n_classes"], denied=ref["denied"],
                                         owner_ids=None)
    mine = sop.groups
    assert np.array_equal(mine.flags, groups_after.flags), seed
    hp = (mine.flags & soa.GROUP_HAS_POD) != 0
    hm = (mine.flags & soa.GROUP_HAS_MINRES) != 0
    assert np.array_equal(mine.cls[hp], groups_after.cls[hp])
    assert np.array_equal(mine.min_resources[:, hm], groups_after.min_resources[:, hm])
    assert np.array_equal(mine.min_resources_present[hm], groups_after.min_resources_present[hm])
    assert np.array_equal(mine.occupied_by != 0, groups_after.occupied_by != 0)


@pytest.mark.parametrize("seed", range(300))
def test_batch_oracle_equals_naive(seed, orc, soa):
    _check_batch(seed, orc, soa)


@pytest.mark.parametrize("seed", range(1000, 1040))
def test_batch_oracle_equals_naive_larger(seed, orc, soa):
    _check_batch(seed, orc, soa, n_nodes=40, n_groups=8, n_pods=60, n_scalars=2, n_classes=3)


@pytest.mark.parametrize("seed", range(2000, 2030))
def test_batch_no_edge_flags(seed, orc, soa):
    _check_batch(seed, orc, soa, n_nodes=25, n_groups=6, n_pods=40, edge=False)


@settings(max_examples=150, deadline=None)
@given(seed=st.integers(min_value=10_000, max_value=10_000_000))
def test_batch_hypothesis_seeds(seed, orc, soa):
    _check_batch(seed, orc, soa)


@pytest.mark.parametrize("seed", range(200))
def test_find_max_pg_and_prealloc(seed, orc, soa):
    sc = random_objects(seed + 5000, n_pods=1)
    for pgs in sc["cache"].values():          # make most groups candidates
        if pgs.pod is None and seed % 3:
            pgs.pod = nv.Pod("rep", pgs.pod_group.name, {"cpu": 1000})
    nodes, fit, groups, pods, gidx = _flatten(sc)
    names = list(sc["cache"].keys())
    try:
        name, mx, fin = nv.find_max_pg(sc["cache"])
        exp = (names.index(name) if mx is not None else -1, fin, False)
    except nv.GoPanic:
        exp = None
    leader, fin, panic = orc.find_max_pg(groups)
    if exp is None:
        assert panic
    else:
        assert (leader, fin, panic) == exp
    S = len(sc["names"])
    for gi, nm in enumerate(names):
        for matched in (0, 1, 3, 9):
            pre = nv.get_pre_allocated(sc["cache"][nm], matched)
            lanes, present = orc.pre_allocated(groups, gi, matched, S)
            exp_lanes, exp_present = nv._lanes(pre, sc["names"])
            assert lanes == exp_lanes and present == exp_present


@settings(max_examples=300, deadline=None)
@given(a=st.integers(min_value=-(2 ** 63), max_value=2 ** 63 - 1),
       pct_bits=st.sampled_from([0x3F333333, 0x3F800000, 0x3F000000, 0x3FC00000, 0x3E99999A]))
def test_scale_hypothesis(a, pct_bits, orc):
    pct = float(np.uint32(pct_bits).view(np.float32))
    assert orc.scale(a, pct) == nv.scale(a, pct)


# ---- structured positional scenes (the states the GPU's positional chain exists for): the C oracle and the independent
# object-level restatement have to agree on them before the GPU is compared with the oracle (tests/test_gpu_epoch.py)
def _check_scene(sc, orc, soa, what):
    nodes, fit, groups, pods, _ = _flatten(sc)
    out = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    ref = naive_batch(sc)
    for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
        assert getattr(out, name).tolist() == ref[name], (what, name)
    return out


def _cold(sc, rng, ladder):
    """nothing seen yet: no group has its pod or MinResources; with `ladder`, later groups have more progress, so the leader
    changes at their captures (core.go:721-724) and reservation checks follow the first checks"""
    names = list(sc["cache"].keys())
    for k, nm in enumerate(names):
        pgs = sc["cache"][nm]
        pgs.pod, pgs.scheduled = None, False
        pgs.pod_group.min_resources = None
        pgs.pod_group.status_scheduled = 0
        pgs.pod_group.min_member = max(pgs.pod_group.min_member, 2) + 6
        pgs.matched = (k if ladder else 0)
    sc["denied"] = set()


@pytest.mark.parametrize("seed", range(7000, 7040))
def test_cold_start_and_leader_ladder_oracle_equals_naive(seed, orc, soa):
    rng = np.random.default_rng(seed)
    sc = random_objects(seed, n_nodes=30, n_groups=7, n_pods=70, n_scalars=seed % 3, n_classes=3)
    _cold(sc, rng, ladder=bool(seed % 2))
    if seed % 4 >= 2:                                            # queue in group order: one capture after the other
        sc["pods"].sort(key=lambda p: (p.group is None, str(p.group)))
    out = _check_scene(sc, orc, soa, f"cold {seed}")
    reached = [c for c in out.pf_code.tolist() if c in (soa.PF_PASS_FIRST_FITS, soa.PF_REJECT_FIRST, soa.PF_PASS_RESERVE_FITS,
                                                         soa.PF_REJECT_RESERVE, soa.PF_PASS_IS_MAX)]
    assert reached, "the scene has to exercise findMaxPG"
    if seed % 2 and seed % 4 >= 2:
        assert len(set(out.pf_leader[out.pf_leader >= 0].tolist())) >= 2, "a ladder has to change the leader"


@pytest.mark.parametrize("seed", range(7100, 7120))
def test_permitted_queue_and_panic_epoch_oracle_equals_naive(seed, orc, soa):
    sc = random_objects(seed, n_nodes=30, n_groups=6, n_pods=60, n_scalars=seed % 3, n_classes=3)
    if seed % 2:
        sc["permitted"] = {p.uid for p in sc["pods"]}            # nobody reaches findMaxPG: Filter sees the carried-in (nil) leader
    else:
        _cold(sc, np.random.default_rng(seed), ladder=False)
        victim = list(sc["cache"].values())[3]
        victim.pod_group.min_member, victim.pod_group.status_scheduled = 0, 1     # uint32(0 - 1) != 0, then / 0: core.go:716-717
    out = _check_scene(sc, orc, soa, f"scene {seed}")
    if seed % 2 == 0 and any(p.group == "ns/g3" for p in sc["pods"]):
        assert (out.pf_code == soa.PF_PANIC_DIV0).any()


def test_threaded_group_subsets_equal_the_single_batch(bsa, soa, orc):
    """bench.py's all-cores CPU baseline (orc_batch_threads: whole groups per thread, one orc_sop each) computes exactly what the
    single sequential batch computes in the steady state: per-pod codes, first_k, Filter results, per-group admit counts."""
    import numpy as np
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    snap = orc.Snapshot(nodes, fit)
    full = orc.Sop(snap, groups).batch(pods, soa.STAGE_ALL, bitmap=False)
    for n in (3, 8):
        own = bsa.dist.owner_ranks(pods.group, groups.g, n)
        idx = [np.nonzero(own == r)[0] for r in range(n)]
        wall, iters, pairs = orc.batch_threads(snap, groups, [pods.take(i) for i in idx], soa.STAGE_ALL)
        assert wall > 0 and iters > 0
        admit = np.zeros(groups.g, np.uint32)
        for i, (_sop, out) in zip(idx, pairs):
            for name in ("pf_code", "pf_first_k", "fl_code", "fl_feasible"):
                assert np.array_equal(getattr(out, name), getattr(full, name)[i]), (n, name)
            admit += out.group_admit
        assert np.array_equal(admit, full.group_admit)
