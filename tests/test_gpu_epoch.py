"""GPU tests of the positional three-launch chain (bs_epoch.hpp): batches with first-pod captures, MinResources defaults
or a leader without matched pods.  Everything goes through the C ABI; the oracle is only the checker.  The general chain
(BS_NO_EPOCH=1) has to give the same answers on the same scenes."""
import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import assert_batch_equal, load_ctx

pytestmark = pytest.mark.gpu

STAGE_SETS = ("PREFILTER", "PREFILTER|TALLY", "PREFILTER|FILTER", "ALL")


def _stages(soa, name):
    return {"PREFILTER": soa.STAGE_PREFILTER, "PREFILTER|TALLY": soa.STAGE_PREFILTER | soa.STAGE_TALLY,
            "PREFILTER|FILTER": soa.STAGE_PREFILTER | soa.STAGE_FILTER, "ALL": soa.STAGE_ALL}[name]


def _check_all_stage_sets(ctx, nodes, fit, groups, pods, soa, orc, what):
    for name in STAGE_SETS:
        stages = _stages(soa, name)
        e = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, stages)
        g = ctx.batch(stages)
        for f in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "fl_bitmap"):
            assert np.array_equal(getattr(g, f), getattr(e, f)), (what, name, f)
        if stages & soa.STAGE_TALLY:
            assert np.array_equal(g.group_admit, e.group_admit) and np.array_equal(g.group_ready, e.group_ready), (what, name)


@pytest.mark.parametrize("config", ["tiny", "cfg2"])
def test_cold_start_takes_the_positional_chain(config, monkeypatch, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make(config, "cold")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for _ in range(3):                                        # stamps / inverted sequence keys: nothing is reset in between
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{config}/cold positional")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 2 and st["launches"] == 3 and st["fast_path"] == 0
        assert st["scan_queries"] <= groups.g                     # one first check per group, however many pods ask it
        _check_all_stage_sets(ctx, nodes, fit, groups, pods, soa, orc, config)
        assert ctx.read_groups().state_equal(groups)              # what-if batches leave the group state alone
    monkeypatch.setenv("BS_NO_EPOCH", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{config}/cold general")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 0 and st["launches"] > 3


def _mixed_scene(bsa, soa, seed, config="cfg2"):
    """A cold snapshot made awkward: some gangs already have progress (so the leader changes as their first pods are
    captured, and reservation checks mix with first checks), some already have their pod but no MinResources (the
    default becomes visible in the middle of the queue), some pods come back from Permit, some groups are denied or
    latched, and the queue is reshuffled."""
    rng = np.random.default_rng(seed)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, "cold", seed=seed)
    g = groups.g
    prog = rng.choice(g, max(2, g // 8), replace=False)
    groups.matched[prog] = rng.integers(1, 3, prog.size)
    havepod = rng.choice(g, max(2, g // 6), replace=False)
    groups.flags[havepod] |= soa.GROUP_HAS_POD
    groups.cls[havepod] = rng.integers(0, fit.n_classes, havepod.size)
    groups.occupied_by[havepod[::2]] = havepod[::2].astype(np.uint64) + np.uint64(1)
    groups.flags[rng.choice(g, max(1, g // 20), replace=False)] |= soa.GROUP_DENIED
    groups.flags[rng.choice(g, max(1, g // 20), replace=False)] |= soa.GROUP_SCHEDULED_LATCH
    pods.flags[rng.random(pods.p) < 0.03] |= soa.POD_LAST_PERMITTED
    order = np.argsort(np.arange(pods.p) // 24 + rng.integers(0, 3, pods.p), kind="stable")
    pods = pods.take(order)
    return nodes, fit, groups, pods


@pytest.mark.parametrize("seed", range(9100, 9124))
def test_mixed_positional_scenes(seed, monkeypatch, bsa, soa, orc):
    nodes, fit, groups, pods = _mixed_scene(bsa, soa, seed, "tiny" if seed % 3 == 0 else "cfg2")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed} positional")
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed} positional again")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] in (0, 2)                              # 0: more leader runs than the chain takes
        if st["chain"] == 2:
            assert st["launches"] == 3
        _check_all_stage_sets(ctx, nodes, fit, groups, pods, soa, orc, f"seed {seed}")
        # a committing batch takes the general chain and must leave the sequential reference's state behind
        sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
        exp_c = sop.batch(pods, soa.STAGE_ALL)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), exp_c, f"seed {seed} commit")
        assert ctx.read_groups().state_equal(sop.groups)
        # ... and the next what-if batch runs on the committed state (analysis redone)
        exp_d = sop.batch(pods, soa.STAGE_ALL)                    # (same Sop: it carries sop.maxFinishedPG over, as the context does)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp_d, f"seed {seed} after commit")
    monkeypatch.setenv("BS_NO_EPOCH", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed} general")


def test_positional_chain_is_the_common_case_on_mixed_scenes(bsa, soa, orc):
    taken = 0
    for seed in range(9100, 9112):
        nodes, fit, groups, pods = _mixed_scene(bsa, soa, seed, "tiny")
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            taken += ctx.stats(soa.STAGE_ALL)["chain"] == 2
    assert taken >= 6


@pytest.mark.parametrize("seed", range(9200, 9230))
def test_random_object_scenes_positional_vs_general(seed, monkeypatch, bsa, soa, orc):
    """The object-level random scenes of the parity suite (captures, defaults, owners, denied groups, permitted pods):
    positional chain == oracle == general chain, and group patches between batches re-run the analysis."""
    sc = random_objects(seed, n_nodes=80 + seed % 150, n_groups=12, n_pods=260, n_scalars=seed % 3, n_classes=4)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    rng = np.random.default_rng(seed)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed}")
        cur = groups.copy()
        for rnd in range(3):                                      # per-cycle patches: matched / scheduled / deny flags
            idx = rng.choice(groups.g, 4, replace=False)
            deltas = []
            for i in idx:
                cur.matched[i] = rng.integers(0, cur.min_member[i] + 1)
                cur.status_scheduled[i] = rng.integers(0, 3)
                cur.flags[i] = (cur.flags[i] & 0x6) | int(rng.integers(0, 2)) | (8 * int(rng.integers(0, 2)))
                deltas.append((i, cur.matched[i], cur.status_scheduled[i], cur.flags[i]))
            ctx.apply_group_deltas(deltas)
            e2 = orc.Sop(orc.Snapshot(nodes, fit), cur).batch(pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), e2, f"seed {seed} patch {rnd}")
    monkeypatch.setenv("BS_NO_EPOCH", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed} general")


def test_pod_reload_between_batches(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for k in range(3):
            sub = pods.take(np.arange(k * 97, pods.p - k * 31))
            ctx.load_pods(sub)
            exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(sub, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"reload {k}")
            assert ctx.stats(soa.STAGE_ALL)["chain"] == 2
