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
        # a committing batch (same chain + k_commit) must leave the sequential reference's state behind
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


_TAKEN = []


@pytest.mark.parametrize("seed", range(9200, 9230))
def test_random_object_scenes_positional_vs_general(seed, monkeypatch, bsa, soa, orc):
    """The object-level random scenes of the parity suite (captures, defaults, owners, denied groups, permitted pods):
    positional chain == oracle == general chain, and group patches between batches re-run the analysis."""
    sc = random_objects(seed, n_nodes=80 + seed % 150, n_groups=12, n_pods=260, n_scalars=seed % 3, n_classes=4)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    if seed % 2:
        pods = _share_templates(pods, np.random.default_rng(seed + 1))        # few request classes: the chain takes the batch
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    rng = np.random.default_rng(seed)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"seed {seed}")
        if seed % 2:
            _TAKEN.append(ctx.stats(soa.STAGE_ALL)["chain"])
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


def test_random_object_scenes_mostly_take_the_positional_chain():
    assert len(_TAKEN) == 15 and sum(c == 2 for c in _TAKEN) >= 8, _TAKEN


def test_pod_reload_between_batches(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for k in range(3):
            sub = pods.take(np.arange(k * 97, pods.p - k * 31))
            ctx.load_pods(sub)
            exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(sub, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"reload {k}")
            assert ctx.stats(soa.STAGE_ALL)["chain"] == 2


def test_node_churn_between_positional_batches(bsa, soa, orc):
    """BASELINE config 5's node events on a cold snapshot: the analysis of (groups, pods) stays, the tables are rebuilt from
    the patched node list in every batch."""
    from test_gpu_parity import _churn
    _churn(bsa, soa, orc, "cfg2", "cold", rounds=6, events=30, stages=soa.STAGE_ALL, seed=15)


def _share_templates(pods, rng):
    """Pods of a gang share a template in real queues (that is what request classes live on); the object-level random
    scenes draw every pod's request on its own.  Give every group two templates (two of its own pods' requests)."""
    req, pres = pods.req.copy(), pods.req_present.copy()
    for g in np.unique(pods.group[pods.group >= 0]):
        idx = np.nonzero(pods.group == g)[0]
        reps = rng.choice(idx, 2)
        pick = reps[rng.integers(0, 2, idx.size)]
        req[:, idx] = pods.req[:, pick]
        pres[idx] = pods.req_present[pick]
    pods.req[:, :] = req
    pods.req_present[:] = pres
    return pods


def _widen(soa, nodes, groups, pods, S, rng):
    """The same scene on 4 + S resource lanes: extra scalar keys with random presence and small random amounts (pods of a
    gang keep sharing their template: the extra lanes are a function of the pod's group)."""
    L0, L = nodes.lanes, 4 + S
    if L <= L0:
        return nodes, groups, pods
    n, g, p = nodes.n, groups.g, pods.p
    extra = ((1 << S) - 1) & ~((1 << (L0 - 4)) - 1)
    def grow(a, cols, hi):
        return np.concatenate([a, rng.integers(0, hi, size=(L - L0, cols)).astype(np.int64)], 0)
    nodes2 = soa.Nodes(grow(nodes.allocatable, n, 64), grow(nodes.requested, n, 24),
                       nodes.allocatable_present | (rng.integers(0, 1 << S, n).astype(np.uint32) & np.uint32(extra)),
                       nodes.requested_present | (rng.integers(0, 1 << S, n).astype(np.uint32) & np.uint32(extra)), nodes.flags.copy())
    gx = rng.integers(0, 3, size=(L - L0, g)).astype(np.int64)
    gp = rng.integers(0, 1 << S, g).astype(np.uint32) & np.uint32(extra) & np.where(rng.random(g) < 0.5, 0xFFFFFFFF, 0).astype(np.uint32)
    groups2 = soa.Groups(groups.min_member, groups.status_scheduled, groups.matched, groups.flags, groups.cls,
                         np.concatenate([groups.min_resources, gx], 0), groups.min_resources_present | gp, groups.occupied_by)
    gi = np.clip(pods.group, 0, g - 1)
    pods2 = soa.Pods(pods.group, np.concatenate([pods.req, gx[:, gi]], 0), pods.req_present | gp[gi], pods.cls, pods.owner, pods.flags)
    return nodes2, groups2.copy(), pods2.copy()


@pytest.mark.parametrize("n_scalars", [0, 2, 5, 12])
def test_positional_chain_with_scalar_lanes(n_scalars, bsa, soa, orc):
    """Scalar resource lanes 0 / 2 (compile-time shapes), 5 and 12 (wider rows, LP = 16) through the positional chain."""
    taken = 0
    for seed in (9300, 9301, 9302):
        sc = random_objects(seed, n_nodes=140, n_groups=10, n_pods=300, n_scalars=min(n_scalars, 2), n_classes=3)
        nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                                denied=sc["denied"], permitted=sc["permitted"])
        pods = _share_templates(pods, np.random.default_rng(seed + 1))
        nodes, groups, pods = _widen(soa, nodes, groups, pods, n_scalars, np.random.default_rng(seed))
        assert nodes.lanes == 4 + n_scalars
        exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"S={n_scalars} seed {seed}")
            taken += ctx.stats(soa.STAGE_ALL)["chain"] == 2
    assert taken >= 1


@pytest.mark.parametrize("scene", ["cold", "mixed"])
def test_sharded_positional_batches(scene, monkeypatch, bsa, soa, orc):
    """Replicated mode (bs_shard_set) on a positional state: every rank runs the positional chain over the whole queue and
    reports its own groups' pods.  Decisions of owned pods equal the single-context batch, the admit counters of the ranks add up
    to the single-context ones, and rank by rank EVERYTHING equals what the general chain reports for the same shard."""
    if scene == "cold":
        nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    else:
        nodes, fit, groups, pods = _mixed_scene(bsa, soa, 9107, "cfg2")
    parts = {}
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        single = ctx.batch(soa.STAGE_ALL)
        reach_codes = [soa.PF_PASS_NO_MAX, soa.PF_PASS_FIRST_FITS, soa.PF_PASS_IS_MAX, soa.PF_PASS_RESERVE_FITS, soa.PF_REJECT_FIRST, soa.PF_REJECT_RESERVE]
        for nranks in (2, 3):
            admit = np.zeros(groups.g, np.uint32)
            owned = np.zeros(pods.p, np.int32)
            for r in range(nranks):
                ctx.set_shard(r, nranks)
                part = ctx.batch(soa.STAGE_ALL)
                assert ctx.stats(soa.STAGE_ALL)["chain"] == 2
                parts[(nranks, r)] = part
                mine = part.pf_code != 0xFF
                owned += mine
                assert np.array_equal(part.pf_code[mine], single.pf_code[mine])
                assert np.array_equal(part.pf_first_k[mine], single.pf_first_k[mine])
                # the stale shared leader (and the Filter result that hangs on it) is a per-rank view while captures can occur
                fresh = mine & np.isin(part.pf_code, reach_codes)
                assert np.array_equal(part.pf_leader[fresh], single.pf_leader[fresh])
                assert np.array_equal(part.fl_code[fresh], single.fl_code[fresh])
                assert np.array_equal(part.fl_feasible[fresh], single.fl_feasible[fresh])
                assert np.array_equal(part.fl_bitmap[:, fresh], single.fl_bitmap[:, fresh])
                admit += part.group_admit
            assert np.all(owned == 1), "every pod is evaluated by exactly one rank"
            stale_free = not np.any((single.pf_code == soa.PF_PASS_LAST_PERMITTED))
            if stale_free:
                assert np.array_equal(admit, single.group_admit)
        ctx.set_shard(0, 1)
    monkeypatch.setenv("BS_NO_EPOCH", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for (nranks, r), part in parts.items():
            ctx.set_shard(r, nranks)
            gen = ctx.batch(soa.STAGE_ALL)
            assert ctx.stats(soa.STAGE_ALL)["chain"] == 0
            for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "fl_bitmap", "group_admit"):
                assert np.array_equal(getattr(part, name), getattr(gen, name)), (nranks, r, name)


def _ladder_scene(bsa, soa, steps, seed=4):
    """A cold snapshot in which the leader changes `steps - 1` times inside the queue: the gangs arrive in group order and
    each later one of the first `steps` has more progress than everything before it (core.go:721-724 strict '>')."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold", seed=seed)
    order = np.argsort(np.where(pods.group < 0, 0, pods.group), kind="stable")        # queue in group order
    pods = pods.take(order)
    # eight request templates for the whole queue (few request classes: the slot count, not the run count, decides otherwise)
    first = {int(g): int(np.nonzero(pods.group == g)[0][0]) for g in range(8)}
    for i in np.nonzero(pods.group >= 0)[0]:
        src = first[int(pods.group[i]) % 8]
        pods.req[:, i] = pods.req[:, src]
        pods.req_present[i] = pods.req_present[src]
    groups.min_member[:steps] = 50
    groups.matched[:steps] = np.arange(steps, dtype=np.uint32) * 3                    # progress 0, 60, 120, ... per mille
    return nodes, fit, groups, pods


# (the stretch before the first capture, with no leader at all, is a run of its own: `steps` leaders = steps + 1 runs)
@pytest.mark.parametrize("steps,chain", [(2, 2), (3, 2), (4, 2), (9, 2), (14, 2), (15, 2), (16, 0), (24, 0)])
def test_leader_ladder_and_the_run_limit(steps, chain, bsa, soa, orc, monkeypatch):
    """round 4: the positional chain takes up to SIXTEEN leader runs (round 3: four — a queue with more leader changes fell back to
    the eleven-launch general chain); beyond that the general chain still takes the batch.  Every ladder also against the general
    chain itself (BS_NO_EPOCH=1)."""
    nodes, fit, groups, pods = _ladder_scene(bsa, soa, steps)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    assert len(set(exp.pf_leader[exp.pf_leader >= 0].tolist())) >= steps - 1
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"ladder {steps}")
        assert ctx.stats(soa.STAGE_ALL)["chain"] == chain           # more than sixteen leader runs: the general chain takes the batch
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS), exp, f"ladder {steps}, latency mode")
    monkeypatch.setenv("BS_NO_EPOCH", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"ladder {steps}, general chain")
        assert ctx.stats(soa.STAGE_ALL)["chain"] == 0


def test_every_pod_back_from_permit_and_a_panic_epoch(bsa, soa, orc):
    """(a) every pod carries a live lastPermittedPod entry: nobody reaches findMaxPG, Filter runs against the leader carried
    into the batch; (b) a gang with MinMember == 0 is captured in mid-queue: findMaxPG panics from that epoch on."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold", seed=8)
    p2 = pods.copy()
    p2.flags[:] |= soa.POD_LAST_PERMITTED
    g2 = groups.copy()
    g2.flags[3] |= soa.GROUP_HAS_POD | soa.GROUP_HAS_MINRES                 # somebody the carried-in leader can be
    g2.min_resources[0, 3], g2.min_resources[1, 3] = 500, 1 << 30
    sop = orc.Sop(orc.Snapshot(nodes, fit), g2)
    warm = sop.batch(pods.take(np.arange(64)), soa.STAGE_ALL)               # leaves sop.maxFinishedPG behind
    with load_ctx(bsa, nodes, fit, g2, pods.take(np.arange(64))) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), warm, "first batch")
        ctx.load_pods(p2)
        exp = sop.batch(p2, soa.STAGE_ALL)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "all permitted")
    g3 = groups.copy()
    g3.min_member[groups.g // 2] = 0
    g3.status_scheduled[groups.g // 2] = 1                                   # uint32(MinMember - Scheduled) != 0 and MinMember == 0: divide by zero
    exp = orc.Sop(orc.Snapshot(nodes, fit), g3).batch(pods, soa.STAGE_ALL)
    assert (exp.pf_code == soa.PF_PANIC_DIV0).any() and (exp.pf_code == soa.PF_PASS_FIRST_FITS).any()
    with load_ctx(bsa, nodes, fit, g3, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "panic epoch")
        assert ctx.stats(soa.STAGE_ALL)["chain"] == 2
