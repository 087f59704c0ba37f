"""GPU parity in the THROUGHPUT regime of the steady-state chain: more than 16 tiles of class slots in a batch (thousands of
distinct requests), where launch B's two roles run one scan item per wave and — with BS_TP_FILTER — as launches of their own:
k_fast_scan, then a lean Filter kernel (1..4: filter_item at 109 / 93 / 75 / 72 VGPRs; 5: the transposed item of
csrc/bs_filter_t.hpp — lanes are request slots, nodes come through the scalar cache; 6 / 7: one launch again, the Filter role taken
by the transposed item).  BS_TP_SHARE bounds the scan shares per tile.
Every form against the oracle, bit for bit: codes, first_k, leaders, Filter codes, feasible counts, slot rows, expanded bitmap,
admit / ready.  Scenes: all-distinct requests on synthetic clusters (S = 0, 1, 2), random object scenes with nil / unschedulable
nodes and a STALE leader (both leader halves of the slot array in use; the tile across their boundary names two leaders),
latency mode (rows written home), Filter's deny entry inside the batch, commit.

Every test needs a real MI355X (`-m gpu`); nothing falls back to the CPU."""
import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import assert_batch_equal, load_ctx, _force_class_mode

pytestmark = pytest.mark.gpu

FORMS = [0, 1, 2, 3, 4, 5, 6, 7, 8]
ONE_LAUNCH = (0, 6, 7)          # forms that keep both roles of launch B in one launch


def _distinct(bsa, config, scenario, **over):
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario, **over)
    pods.req[0, :] += np.arange(pods.p, dtype=np.int64)          # cpu-milli differs pod by pod: no request is shared
    return nodes, fit, groups, pods


def _check_split(st, form):
    assert st["chain"] == 1 and st["class_mode"] == 1
    assert st["launches"] == (3 if form in ONE_LAUNCH else 4), "the batch did not take the throughput-regime launches"


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("config,scenario,over", [
    ("cfg2", "tail", dict(pods=2300, groups=300, nodes=700)),
    ("cfg2", "warm", dict(pods=1500, groups=200, nodes=449)),
    ("cfg3", "tail", dict(pods=3000, groups=500, nodes=1300, classes=16)),
    ("cfg3", "busy", dict(pods=2000, groups=400, nodes=1000, classes=8, scalars=2)),
])
def test_all_distinct_requests_every_form(form, config, scenario, over, bsa, soa, orc, monkeypatch):
    nodes, fit, groups, pods = _distinct(bsa, config, scenario, **over)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"form {form}")
        _check_split(ctx.stats(soa.STAGE_ALL), form)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"form {form}, again (stamps)")


@pytest.mark.parametrize("config,scenario,over", [
    ("cfg2", "tail", dict(pods=2300, groups=300, nodes=700)),
    ("cfg3", "tail", dict(pods=3000, groups=500, nodes=1300, classes=16)),
    ("cfg3", "warm", dict(pods=2000, groups=400, nodes=1000, classes=8, scalars=2)),
])
def test_all_distinct_requests_library_defaults(config, scenario, over, bsa, soa, orc, monkeypatch):
    """No switch set: the throughput regime takes one launch for both roles of launch B (the Filter role by the transposed item)."""
    for k in ("BS_TP_FILTER", "BS_TP_SHARE", "BS_TP_FWAVES", "BS_FILTER_WAVES", "BS_TP_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    nodes, fit, groups, pods = _distinct(bsa, config, scenario, **over)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "defaults")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 1 and st["class_mode"] == 1 and st["launches"] == 3
        ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert_batch_equal(ctx.read(), exp, "defaults, latency mode")


@pytest.mark.parametrize("share", [1, 4, 16])
@pytest.mark.parametrize("form", [0, 3, 5, 6, 7])
def test_scan_shares_in_the_throughput_regime(form, share, bsa, soa, orc, monkeypatch):
    nodes, fit, groups, pods = _distinct(bsa, "cfg3", "tail", pods=2600, groups=450, nodes=1100, classes=16)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    monkeypatch.setenv("BS_TP_SHARE", str(share))
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"form {form}, {share} shares")
        _check_split(ctx.stats(soa.STAGE_ALL), form)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("seed", [8101, 8103, 8108, 8133])        # (S = 1, 0, 2, 0; the oracle's batch B names two leaders among the evaluated pods)
def test_random_scenes_with_stale_leader_every_form(seed, form, bsa, soa, orc, monkeypatch):
    """Batch A commits and leaves a leader behind; the group state is shuffled so that batch B computes another one, and the first
    150 pods of the queue carry a lastPermittedPod entry (core.go:95-98): they pass PreFilter without reaching findMaxPG and evaluate
    Filter against the STALE leader (slot class + K) — both halves of the slot array are in use and the tile across their boundary
    holds slots of two leaders (the non-uniform case 3)."""
    rng = np.random.default_rng(seed)
    n_classes = 4
    sc = random_objects(seed, n_nodes=150 + seed % 90, n_groups=40, n_pods=1400, n_scalars=seed % 3, n_classes=n_classes)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    pods.req[0, :] += np.arange(pods.p, dtype=np.int64)
    _force_class_mode(groups, rng, n_classes)
    groups.matched[:] = rng.integers(0, 4, groups.g)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp_a = sop.batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), exp_a, "batch A")
        g2 = ctx.read_groups()
        assert g2.state_equal(sop.groups)
        new_matched = rng.integers(0, 6, groups.g).astype(np.uint32)
        for gs in (g2, sop.groups):
            gs.flags &= ~np.uint8(soa.GROUP_DENIED)
            gs.matched[:] = new_matched
        ctx.load_groups(g2)
        pods.flags[:150] |= soa.POD_LAST_PERMITTED
        ctx.load_pods(pods)
        exp_b = sop.batch(pods, soa.STAGE_ALL)
        ev = exp_b.fl_code == soa.FL_EVALUATED
        assert len(np.unique(exp_b.pf_leader[ev])) == 2, "the scene lost its second leader"
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp_b, f"batch B, form {form}")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 1, "batch B left the steady-state chain"
        assert st["launches"] == (3 if form in ONE_LAUNCH else 4)


@pytest.mark.parametrize("form", [0, 2, 4, 5, 6, 7])
def test_latency_mode_and_filter_deny_in_the_throughput_regime(form, bsa, soa, orc, monkeypatch):
    nodes, fit, groups, pods = _distinct(bsa, "cfg3", "tail", pods=2400, groups=400, nodes=900, classes=8)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp = sop.batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        for it in range(3):
            ctx.run(soa.STAGE_ALL | (soa.BATCH_HOST_RESULTS if it != 1 else 0))
            out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
            ctx.read(out=out)
            assert_batch_equal(out, exp, f"latency mode, cycle {it}, form {form}", bitmap=False)
            assert np.array_equal(out.bitmap_from_rows(), exp.fl_bitmap)
        ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert_batch_equal(ctx.read(), exp, "bitmap after latency mode")
        exp_fd = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL | soa.BATCH_FILTER_DENY, bitmap=False)
        ctx.run(soa.STAGE_ALL | soa.BATCH_FILTER_DENY)
        assert_batch_equal(ctx.read(bitmap=False, rows=False), exp_fd, f"Filter's deny entry, form {form}", bitmap=False)
        assert not np.array_equal(exp_fd.pf_code, exp.pf_code), "the scene lost its Filter-deny events"


@pytest.mark.parametrize("split", [2, 5])
@pytest.mark.parametrize("form", [5, 6, 8])
def test_item_order_by_tile_quads_on_one_context(form, split, bsa, soa, orc, monkeypatch):
    """BS_TP_SPLIT on a single context: the transposed Filter items cut finer than the launched waves and numbered tile quad by tile
    quad (a tile count that is not a multiple of four: the last quad has idle members)."""
    nodes, fit, groups, pods = _distinct(bsa, "cfg3", "tail", pods=2950, groups=500, nodes=1300, classes=16)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    monkeypatch.setenv("BS_TP_SPLIT", str(split))
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"form {form}, split {split}")
        assert ctx.stats(soa.STAGE_ALL)["chain"] == 1


@pytest.mark.parametrize("split", [0, 1, 3, 8])       # BS_TP_SPLIT: a rank cuts its Filter items finer and deals them out tile quad by tile quad (0: the library's rule)
@pytest.mark.parametrize("form", [0, 5, 6, 8])
@pytest.mark.parametrize("nranks", [2, 3])
def test_shards_in_the_throughput_regime(nranks, form, split, bsa, soa, orc, monkeypatch):
    """Pod-axis shard (bs_shard_set, the whole queue on every rank): only the pods a rank owns stamp their class / Filter slots, so a
    rank's launch B evaluates the slots of ITS pods — with distinct requests that is 1 / nranks of the work.  Every rank's owned pods
    == the single batch's, the union of the admit counters == the single batch's."""
    nodes, fit, groups, pods = _distinct(bsa, "cfg3", "tail", pods=3000, groups=500, nodes=1300, classes=16)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    monkeypatch.setenv("BS_TP_FILTER", str(form))
    if split:
        monkeypatch.setenv("BS_TP_SPLIT", str(split))
    else:
        monkeypatch.delenv("BS_TP_SPLIT", raising=False)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        admit = np.zeros(groups.g, np.uint32)
        owned = np.zeros(pods.p, np.uint32)
        for r in range(nranks):
            ctx.set_shard(r, nranks)
            ctx.run(soa.STAGE_ALL)
            part = ctx.read()
            mine = part.pf_code != 0xFF
            owned += mine
            for name in ("pf_code", "pf_first_k", "fl_code", "fl_feasible"):
                assert np.array_equal(getattr(part, name)[mine], getattr(exp, name)[mine]), (name, r)
            assert np.array_equal(part.fl_bitmap[:, mine], exp.fl_bitmap[:, mine]), r
            admit += part.group_admit
            st = ctx.stats(soa.STAGE_ALL)
            assert st["chain"] == 1
            assert st["filter_evals_executed"] < 0.75 * int((exp.fl_code == soa.FL_EVALUATED).sum()) * nodes.n, "a rank evaluated (almost) every slot"
        assert np.array_equal(owned, np.ones(pods.p, np.uint32)), "every pod is owned by exactly one rank"
        assert np.array_equal(admit, exp.group_admit)
        ctx.set_shard(0, 1)


def _golden():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "throughput_digests.json")))


@pytest.mark.parametrize("config,scenario", [("cfg3", "tail"), ("cfg3", "busy"), ("cfg4", "tail")])
def test_full_size_all_distinct_equals_the_oracles_digest(config, scenario, bsa, soa):
    """BASELINE's full sizes with every request distinct (10k x 5k: 4.8e7 evaluated pod x node pairs per batch; 50k x 20k: 9.6e8) against
    the digest of the ORACLE's batch committed under tests/golden/ (make_throughput_golden.py; the oracle needs ~3 minutes for cfg4)."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_throughput_golden as mk
    want = _golden().get(f"{config}/{scenario}/distinct")
    if want is None:
        pytest.skip("no golden digest for this scene")
    nodes, fit, groups, pods = mk.scene(bsa, config, scenario)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        got = ctx.batch(soa.STAGE_ALL, bitmap=False)
        assert mk.digest(got) == want["digest"], f"{config}/{scenario} all-distinct: the device's result arrays differ from the oracle's"
        assert int((got.fl_code == soa.FL_EVALUATED).sum()) == want["evaluated_pods"] and int(got.group_ready.sum()) == want["groups_ready"]
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 1 and st["launches"] == 3


@pytest.mark.parametrize("scalars", [5, 7])
def test_wide_contexts_take_the_split_launches(scalars, bsa, soa, orc):
    """More than four scalar lanes: the combined transposed kernel is not instantiated (it ran out of SGPRs there and reserved scratch
    memory, tools/kernel_resources.py); run_fast sends such contexts through k_fast_scan + k_fast_filter_t.  Same answers."""
    nodes, fit, groups, pods = _distinct(bsa, "cfg3", "busy", pods=2000, groups=400, nodes=1000, classes=8, scalars=scalars)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{scalars} scalar lanes, default form")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 1 and st["launches"] == 4, st          # query+tables | scan | transposed Filter | final


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_full_size_shards_equal_the_single_batch(nranks, bsa, soa):
    """BASELINE configs[3] (50k pods / 5k groups / 20k nodes) with every request distinct, pod-axis shard on ONE context: every rank's
    owned pods == the single batch's (which tests/golden/throughput_digests.json pins on the oracle), every pod owned exactly once, the
    union of the admit counters == the single batch's.  (2 / 4 / 8 ranks.)"""
    nodes, fit, groups, pods = _distinct(bsa, "cfg4", "tail")
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        full = ctx.batch(soa.STAGE_ALL, bitmap=False, rows=True)
        owned = np.zeros(pods.p, np.uint32)
        admit = np.zeros(groups.g, np.uint32)
        for r in range(nranks):
            ctx.set_shard(r, nranks)
            part = ctx.batch(soa.STAGE_ALL, bitmap=False, rows=True)
            mine = part.pf_code != 0xFF
            owned += mine
            admit += part.group_admit
            for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible"):
                assert np.array_equal(getattr(part, name)[mine], getattr(full, name)[mine]), (name, r)
            ev = mine & (part.fl_code == soa.FL_EVALUATED)
            if ev.any():                                            # the Filter rows of the owned, evaluated pods (slot numbering is the rank's own)
                assert np.array_equal(part.fl_rows[:, part.fl_slot[ev]], full.fl_rows[:, full.fl_slot[ev]]), r
        assert np.array_equal(owned, np.ones(pods.p, np.uint32)) and np.array_equal(admit, full.group_admit)
        ctx.set_shard(0, 1)
