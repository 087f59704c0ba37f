"""GPU tests of the steady-state fast path (three launches), the per-cycle group patches, the Filter slot rows
and the argument checks added with them.  Everything goes through the C ABI; the oracle is only the checker."""
import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import assert_batch_equal, load_ctx, _force_class_mode

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,scenario", [("cfg2", "warm"), ("cfg2", "busy"), ("cfg2", "tail"), ("tiny", "warm"), ("tiny", "tail")])
def test_fast_path_is_taken_and_equals_general_chain(config, scenario, monkeypatch, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for _ in range(3):                                         # nothing is reset between batches: stamps / inverted sequence keys
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{config}/{scenario} fast")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["fast_path"] == 1 and st["launches"] == 2 and st["class_mode"] == 1
        assert 0 < st["filter_distinct"] and st["filter_evals_executed"] == st["filter_distinct"] * nodes.n
        for stages in (soa.STAGE_PREFILTER, soa.STAGE_PREFILTER | soa.STAGE_TALLY, soa.STAGE_PREFILTER | soa.STAGE_FILTER):
            e = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, stages)
            g = ctx.batch(stages)
            for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "fl_bitmap"):
                assert np.array_equal(getattr(g, name), getattr(e, name)), (stages, name)
            if stages & soa.STAGE_TALLY:
                assert np.array_equal(g.group_admit, e.group_admit) and np.array_equal(g.group_ready, e.group_ready)
    monkeypatch.setenv("BS_NO_FAST", "1")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{config}/{scenario} general")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["fast_path"] == 0 and st["launches"] > 3


@pytest.mark.parametrize("seed", range(8100, 8140))
def test_fast_path_random_scenes_with_commit(seed, bsa, soa, orc):
    """Random scenes forced into the steady state (every group has its pod and MinResources, the leader has
    matched pods): deny flags, permitted pods, owners and occupancy as generated; the first batch commits
    (OccupiedBy, deny entries, the carried leader), the second runs against a reshuffled group state."""
    rng = np.random.default_rng(seed)
    n_classes = 3
    sc = random_objects(seed, n_nodes=50 + seed % 120, n_groups=9, n_pods=200, n_scalars=seed % 3, n_classes=n_classes)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    _force_class_mode(groups, rng, n_classes)
    groups.matched[:] = rng.integers(1, 4, groups.g)               # whoever leads has matched pods: reservation checks only
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp_a = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        st = ctx.stats(soa.STAGE_ALL)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), exp_a, "batch A")
        g2 = ctx.read_groups()
        assert g2.state_equal(sop.groups), "committed group state must equal the sequential reference's"
        if st["fast_path"]:
            # the group state changes through bs_groups_apply (what a scheduling cycle does), not a reload
            new_matched = rng.integers(1, 6, groups.g).astype(np.uint32)
            new_flags = g2.flags & ~np.uint8(soa.GROUP_DENIED)
            sop.groups.flags[:] = new_flags
            sop.groups.matched[:] = new_matched
            ctx.apply_group_deltas([(i, new_matched[i], g2.status_scheduled[i], new_flags[i]) for i in range(groups.g)])
            exp_b = sop.batch(pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp_b, "batch B")
            assert ctx.read_groups().state_equal(_with(g2, matched=new_matched, flags=new_flags))


def _with(groups, **kw):
    g = groups.copy()
    for k, v in kw.items():
        getattr(g, k)[:] = v
    return g


def test_groups_apply_equals_reload_and_validates(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "warm", seed=11)
    rng = np.random.default_rng(3)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx, load_ctx(bsa, nodes, fit, groups, pods) as ref:
        cur = groups.copy()
        for rnd in range(6):
            idx = rng.choice(groups.g, 12, replace=False)
            deltas = []
            for i in idx:
                cur.matched[i] = rng.integers(0, cur.min_member[i] + 1)
                cur.status_scheduled[i] = rng.integers(0, 3)
                cur.flags[i] = (cur.flags[i] & 0x6) | int(rng.integers(0, 2)) | (8 * int(rng.integers(0, 2)))
                deltas.append((i, cur.matched[i], cur.status_scheduled[i], cur.flags[i]))
            ctx.apply_group_deltas(deltas)
            ref.load_groups(cur)
            exp = orc.Sop(orc.Snapshot(nodes, fit), cur).batch(pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"apply round {rnd}")
            assert_batch_equal(ref.batch(soa.STAGE_ALL), exp, f"reload round {rnd}")
            assert ctx.read_groups().state_equal(cur)
        # a bad delta anywhere in the list leaves everything untouched
        before = ctx.read_groups()
        for bad in ([(0, 1, 0, int(cur.flags[0])), (groups.g, 0, 0, 0)],                   # index out of range
                    [(1, 5, 0, int(cur.flags[1]) & ~soa.GROUP_HAS_POD)],                   # HAS_POD may not change
                    [(2, 5, 0, 0x100)]):                                                   # not a flag byte
            with pytest.raises(bsa.BsError) as e:
                ctx.apply_group_deltas(bad)
            assert e.value.status == -1
        assert ctx.read_groups().state_equal(before)


def test_load_order_is_free_and_class_indices_are_checked(bsa, soa, orc):
    """pods before groups on a fresh context, and fit-class indices out of range refuse the batch
    instead of faulting the GPU."""
    nodes, fit, groups, pods, _ = bsa.synth.make("tiny", "warm")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_pods(pods)
        ctx.load_groups(groups)
        ctx.load_nodes(nodes, fit)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "pods, groups, nodes")
        bad = pods.copy()
        bad.cls[5] = fit.n_classes
        ctx.load_pods(bad)
        with pytest.raises(bsa.BsError) as e:
            ctx.run(soa.STAGE_ALL)
        assert e.value.status == -1
        ctx.load_pods(pods)
        gbad = groups.copy()
        gbad.cls[0] = fit.n_classes + 7
        ctx.load_groups(gbad)
        with pytest.raises(bsa.BsError) as e:
            ctx.run(soa.STAGE_ALL)
        assert e.value.status == -1
        ctx.load_groups(groups)
        # fewer classes after the groups / pods are in: same refusal
        ctx.load_fit(soa.FitMasks(fit.bits[:1].copy(), fit.n))
        if int(groups.cls.max()) >= 1 or int(pods.cls.max()) >= 1:
            with pytest.raises(bsa.BsError):
                ctx.run(soa.STAGE_ALL)
        ctx.load_fit(fit)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "after the refusals")
        with pytest.raises(bsa.BsError) as e:
            ctx.filter_one(0, pods.req[:, 0].tolist(), 0, groups.g, 0)
        assert e.value.status == -1


def test_nodes_apply_is_atomic_on_error(bsa, soa, orc):
    capi = bsa.capi
    nodes, fit, groups, pods, _ = bsa.synth.make("tiny", "tail")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        good = capi.NodeDelta()
        good.kind, good.index = capi.DELTA_REMOVE, 3
        for bad_kind, bad_index, exc in ((capi.DELTA_REMOVE, 10 ** 6, 0), (capi.DELTA_UPDATE, 10 ** 6, 0), (7, 0, 0), (capi.DELTA_UPDATE, 1, 9)):
            bad = capi.NodeDelta()
            bad.kind, bad.index, bad.n_fit_exceptions = bad_kind, bad_index, exc
            with pytest.raises(bsa.BsError) as e:
                ctx.apply_node_deltas([good, bad])                # the valid REMOVE in front must not stick
            assert e.value.status == -1
            assert ctx.n == nodes.n
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "after a refused delta list")


def test_filter_rows_capacity_and_rows_only_read(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.run(soa.STAGE_ALL)
        need = ctx.filter_rows_count()
        assert 0 < need <= 2 * pods.p
        out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=need)      # rows without the expanded bitmap
        ctx.read(out=out)
        assert np.array_equal(out.bitmap_from_rows(), exp.fl_bitmap)
        rng = np.random.default_rng(0)
        for _ in range(200):
            p_, n_ = int(rng.integers(0, pods.p)), int(rng.integers(0, nodes.n))
            assert out.node_passes(p_, n_) == bool((int(exp.fl_bitmap[n_ >> 6, p_]) >> (n_ & 63)) & 1)
        small = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(1, int(out.fl_rows_n[0]) - 1))
        with pytest.raises(bsa.BsError) as e:
            ctx.read(out=small)
        assert e.value.status == -5 and int(small.fl_rows_n[0]) == int(out.fl_rows_n[0])


@pytest.mark.parametrize("config,scenario,distinct", [("cfg2", "tail", False), ("cfg2", "warm", False), ("tiny", "warm", False), ("cfg2", "busy", True)])
def test_latency_mode_zero_copy_in_and_out(config, scenario, distinct, bsa, soa, orc):
    """bs_pods_map (the queue is marshalled straight into the pinned upload buffer) + BS_BATCH_HOST_RESULTS (the last launch
    writes the results into pinned host memory, bs_batch_read polls a completion word: no copy, no stream wait) give the
    same bits as the ordinary calls; `distinct`: more Filter rows than the pinned window holds -> the rows take the copy path."""
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    if distinct:
        pods.req[0, :] += np.arange(pods.p, dtype=np.int64)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        for it in range(4):
            view = ctx.map_pods(pods.p)
            for name in ("group", "req", "req_present", "cls", "owner", "flags"):
                getattr(view, name)[...] = getattr(pods, name)
            ctx.load_pods(view)
            ctx.run(soa.STAGE_ALL | (soa.BATCH_HOST_RESULTS if it != 2 else 0))      # one ordinary batch in between
            out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
            ctx.read(out=out)
            assert_batch_equal(out, exp, f"latency mode, cycle {it}", bitmap=False)
            assert np.array_equal(out.bitmap_from_rows(), exp.fl_bitmap)
            assert np.array_equal(out.fl_rows_feasible[out.fl_slot[out.fl_code == 3]], exp.fl_feasible[out.fl_code == 3])
        # the expanded bitmap can still be had after a latency-mode batch
        ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert_batch_equal(ctx.read(), exp, "bitmap after latency mode")
        st = ctx.stats(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert st["fast_path"] == 1
    # a batch that is not on the steady-state chain ignores the flag
    n2, f2, g2, p2, _ = bsa.synth.make("tiny", "cold")
    e2 = orc.Sop(orc.Snapshot(n2, f2), g2).batch(p2, soa.STAGE_ALL)
    with load_ctx(bsa, n2, f2, g2, p2) as ctx:
        ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert_batch_equal(ctx.read(), e2, "cold batch with the latency flag")


@pytest.mark.parametrize("config,scenario", [("cfg2", "tail"), ("cfg2", "warm"), ("cfg3", "tail")])
def test_unfused_final_launch_is_selectable_and_equal(config, scenario, bsa, soa, orc, monkeypatch):
    """BS_NO_FUSE_FINAL=1: the final blocks run as their own launch (k_fast_scan_filter + k_fast_final) instead of waiting for the
    producer blocks inside one launch — the form the library also falls back to by itself when the fused grid would not be resident
    at once.  Same results, three launches."""
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL, bitmap=False)
    monkeypatch.setenv("BS_NO_FUSE_FINAL", "1")
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        st = ctx.stats(soa.STAGE_ALL)
        assert st["chain"] == 1 and st["launches"] == 3
        for mode in (soa.STAGE_ALL, soa.STAGE_ALL | soa.BATCH_HOST_RESULTS):
            ctx.run(mode)
            got = ctx.read(bitmap=False, rows=False)
            for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
                assert np.array_equal(getattr(got, name), getattr(exp, name)), name
    monkeypatch.delenv("BS_NO_FUSE_FINAL")
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        assert ctx.stats(soa.STAGE_ALL)["launches"] == 2          # the fused form: the whole grid is resident on this chip


@pytest.mark.parametrize("form", ["1", "2", "3"])
@pytest.mark.parametrize("config,scenario", [("cfg2", "warm"), ("cfg2", "tail"), ("cfg3", "tail"), ("tiny", "busy")])
def test_one_launch_form_of_the_step_equals_the_oracle(config, scenario, form, monkeypatch, bsa, soa, orc):
    """BS_STEP_A=1 (round 5's experiment, off by default because it measured slower): launch A and the scan / Filter roles of launch B as
    ONE launch — the block that builds a table chunk keeps its rows in registers and scans them itself, slots handed over inside the
    launch (k_fast_step_a) — then k_fast_final.  BS_STEP_A=2 (round 6): the slots come from the class directory (class_slots_block), the pod
    blocks publish nothing.  BS_STEP_A=3 (the default): the whole step in that one launch — table and Filter blocks derive their slots themselves, the
    pod blocks finish their own pods after an in-launch hand-over, no k_fast_final.  Same answers, batch after batch, with queue patches in between (pods leaving; pods with NEW requests arriving:
    class ids the directory learns from the insert wave)."""
    monkeypatch.setenv("BS_STEP_A", form)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for _ in range(4):
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"{config}/{scenario} one-launch step")
        assert ctx.stats(soa.STAGE_ALL)["launches"] == 2
        rem = np.arange(0, pods.p, 7, dtype=np.uint32)
        ctx.apply_pods(remove=rem)
        left = pods.take(np.setdiff1d(np.arange(pods.p), rem))
        exp2 = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(left, soa.STAGE_ALL)
        for _ in range(2):
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp2, f"{config}/{scenario} one-launch step after a queue patch")
        new = left.take(np.arange(min(6, left.p)))
        new.req[0, :] += 13 + np.arange(new.p)                      # requests nobody had: new class ids
        ctx.apply_pods(insert=new)
        grown = left.patched(insert=new, insert_at=None)
        exp3 = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(grown, soa.STAGE_ALL)
        for _ in range(2):
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp3, f"{config}/{scenario} one-launch step after new classes arrived")


@pytest.mark.gpu
def test_a_timed_out_hand_over_voids_the_batch_and_the_context_goes_back_to_separate_launches(monkeypatch, bsa, soa, orc):
    """The in-launch waits of the one-launch step are bounded (bs_fast.hpp, kSpinBound): a block whose wait runs out raises the context's error word.  The
    host's side of that, with the word raised by a test hook instead of a stuck GPU: the batch is void (BS_ERR_RETRY from the call that would have handed
    its results out), the next bs_batch_run answers like the oracle again, and the context no longer takes the one-launch forms."""
    monkeypatch.setenv("BS_TEST_HANDOVER_TIMEOUT", "2")            # the second one-launch step "times out"
    monkeypatch.delenv("BS_STEP_A", raising=False)                  # (the library's default forms: a suite run under BS_STEP_A=0 has no one-launch step to time out)
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        # Which batch is the second one-launch step depends on a race the library leaves open on purpose: the FIRST batch over a fresh queue takes the
        # one-launch form only if the pod load's class count has already landed on the host (it never waits for it) — so the void batch is the second
        # or the third.  Every batch in front of it answers like the oracle.
        void_at = None
        for k in range(3):
            try:
                got = ctx.batch(soa.STAGE_ALL)
            except bsa.capi.BsError as e:
                assert e.status == -8                               # BS_ERR_RETRY (include/bsched.h)
                void_at = k
                break
            assert_batch_equal(got, exp, f"batch {k} (in front of the time-out)")
        assert void_at in (1, 2), void_at
        for _ in range(3):
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "after the time-out")
