"""GPU tests of the queue-resident cycle: bs_pods_apply patches pods, request classes, (group, class) pairs and per-group
minima on the device; every batch after a patch must equal the batch after a full reload of the same queue — and the
oracle's sequential PreFilter calls (core.go:88-167) on it.  Everything goes through the C ABI."""
import json
import os

import numpy as np
import pytest

import fullsize
import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import assert_batch_equal, load_ctx, _force_class_mode

pytestmark = pytest.mark.gpu
DIGESTS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")))


def random_delta(rng, cur, soa, max_events=12, novel_base=0):
    """one random delta against `cur`: stable removals, flag flips, insertions anywhere (or None = append)"""
    n_rem = int(rng.integers(0, min(max_events, cur.p) + 1))
    remove = np.sort(rng.choice(cur.p, n_rem, replace=False)).astype(np.uint32)
    stay = np.setdiff1d(np.arange(cur.p, dtype=np.uint32), remove)
    n_flag = int(rng.integers(0, min(max_events, len(stay)) + 1))
    flag_index = np.sort(rng.choice(stay, n_flag, replace=False)).astype(np.uint32) if n_flag else np.zeros(0, np.uint32)
    flag_value = (cur.flags[flag_index] ^ np.uint8(soa.POD_LAST_PERMITTED)).astype(np.uint8)
    ni = int(rng.integers(0, max_events + 1)) if cur.p else 0
    ins, at = None, None
    if ni:
        ins = cur.take(rng.integers(0, cur.p, ni))
        ins.flags[:] = rng.integers(0, 2, ni).astype(np.uint8) * np.uint8(soa.POD_LAST_PERMITTED) * (rng.random(ni) < 0.2)
        for k in np.nonzero(rng.random(ni) < 0.3)[0]:                  # requests nobody made before, sometimes shared by two new pods
            ins.req[0, k] = 77 + 13 * (novel_base + int(k) // 2)
        pn = cur.p - n_rem + ni
        if rng.random() < 0.7:
            at = np.sort(rng.choice(pn, ni, replace=False)).astype(np.uint32)
    return dict(remove=remove, flag_index=flag_index, flag_value=flag_value, insert=ins, insert_at=at)


def scene(seed, bsa, soa, steady):
    rng = np.random.default_rng(seed)
    n_classes = 3
    sc = random_objects(seed, n_nodes=40 + seed % 90, n_groups=9, n_pods=160, n_scalars=seed % 3, n_classes=n_classes)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    if steady:
        _force_class_mode(groups, rng, n_classes)
        groups.matched[:] = rng.integers(1, 4, groups.g)
    return rng, nodes, fit, groups, pods


@pytest.mark.parametrize("steady", [True, False], ids=["steady", "positional"])
@pytest.mark.parametrize("seed", range(9100, 9124))
def test_pods_apply_equals_reload_and_oracle(seed, steady, bsa, soa, orc):
    """Rounds of random deltas on random scenes: the resident queue == the host-patched queue (bs_pods_read), the batch ==
    the oracle on that queue == a context that reloads it.  steady: every batch takes the three-launch class chain;
    positional: first-pod captures and MinResources defaults are still possible (positional / general chain)."""
    rng, nodes, fit, groups, pods = scene(seed, bsa, soa, steady)
    snap = orc.Snapshot(nodes, fit)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx, load_ctx(bsa, nodes, fit, groups, pods) as ref:
        cur = pods
        for rnd in range(7):
            if rnd != 3:                                                 # round 3: patch without a batch in between (two deltas back to back)
                exp = orc.Sop(snap, groups).batch(cur, soa.STAGE_ALL)
                got = ctx.batch(soa.STAGE_ALL)
                assert_batch_equal(got, exp, f"seed {seed} round {rnd} (apply)")
                if rnd in (0, 4):
                    ref.load_pods(cur)
                    assert_batch_equal(ref.batch(soa.STAGE_ALL), exp, f"seed {seed} round {rnd} (reload)")
            d = random_delta(rng, cur, soa, novel_base=100 * rnd)
            ctx.apply_pods(**d)
            cur = cur.patched(**d)
            assert ctx.p == cur.p
            assert ctx.read_pods().equal(cur), f"seed {seed} round {rnd}: resident queue differs from the patched queue"
        applies, rederives = ctx.apply_stats()
        assert applies >= 6 and rederives == 0, "small deltas must be patched, not re-derived"


@pytest.mark.parametrize("knob", ["BS_ID_ROOM=3", "BS_SERIAL_INSERT_MAX=0", "BS_HASH_BITS=2", "BS_HASH_SLOT_BITS=2", "BS_HASH_SLOT_BITS=0"])
def test_pods_apply_rederive_paths(knob, monkeypatch, bsa, soa, orc):
    """A used-up id space and a delta too large for the insert wave take the re-derivation path; a 2-bit hash makes every
    directory probe collide, 2 (0) slot bits start every probe at one of four slots (the same slot): long probe paths, lanes
    of one insert wave ending on the same empty slot.  Results stay those of a reload."""
    k, v = knob.split("=")
    monkeypatch.setenv(k, v)
    rng, nodes, fit, groups, pods = scene(9300, bsa, soa, True)
    snap = orc.Snapshot(nodes, fit)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        cur = pods
        for rnd in range(6):
            d = random_delta(rng, cur, soa, novel_base=50 * rnd)
            ctx.apply_pods(**d)
            cur = cur.patched(**d)
            assert ctx.read_pods().equal(cur)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), orc.Sop(snap, groups).batch(cur, soa.STAGE_ALL), f"{knob} round {rnd}")
        applies, rederives = ctx.apply_stats()
        assert (rederives > 0) == (not k.startswith("BS_HASH"))


def test_pods_apply_validates_and_is_atomic(bsa, soa, orc):
    rng, nodes, fit, groups, pods = scene(9400, bsa, soa, True)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        one = pods.take([0])
        two = pods.take([0, 1])
        bad = [dict(remove=[pods.p]), dict(remove=[3, 3]), dict(remove=[4, 2]), dict(flag_index=[pods.p], flag_value=[0]),
               dict(flag_index=[2, 1], flag_value=[0, 0]), dict(insert=one, insert_at=[pods.p + 1]), dict(insert=two, insert_at=[5, 5]),
               dict(remove=[0], insert=one, insert_at=[pods.p])]
        for b in bad:
            with pytest.raises(bsa.BsError) as e:
                ctx.apply_pods(**b)
            assert e.value.status == -1, b
        assert ctx.read_pods().equal(pods), "a refused delta must leave the queue untouched"
        ctx.apply_pods()                                                   # the empty delta is fine
        assert ctx.apply_stats()[0] == 0
        # drain the queue completely, then fill it again
        ctx.apply_pods(remove=np.arange(pods.p))
        assert ctx.p == 0 and ctx.read_pods().p == 0
        out = ctx.batch(soa.STAGE_ALL, bitmap=False)
        assert out.group_admit.sum() == 0
        ctx.apply_pods(insert=pods)
        assert ctx.read_pods().equal(pods)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL), "refilled queue")


def test_nodes_assume_equals_node_update(bsa, soa, orc):
    """bs_nodes_assume (device-side scatter of new requested vectors) == bs_nodes_apply UPDATE of the same nodes == the oracle
    on the patched snapshot; validation is atomic."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail", seed=8)
    rng = np.random.default_rng(8)
    cur = nodes.copy()
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx, load_ctx(bsa, nodes, fit, groups, pods) as ref:
        for rnd in range(5):
            idx = rng.choice(cur.n, 9, replace=False)
            reqs, deltas = [], []
            for n in idx:
                cur.requested[:3, n] += rng.integers(0, 1 + cur.allocatable[:3, n] // 8)
                cur.requested[3, n] += 1
                if cur.lanes > 4:
                    cur.requested[4, n] += int(rng.integers(0, 2))
                    cur.requested_present[n] |= 1
                reqs.append((int(n), cur.requested[:, n].tolist(), int(cur.requested_present[n])))
                d = bsa.capi.NodeDelta()
                d.kind, d.index = bsa.capi.DELTA_UPDATE, int(n)
                for j in range(cur.lanes):
                    d.allocatable[j], d.requested[j] = int(cur.allocatable[j, n]), int(cur.requested[j, n])
                d.allocatable_present, d.requested_present, d.flags = int(cur.allocatable_present[n]), int(cur.requested_present[n]), int(cur.flags[n])
                fb = fit.to_bool()[:, n]
                d.fit_default = 1
                exc = np.nonzero(~fb)[0][:8]
                d.n_fit_exceptions = len(exc)
                for k, e in enumerate(exc):
                    d.fit_exceptions[k] = int(e)
                if (~fb).sum() > 8:
                    continue
                deltas.append(d)
            ctx.assume_nodes(reqs)
            exp = orc.Sop(orc.Snapshot(cur, fit), groups).batch(pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"assume round {rnd}")
            if len(deltas) == len(reqs):
                ref.apply_node_deltas(deltas)
                assert_batch_equal(ref.batch(soa.STAGE_ALL), exp, f"node update round {rnd}")
            else:
                ref.load_nodes(cur, fit)
        for bad in ([(cur.n, [0] * cur.lanes, 0)], [(1, [0] * cur.lanes, 0), (1, [0] * cur.lanes, 0)], [(0, [0] * cur.lanes, 1 << (cur.lanes - 4))]):
            with pytest.raises(bsa.BsError) as e:
                ctx.assume_nodes(bad)
            assert e.value.status == -1
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), orc.Sop(orc.Snapshot(cur, fit), groups).batch(pods, soa.STAGE_ALL), "after refused requests")
        # a later list surgery starts from the assumed state (host mirror kept in step)
        d = bsa.capi.NodeDelta()
        d.kind, d.index = bsa.capi.DELTA_REMOVE, 0
        ctx.apply_node_deltas([d])
        cut = soa.Nodes(cur.allocatable[:, 1:], cur.requested[:, 1:], cur.allocatable_present[1:], cur.requested_present[1:], cur.flags[1:])
        cfit = soa.FitMasks.from_bool(fit.to_bool()[:, 1:])
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), orc.Sop(orc.Snapshot(cut, cfit), groups).batch(pods, soa.STAGE_ALL), "remove after assume")


def test_pods_apply_with_group_and_node_changes_in_between(bsa, soa, orc):
    """One scheduling cycle after another the way a shim drives them: group counters, node requests and the queue all move."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail", seed=5)
    stream = fullsize.PodChurnStream(pods, 5)
    rng = np.random.default_rng(5)
    cur_groups = groups.copy()
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd in range(8):
            idx = rng.choice(groups.g, 10, replace=False)
            deltas = []
            for i in idx:
                cur_groups.matched[i] = rng.integers(0, cur_groups.min_member[i] + 1)
                cur_groups.flags[i] = (cur_groups.flags[i] & 0x6) | (8 * int(rng.integers(0, 2)))
                deltas.append((i, cur_groups.matched[i], cur_groups.status_scheduled[i], cur_groups.flags[i]))
            ctx.apply_group_deltas(deltas)
            ctx.apply_pods(**stream.next_delta(40))
            if rnd == 4:                                                 # G changes: pairs and per-group minima are re-derived from the resident queue
                cur_groups = soa.Groups(*[np.concatenate([getattr(cur_groups, k), getattr(cur_groups, k)[..., :1]], axis=-1) for k in
                                          ("min_member", "status_scheduled", "matched", "flags", "cls", "min_resources", "min_resources_present", "occupied_by")])
                ctx.load_groups(cur_groups)
            exp = orc.Sop(orc.Snapshot(nodes, fit), cur_groups).batch(stream.pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"cycle {rnd}")


def test_pod_churn_stream_10000_events(bsa, soa, orc):
    """cfg3 + 10 000 pod events, a re-score after every 100 through bs_pods_apply: 100 incremental re-scores, each equal to
    a full oracle recompute on the patched queue (digest of every output array; every 10th round re-computed live)."""
    c = DIGESTS["pod_churn"]["params"]
    assert c == fullsize.POD_CHURN
    nodes, fit, groups, pods, _ = bsa.synth.make(c["config"], c["scenario"], seed=c["seed"])
    stream = fullsize.PodChurnStream(pods, c["seed"])
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd, want in enumerate(DIGESTS["pod_churn"]["rounds"]):
            ctx.apply_pods(**stream.next_delta(c["events"]))
            assert ctx.p == want["p"]
            got = ctx.batch(st, bitmap=False)
            have = fullsize.sha(np.concatenate([getattr(got, a).view(np.uint8).ravel() for a in fullsize.ARRAYS]))
            if have != want["all"] or rnd % 10 == 9:
                exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(stream.pods, st, bitmap=False)
                assert_batch_equal(got, exp, f"pod churn round {rnd}", bitmap=False)
            assert have == want["all"], f"pod churn round {rnd}: digest differs although the live oracle agrees -> stale golden file"
        assert ctx.read_pods().equal(stream.pods)
        assert ctx.stats(st)["fast_path"] == 1, "the patched queue must still take the three-launch steady-state chain"
        applies, rederives = ctx.apply_stats()
        assert applies == len(DIGESTS["pod_churn"]["rounds"]) and rederives <= 2


@pytest.mark.parametrize("config,scenario", [("cfg2", "cold"), ("tiny", "cold"), ("cfg2", "tail")])
def test_latency_mode_on_both_chains_with_resident_queue(config, scenario, bsa, soa, orc):
    """BS_BATCH_HOST_RESULTS on the positional chain too (cold: first-pod captures), and combined with bs_pods_apply: the
    cycle a shim runs in latency mode — patch the queue, run, poll the completion word — gives the bits of the plain calls."""
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    stream = fullsize.PodChurnStream(pods, 3)
    snap = orc.Snapshot(nodes, fit)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for it in range(5):
            if it:
                ctx.apply_pods(**stream.next_delta(20))
            exp = orc.Sop(snap, groups).batch(stream.pods, soa.STAGE_ALL)
            ctx.run(soa.STAGE_ALL | (soa.BATCH_HOST_RESULTS if it != 3 else 0))
            out = soa.BatchOut.alloc(stream.pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
            ctx.read(out=out)
            assert_batch_equal(out, exp, f"{config}/{scenario} latency cycle {it}", bitmap=False)
            assert np.array_equal(out.bitmap_from_rows(), exp.fl_bitmap)
        st = ctx.stats(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        assert st["chain"] == (2 if scenario == "cold" else 1)


@pytest.mark.parametrize("config,scenario", [("cfg2", "tail"), ("cfg2", "cold"), ("tiny", "warm")])
def test_batch_map_is_the_read_without_a_copy(config, scenario, bsa, soa, orc):
    """bs_batch_map: the pointers into the pinned result memory of a latency-mode batch hold exactly what bs_batch_read copies
    out (both chains, resident queue patched in between), stay unchanged until the next run, and the call refuses batches
    that did not write host results."""
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    stream = fullsize.PodChurnStream(pods, 5)
    snap = orc.Snapshot(nodes, fit)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.run(soa.STAGE_ALL)
        with pytest.raises(bsa.BsError) as e:
            ctx.map_results()
        assert e.value.status == -4                                              # BS_ERR_STATE: use bs_batch_read
        for it in range(4):
            if it:
                ctx.apply_pods(**stream.next_delta(15))
            exp = orc.Sop(snap, groups).batch(stream.pods, soa.STAGE_ALL)
            ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
            v = ctx.map_results()
            out = soa.BatchOut.alloc(stream.pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
            ctx.read(out=out)
            assert v["p"] == stream.pods.p and v["g"] == groups.g
            for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
                assert np.array_equal(v[name], getattr(out, name)), f"{name} (cycle {it})"
                assert np.array_equal(v[name], getattr(exp, name)), f"{name} vs oracle (cycle {it})"
            ev = out.fl_code == 3
            assert np.array_equal(v["fl_slot"][ev], out.fl_slot[ev])
            n = int(out.fl_rows_n[0])
            assert v["fl_rows_n"] == n
            if n:
                assert np.array_equal(v["fl_rows"], out.fl_rows[:, :n]) and np.array_equal(v["fl_rows_feasible"], out.fl_rows_feasible[:n])
                # the Filter answer of a pod on a node is a bit test in the mapped rows
                rng = np.random.default_rng(it)
                for _ in range(200):
                    p_, n_ = int(rng.integers(0, stream.pods.p)), int(rng.integers(0, nodes.n))
                    if v["fl_code"][p_] == 3:
                        got = bool((int(v["fl_rows"][n_ >> 6, v["fl_slot"][p_]]) >> (n_ & 63)) & 1)
                        assert got == bool((int(exp.fl_bitmap[n_ >> 6, p_]) >> (n_ & 63)) & 1)
        # prefilter + tally only: no rows, the rest as usual
        ctx.run(soa.STAGE_PREFILTER | soa.STAGE_TALLY | soa.BATCH_HOST_RESULTS)
        v = ctx.map_results()
        e2 = orc.Sop(snap, groups).batch(stream.pods, soa.STAGE_PREFILTER | soa.STAGE_TALLY)
        assert v["fl_rows"] is None and np.array_equal(v["pf_code"], e2.pf_code) and np.array_equal(v["group_ready"], e2.group_ready)
        # a sharded context never writes host results
        ctx.set_shard(0, 2)
        ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
        ctx.finish()
        with pytest.raises(bsa.BsError):
            ctx.map_results()


def test_deferred_group_patch_reaches_every_reader(bsa, soa, orc):
    """bs_groups_apply leaves its launch to the next call (a bs_pods_apply of the same cycle takes the patch along in its own
    launch): whatever comes next — another patch, findMaxPG, a read-back, a batch, a queue patch, an empty queue patch — sees
    the patched groups."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail", seed=11)
    stream = fullsize.PodChurnStream(pods, 11)
    rng = np.random.default_rng(11)
    cur = groups.copy()
    snap = orc.Snapshot(nodes, fit)

    def patch(ctx, n=6):
        deltas = []
        for i in rng.choice(cur.g, n, replace=False):
            cur.matched[i] = rng.integers(0, cur.min_member[i] + 2)
            cur.status_scheduled[i] = rng.integers(0, 2)
            deltas.append((int(i), int(cur.matched[i]), int(cur.status_scheduled[i]), int(cur.flags[i])))
        ctx.apply_group_deltas(deltas)

    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd in range(12):
            patch(ctx)
            kind = rnd % 6
            if kind == 0:                                                  # patch -> patch -> queue patch
                patch(ctx)
                ctx.apply_pods(**stream.next_delta(10))
            elif kind == 1:                                                # patch -> findMaxPG
                leader, panic = ctx.find_max_pg()
                exp_leader, _fin, exp_panic = orc.find_max_pg(cur)
                assert (leader, panic) == (exp_leader, exp_panic)
            elif kind == 2:                                                # patch -> read-back
                back = ctx.read_groups()
                assert np.array_equal(back.matched, cur.matched) and np.array_equal(back.status_scheduled, cur.status_scheduled)
            elif kind == 3:                                                # patch -> empty queue patch -> batch
                ctx.apply_pods()
            elif kind == 4:                                                # patch -> queue patch -> queue patch
                ctx.apply_pods(**stream.next_delta(10))
                ctx.apply_pods(**stream.next_delta(5))
            exp = orc.Sop(snap, cur).batch(stream.pods, soa.STAGE_ALL)
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"round {rnd} kind {kind}")


@pytest.mark.parametrize("n_ins,n_rem", [(700, 300), (1500, 9000), (3000, 100)])
def test_pods_apply_deltas_beyond_the_lds_window(n_ins, n_rem, bsa, soa, orc):
    """Deltas whose blob (700 inserted pods: > 32 KB), whose index lists (9000 removals: > 32 KB) or whose insert count (3000: the
    insert wave is not asked, everything is re-derived) exceed what a block stages in LDS are read in place: same queue, same
    batch as a reload."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", "tail", seed=3)
    rng = np.random.default_rng(n_ins)
    snap = orc.Snapshot(nodes, fit)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        cur = pods
        for rnd in range(2):
            remove = np.sort(rng.choice(cur.p, min(n_rem, cur.p), replace=False)).astype(np.uint32)
            ins = cur.take(rng.integers(0, cur.p, n_ins))
            ins.req[0, ::7] += 1 + rnd                                        # some requests nobody made before
            pn = cur.p - len(remove) + n_ins
            at = np.sort(rng.choice(pn, n_ins, replace=False)).astype(np.uint32) if rnd else None
            d = dict(remove=remove, insert=ins, insert_at=at)
            ctx.apply_pods(**d)
            cur = cur.patched(**d)
            assert ctx.read_pods().equal(cur)
            got = ctx.batch(soa.STAGE_PREFILTER | soa.STAGE_TALLY, bitmap=False)
            exp = orc.Sop(snap, groups).batch(cur, soa.STAGE_PREFILTER | soa.STAGE_TALLY)
            for name in ("pf_code", "pf_first_k", "pf_leader", "group_admit", "group_ready"):
                assert np.array_equal(getattr(got, name), getattr(exp, name)), f"{name} round {rnd}"
