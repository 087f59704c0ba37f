"""GPU parity: the HIP path (through the C ABI, libbsched.so) against the CPU oracle, bit for bit.

Every test here needs a real MI355X (`-m gpu`).  Nothing falls back to the CPU: Context() raises when
the library or the device is missing.
"""
import json
import os

import numpy as np
import pytest

import naive_ref as nv
from scenarios import random_objects

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_batch_equal(got, exp, what="", bitmap=True):
    for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
        a, b = getattr(got, name), getattr(exp, name)
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{what}: {name} differs at {bad[:8].tolist()} ({len(bad)} total): got {a[bad[:8]].tolist()} exp {b[bad[:8]].tolist()}")
    if bitmap and exp.fl_bitmap is not None:
        assert np.array_equal(got.fl_bitmap, exp.fl_bitmap), f"{what}: fl_bitmap differs"
        if got.fl_rows is not None:
            # what the Go plugin's Filter does: a bit test on the row of the pod's slot — no expanded bitmap involved
            assert np.array_equal(got.bitmap_from_rows(), exp.fl_bitmap), f"{what}: slot rows disagree with the reference bitmap"
            ev = got.fl_code == 3
            assert int(got.fl_rows_n[0]) <= got.fl_rows.shape[1]
            assert np.all(got.fl_slot[ev] < got.fl_rows_n[0])
            assert np.array_equal(got.fl_rows_feasible[got.fl_slot[ev]], exp.fl_feasible[ev]), f"{what}: per-row feasible counts"


def load_ctx(bsa, nodes, fit, groups, pods, **kw):
    ctx = bsa.Context(scalar_lanes=nodes.lanes - 4, **kw)
    ctx.load_nodes(nodes, fit)
    ctx.load_groups(groups)
    ctx.load_pods(pods)
    return ctx


def test_scale_kat_on_device(bsa, soa, orc):
    """int64(float32(a)*pct) on gfx950 == x86 golden vectors (core.go:656-659), via bs_node_left."""
    kat = json.load(open(os.path.join(GOLD, "f32_scale_kat.json")))["vectors"]
    by_pct = {}
    for v in kat:
        by_pct.setdefault(v["pct_bits"], []).append(v)
    rng = np.random.default_rng(99)
    extra = rng.integers(-(2 ** 62), 2 ** 62, size=20000)
    for pct_bits, vs in by_pct.items():
        pct = float(np.uint32(pct_bits).view(np.float32))
        a = np.array([v["a"] for v in vs] + extra.tolist() + [2 ** 63 - 1, -(2 ** 63), 2 ** 63 - 2 ** 38, 0], dtype=np.int64)
        n = len(a)
        alloc = np.stack([a, a[::-1], a, a[::-1]])
        nodes = soa.Nodes(alloc, np.zeros((4, n), np.int64), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8))
        fit = soa.FitMasks.from_bool(np.ones((1, n), bool))
        with bsa.Context(scalar_lanes=0) as ctx:
            ctx.load_nodes(nodes, fit)
            left, _ = ctx.node_left(0, pct)
        exp = np.array([v["out"] for v in vs], dtype=np.int64)
        assert np.array_equal(left[0, : len(vs)], exp)
        exp_all = np.array([orc.scale(int(x), pct) for x in a], dtype=np.int64)
        assert np.array_equal(left[0], exp_all)
        assert np.array_equal(left[1], exp_all[::-1])


def test_reference_core_test_vectors_on_device(bsa, soa, orc):
    v = json.load(open(os.path.join(GOLD, "core_test_vectors.json")))
    nd = v["node"]
    alloc, reqd = nv.Resource(), nv.Resource()
    alloc.Add(nd["allocatable"])
    reqd.Add(nd["requested"])
    info = nv.NodeInfo(alloc, reqd, nd["pod_count"])
    names = v["scalar_names"]
    for case in v["cases"]:
        pod = nv.Pod("u", None, case["req"])
        nodes, fit, groups, pods, _ = nv.to_soa([info], {}, [pod], names, 1)
        with bsa.Context(scalar_lanes=2) as ctx:
            ctx.load_nodes(nodes, fit)
            left, present = ctx.node_left(0, v["percent"])
            exp = v["expected_left"]
            assert left[:, 0].tolist() == [exp["cpu"], exp["memory"], exp["ephemeral-storage"], exp["pods"], exp[names[0]], exp[names[1]]]
            assert int(present[0]) == 3
            ok, fk = ctx.cluster_fits(0, v["percent"], pods.req[:, 0].tolist(), int(pods.req_present[0]))
            assert ok is case["desire"]
            assert fk == (0 if ok else soa.K_NONE)


@pytest.mark.parametrize("seed", range(60))
def test_single_queries_random(seed, bsa, soa, orc):
    sc = random_objects(seed + 700, n_nodes=int(np.random.default_rng(seed).integers(1, 200)))
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"])
    snap = orc.Snapshot(nodes, fit)
    S = len(sc["names"])
    with bsa.Context(scalar_lanes=S) as ctx:
        ctx.load_nodes(nodes, fit)
        for cls in range(sc["n_classes"]):
            for pct in (1.0, 0.7, 0.35):
                l_g, p_g = ctx.node_left(cls, pct)
                l_o, p_o = snap.node_left(cls, pct)
                assert np.array_equal(l_g, l_o) and np.array_equal(p_g, p_o)
                pre_g, pp_g, idx_g = ctx.scan_prefix(cls, pct)
                pre_o, pp_o, idx_o = snap.scan_prefix(cls, pct)
                assert np.array_equal(idx_g, idx_o)
                assert np.array_equal(pre_g, pre_o), "every prefix element must match exactly"
                assert np.array_equal(pp_g, pp_o)
            assert ctx.cluster_total(cls) == snap.cluster_total(cls)
        rng = np.random.default_rng(seed)
        for i in range(min(pods.p, 12)):
            cls = int(rng.integers(0, sc["n_classes"]))
            pct = float(rng.choice([1.0, 0.7]))
            req = (pods.req[:, i] * int(rng.integers(0, 40))).tolist()
            pres = int(pods.req_present[i])
            ok_o, fk_o, _ = snap.compare_cluster(cls, req, pres, pct)
            assert ctx.cluster_fits(cls, pct, req, pres) == (ok_o, fk_o)


def _batch_case(sc, bsa, soa, orc, commit=True, eph_gate=1):
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    snap = orc.Snapshot(nodes, fit, eph_gate=eph_gate)
    sop = orc.Sop(snap, groups)
    exp = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods, eph_gate=eph_gate) as ctx:
        got = ctx.batch(soa.STAGE_ALL)
        assert_batch_equal(got, exp)
        # what-if runs leave the loaded group state alone
        assert ctx.read_groups().state_equal(groups)
        if commit:
            got2 = ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT)
            assert_batch_equal(got2, exp)
            after = ctx.read_groups()
            assert after.state_equal(sop.groups), "committed group state must equal the sequential reference's"
    return got


@pytest.mark.parametrize("seed", range(150))
def test_batch_random_small(seed, bsa, soa, orc):
    _batch_case(random_objects(seed), bsa, soa, orc)


@pytest.mark.parametrize("seed", range(3000, 3040))
def test_batch_random_medium(seed, bsa, soa, orc):
    _batch_case(random_objects(seed, n_nodes=300, n_groups=20, n_pods=400, n_scalars=seed % 3, n_classes=4), bsa, soa, orc)


@pytest.mark.parametrize("seed", range(4000, 4010))
def test_batch_eph_gate_off(seed, bsa, soa, orc):
    _batch_case(random_objects(seed, n_nodes=60, n_groups=8, n_pods=80), bsa, soa, orc, eph_gate=0)


@pytest.mark.parametrize("scenario", ["cold", "warm", "busy", "tail"])
def test_batch_cfg2_all_scenarios(scenario, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, scenario)


@pytest.mark.parametrize("scenario,seed", [("cold", 1), ("warm", 2), ("busy", 3), ("tail", 20260921)])
def test_batch_cfg3_prefilter(scenario, seed, bsa, soa, orc):
    """BASELINE config 3 (10k pods / 2k groups / 5k nodes, 4 resource dims): admit/reject bit-identical."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", scenario, seed=seed)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    exp = sop.batch(pods, st, bitmap=False)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        got = ctx.batch(st, bitmap=False)
        assert_batch_equal(got, exp, scenario, bitmap=False)


def test_batch_cfg3_tail_with_filter(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", "tail")
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "cfg3 tail")


def test_find_max_pg_and_filter_one(bsa, soa, orc):
    for seed in range(40):
        sc = random_objects(seed + 9000, n_nodes=20)
        nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"])
        snap = orc.Snapshot(nodes, fit)
        sop = orc.Sop(snap, groups)
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            leader, fin, panic = orc.find_max_pg(groups)
            gl, gp = ctx.find_max_pg()
            assert gp == panic
            if not panic:
                assert gl == leader
            rng = np.random.default_rng(seed)
            for _ in range(10):
                i = int(rng.integers(0, pods.p))
                ld = int(rng.integers(-1, groups.g))
                k = int(rng.integers(0, nodes.n))
                if pods.group[i] == soa.POD_GROUP_MISSING:
                    continue
                exp = sop.filter_node(pods, i, ld, k)
                got = ctx.filter_one(int(pods.group[i]), pods.req[:, i].tolist(), int(pods.req_present[i]), ld, k)
                assert got[0] == exp[0]
                if exp[0] == soa.FL_EVALUATED:
                    assert got[1] == exp[1]


def test_edge_sizes(bsa, soa, orc):
    """empty and ragged inputs: zero pods, zero nodes, one node, 63/64/65 pods and nodes."""
    for n_nodes, n_pods in [(0, 5), (1, 1), (63, 65), (64, 64), (65, 63), (129, 1), (5, 0)]:
        sc = random_objects(n_nodes * 131 + n_pods, n_nodes=n_nodes, n_pods=max(n_pods, 1), n_groups=3)
        if n_pods == 0:
            sc["pods"] = []
        _batch_case(sc, bsa, soa, orc, commit=False)


def test_max_scalar_lanes(bsa, soa, orc):
    """S = 12 (BS_MAX_SCALARS): the widest row layout."""
    rng = np.random.default_rng(5)
    S, n, L = 12, 300, 16
    alloc = rng.integers(0, 1000, size=(L, n)).astype(np.int64)
    req = rng.integers(0, 800, size=(L, n)).astype(np.int64)
    ap = rng.integers(0, 1 << S, size=n).astype(np.uint32)
    rp = rng.integers(0, 1 << S, size=n).astype(np.uint32)
    nodes = soa.Nodes(alloc, req, ap, rp, np.zeros(n, np.uint8))
    fit = soa.FitMasks.from_bool(rng.random((2, n)) < 0.9)
    snap = orc.Snapshot(nodes, fit)
    with bsa.Context(scalar_lanes=S) as ctx:
        ctx.load_nodes(nodes, fit)
        for cls in (0, 1):
            a, b, c = ctx.scan_prefix(cls, 0.7)
            x, y, z = snap.scan_prefix(cls, 0.7)
            assert np.array_equal(a, x) and np.array_equal(b, y) and np.array_equal(c, z)
            for _ in range(30):
                reqv = rng.integers(-50, 4000, size=L).tolist()
                pres = int(rng.integers(0, 1 << S))
                ok, fk, _ = snap.compare_cluster(cls, reqv, pres, 0.7)
                assert ctx.cluster_fits(cls, 0.7, reqv, pres) == (ok, fk)


def _churn(bsa, soa, orc, config, scenario, rounds, events, stages, seed):
    """A stream of node events (40 % requested-update, 30 % append, 30 % stable remove — BASELINE config 5);
    after every `events` of them the batch is re-scored and compared with a full oracle recompute."""
    capi = bsa.capi
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario, seed=seed)
    rng = np.random.default_rng(seed + 11)
    L = nodes.lanes
    alloc, req = nodes.allocatable.copy(), nodes.requested.copy()
    ap, rp, fl = nodes.allocatable_present.copy(), nodes.requested_present.copy(), nodes.flags.copy()
    fitb = fit.to_bool()
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd in range(rounds):
            deltas = []
            for _ in range(events):
                kind = int(rng.choice([capi.DELTA_UPDATE, capi.DELTA_APPEND, capi.DELTA_REMOVE], p=[0.4, 0.3, 0.3]))
                n = alloc.shape[1]
                d = capi.NodeDelta()
                d.kind = kind
                if kind == capi.DELTA_REMOVE:
                    idx = int(rng.integers(0, n))
                    d.index = idx
                    alloc, req = np.delete(alloc, idx, 1), np.delete(req, idx, 1)
                    ap, rp, fl = np.delete(ap, idx), np.delete(rp, idx), np.delete(fl, idx)
                    fitb = np.delete(fitb, idx, 1)
                else:
                    src = int(rng.integers(0, n))
                    col_a, col_r = alloc[:, src].copy(), req[:, src].copy()
                    col_r[0] = int(col_a[0] * rng.random())
                    col_r[1] = int(col_a[1] * rng.random())
                    a_p, r_p = int(ap[src]), int(rp[src])
                    for j in range(L):
                        d.allocatable[j], d.requested[j] = int(col_a[j]), int(col_r[j])
                    d.allocatable_present, d.requested_present = a_p, r_p
                    d.fit_default, d.n_fit_exceptions = 1, 1
                    exc = int(rng.integers(0, fitb.shape[0]))
                    d.fit_exceptions[0] = exc
                    fcol = np.ones(fitb.shape[0], bool)
                    fcol[exc] = False
                    if kind == capi.DELTA_UPDATE:
                        idx = int(rng.integers(0, n))
                        d.index = idx
                        alloc[:, idx], req[:, idx], ap[idx], rp[idx], fl[idx] = col_a, col_r, a_p, r_p, 0
                        fitb[:, idx] = fcol
                    else:
                        alloc, req = np.concatenate([alloc, col_a[:, None]], 1), np.concatenate([req, col_r[:, None]], 1)
                        ap, rp = np.append(ap, a_p).astype(np.uint32), np.append(rp, r_p).astype(np.uint32)
                        fl = np.append(fl, 0).astype(np.uint8)
                        fitb = np.concatenate([fitb, fcol[:, None]], 1)
                deltas.append(d)
            ctx.apply_node_deltas(deltas)
            cur_nodes = soa.Nodes(alloc, req, ap, rp, fl)
            cur_fit = soa.FitMasks.from_bool(fitb)
            assert ctx.n == cur_nodes.n
            sop = orc.Sop(orc.Snapshot(cur_nodes, cur_fit), groups)
            exp = sop.batch(pods, stages, bitmap=bool(stages & soa.STAGE_FILTER))
            got = ctx.batch(stages, bitmap=bool(stages & soa.STAGE_FILTER))
            assert_batch_equal(got, exp, f"churn round {rnd}", bitmap=bool(stages & soa.STAGE_FILTER))


def test_churn_apply_equals_reload(bsa, soa, orc):
    """BASELINE config 5 in miniature: update / append / stable-remove edits, full batch compared each round."""
    _churn(bsa, soa, orc, "cfg2", "warm", rounds=8, events=25, stages=soa.STAGE_ALL, seed=5)
    _churn(bsa, soa, orc, "cfg2", "tail", rounds=4, events=40, stages=soa.STAGE_ALL, seed=6)


def test_churn_cfg3_incremental_rescore(bsa, soa, orc):
    """BASELINE config 5 at size: 10k pods / 5k nodes, 100 events between re-scores (the suffix of the node
    list from the first changed index is re-uploaded and re-derived), admit / reject identical each time."""
    _churn(bsa, soa, orc, "cfg3", "tail", rounds=5, events=100, stages=soa.STAGE_PREFILTER | soa.STAGE_TALLY, seed=7)


def test_sharded_union_equals_single(bsa, soa, orc):
    """Pod-axis sharding: ranks own whole groups, per-group admit counters are disjoint, their sum (what
    the RCCL all-reduce produces) equals the single-GPU counters and the decisions agree pod by pod."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "busy", seed=3)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        single = ctx.batch(soa.STAGE_ALL)
        for nranks in (2, 3, 8):
            admit = np.zeros(groups.g, np.uint32)
            owned = np.zeros(pods.p, np.int32)
            for r in range(nranks):
                ctx.set_shard(r, nranks)
                ctx.run(soa.STAGE_ALL)
                part = ctx.read()
                mine = part.pf_code != 0xFF
                owned += mine
                assert np.array_equal(part.pf_code[mine], single.pf_code[mine])
                assert np.array_equal(part.pf_first_k[mine], single.pf_first_k[mine])
                assert np.array_equal(part.fl_feasible[mine], single.fl_feasible[mine])
                assert np.array_equal(part.fl_bitmap[:, mine], single.fl_bitmap[:, mine])
                # no first-pod capture can occur here: the replicated mode is exact down to the stale shared leader field
                assert np.array_equal(part.pf_leader[mine], single.pf_leader[mine]) and np.array_equal(part.fl_code[mine], single.fl_code[mine])
                admit += part.group_admit
            assert np.all(owned == 1), "every pod is evaluated by exactly one rank"
            assert np.array_equal(admit, single.group_admit)
        ctx.set_shard(0, 1)


def test_native_rccl_single_rank():
    """bs_comm_init + in-library ncclAllReduce at world size 1 (the only size one GPU allows), in its own process with a
    deadline: ncclCommInitRank takes 2 s on a good day and has been seen to take 160 s on these single-GPU boxes."""
    import subprocess
    import sys
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", RCCL_MSCCL_ENABLE="0", RCCL_MSCCLPP_ENABLE="0", NCCL_IB_DISABLE="1")
    try:
        res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_worker.py")], env=env,
                             capture_output=True, text=True, timeout=40)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL bootstrap (ncclCommInitRank, 1 rank) did not finish within 40 s on this box")
    assert res.returncode == 0 and "RCCL_SINGLE_OK" in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]


def test_full_size_properties_cfg4(bsa, soa, orc):
    """BASELINE config 4 sizes (50k pods / 5k groups / 20k nodes): size-independent properties."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg4", "tail", seed=4)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        a = ctx.batch(soa.STAGE_ALL)
        b = ctx.batch(soa.STAGE_ALL)
        assert_batch_equal(a, b, "determinism")                      # idempotent / deterministic
        # feasible count == popcount of the pod's bitmap column
        lut = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint8)
        pop = lut[a.fl_bitmap.view(np.uint8)].reshape(a.fl_bitmap.shape[0], pods.p, 8).sum(axis=(0, 2), dtype=np.uint32)
        assert np.array_equal(pop, a.fl_feasible)
        # admit counters == per-group count of admitted pods; ready == quorum predicate core.go:303
        passed = (a.pf_code < 16) & (a.fl_feasible > 0) & (pods.group >= 0)
        cnt = np.bincount(pods.group[passed], minlength=groups.g).astype(np.uint32)
        assert np.array_equal(cnt, a.group_admit)
        ready = (groups.matched + cnt).astype(np.uint32) >= (groups.min_member - groups.status_scheduled).astype(np.uint32)
        assert np.array_equal(ready.astype(np.uint8), a.group_ready)
        # deny replay: behind a group's first REJECT every later pod of the group is ERR_DENIED
        rej = np.isin(a.pf_code, [soa.PF_REJECT_FIRST, soa.PF_REJECT_RESERVE])
        first_rej = np.full(groups.g, pods.p, np.int64)
        idx = np.nonzero(rej)[0]
        np.minimum.at(first_rej, pods.group[idx], idx)
        grouped = pods.group >= 0
        later = grouped & (np.arange(pods.p) > first_rej[np.maximum(pods.group, 0)])
        assert np.all(a.pf_code[later] == soa.PF_ERR_DENIED)
        assert rej.sum() == (first_rej < pods.p).sum(), "exactly one REJECT per denied group"
        # spot-check decisions against the oracle's single-query path (finishes in seconds)
        snap = orc.Snapshot(nodes, fit)
        sop = orc.Sop(snap, groups)
        rng = np.random.default_rng(0)
        for i in rng.choice(pods.p, 40, replace=False):
            code = int(a.pf_code[i])
            if code in (soa.PF_PASS_RESERVE_FITS, soa.PF_REJECT_RESERVE):
                ld = int(a.pf_leader[i])
                pre, pres = orc.pre_allocated(groups, ld, int(groups.matched[ld]), 1)
                req = [pre[j] + int(pods.req[j, i]) for j in range(5)]
                ok, fk, _ = snap.compare_cluster(int(groups.cls[ld]), req, pres | int(pods.req_present[i]), 0.7)
                assert ok == (code == soa.PF_PASS_RESERVE_FITS) and fk == int(a.pf_first_k[i])


@pytest.mark.parametrize("scenario", ["warm", "busy", "tail"])
def test_partitioned_ranks_equal_single(scenario, bsa, soa, orc):
    """Partitioned mode (bench.py --gpus N in steady state): each rank loads ONLY the pods of the groups it
    owns; codes, first_k, bitmaps per pod equal the single-context batch, the admit counters add up, and
    the quorum bits computed from the reduced counters (bs_batch_finish) equal the single-context ones."""
    import importlib
    bdist = importlib.import_module("batch-scheduler_amd.dist")
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario, seed=9)
    assert (groups.flags & soa.GROUP_HAS_POD).all()
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        single = ctx.batch(soa.STAGE_ALL)
        # the device-side ownership rule (bs_shard_set) and its host mirror agree
        ctx.set_shard(1, 3)
        dev_owned = ctx.batch(soa.STAGE_ALL).pf_code != 0xFF
        assert np.array_equal(dev_owned, bdist.owner_ranks(pods.group, groups.g, 3) == 1)
        ctx.set_shard(0, 1)
    for nranks in (2, 4):
        own = bdist.owner_ranks(pods.group, groups.g, nranks)
        admit = np.zeros(groups.g, np.uint32)
        ctxs = []
        for r in range(nranks):
            idx = np.nonzero(own == r)[0]
            sub = pods.take(idx)
            c = load_ctx(bsa, nodes, fit, groups, sub)
            c.reduce_external(True)
            part = c.batch(soa.STAGE_ALL)
            assert np.array_equal(part.pf_code, single.pf_code[idx])
            assert np.array_equal(part.pf_first_k, single.pf_first_k[idx])
            # pf_leader of a pod that returned before findMaxPG is the stale shared field (whatever the previous
            # pod left, core.go:121) — by design only defined within one context's queue; compare the others
            reached = np.isin(part.pf_code, [soa.PF_PASS_NO_MAX, soa.PF_PASS_FIRST_FITS, soa.PF_PASS_IS_MAX, soa.PF_PASS_RESERVE_FITS,
                                             soa.PF_REJECT_FIRST, soa.PF_REJECT_RESERVE])
            assert np.array_equal(part.pf_leader[reached], single.pf_leader[idx][reached])
            assert np.array_equal(part.fl_feasible, single.fl_feasible[idx])
            assert np.array_equal(part.fl_bitmap, single.fl_bitmap[:, idx])
            admit += part.group_admit
            ctxs.append(c)
        assert np.array_equal(admit, single.group_admit)
        # what the all-reduce leaves in every rank's buffer, then the quorum pass
        import ctypes
        for c in ctxs:
            ptr, n = c.admit_devptr()
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            assert hip.hipMemcpy(ctypes.c_void_p(ptr), admit.ctypes.data_as(ctypes.c_void_p), n * 4, 1) == 0   # H2D
            c.finish()
            c.sync()
            assert np.array_equal(c.read(bitmap=False).group_ready, single.group_ready)
            c.close()


# ---- early Filter (Filter on its own stream beside the node scan; rows of pods PreFilter turns down are
# voided afterwards).  Production switches it on from 2e8 pod x node pairs; here it is forced on.
@pytest.fixture
def early(monkeypatch):
    monkeypatch.setenv("BS_EARLY_FILTER_MIN", "0")


@pytest.mark.parametrize("scenario", ["warm", "busy", "tail"])
def test_early_filter_cfg2(scenario, early, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for _ in range(3):                       # back-to-back batches reuse the streams and events
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, scenario)
        # without the tally stage the rows are voided all the same
        got = ctx.batch(soa.STAGE_PREFILTER | soa.STAGE_FILTER)
        for name in ("pf_code", "fl_code", "fl_feasible", "fl_bitmap"):
            assert np.array_equal(getattr(got, name), getattr(exp, name)), name


@pytest.mark.parametrize("seed", range(5000, 5060))
def test_early_filter_random_after_commit(seed, early, bsa, soa, orc):
    """First batch commits (every group gets its pod, denials persist); the second batch then has no
    capture left and takes the early-Filter path.  Both must equal the sequential reference run twice."""
    sc = random_objects(seed, n_nodes=90 + seed % 80, n_groups=10, n_pods=150, n_scalars=seed % 3, n_classes=3)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp1 = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), exp1, "committing batch")
        # the deny entries of batch 1 would stop almost every pod: clear them on both sides (20 s later)
        g2 = ctx.read_groups()
        assert g2.state_equal(sop.groups)
        g2.flags &= ~np.uint8(soa.GROUP_DENIED)
        sop.groups.flags &= ~np.uint8(soa.GROUP_DENIED)
        ctx.load_groups(g2)
        exp2 = sop.batch(pods, soa.STAGE_ALL)
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp2, "early batch")


def test_early_filter_cfg3_tail_and_shards(early, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", "tail")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "cfg3 tail early")
        admit = np.zeros(groups.g, np.uint32)
        for r in range(2):
            ctx.set_shard(r, 2)
            ctx.run(soa.STAGE_ALL)
            part = ctx.read()
            mine = part.pf_code != 0xFF
            assert np.array_equal(part.fl_feasible[mine], exp.fl_feasible[mine])
            assert np.array_equal(part.fl_code[mine], exp.fl_code[mine])
            assert np.array_equal(part.fl_bitmap[:, mine], exp.fl_bitmap[:, mine])
            admit += part.group_admit
        assert np.array_equal(admit, exp.group_admit)
        ctx.set_shard(0, 1)


# ---- request de-duplication (identical derived requests are evaluated once).  BS_HASH_BITS=0 drops every
# hash bit from the slots, so every probe that meets an occupied slot has to compare full keys and walk on.
@pytest.fixture
def weak_hash(monkeypatch):
    monkeypatch.setenv("BS_HASH_BITS", "0")


@pytest.mark.parametrize("seed", range(6000, 6030))
def test_dedupe_with_colliding_hashes_random(seed, weak_hash, bsa, soa, orc):
    _batch_case(random_objects(seed, n_nodes=200, n_groups=16, n_pods=400, n_scalars=seed % 3, n_classes=4), bsa, soa, orc)


@pytest.mark.parametrize("scenario", ["cold", "warm", "tail"])
def test_dedupe_with_colliding_hashes_cfg2(scenario, weak_hash, early, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, scenario)
        st = ctx.stats(soa.STAGE_ALL)
        assert 0 < st["filter_distinct"] <= pods.p
        assert st["filter_evals_executed"] == st["filter_distinct"] * nodes.n


def test_dedupe_all_requests_distinct(bsa, soa, orc):
    """Worst case for the de-duplication: every pod asks for something else."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "warm")
    pods.req[0, :] += np.arange(pods.p, dtype=np.int64)          # cpu-milli differs pod by pod
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp)
        st = ctx.stats(soa.STAGE_ALL)
        evaluated = int((exp.fl_code == soa.FL_EVALUATED).sum())
        assert st["class_mode"] == 1 and st["filter_distinct"] >= evaluated      # both leader slots of a class are filled


@pytest.mark.parametrize("scenario", ["warm", "tail"])
def test_class_mode_without_fused_scan_filter(scenario, monkeypatch, bsa, soa, orc):
    """Class mode normally evaluates the Filter slots inside the scan launch (k_scan_filter); the separate
    k_filter path (used with early Filter and when slot = pod) must agree."""
    monkeypatch.setenv("BS_NO_FUSE_FILTER", "1")
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, scenario)


def test_no_schedulable_node_still_filters(bsa, soa, orc):
    """M = 0 (every node unschedulable): nothing to scan, every reserve check fails, Filter still runs for
    the pods that pass without a scan."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "warm")
    nodes.flags[:] = soa.NODE_UNSCHEDULABLE
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp)


# ---- class mode on random scenes.  Random group states rarely qualify for request classes (every group
# needs its pod and its MinResources), so they are forced here; deny flags, permitted pods, owners and
# occupancy stay as generated.  Batch A commits (leaves a leader behind), then the group state is shuffled so
# that batch B computes a different leader: pods that pass PreFilter without reaching findMaxPG (last
# permitted ones at the head of the queue) then evaluate Filter against the STALE leader (slot class + K).
def _force_class_mode(groups, rng, n_classes):
    L, g = groups.min_resources.shape
    missing = (groups.flags & soa_mod.GROUP_HAS_MINRES) == 0
    groups.min_resources[0, missing] = rng.integers(100, 4000, int(missing.sum()))
    groups.min_resources[1, missing] = rng.integers(2 ** 20, 2 ** 32, int(missing.sum()))
    groups.flags |= np.uint8(soa_mod.GROUP_HAS_POD | soa_mod.GROUP_HAS_MINRES)
    groups.cls[:] = rng.integers(0, n_classes, g)


import importlib as _il
soa_mod = _il.import_module("batch-scheduler_amd.soa")


@pytest.mark.parametrize("seed", range(8000, 8060))
def test_class_mode_random_with_stale_leader(seed, bsa, soa, orc):
    rng = np.random.default_rng(seed)
    n_classes = 3
    sc = random_objects(seed, n_nodes=60 + seed % 100, n_groups=9, n_pods=180, n_scalars=seed % 3, n_classes=n_classes)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"],
                                            denied=sc["denied"], permitted=sc["permitted"])
    _force_class_mode(groups, rng, n_classes)
    groups.matched[:] = rng.integers(0, 4, groups.g)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp_a = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT), exp_a, "batch A")
        g2 = ctx.read_groups()
        assert g2.state_equal(sop.groups)
        new_matched = rng.integers(0, 6, groups.g).astype(np.uint32)
        for gs in (g2, sop.groups):
            gs.flags &= ~np.uint8(soa.GROUP_DENIED)
            gs.matched[:] = new_matched
        ctx.load_groups(g2)
        exp_b = sop.batch(pods, soa.STAGE_ALL)
        got_b = ctx.batch(soa.STAGE_ALL)
        assert_batch_equal(got_b, exp_b, "batch B")
        st = ctx.stats(soa.STAGE_ALL)
        assert st["class_mode"] == 1 and st["filter_distinct"] <= 2 * pods.p
