"""What bs_batch_run's admit / ready mean in terms of the sequential reference (GPU).

The batch answers, for every pod of the queue, "what would PreFilter return if the pods were offered in queue
order against the FROZEN node snapshot and group counters" (in-batch captures, OccupiedBy and deny entries
replayed).  The sequential reference additionally ASSUMES every admitted pod on a node (less headroom for the
pods behind it) and PERMITS it (matched counters move, findMaxPG may elect another leader).  Relations:

  R1  equality   batch codes == the host mirror's PreFilter calls in queue order when nothing is assumed or
                 permitted in between (two different device paths: bs_batch_run vs bs_find_max_pg +
                 bs_cluster_fits one pod at a time, with the real TTL deny cache).
  R2  inclusion  with Filter off (the shipped configuration), a leader that keeps leading and has no pod in
                 the queue, and no assumed pod creating a scalar key on a node:
                     sequential pass(i)  =>  batch pass(i)          for every pod i
                 (headroom only shrinks, so a scan that passes later passes on the frozen snapshot; by
                 induction over the queue the deny entries follow) — hence admit_seq <= admit_batch per group
                 and every gang the sequential run releases is ready in the batch.  On a tight cluster the
                 inclusion is strict: the batch over-admits, it is a pre-screen, not a reservation.
  R1F equality, Filter on   batch (PREFILTER|FILTER|TALLY|BS_BATCH_FILTER_DENY: Filter's deny entry, core.go:183-185, replayed
                 inside the batch on the device — round 3 needed a host-side pass for it) == the host mirror's PreFilter
                 followed by Filter on EVERY node, pod by pod, with the real TTL deny cache.
  R3  the README race (README.md:78-188): frozen, both gangs of 5 fit the node on their own -> the batch
      reports both ready; sequentially the second gang is denied.  The canonical over-admission.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GI = 1 << 30


def _scene(soa, n_nodes, seed, n_groups=24, size=5, cpu_choices=(1000, 1500, 2000)):
    rng = np.random.default_rng(seed)
    alloc = np.zeros((4, n_nodes), np.int64)
    alloc[0], alloc[1], alloc[3] = 8000, 32 * GI, 40
    req = np.zeros((4, n_nodes), np.int64)
    req[0] = rng.integers(0, 3000, n_nodes)
    req[1] = rng.integers(0, 8, n_nodes) * GI
    req[3] = rng.integers(0, 10, n_nodes)
    nodes = soa.Nodes(alloc, req, np.zeros(n_nodes, np.uint32), np.zeros(n_nodes, np.uint32), np.zeros(n_nodes, np.uint8))
    fit = soa.FitMasks.from_bool(np.ones((2, n_nodes), bool))
    gcpu = rng.choice(list(cpu_choices), n_groups)
    gmem = rng.choice([1, 2, 4], n_groups) * GI
    order = rng.permutation(n_groups * size)
    pgroup = (order // size).astype(np.int32) + 1              # group 0 is the leader: none of its pods is in the queue
    preq = np.zeros((4, len(order)), np.int64)
    preq[0], preq[1] = gcpu[pgroup - 1], gmem[pgroup - 1]
    pods = soa.Pods(pgroup, preq, np.zeros(len(order), np.uint32), np.zeros(len(order), np.uint32), np.zeros(len(order), np.uint64),
                    np.zeros(len(order), np.uint8))
    return nodes, fit, pods, gcpu, gmem, size


def _mirror_with_leader(bsa, ctx, gcpu, gmem, size):
    """PodGroup cache as the per-pod entry points build it: the leader (group 0, MinMember 10) has its pod and 9
    permitted members; the other gangs are known to the controller (MinResources set) but have not been seen."""
    sop = bsa.plugin.ScheduleOperation(ctx)
    assert sop.add_group(10, min_resources=[1000, GI, 0, 0], name_rank=0) == 0
    for g in range(len(gcpu)):
        assert sop.add_group(size, min_resources=[int(gcpu[g]), int(gmem[g]), 0, 0], name_rank=g + 1) == g + 1
    code, _ = sop.PreFilter(10 ** 6, 10 ** 6, 0, [1000, GI, 0, 0])          # first-pod capture of the leader
    assert code < 16
    for k in range(9):
        assert sop.Permit(10 ** 6 + k, 10 ** 6 + k, 0, 0) == (False, 1)     # ErrorWaiting: 9 of 10
    sop.sync()
    return sop


def _sequential(bsa, soa, nodes, fit, pods, gcpu, gmem, size, assume):
    """the reference's loop through the host mirror: PreFilter -> (plain capacity node pick -> assume -> Permit ->
    release) per pod; with assume=False only the PreFilter calls."""
    capi = bsa.capi
    codes, fks = [], []
    released = set()
    alloc, req = nodes.allocatable, nodes.requested.copy()
    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = _mirror_with_leader(bsa, ctx, gcpu, gmem, size)
        ctx.g = len(gcpu) + 1                                      # the mirror loaded the groups through the C ABI directly
        groups0 = ctx.read_groups()
        for i in range(pods.p):
            g = int(pods.group[i])
            rq = pods.req[:, i]
            code, fk = sop.PreFilter(i + 1, i + 1, g, rq.tolist())
            codes.append(code)
            fks.append(fk)
            if code >= 16 or not assume:
                continue
            ok = np.nonzero(np.all(alloc[:3] - req[:3] >= rq[:3, None], axis=0) & (alloc[3] - req[3] >= 1))[0]
            if not len(ok):
                continue                                           # the default scheduler finds no node: pod stays pending
            k = int(ok[0])
            req[:3, k] += rq[:3]
            req[3, k] += 1
            d = capi.NodeDelta()
            d.kind, d.index = capi.DELTA_UPDATE, k
            for j in range(4):
                d.allocatable[j], d.requested[j] = int(alloc[j, k]), int(req[j, k])
            d.fit_default = 1
            ctx.apply_node_deltas([d])                             # assume: NodeInfo.requested += pod
            ready, _ = sop.Permit(i + 1, i + 1, g, k)
            if ready:
                for _ in sop.StartBatchSchedule(g):
                    sop.PostBind(g)
                released.add(g)
            assert ctx.find_max_pg()[0] == 0, "the regime of R2: group 0 keeps leading"
        sop.close()
    return np.array(codes, np.uint8), np.array(fks, np.uint32), released, groups0


def _batch(bsa, soa, nodes, fit, groups0, pods):
    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups0)
        ctx.load_pods(pods)
        return ctx.batch(soa.STAGE_PREFILTER | soa.STAGE_TALLY, bitmap=False)


@pytest.mark.parametrize("n_nodes,seed,expect_strict", [(36, 1, True), (40, 2, True), (160, 3, False)])
def test_batch_admit_vs_sequential_mirror(n_nodes, seed, expect_strict, bsa, soa, orc):
    nodes, fit, pods, gcpu, gmem, size = _scene(soa, n_nodes, seed)
    dry_codes, dry_fk, _, groups0 = _sequential(bsa, soa, nodes, fit, pods, gcpu, gmem, size, assume=False)
    seq_codes, _, released, groups0b = _sequential(bsa, soa, nodes, fit, pods, gcpu, gmem, size, assume=True)
    assert groups0.state_equal(groups0b)
    out = _batch(bsa, soa, nodes, fit, groups0, pods)
    # R1: nothing assumed, nothing permitted -> identical, code by code, early-exit index by index
    assert np.array_equal(out.pf_code, dry_codes)
    assert np.array_equal(out.pf_first_k, dry_fk)
    # ... and the oracle agrees with both
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups0).batch(pods, soa.STAGE_PREFILTER | soa.STAGE_TALLY, bitmap=False)
    assert np.array_equal(exp.pf_code, out.pf_code) and np.array_equal(exp.group_admit, out.group_admit)
    # R2: per pod, group and gang
    seq_pass, bat_pass = seq_codes < 16, out.pf_code < 16
    assert not np.any(seq_pass & ~bat_pass), "a pod the sequential reference admits is admitted by the batch"
    G = groups0.g
    admit_seq = np.bincount(pods.group[seq_pass], minlength=G).astype(np.uint32)
    assert np.array_equal(np.bincount(pods.group[bat_pass], minlength=G).astype(np.uint32), out.group_admit)
    assert np.all(admit_seq <= out.group_admit)
    assert all(out.group_ready[g] == 1 for g in released), "a gang the sequential run releases is ready in the batch"
    strict = int((out.group_admit > admit_seq).sum())
    if expect_strict:
        assert strict > 0, "tight cluster: the frozen-snapshot batch over-admits (it does not reserve capacity between gangs)"
        assert int(out.group_ready.sum()) > len(released | {0}) - 1
    else:
        assert strict == 0 and np.array_equal(out.pf_code, seq_codes), "slack cluster: nothing is ever rejected, batch == sequential"


def test_readme_race_is_the_canonical_over_admission(bsa, soa, orc):
    """R3.  Sequentially the README scene ends 5/5 + 0/5 (tests/test_host_mirror.py); the batch over the same ten pods
    on the frozen snapshot finds each gang feasible on its own and reports both ready."""
    scene = json.load(open(os.path.join(GOLD, "readme_race_scene.json")))
    nd = scene["node"]
    alloc = np.array([[nd["allocatable_cpu"]], [64 << 30], [0], [nd["allocatable_pods"]]], np.int64)
    req = np.array([[nd["requested_cpu"]], [0], [0], [nd["pod_count"]]], np.int64)
    nodes = soa.Nodes(alloc, req, np.zeros(1, np.uint32), np.zeros(1, np.uint32), np.zeros(1, np.uint8))
    fit = soa.FitMasks.from_bool(np.ones((1, 1), bool))
    groups = soa.Groups.empty(2, 4)
    groups.min_member[:] = 5
    order = [0, 0, 1, 0, 1, 0, 1, 0, 1, 1]
    preq = np.zeros((4, 10), np.int64)
    preq[0] = 1000
    pods = soa.Pods(np.array(order, np.int32), preq, np.zeros(10, np.uint32), np.zeros(10, np.uint32), np.zeros(10, np.uint64), np.zeros(10, np.uint8))
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_PREFILTER | soa.STAGE_TALLY, bitmap=False)
    out = _batch(bsa, soa, nodes, fit, groups, pods)
    assert np.array_equal(out.pf_code, exp.pf_code) and np.array_equal(out.group_ready, exp.group_ready)
    # frozen snapshot: nobody has matched pods -> every pod takes the first-fit branch (core.go:136-147):
    # 5 x 1000 m <= scale(8000, 1) - 900 for either gang
    # (the very first pod captures its group at core.go:486 before findMaxPG runs, so it already finds a leader)
    assert set(out.pf_code.tolist()) == {soa.PF_PASS_FIRST_FITS}
    assert out.group_admit.tolist() == [5, 5] and out.group_ready.tolist() == [1, 1]
    assert scene["expected_end_state"] == {"group1": "5/5 admitted", "group2": "0/5"}   # the sequential outcome (tests/test_host_mirror.py)


@pytest.mark.parametrize("n_nodes,seed,big", [(36, 1, True), (40, 2, True), (64, 5, True), (48, 7, False), (160, 3, False)])
def test_filter_on_batch_plus_deny_pass_equals_sequential_prefilter_and_filter(n_nodes, seed, big, bsa, soa, orc):
    """R1F.  Sequential: PreFilter(pod) and, if it passes, Filter(pod, node) for every node (a failing node deny-lists the group,
    core.go:183-185; the first pod of a gang whose Filter fails somewhere turns every later pod of the gang into ERR_DENIED).
    Batch: ONE bs_batch_run with Filter on and BS_BATCH_FILTER_DENY — plain equality.  Code by code, Filter code by Filter code,
    feasible count by feasible count, admit / ready."""
    # big: pods of 4.5 - 6.5 cores on nodes with 5 - 8 cores left: pod + a leader member does not fit many nodes that could still
    # take the leader member alone -> Filter fails there (neither case 2 nor case 3 of core.go:551-561)
    nodes, fit, pods, gcpu, gmem, size = _scene(soa, n_nodes, seed, cpu_choices=(4500, 5500, 6500) if big else (1000, 1500, 2000))
    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = _mirror_with_leader(bsa, ctx, gcpu, gmem, size)
        ctx.g = len(gcpu) + 1
        groups0 = ctx.read_groups()
        seq_pf, seq_fl, seq_feas = [], [], []
        for i in range(pods.p):
            g, rq = int(pods.group[i]), pods.req[:, i].tolist()
            code, _ = sop.PreFilter(i + 1, i + 1, g, rq)
            seq_pf.append(code)
            if code >= 16:
                seq_fl.append(soa.FL_NOT_RUN)
                seq_feas.append(0)
                continue
            fl, feas = soa.FL_NOT_RUN, 0
            for k in range(nodes.n):
                fl, fn = sop.Filter(i + 1, g, rq, 0, k)
                feas += 1 if (fl != soa.FL_EVALUATED and fl < 16) or (fl == soa.FL_EVALUATED and fn < 16) else 0
            seq_fl.append(fl)
            seq_feas.append(feas)
        sop.close()
    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups0)
        ctx.load_pods(pods)
        raw = ctx.batch(soa.STAGE_ALL, bitmap=False)                                   # Filter as a what-if: the entry is not written
        out = ctx.batch(soa.STAGE_ALL | soa.BATCH_FILTER_DENY, bitmap=False)
    seq_pf, seq_fl, seq_feas = np.array(seq_pf, np.uint8), np.array(seq_fl, np.uint8), np.array(seq_feas, np.uint32)
    assert np.array_equal(out.pf_code, seq_pf)
    assert np.array_equal(out.fl_code, seq_fl)
    assert np.array_equal(out.fl_feasible, seq_feas)
    passed = (seq_pf < 16) & (seq_feas > 0)
    assert np.array_equal(out.group_admit, np.bincount(pods.group[passed], minlength=groups0.g).astype(np.uint32))
    denied_by_filter = int(((raw.pf_code < 16) & (out.pf_code == soa.PF_ERR_DENIED)).sum())
    if big:
        assert denied_by_filter > 0, "big pods: some gang's Filter fails on a node and the deny entry bites"
    # the batch WITHOUT the flag is exact up to and including each group's first failing pod
    for g in range(groups0.g):
        idx = np.nonzero(pods.group == g)[0]
        ev = [i for i in idx if raw.fl_code[i] == soa.FL_EVALUATED and raw.fl_feasible[i] < nodes.n]
        upto = ev[0] if ev else (idx[-1] if len(idx) else -1)
        sel = idx[idx <= upto]
        assert np.array_equal(raw.pf_code[sel], seq_pf[sel]) and np.array_equal(raw.fl_feasible[sel], seq_feas[sel])
