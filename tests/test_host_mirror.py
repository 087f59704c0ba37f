"""The C++ host mirror of ScheduleOperation (libbsched_host.so): CPU-side surface/TTL tests and, on the
GPU, sequential replays against the stateful naive restatement — including the reference's own README
scene (BASELINE config 1: 2 PodGroups x 5 pods, 1 node, 8 CPU)."""
import ctypes
import functools
import json
import os

import numpy as np
import pytest

import naive_ref as nv
import naive_seq as ns

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_host_library_exports(bsa):
    bsa.build.build_host()
    lib = ctypes.CDLL(bsa.build.HOST_LIB_PATH)
    for name in bsa.plugin.HOST_SYMBOLS:
        assert hasattr(lib, name), name


def test_host_ttl_is_go_cache(bsa):
    L = bsa.plugin.load_host_library()
    S = 1_000_000_000
    t = L.bsh_ttl_new()
    v = ctypes.c_uint64(0)
    assert L.bsh_ttl_add(t, 1, 7, 0, 20 * S) == 0
    assert L.bsh_ttl_add(t, 1, 8, 5 * S, 20 * S) == -1          # live: Add fails, window not extended (Q16)
    assert L.bsh_ttl_get(t, 1, 20 * S, ctypes.byref(v)) == 1 and v.value == 7
    assert L.bsh_ttl_get(t, 1, 20 * S + 1, ctypes.byref(v)) == 0  # expired strictly after the deadline
    assert L.bsh_ttl_add(t, 1, 9, 21 * S, 20 * S) == 0
    L.bsh_ttl_set(t, 2, 1, 0, 3 * S)
    assert L.bsh_ttl_count(t, 2 * S) == 2
    assert L.bsh_ttl_count(t, 22 * S) == 1
    L.bsh_ttl_delete(t, 1)
    assert L.bsh_ttl_get(t, 1, 22 * S, ctypes.byref(v)) == 0
    L.bsh_ttl_free(t)


def _update_node(bsa, ctx, soa, idx, alloc_col, req_col, ap=0, rp=0):
    d = bsa.capi.NodeDelta()
    d.kind, d.index = bsa.capi.DELTA_UPDATE, idx
    for j in range(len(alloc_col)):
        d.allocatable[j], d.requested[j] = int(alloc_col[j]), int(req_col[j])
    d.allocatable_present, d.requested_present, d.flags = ap, rp, 0
    d.fit_default, d.n_fit_exceptions = 1, 0
    ctx.apply_node_deltas([d])


@pytest.mark.gpu
def test_readme_race_scene_end_to_end(bsa, soa, orc):
    """README.md:78-188: two gangs of 5 x 1 CPU race for one 8-CPU node with 0.9 CPU in use.
    Expected end state: exactly one gang admitted 5/5, the other 0/5."""
    scene = json.load(open(os.path.join(GOLD, "readme_race_scene.json")))
    nd = scene["node"]
    alloc = np.array([[nd["allocatable_cpu"]], [64 << 30], [0], [nd["allocatable_pods"]]], np.int64)
    req = np.array([[nd["requested_cpu"]], [0], [0], [nd["pod_count"]]], np.int64)
    nodes = soa.Nodes(alloc, req, [0], [0], [0])
    fit = soa.FitMasks.from_bool(np.ones((1, 1), bool))
    # naive sequential twin
    a, r = nv.Resource(), nv.Resource()
    a.Add({"cpu": nd["allocatable_cpu"], "memory": 64 << 30, "pods": nd["allocatable_pods"]})
    r.Add({"cpu": nd["requested_cpu"]})
    info = nv.NodeInfo(a, r, nd["pod_count"])
    cache = {g: ns.SeqGroup(nv.PodGroup(g, 5), creation_ts=i) for i, g in enumerate(["group1", "group2"])}
    ref = ns.SeqOperation([info], cache)

    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = bsa.plugin.ScheduleOperation(ctx)
        gidx = {"group1": sop.add_group(5, creation_ts=0, name_rank=1), "group2": sop.add_group(5, creation_ts=1, name_rank=2)}
        pods = [(f"g{g}-p{i}", f"group{g}") for i in range(5) for g in (1, 2)]      # interleaved arrival
        # QueueSort: Less orders by group creation time, then queue time (core.go:368-411)
        keyed = [(gidx[grp], 0, t) for t, (_, grp) in enumerate(pods)]
        order = sorted(range(len(pods)), key=functools.cmp_to_key(lambda x, y: -1 if sop.Less(keyed[x], keyed[y]) else (1 if sop.Less(keyed[y], keyed[x]) else 0)))
        ref_order = sorted(range(len(pods)), key=functools.cmp_to_key(
            lambda x, y: -1 if ref.less((pods[x][1], 0, x), (pods[y][1], 0, y)) else (1 if ref.less((pods[y][1], 0, y), (pods[x][1], 0, x)) else 0)))
        assert order == ref_order
        bound = {"group1": 0, "group2": 0}
        node_req = req[:, 0].copy()
        t = 0.0
        for qi in order:
            name, grp = pods[qi]
            uid = 1000 + qi
            t += 0.05
            sop.set_time(t)
            ref.set_time(t)
            reqv = [1000, 0, 0, 0]
            pod = nv.Pod(uid, grp, {"cpu": 1000})
            code, fk = sop.PreFilter(uid, qi, gidx[grp], reqv)
            rcode, rfk = ref.prefilter(pod)
            assert (code, fk) == (rcode, rfk), (name, soa.PF_NAMES[code], soa.PF_NAMES[rcode])
            if code >= 16:
                continue
            # default scheduler: the pod fits the node? (cpu only here); assume it there
            if node_req[0] + 1000 > alloc[0, 0]:
                continue
            ready, pc = sop.Permit(uid, qi, gidx[grp], 0)
            rready, rpc = ref.permit(pod, qi, 0)
            assert (ready, pc) == (rready, rpc)
            node_req[0] += 1000
            node_req[3] += 1
            _update_node(bsa, ctx, soa, 0, alloc[:, 0], node_req)
            info.requested.MilliCPU += 1000
            info.pod_count += 1
            if ready:
                # a buffer smaller than the gang releases nothing (no pod may leave the cache unreturned)
                with pytest.raises(ValueError):
                    sop.StartBatchSchedule(gidx[grp], cap=1)
                assert sop.group_state(gidx[grp])["matched"] == 5
                released = sop.StartBatchSchedule(gidx[grp])
                assert released == ref.start_batch(grp)
                for _ in released:
                    sop.PostBind(gidx[grp])
                    ref.postbind(grp)
                    bound[grp] += 1
        assert sorted(bound.values()) == [0, 5], bound          # README.md:176-188
        assert bound["group1"] == 5
        st1, st2 = sop.group_state(gidx["group1"]), sop.group_state(gidx["group2"])
        assert st1["status_scheduled"] == 5 and st1["scheduled_latch"] and st2["status_scheduled"] == 0
        assert sop.gpu_calls > 0
        sop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_sequential_replay_random(seed, bsa, soa, orc):
    """Random small clusters: PreFilter -> Filter on a few nodes -> Permit -> release, with the TTL clock
    moving (deny entries expire after 20 s, permitted entries after 2 s): every return value equals the
    stateful naive restatement's."""
    rng = np.random.default_rng(seed)
    from scenarios import random_objects
    sc = random_objects(seed + 31000, n_nodes=int(rng.integers(2, 30)), n_groups=4, n_pods=30, n_scalars=int(rng.integers(0, 2)), edge=False)
    names = sc["names"]
    gnames = list(sc["cache"].keys())
    # fresh groups (controller-created): no pod seen, nothing matched
    cache = {g: ns.SeqGroup(nv.PodGroup(g, int(rng.integers(1, 5))), creation_ts=int(rng.integers(0, 3))) for g in gnames}
    nodes_soa, fit, _, pods_soa, gidx = nv.to_soa(sc["nodes"], {g: nv.PGS(cache[g].pod_group) for g in gnames}, sc["pods"], names, sc["n_classes"])
    ref = ns.SeqOperation(sc["nodes"], cache)
    with bsa.Context(scalar_lanes=len(names)) as ctx:
        ctx.load_nodes(nodes_soa, fit)
        sop = bsa.plugin.ScheduleOperation(ctx)
        for rank, g in enumerate(gnames):
            assert sop.add_group(cache[g].pod_group.min_member, creation_ts=cache[g].creation_ts, name_rank=rank) == gidx[g]
        t = 0.0
        for i, pod in enumerate(sc["pods"]):
            t += float(rng.choice([0.1, 0.5, 3.0, 25.0], p=[0.5, 0.3, 0.15, 0.05]))
            sop.set_time(t)
            ref.set_time(t)
            grp = int(pods_soa.group[i])
            reqv = pods_soa.req[:, i].tolist()
            pres = int(pods_soa.req_present[i])
            uid = i + 1
            pod.uid = uid
            code, fk = sop.PreFilter(uid, uid, grp, reqv, pres, int(pods_soa.cls[i]), int(pods_soa.owner[i]))
            assert (code, fk) == ref.prefilter(pod), (seed, i)
            if code >= 16 or code == soa.PF_PANIC_DIV0:
                continue
            ok_node = None
            for k in rng.choice(len(sc["nodes"]), size=min(3, len(sc["nodes"])), replace=False):
                got = sop.Filter(uid, grp, reqv, pres, int(k))
                exp = ref.filter(pod, int(k))
                assert got[0] == exp[0] and (got[0] != soa.FL_EVALUATED or got[1] == exp[1]), (seed, i, k)
                if got[0] < 16 and (got[0] != soa.FL_EVALUATED or got[1] < 16) and ok_node is None:
                    ok_node = int(k)
            if ok_node is None or grp < 0:
                continue
            got = sop.Permit(uid, uid, grp, ok_node)
            assert got == ref.permit(pod, uid, ok_node)
            if got[0]:
                rel = sop.StartBatchSchedule(grp)
                assert rel == ref.start_batch(pod.group)
                for _ in rel:
                    sop.PostBind(grp)
                    ref.postbind(pod.group)
        for g in gnames:
            st = sop.group_state(gidx[g])
            assert st["status_scheduled"] == cache[g].pod_group.status_scheduled
            assert st["scheduled_latch"] == cache[g].scheduled
        sop.close()
