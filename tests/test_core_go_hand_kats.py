"""tests/golden/core_go_hand_kats.json: known answers derived by hand from pkg/scheduler/core/core.go (each vector cites its lines) for the functions
the reference holds no test for — findMaxPG, getPreAllocatedResource, compareClusterResourceAndRequire / singleNodeResource /
compareResourceAndRequire, PreFilter's branch selection with its in-queue side effects, computeResourceSatisfied.  Every vector goes to BOTH
restatements — the C oracle (oracle/bs_oracle.c, flat lanes) and the independent object-level one (oracle/naive_ref.py) — on the CPU and, with
-m gpu, through the C ABI to the HIP path.  The expectations are literals in the JSON: no restatement produced them."""
import json
import os

import numpy as np
import pytest

import naive_ref as nv

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "core_go_hand_kats.json")))
SCALARS = KATS["scalars"]
BY_KIND = {k: [v for v in KATS["vectors"] if v["kind"] == k] for k in ("find_max_pg", "pre_allocated", "cluster", "scene")}
ids = lambda vs: [v["id"] for v in vs]


def test_the_file_is_what_it_says():
    vs = KATS["vectors"]
    assert len(vs) >= 40 and len({v["id"] for v in vs}) == len(vs)
    assert all(v["cites"].strip() and v["why"].strip() for v in vs), "every vector cites core.go lines and says why"
    assert all(len(BY_KIND[k]) >= 10 for k in BY_KIND)


# ---- JSON -> the objects of naive_ref -> (through its marshaller) the lanes of the ABI
def _res(d):
    r = nv.Resource()
    r.Add(dict(d))
    return r


def _nodes(specs):
    out = []
    for s in specs:
        info = nv.NodeInfo(_res(s["alloc"]), _res(s["req"]), int(s.get("pods", 0)), nil=bool(s.get("nil")), has_node=not s.get("no_node"),
                           unschedulable=bool(s.get("unschedulable")), taint_err=bool(s.get("taint_err")))
        for c in s.get("nofit", []):
            info.labels_fit[int(c)] = False
        out.append(info)
    return out


def _cache(groups):
    cache = {}
    for g in groups:
        pg = nv.PodGroup(g["name"], int(g["min_member"]), int(g["scheduled"]), dict(g["min_resources"]) if g["min_resources"] is not None else None, g["occupied_by"])
        pod = nv.Pod("rep-" + g["name"], g["name"], {}, cls=int(g["pod"]["cls"])) if g["pod"] is not None else None
        cache[g["name"]] = nv.PGS(pg, matched=int(g["matched"]), pod=pod, scheduled=bool(g["latch"]))
    return cache


def _pods(specs):
    return [nv.Pod(p["uid"], p["group"], dict(p["requests"]), cls=int(p["cls"]), owner_refs=tuple(p["owner_refs"])) for p in specs]


def _n_classes(v):
    cl = [0] + [int(g["pod"]["cls"]) for g in v.get("groups", []) if g.get("pod")] + [int(p["cls"]) for p in v.get("pods", [])] + [int(v.get("cls", 0))]
    cl += [int(c) for s in v.get("nodes", []) for c in s.get("nofit", [])]
    return max(cl) + 1


def _lanes(d):
    vals, present = nv._lanes(dict(d), SCALARS)
    return vals, present


def _expect_lanes(e):
    vals = [e["cpu"], e["memory"], e["ephemeral-storage"], e["pods"]] + [e["scalars"].get(s, 0) for s in SCALARS]
    present = sum(1 << i for i, s in enumerate(SCALARS) if s in e["scalars"])
    return vals, present


# ---- findMaxPG (core.go:701-739)
@pytest.mark.parametrize("v", BY_KIND["find_max_pg"], ids=ids(BY_KIND["find_max_pg"]))
def test_find_max_pg(v, orc, soa):
    cache, e = _cache(v["groups"]), v["expect"]
    names = list(cache)
    if e["panic"]:
        with pytest.raises(nv.GoPanic):
            nv.find_max_pg(cache)
    else:
        name, pgs, fin = nv.find_max_pg(cache)
        assert (name or None, fin) == (e["leader"], e["finished"]), "naive_ref"
        assert (pgs is None) == (e["leader"] is None)
    _, _, groups, _, _ = nv.to_soa([], cache, [], SCALARS, 1)
    leader, fin, panic = orc.find_max_pg(groups)
    assert panic == e["panic"], "oracle"
    if not e["panic"]:
        assert (names[leader] if leader >= 0 else None, fin) == (e["leader"], e["finished"]), "oracle"


# ---- getPreAllocatedResource (core.go:774-793)
@pytest.mark.parametrize("v", BY_KIND["pre_allocated"], ids=ids(BY_KIND["pre_allocated"]))
def test_pre_allocated(v, orc, soa):
    cache = _cache([v["group"]])
    want, wpres = _expect_lanes(v["expect"])
    r = nv.get_pre_allocated(cache[v["group"]["name"]], int(v["matched"]))
    got, gpres = nv._lanes(r, SCALARS)
    assert (got, gpres) == (want, wpres), "naive_ref"
    _, _, groups, _, _ = nv.to_soa([], cache, [], SCALARS, 1)
    lanes, present = orc.pre_allocated(groups, 0, int(v["matched"]), len(SCALARS))
    assert ([int(x) for x in lanes], present) == (want, wpres), "oracle"


# ---- compareClusterResourceAndRequire (core.go:595-632) with singleNodeResource (:634-670) and compareResourceAndRequire (:672-699)
def _cluster_inputs(v):
    nodes = _nodes(v["nodes"])
    nodes_soa, fit, _, _, _ = nv.to_soa(nodes, {}, [], SCALARS, _n_classes(v))
    req, present = _lanes(v["req"])
    return nodes, nodes_soa, fit, req, present


@pytest.mark.parametrize("v", BY_KIND["cluster"], ids=ids(BY_KIND["cluster"]))
def test_cluster(v, orc, soa):
    nodes, nodes_soa, fit, req, present = _cluster_inputs(v)
    e = v["expect"]
    ok, k = nv.compare_cluster(nodes, nv.Pod("rep", "g", {}, cls=int(v["cls"])), _res(v["req"]), float(v["pct"]))
    assert (ok, k) == (e["fits"], e["first_k"]), "naive_ref"
    fits, fk, _ = orc.Snapshot(nodes_soa, fit, scalar_lanes=len(SCALARS)).compare_cluster(int(v["cls"]), req, present, float(v["pct"]))
    assert (bool(fits), (int(fk) if fits else None)) == (e["fits"], e["first_k"]), "oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("v", BY_KIND["cluster"], ids=ids(BY_KIND["cluster"]))
def test_cluster_on_the_device(v, bsa, soa):
    _, nodes_soa, fit, req, present = _cluster_inputs(v)
    e = v["expect"]
    with bsa.Context(scalar_lanes=len(SCALARS)) as ctx:
        ctx.load_nodes(nodes_soa, fit)
        fits, fk = ctx.cluster_fits(int(v["cls"]), float(v["pct"]), req, present)
        assert (fits, (fk if fits else None)) == (e["fits"], e["first_k"])


# ---- scenes: PreFilter pod by pod in queue order (core.go:88-167), Filter on the named nodes (:170-191, :514-564)
def _k(soa, k):
    return {"not_scanned": soa.K_NOT_SCANNED, "none": soa.K_NONE}.get(k, k)


def _scene(v):
    nodes, cache, pods = _nodes(v["nodes"]), _cache(v["groups"]), _pods(v["pods"])
    flat = nv.to_soa(nodes, cache, pods, SCALARS, _n_classes(v), denied=v["denied"], permitted=v["permitted"])
    return nodes, cache, pods, flat


@pytest.mark.parametrize("v", BY_KIND["scene"], ids=ids(BY_KIND["scene"]))
def test_scene_naive_ref(v, soa):
    nodes, cache, pods, _ = _scene(v)
    sop = nv.ScheduleOperation(nodes, cache)
    sop.denied, sop.permitted = set(v["denied"]), set(v["permitted"])
    for i, (pod, e) in enumerate(zip(pods, v["expect"])):
        code, k = sop.prefilter(pod)
        assert (soa.PF_NAMES[code], k, sop.max_finished_pg or None) == (e["pf"], _k(soa, e["first_k"]), e["leader"]), f"pod {i}"
        for f in e["filter"]:
            fl, fn = sop.filter_node(pods[f["pod"]], f["node"])
            assert fl == getattr(soa, "FL_" + f["fl"]) and (f["fn"] is None or fn == getattr(soa, "FN_" + f["fn"])), f"Filter(pod {f['pod']}, node {f['node']})"


@pytest.mark.parametrize("v", BY_KIND["scene"], ids=ids(BY_KIND["scene"]))
def test_scene_oracle(v, orc, soa):
    _, cache, _, (nodes_soa, fit, groups, pods_soa, gidx) = _scene(v)
    names = list(cache)
    name_of = lambda g: names[g] if g >= 0 else None
    # pod by pod (orc_prefilter / orc_filter_node) ...
    sop = orc.Sop(orc.Snapshot(nodes_soa, fit, scalar_lanes=len(SCALARS)), groups)
    for i, e in enumerate(v["expect"]):
        code, k = sop.prefilter(pods_soa, i)
        assert (soa.PF_NAMES[code], k, name_of(sop.leader)) == (e["pf"], _k(soa, e["first_k"]), e["leader"]), f"pod {i}"
        for f in e["filter"]:
            fl, fn = sop.filter_node(pods_soa, f["pod"], sop.leader, f["node"])
            assert fl == getattr(soa, "FL_" + f["fl"]) and (f["fn"] is None or fn == getattr(soa, "FN_" + f["fn"])), f"Filter(pod {f['pod']}, node {f['node']})"
    # ... and as one batch (orc_batch: what the device is held against everywhere else)
    _check_batch(v, orc.Sop(orc.Snapshot(nodes_soa, fit, scalar_lanes=len(SCALARS)), groups).batch(pods_soa, soa.STAGE_ALL), names, soa)


def _check_batch(v, out, names, soa):
    name_of = lambda g: names[g] if g >= 0 else None
    for i, e in enumerate(v["expect"]):
        assert (soa.PF_NAMES[int(out.pf_code[i])], int(out.pf_first_k[i]), name_of(int(out.pf_leader[i]))) == (e["pf"], _k(soa, e["first_k"]), e["leader"]), f"pod {i}"
        for f in e["filter"]:
            assert int(out.fl_code[f["pod"]]) == getattr(soa, "FL_" + f["fl"]), f"fl_code of pod {f['pod']}"
            if f["fn"] is not None:
                assert out.node_passes(f["pod"], f["node"]) == f["fn"].startswith("PASS"), f"Filter(pod {f['pod']}, node {f['node']})"


@pytest.mark.gpu
@pytest.mark.parametrize("v", BY_KIND["scene"], ids=ids(BY_KIND["scene"]))
def test_scene_on_the_device(v, bsa, soa):
    _, cache, _, (nodes_soa, fit, groups, pods_soa, gidx) = _scene(v)
    with bsa.Context(scalar_lanes=len(SCALARS)) as ctx:
        ctx.load_nodes(nodes_soa, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods_soa)
        _check_batch(v, ctx.batch(soa.STAGE_ALL), list(cache), soa)
        # the single-query entry point names the case as well (bs_filter_one)
        for e in v["expect"]:
            for f in e["filter"]:
                if f["fn"] is None or e["leader"] is None:
                    continue
                pi = f["pod"]
                fl, fn = ctx.filter_one(int(pods_soa.group[pi]), pods_soa.req[:, pi].tolist(), int(pods_soa.req_present[pi]), gidx[e["leader"]], f["node"])
                assert (fl, fn) == (getattr(soa, "FL_" + f["fl"]), getattr(soa, "FN_" + f["fn"])), f"bs_filter_one(pod {pi}, node {f['node']})"


@pytest.mark.gpu
@pytest.mark.parametrize("v", BY_KIND["find_max_pg"], ids=ids(BY_KIND["find_max_pg"]))
def test_find_max_pg_on_the_device(v, bsa, soa):
    cache, e = _cache(v["groups"]), v["expect"]
    names = list(cache)
    _, _, groups, _, _ = nv.to_soa([], cache, [], SCALARS, 1)
    with bsa.Context(scalar_lanes=len(SCALARS)) as ctx:
        ctx.load_groups(groups)
        leader, panic = ctx.find_max_pg()
        assert panic == e["panic"]
        if not panic:
            assert (names[leader] if leader >= 0 else None) == e["leader"]
