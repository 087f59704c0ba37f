"""Pins the CPU oracle (and the independent naive restatement) to every result-fixing vector the
reference holds for the hot path, plus the derived float32 known-answer table.  CPU only."""
import json
import os

import numpy as np
import pytest

import naive_ref as nv

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def test_f32_constant_is_0x3f333333():
    assert int(np.float32(0.7).view(np.uint32)) == 0x3F333333


def test_scale_kat_oracle_and_naive(orc):
    kat = load("f32_scale_kat.json")
    assert len(kat["vectors"]) > 4000
    for v in kat["vectors"]:
        pct = float(np.uint32(v["pct_bits"]).view(np.float32))
        assert orc.scale(v["a"], pct) == v["out"], v
        assert nv.scale(v["a"], pct) == v["out"], v


def test_scale_listed_values(orc):
    # pct=1 is NOT the identity; 0.7 truncation; tie-to-even
    assert orc.scale(8000, .7) == 5600 and orc.scale(110, .7) == 77 and orc.scale(100, .7) == 70
    assert orc.scale(16777217, 1) == 16777216 and orc.scale(16777219, 1) == 16777220
    assert orc.scale(16655429632, .7) == 11658800128
    assert orc.scale(270255247360, 1) == 270255243264
    assert orc.scale(2 ** 34 + 1024, 1) == 2 ** 34 and orc.scale(2 ** 34 + 1025, 1) == 2 ** 34 + 2048
    assert orc.scale(7, .7) == 4 and orc.scale(3, .7) == 2 and orc.scale(1, .7) == 0 and orc.scale(0, .7) == 0


def test_scale_beyond_2_53_oracle_equals_integer_rounding(orc):
    # above 2**53 numpy's int->double->float32 double-rounds, so the pin is the integer-arithmetic
    # RNE of naive_ref (independent of the C compiler's cvtsi2ss)
    rng = np.random.default_rng(7)
    for _ in range(4000):
        a = int(rng.integers(-(2 ** 62), 2 ** 62))
        for pct in (1.0, 0.7):
            assert orc.scale(a, pct) == nv.scale(a, pct)
    assert orc.scale(2 ** 63 - 1, 1.0) == -(2 ** 63)      # CVTTSS2SQ integer indefinite
    assert nv.scale(2 ** 63 - 1, 1.0) == -(2 ** 63)
    assert orc.scale(-(2 ** 63), 1.0) == -(2 ** 63)


def _core_test_objects():
    v = load("core_test_vectors.json")
    nd = v["node"]
    alloc, reqd = nv.Resource(), nv.Resource()
    alloc.Add(nd["allocatable"])
    reqd.Add(nd["requested"])
    info = nv.NodeInfo(alloc, reqd, nd["pod_count"])
    return v, info


def test_reference_core_test_vectors_naive():
    v, info = _core_test_objects()
    for case in v["cases"]:
        pod = nv.Pod("u", None, case["req"])
        left = nv.single_node_resource(info, pod, v["percent"])
        got = left.ResourceList()
        assert got == v["expected_left"]
        assert nv.compare_resource_and_require(left, nv.pod_resource_require(pod)) is case["desire"]


def test_reference_core_test_vectors_oracle(orc):
    v, info = _core_test_objects()
    names = v["scalar_names"]
    for case in v["cases"]:
        pod = nv.Pod("u", None, case["req"])
        nodes, fit, groups, pods, _ = nv.to_soa([info], {}, [pod], names, 1)
        snap = orc.Snapshot(nodes, fit)
        left, present = snap.single_node_resource(0, 0, v["percent"])
        exp = v["expected_left"]
        assert left == [exp["cpu"], exp["memory"], exp["ephemeral-storage"], exp["pods"], exp[names[0]], exp[names[1]]]
        assert present == 0b11
        req = [int(pods.req[j, 0]) for j in range(6)]
        assert orc.compare_resource_and_require(left, present, req, int(pods.req_present[0]), 2) is case["desire"]
        # the same through the cluster loop (1 node): core.go:595-632
        ok, fk, iters = snap.compare_cluster(0, req, int(pods.req_present[0]), v["percent"])
        assert ok is case["desire"] and iters == 1 and fk == (0 if ok else 0xFFFFFFFF)


def test_q3_idle_scalar_contributes_nothing(orc):
    # a node with 8 idle GPUs and no GPU pod (requested has no gpu key) yields NO gpu key
    a, r = nv.Resource(), nv.Resource()
    a.Add({"cpu": 4000, "pods": 10, "gpu": 8})
    r.Add({"cpu": 0})
    info = nv.NodeInfo(a, r, 0)
    left = nv.single_node_resource(info, nv.Pod("u", None, {}), 1.0)
    assert "gpu" not in left.ScalarResources
    req = nv.Resource()
    req.Add({"gpu": 1})
    assert not nv.compare_resource_and_require(left, req)
    req0 = nv.Resource()
    req0.Add({"gpu": 0})
    assert nv.compare_resource_and_require(left, req0)
    nodes, fit, _, pods, _ = nv.to_soa([info], {}, [nv.Pod("u", None, {"gpu": 1}), nv.Pod("v", None, {"gpu": 0})], ["gpu"], 1)
    snap = orc.Snapshot(nodes, fit)
    lv, lp = snap.single_node_resource(0, 0, 1.0)
    assert lp == 0 and lv[4] == 0
    assert not orc.compare_resource_and_require(lv, lp, pods.req[:, 0].tolist(), 1, 1)
    assert orc.compare_resource_and_require(lv, lp, pods.req[:, 1].tolist(), 1, 1)


def test_present_zero_request_vs_negative_left(orc):
    # req key present with value 0, left present and negative: v1 > v2 -> false (core.go:694)
    assert not orc.compare_resource_and_require([0, 0, 0, 0, -1], 1, [0, 0, 0, 0, 0], 1, 1)
    assert orc.compare_resource_and_require([0, 0, 0, 0, -1], 1, [0, 0, 0, 0, 0], 0, 1)
    # negative request against an absent key: v1 != 0 -> false (core.go:689)
    assert not orc.compare_resource_and_require([0, 0, 0, 0, 0], 0, [0, 0, 0, 0, -5], 1, 1)


def _race_step(step, scene, orc, soa):
    nd = scene["node"]
    a, r = nv.Resource(), nv.Resource()
    a.Add({"cpu": nd["allocatable_cpu"], "pods": nd["allocatable_pods"]})
    r.Add({"cpu": step["node_requested_cpu"]})
    extra = (step["node_requested_cpu"] - nd["requested_cpu"]) // scene["pod_cpu"]
    info = nv.NodeInfo(a, r, nd["pod_count"] + extra)
    cache = {}
    rep = {}
    for name, g in scene["groups"].items():
        st = step["state"][name]
        pg = nv.PodGroup(name, g["min_member"])
        pgs = nv.PGS(pg, matched=st["matched"], scheduled=st.get("scheduled_latch", False))
        if st["has_pod"]:
            rep[name] = nv.Pod(name + "-rep", name, {"cpu": scene["pod_cpu"]})
            pgs.pod = rep[name]
            pg.min_resources = nv.pod_resource_require(pgs.pod).ResourceList()
        cache[name] = pgs
    pod = nv.Pod("p", step["pod_group"], {"cpu": scene["pod_cpu"]})
    return [info], cache, pod


def test_readme_race_scene_decisions(orc, soa):
    scene = load("readme_race_scene.json")
    name_to_code = {v: k for k, v in soa.PF_NAMES.items()}
    for step in scene["steps"]:
        nodes, cache, pod = _race_step(step, scene, orc, soa)
        # naive
        sop = nv.ScheduleOperation(nodes, cache)
        code, fk = sop.prefilter(pod)
        assert soa.PF_NAMES[code] == step["expect_code"], step["what"]
        if "expect_first_k" in step:
            assert fk == step["expect_first_k"]
        # C oracle on the flattened state (fresh objects: the naive run mutated the cache)
        nodes, cache, pod = _race_step(step, scene, orc, soa)
        n_soa, fit, groups, pods, _ = nv.to_soa(nodes, cache, [pod], [], 1)
        osop = orc.Sop(orc.Snapshot(n_soa, fit), groups)
        ocode, ofk = osop.prefilter(pods, 0)
        assert ocode == name_to_code[step["expect_code"]], step["what"]
        assert ofk == fk


def test_permit_quorum_uint32_wrap(orc):
    assert orc.permit_ready(5, 5, 0) and not orc.permit_ready(4, 5, 0)
    assert orc.permit_ready(2, 5, 3)
    # MinMember - Scheduled wraps when Scheduled > MinMember (core.go:303, Q7): 5-6 = 0xFFFFFFFF
    assert not orc.permit_ready(1000, 5, 6)
    assert orc.permit_ready(0, 5, 5)


def test_ttl_go_cache_semantics(orc):
    t = orc.TTL()
    S = 1_000_000_000
    assert t.add(1, 7, 0, 20 * S)
    assert not t.add(1, 8, 5 * S, 20 * S)            # Add fails while live: window NOT extended (Q16)
    assert t.get(1, 20 * S) == 7                     # now == expiration is still live (now > exp expires)
    assert t.get(1, 20 * S + 1) is None
    assert t.add(1, 9, 21 * S, 20 * S) and t.get(1, 22 * S) == 9
    t.set(2, 1, 0, 3 * S)
    t.set(2, 2, 1 * S, 3 * S)
    assert t.get(2, 4 * S) == 2 and t.count(2 * S) == 2 and t.count(30 * S) == 1
    t.delete(2)
    assert t.get(2, 0) is None
    for k in range(100, 400):
        t.set(k, k, 0, S)
    assert t.count(0) == 301 and t.get(250, S) == 250
