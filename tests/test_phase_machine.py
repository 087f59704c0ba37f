"""SURVEY.md section 8(f)-4, second half (CPU only): the PodGroup phase machine and the merge-patch writer of host/bs_phase.cpp (include/bsched_host.h)
against (i) the reference's own two expected strings (pkg/util/k8s_test.go:31-78), (ii) hand-derived known answers that cite controller.go / core.go /
batchscheduler.go lines (tests/golden/podgroup_phase_kats.json), (iii) an independent Python restatement (oracle/naive_phase.py) on random inputs."""
import ctypes
import importlib
import json
import os
import re

import numpy as np
import pytest
from hypothesis import example, given, settings, strategies as st

import naive_phase as nph

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KATS = json.load(open(os.path.join(HERE, "golden", "podgroup_phase_kats.json")))
ids = lambda vs: [v["id"] for v in vs]
ACTS = {"patch_recover": 1, "patch": 2, "cache_delete": 4, "no_requeue": 8, "listed": 16}


@pytest.fixture(scope="module")
def pg(bsa):
    bsa.build.build_host()
    return importlib.import_module("batch-scheduler_amd.podgroup")


def _st(pg, d):
    return pg.PodGroupStatus(d["phase"], d["scheduled"], d["running"], d["succeeded"], d["failed"], d["scheduleStartTime"], d.get("occupiedBy", ""))


def _dict(s):
    return {"phase": s.phase, "scheduled": s.scheduled, "running": s.running, "succeeded": s.succeeded, "failed": s.failed, "scheduleStartTime": s.schedule_start_ns}


def test_host_library_exports_every_symbol_the_header_declares(bsa, pg):
    lib = ctypes.CDLL(bsa.build.build_host())
    header = open(os.path.join(ROOT, "include", "bsched_host.h")).read()
    declared = set(re.findall(r"^\s*(?:int|uint32_t|void|const char\*|bsh_pg\*)\s+(bsh_[a-z_0-9]+)\s*\(", header, re.M))
    assert declared == set(pg.HOST_PHASE_SYMBOLS), declared ^ set(pg.HOST_PHASE_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("v", KATS["merge_patch"], ids=ids(KATS["merge_patch"]))
def test_merge_patch_known_answers(v, pg):
    assert pg.create_merge_patch(v["original"], v["modified"]) == v["expected"]
    assert nph.create_merge_patch(v["original"], v["modified"]) == v["expected"], "naive restatement"


def test_merge_patch_refuses_what_is_not_an_object(pg, bsa):
    for bad in ("[1]", "3", "{", "{\"a\":}", "{\"a\":1} x", ""):
        with pytest.raises(bsa.capi.BsError) as e:
            pg.create_merge_patch(bad, "{}")
        assert e.value.status == -1


_leaf = st.one_of(st.none(), st.booleans(), st.integers(-10 ** 6, 10 ** 6), st.sampled_from([0.5, -1.25, 1e21, 2.5e-7, 3.0]), st.text(alphabet="ab<>&\"\\\n\t é ", max_size=6))
_json = st.recursive(_leaf, lambda c: st.one_of(st.lists(c, max_size=3), st.dictionaries(st.sampled_from(["a", "b", "c", "k<", "é"]), c, max_size=4)), max_leaves=12)
_obj = st.dictionaries(st.sampled_from(["a", "b", "c", "status", "z"]), _json, max_size=5)


@settings(max_examples=300, deadline=None)
@given(_obj, _obj)
@example({"c": [{}]}, {"c": [{"a": None}]})              # a null inside an array's element: both statements say "no difference" (matchesValue), no round trip
def test_merge_patch_equals_the_naive_restatement(pg, a, b):
    ta, tb = json.dumps(a), json.dumps(b)
    got = pg.create_merge_patch(ta, tb)
    assert got == nph.create_merge_patch(ta, tb)
    # RFC 7386 round trip where it is defined (no nulls anywhere inside the target — matchesValue takes a missing key and a null for the same thing,
    # also inside an array's elements): applying the patch to a gives b
    def has_null(x):
        return x is None or (isinstance(x, dict) and any(has_null(v) for v in x.values())) or (isinstance(x, list) and any(has_null(v) for v in x))
    if not has_null(b):
        def apply(t, p):
            if not isinstance(p, dict):
                return p
            t = dict(t) if isinstance(t, dict) else {}
            for k, v in p.items():
                if v is None:
                    t.pop(k, None)
                else:
                    t[k] = apply(t.get(k), v)
            return t
        assert apply(a, json.loads(got)) == b


@pytest.mark.parametrize("v", KATS["status_patch"], ids=ids(KATS["status_patch"]))
def test_status_patch_known_answers(v, pg):
    assert pg.status_patch(_st(pg, v["from"]), _st(pg, v["to"])) == v["expected"]
    assert nph.status_patch(nph.new_status(**v["from"]), nph.new_status(**v["to"])) == v["expected"]


def test_status_json_follows_the_struct_tags(pg):
    s = pg.PodGroupStatus("Scheduling", 2, 1, 0, 0, 1600000000 * 10 ** 9 + 999, "owner-a")
    assert pg.status_json(s) == '{"phase":"Scheduling","occupiedBy":"owner-a","scheduled":2,"running":1,"succeeded":0,"failed":0,"scheduleStartTime":"2020-09-13T12:26:40Z"}'
    assert pg.status_json(pg.PodGroupStatus()) == '{"phase":"","scheduled":0,"running":0,"succeeded":0,"failed":0,"scheduleStartTime":null}'
    assert json.loads(pg.status_json(s)) == nph.status_doc(nph.new_status(phase="Scheduling", scheduled=2, running=1, scheduleStartTime=1600000000 * 10 ** 9 + 999, occupiedBy="owner-a"))


@pytest.mark.parametrize("v", KATS["sync"], ids=ids(KATS["sync"]))
def test_sync_handler_known_answers(v, pg):
    pods = [(u, p) for u, p in v["pods"]]
    # the product
    c = pg.PodGroupController()
    if v["sets_before"][0] or v["sets_before"][1]:          # earlier syncs: feed the sets through a sync of their own (phase Running, those pods listed)
        warm = [(u, "Succeeded") for u in v["sets_before"][0]] + [(u, "Failed") for u in v["sets_before"][1]]
        c.sync_handler(10 ** 6, 0, pg.PodGroupStatus("Running"), warm)
    rec, out, act = c.sync_handler(v["min_member"], v["creation"], _st(pg, v["in"]), pods)
    assert _dict(out) == v["out"]
    assert (None if rec is None else _dict(rec)) == v["recovered"]
    assert act == sum(ACTS[a] for a in v["actions"]), (act, v["actions"])
    # the naive restatement
    n = nph.Controller()
    n.succeed, n.failed = set(v["sets_before"][0]), set(v["sets_before"][1])
    nrec, nout, nact = n.sync_handler(v["min_member"], v["creation"], nph.new_status(**v["in"]), pods)
    strip = lambda d: None if d is None else {k: d[k] for k in v["out"]}
    assert strip(nout) == v["out"] and strip(nrec) == v["recovered"] and nact == set(v["actions"])


@pytest.mark.parametrize("v", KATS["post_bind"], ids=ids(KATS["post_bind"]))
def test_post_bind_known_answers(v, pg):
    out, patch = pg.post_bind(v["min_member"], _st(pg, v["in"]), v["now"])
    assert _dict(out) == v["out"] and patch == v["patch"]
    nout, npatch = nph.post_bind(v["min_member"], nph.new_status(**v["in"]), v["now"])
    assert {k: nout[k] for k in v["out"]} == v["out"] and npatch == v["patch"]


@pytest.mark.parametrize("v", KATS["gates"], ids=ids(KATS["gates"]))
def test_phase_gates_known_answers(v, pg, soa):
    assert pg.permit_phase(v["phase"]) == v["permit"]
    assert pg.phase_closed(v["phase"]) == v["closed"]
    assert pg.start_gate(3, pg.PodGroupStatus(v["phase"], 0))[0] == v["release"]
    assert nph.start_gate(3, nph.new_status(phase=v["phase"]))[0] == v["release"]
    # what is open for release is never closed; the closed bit is what the device is told (BS_GROUP_PHASE_CLOSED)
    assert not (v["release"] and v["closed"])


def test_start_gate_stamps_the_start_time_at_the_quorum(pg):
    assert pg.start_gate(3, pg.PodGroupStatus("Scheduling", 3)) == (True, True)          # batchscheduler.go:264
    assert pg.start_gate(3, pg.PodGroupStatus("Scheduling", 2)) == (True, False)
    assert pg.start_gate(3, pg.PodGroupStatus("Scheduled", 3)) == (False, False)


@pytest.mark.parametrize("seed", range(400))
def test_sync_handler_equals_the_naive_restatement(seed, pg):
    rng = np.random.default_rng(seed)
    mm = int(rng.integers(0, 5))
    creation = int(rng.integers(0, 10 ** 6))
    phases = ["", "Pending", "PreScheduling", "Scheduling", "Scheduled", "Running", "Unknown", "Finished", "Failed"]
    c, n = pg.PodGroupController(), nph.Controller()
    status = nph.new_status(phase=str(rng.choice(phases)), scheduled=int(rng.integers(0, 5)), running=int(rng.integers(0, 3)),
                            scheduleStartTime=int(rng.choice([0, 5000, creation + nph.H48, creation + nph.H48 + 1])))
    for step in range(4):                                     # a few syncs in a row: the uid sets carry over, the status is what the last sync left
        pods = [(int(u), str(rng.choice(["Pending", "Running", "Succeeded", "Failed", "Unknown"]))) for u in rng.choice(12, int(rng.integers(0, 7)), replace=False)]
        rec, out, act = c.sync_handler(mm, creation, _st(pg, status), pods)
        nrec, nout, nact = n.sync_handler(mm, creation, status, pods)
        keys = ("phase", "scheduled", "running", "succeeded", "failed", "scheduleStartTime")
        assert _dict(out) == {k: nout[k] for k in keys}, (seed, step)
        assert (rec is None) == (nrec is None) and (rec is None or _dict(rec) == {k: nrec[k] for k in keys})
        assert act == sum(ACTS[a] for a in nact), (seed, step, act, nact)
        assert c.counts == (len(n.succeed), len(n.failed))
        assert pg.enqueue(mm, creation, out) == nph.enqueue(mm, creation, nout)
        if "patch" in nact:
            assert pg.status_patch(_st(pg, nrec or status), out) == nph.status_patch(nrec or status, nout) != "{}"
        status = nout
