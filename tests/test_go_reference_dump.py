"""The oracle against golden vectors produced by the REFERENCE's own Go functions (CPU).

tests/golden/go_reference_dump.json is written by go/pkg/scheduler/core/golden_dump_test.go run next to the reference's
core.go (needs a Go toolchain and the k8s.io modules — neither exists where this library is developed).  While the file
is absent the pinning test is skipped and findMaxPG / getPreAllocatedResource / the prefix early exit /
computeResourceSatisfied / the PreFilter sequence stay "parity unpinned by the reference" (DESIGN.md section 6).  The
format, the loader and the comparison are exercised regardless, oracle against oracle."""
import copy
import json
import os

import pytest

import go_golden

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IN, DUMP = os.path.join(GOLD, "go_reference_input.json"), os.path.join(GOLD, "go_reference_dump.json")


def test_committed_input_is_what_the_generator_writes():
    assert json.load(open(IN)) == json.loads(json.dumps(go_golden.build_input()))


def test_loader_and_comparison_on_the_oracles_own_dump(orc):
    inp = json.load(open(IN))
    mine = go_golden.oracle_dump(inp, orc)
    assert len(mine["find_max_pg"]["leaders_seen"]) == 1 and mine["find_max_pg"]["leaders_seen"][0] != ""
    assert len(mine["single_node"]) == inp["classes"] * len(inp["nodes"]) * 2
    fits = [e["fits"] for e in mine["cluster_fits"]]
    assert any(fits) and not all(fits), "the query set exercises both outcomes of compareClusterResourceAndRequire"
    errs = {go_golden._err_class(e["err"]) for e in mine["prefilter_sequence"]}
    assert {"", "can not found pod group", "cluster resource not enough", "last failed in 20s"} <= errs, errs
    assert {e["err"] for e in mine["filter"]} >= {"", "resource not enough"}
    through_json = json.loads(json.dumps(mine))
    assert go_golden.compare(through_json, mine) == []
    # the comparison is not vacuous: any single perturbed entry is reported
    for sec, field in (("pre_allocated", "lanes"), ("single_node", "lanes"), ("cluster_fits", "first_k"), ("left_resource", "lanes")):
        bad = copy.deepcopy(through_json)
        e = next(x for x in bad[sec] if x[field] not in (None, -1))
        e[field] = [v + 1 for v in e[field]] if isinstance(e[field], list) else e[field] + 1
        assert go_golden.compare(bad, mine), sec
    bad = copy.deepcopy(through_json)
    bad["prefilter_sequence"][3]["err"] = "cluster resource not enough" if bad["prefilter_sequence"][3]["err"] == "" else ""
    assert go_golden.compare(bad, mine)


@pytest.mark.skipif(not os.path.exists(DUMP), reason="tests/golden/go_reference_dump.json absent: nobody has run golden_dump_test.go beside the reference yet")
def test_oracle_equals_the_references_own_outputs(orc):
    inp = json.load(open(IN))
    diffs = go_golden.compare(json.load(open(DUMP)), go_golden.oracle_dump(inp, orc))
    assert not diffs, diffs[:10]
