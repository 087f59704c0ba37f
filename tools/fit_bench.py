"""Times bs_fit_build (checkFit for every (class, node), core.go:741-759) on a seeded scene.  Parity with
the oracle is asserted in tests/test_fit_build.py (which also times the oracle); nothing under oracle/ is
used here.  Usage: python tools/fit_bench.py [nodes classes]...   (GPU box)"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: F401

capi = importlib.import_module("batch-scheduler_amd.capi")
synth = importlib.import_module("batch-scheduler_amd.synth")
fitspec = importlib.import_module("batch-scheduler_amd.fitspec")


def run(n, c, seed=20260921):
    nodes = synth.make_nodes(seed, n, 1, "warm")
    scene_nodes, templates = synth.make_fit_scene(seed, n, c)
    t0 = time.perf_counter()
    nl, ft = fitspec.marshal(scene_nodes, templates)
    t_marshal = time.perf_counter() - t0
    ctx = capi.Context(scalar_lanes=1)
    ctx.load_nodes(nodes)
    ctx.build_fit(nl, ft)                       # warm-up (allocations)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.build_fit(nl, ft)
    t_gpu = (time.perf_counter() - t0) / reps
    fits = int(ctx.read_fit().to_bool().sum())
    return {"nodes": n, "classes": c, "pairs": n * c, "labels": int(nl.label_off[-1]), "exprs": int(len(ft.exprs.key)),
            "bs_fit_build_ms": round(t_gpu * 1e3, 3), "pairs_per_s": round(n * c / t_gpu),
            "marshal_python_ms": round(t_marshal * 1e3, 1), "pairs_that_fit": fits,
            "note": "bs_fit_build time is host-observed: packing + one H2D + two kernels + mask D2H + table rebuild"}


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:]] or [5000, 200, 20000, 500]
    for i in range(0, len(args), 2):
        print(json.dumps(run(args[i], args[i + 1])))
