"""bs_seq_run (the reference's pod-by-pod cycle on the device) timed on a synthetic configuration; bench.py puts the CPU port's
sequential pass beside it (`drain.sequential_on_device`).  Usage: python tools/seq_bench.py [config] [scenario] [--filter]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    config = args[0] if args else "cfg3"
    scenario = args[1] if len(args) > 1 else "tail"
    st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if "--filter" in sys.argv else 0)
    if "--probe" in sys.argv:            # the -DBS_SEQ_PROBE build: cycles per phase go to stderr
        bsa.capi.LIB_PATH = os.environ.get("BS_SEQ_PROBE_LIB") or os.path.join(ROOT, "tools", "ubench", "libbsched_seqprobe.so")
        os.environ["BS_SEQ_PROBE_PRINT"] = "1"
    if os.environ.get("BS_AB_LIB"):        # A/B runs of build variants
        bsa.capi.LIB_PATH = os.path.abspath(os.environ["BS_AB_LIB"])
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    pods = pods.take(np.argsort(pods.group, kind="stable"))           # Compare order
    runs = []
    for rep in range(3):
        with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
            ctx.load_nodes(nodes, fit)
            ctx.load_groups(groups)
            ctx.load_pods(pods)
            t = time.perf_counter()
            r = ctx.seq_run(st)
            wall = time.perf_counter() - t
        runs.append((wall, r))
    wall, r = min(runs, key=lambda x: x[0])
    lat = (r["ready_ns"] - r["first_ns"]) * 1e-6
    print(json.dumps({
        "config": f"{config}/{scenario}", "filter": bool(st & soa.STAGE_FILTER), "pods": int(pods.p), "nodes": int(nodes.n), "groups": int(groups.g),
        "gangs_released": r["n_released"],
        "gpu": {"wall_ms": wall * 1e3, "device_ms": r["total_ns"] * 1e-6, "us_per_pod": r["total_ns"] * 1e-3 / max(pods.p, 1), "node_picks": r["node_picks"], "node_scans": r["node_scans"], "scan_rounds": r["scan_rounds"], "pick_rounds": r["pick_rounds"], "leader_folds": r["leader_folds"], "table_builds": r["table_builds"],
                "gang_admit_latency_ms_p50": float(np.median(lat)) if lat.size else None, "gang_admit_latency_ms_p95": float(np.percentile(lat, 95)) if lat.size else None},
    }))


if __name__ == "__main__":
    main()
