"""bs_seq_run (the reference's pod-by-pod cycle on the device) timed beside the oracle's sequential pass on the same inputs.
Usage: python tools/seq_bench.py [config] [scenario] [--filter]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
import orc  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    config = args[0] if args else "cfg3"
    scenario = args[1] if len(args) > 1 else "tail"
    st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if "--filter" in sys.argv else 0)
    orc.build()
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    pods = pods.take(np.argsort(pods.group, kind="stable"))           # Compare order
    s = orc.seq_replay(nodes, fit, groups, pods, st)
    lat_cpu = (s["ready_ns"] - s["first_ns"]) * 1e-6
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
    same = r["released_group"].tolist() == s["released_group"].tolist() and np.array_equal(r["pod_node"], s["pod_node"]) and np.array_equal(r["pf_code"], s["pf_code"])
    print(json.dumps({
        "config": f"{config}/{scenario}", "filter": bool(st & soa.STAGE_FILTER), "pods": int(pods.p), "nodes": int(nodes.n), "groups": int(groups.g),
        "same_as_cpu_pass": bool(same), "gangs_released": r["n_released"],
        "gpu": {"wall_ms": wall * 1e3, "device_ms": r["total_ns"] * 1e-6, "us_per_pod": r["total_ns"] * 1e-3 / max(pods.p, 1), "node_passes": r["node_passes"], "node_scans": r["node_scans"],
                "gang_admit_latency_ms_p50": float(np.median(lat)) if lat.size else None, "gang_admit_latency_ms_p95": float(np.percentile(lat, 95)) if lat.size else None},
        "cpu_port_1_core": {"total_ms": s["total_ns"] * 1e-6, "us_per_pod": s["total_ns"] * 1e-3 / max(pods.p, 1),
                            "gang_admit_latency_ms_p50": float(np.median(lat_cpu)) if lat_cpu.size else None, "gang_admit_latency_ms_p95": float(np.percentile(lat_cpu, 95)) if lat_cpu.size else None},
    }))


if __name__ == "__main__":
    main()
