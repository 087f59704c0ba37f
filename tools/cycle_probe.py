#!/usr/bin/env python3
"""The queue-resident scheduling cycle in a tight loop (what bench.py's host_cycle 'resident' mode times), for profiling:
  python tools/cycle_probe.py [config=cfg3] [scenario=tail] [cycles=200]
prints host-observed p50 / p95 per call and in total; run it under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402

bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    scenario = sys.argv[2] if len(sys.argv) > 2 else "tail"
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    stages = soa.STAGE_ALL if os.environ.get("PROBE_FILTER", "1") == "1" else (soa.STAGE_PREFILTER | soa.STAGE_TALLY)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        ctx.run(stages)
        out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
        idx = np.random.default_rng(1).choice(groups.g, min(32, groups.g), replace=False)
        darr = (soa.GroupDelta * len(idx))(*[soa.GroupDelta(int(i), int(groups.matched[i]), int(groups.status_scheduled[i]), int(groups.flags[i])) for i in idx])
        structs, keep = bench.make_pod_deltas(bsa, pods, cycles + 10, max(2, pods.p // 100))
        parts = []
        zero_copy = os.environ.get("PROBE_READ", "map") == "map"      # bs_batch_map (no host copy) | bs_batch_read
        view = soa.BatchViewStruct()
        for it in range(cycles + 10):
            t0 = time.perf_counter()
            ctx.apply_group_deltas_raw(darr, len(idx))
            t1 = time.perf_counter()
            ctx.apply_pods_raw(structs[it])
            t2 = time.perf_counter()
            ctx.run(stages | soa.BATCH_HOST_RESULTS)
            t3 = time.perf_counter()
            if zero_copy:
                ctx.map_raw(view)
            else:
                ctx.read(out=out)
            t4 = time.perf_counter()
            if it >= 10:
                parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
        a = np.array(parts) * 1e6
        names = ("groups_apply", "pods_apply", "run", "read", "total")
        print(json.dumps({"workload": f"{config}/{scenario}", "cycles": cycles, "stages": int(stages), "results": "bs_batch_map" if zero_copy else "bs_batch_read",
                          "us_p50": {n: round(float(np.percentile(a[:, k], 50)), 1) for k, n in enumerate(names)},
                          "us_p95": {n: round(float(np.percentile(a[:, k], 95)), 1) for k, n in enumerate(names)},
                          "apply_stats": ctx.apply_stats()}))


if __name__ == "__main__":
    main()
