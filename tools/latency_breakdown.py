"""Host-observed breakdown of one scheduling cycle (what host_cycle in bench.py adds up): bs_groups_apply,
bs_pods_load (pack + H2D + class / pair kernels, asynchronous), bs_batch_run, bs_batch_read — once as the cycle
runs (no wait between the calls) and once with a stream wait after every call to see what each one costs the GPU.
Usage (GPU box): python tools/latency_breakdown.py [config=cfg3] [scenario=tail]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np

bsa = importlib.import_module("batch-scheduler_amd")
soa, synth = bsa.soa, bsa.synth


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    scenario = sys.argv[2] if len(sys.argv) > 2 else "tail"
    nodes, fit, groups, pods, _ = synth.make(config, scenario)
    ctx = bsa.Context(scalar_lanes=nodes.lanes - 4)
    ctx.load_nodes(nodes, fit)
    ctx.load_groups(groups)
    ctx.load_pods(pods)
    ctx.run(soa.STAGE_ALL)
    out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
    small = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False)
    idx = np.random.default_rng(1).choice(groups.g, min(32, groups.g), replace=False)
    deltas = [(int(i), int(groups.matched[i]), int(groups.status_scheduled[i]), int(groups.flags[i])) for i in idx]
    names = ["groups_apply", "pods_load", "batch_run", "batch_read"]
    res = {}
    for mode in ("as_run", "drained_after_each_call", "read_without_rows"):
        rows = []
        for it in range(50):
            ts = [time.perf_counter()]
            ctx.apply_group_deltas(deltas)
            if mode == "drained_after_each_call":
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.load_pods(pods)
            if mode == "drained_after_each_call":
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.run(soa.STAGE_ALL)
            if mode == "drained_after_each_call":
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.read(out=small if mode == "read_without_rows" else out)
            ts.append(time.perf_counter())
            if it >= 10:
                rows.append([ts[k + 1] - ts[k] for k in range(4)] + [ts[4] - ts[0]])
        med = np.median(np.array(rows), axis=0) * 1e6
        res[mode] = {n: round(float(m), 1) for n, m in zip(names + ["total"], med)}
    # the queue-resident cycle: bs_pods_apply (1 % of the queue per cycle) instead of bs_pods_load, results in latency mode
    sys.path.insert(0, ROOT)
    import bench
    structs, keep = bench.make_pod_deltas(bsa, pods, 120, max(2, pods.p // 100))
    k = 0
    for mode in ("resident_as_run", "resident_drained_after_each_call"):
        rows = []
        for it in range(50):
            ts = [time.perf_counter()]
            ctx.apply_group_deltas(deltas)
            if mode.endswith("each_call"):
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.apply_pods_raw(structs[k]); k += 1
            if mode.endswith("each_call"):
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
            if mode.endswith("each_call"):
                ctx.sync()
            ts.append(time.perf_counter())
            ctx.read(out=out)
            ts.append(time.perf_counter())
            if it >= 10:
                rows.append([ts[j + 1] - ts[j] for j in range(4)] + [ts[4] - ts[0]])
        med = np.median(np.array(rows), axis=0) * 1e6
        res[mode] = {n: round(float(m), 1) for n, m in zip(["groups_apply", "pods_apply", "batch_run", "batch_read", "total"], med)}
    print(json.dumps({"workload": f"{config}/{scenario}", "us_p50": res,
                      "note": "drained_after_each_call: a stream wait after every call (only to split the phases; the cycle itself has one wait, inside bs_batch_read)"}))


if __name__ == "__main__":
    main()
