"""Host-observed breakdown of one admit cycle (what gang_admit_latency_ms_p50 in bench.py adds up):
bs_pods_load (pack + H2D + request classes, asynchronous), bs_batch_run + sync, bs_batch_read.
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
    out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False)
    rows = []
    for it in range(40):
        t0 = time.perf_counter()
        ctx.load_pods(pods)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        ctx.run(soa.STAGE_ALL)
        t3 = time.perf_counter()
        ctx.sync()
        t4 = time.perf_counter()
        ctx.read(bitmap=False, out=out)
        t5 = time.perf_counter()
        if it >= 10:
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0))
    med = np.median(np.array(rows), axis=0) * 1e6
    print(json.dumps({"workload": f"{config}/{scenario}", "us_p50": {
        "pods_load_call (pack + enqueue)": round(float(med[0]), 1), "upload + class kernels drain": round(float(med[1]), 1),
        "batch_run call (enqueue)": round(float(med[2]), 1), "batch drain": round(float(med[3]), 1),
        "batch_read (D2H + unpack)": round(float(med[4]), 1), "total with the two extra syncs": round(float(med[5]), 1)},
        "note": "the extra ctx.sync() calls exist only to split the phases; bench.py's latency loop has none between load, run and read"}))


if __name__ == "__main__":
    main()
