#!/usr/bin/env python3
"""Sequential 1:1 drop-in mode timed: one reference-style PreFilter call per pod through the C++ host mirror
(libbsched_host.so), every node loop on the GPU through the ABI (bs_find_max_pg + bs_cluster_fits).
This is the 'sequential replay with the GPU node loop' variant of SURVEY.md 8(d) — it shows what per-call
launch latency costs and why the batched path exists.  usage: seq_replay_bench.py [config] [scenario]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

bsa = importlib.import_module("batch-scheduler_amd")


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    sc = sys.argv[2] if len(sys.argv) > 2 else "tail"
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, sc)
    L = nodes.lanes
    with bsa.Context(scalar_lanes=L - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = bsa.plugin.ScheduleOperation(ctx)
        for g in range(groups.g):
            has_mr = bool(groups.flags[g] & bsa.soa.GROUP_HAS_MINRES)
            sop.add_group(int(groups.min_member[g]), int(groups.status_scheduled[g]), creation_ts=g, name_rank=g,
                          min_resources=groups.min_resources[:, g].tolist() if has_mr else None,
                          min_resources_present=int(groups.min_resources_present[g]))
        n = min(pods.p, 2000)
        lat = []
        t0 = time.perf_counter()
        for i in range(n):
            a = time.perf_counter()
            sop.PreFilter(i + 1, i + 1, int(pods.group[i]), pods.req[:, i].tolist(), int(pods.req_present[i]), int(pods.cls[i]), int(pods.owner[i]))
            lat.append(time.perf_counter() - a)
        dt = time.perf_counter() - t0
        print(json.dumps({"mode": "sequential PreFilter via host mirror + GPU node loop", "config": cfg, "scenario": sc, "pods": n, "nodes": nodes.n,
                          "pods_per_s": n / dt, "logical_evals_per_s": n * nodes.n / dt, "prefilter_latency_us_p50": float(np.percentile(lat, 50) * 1e6),
                          "gpu_calls": sop.gpu_calls}))
        sop.close()


if __name__ == "__main__":
    main()
