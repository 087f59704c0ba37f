"""Throughput regime, what the SCAN role has to do: per config the distribution of first_k over the batch's scan queries (how many 64-row
groups a tile of 64 distinct requests has to walk before its last lane has its row), rows walked (work counters) and per-kernel times.
usage: python tools/tp_scan_probe.py [cfg3 cfg4]"""
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa

for cfg in (sys.argv[1:] or ["cfg3", "cfg4"]):
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, "tail")
    pods = pods.copy()
    pods.req[0, :] += np.arange(pods.p, dtype=np.int64)
    with bsa.Context(scalar_lanes=nodes.lanes - 4, enable_timing=1) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        out = ctx.batch(soa.STAGE_ALL, bitmap=False)
        fk = out.pf_first_k.astype(np.int64)
        scanned = (out.pf_code == soa.PF_PASS_RESERVE_FITS) | (out.pf_code == soa.PF_REJECT_RESERVE) | (out.pf_code == soa.PF_PASS_FIRST_FITS) | (out.pf_code == soa.PF_REJECT_FIRST)
        found = scanned & (fk < nodes.n)
        q = fk[found]
        line = {"config": cfg, "pods": int(pods.p), "nodes": int(nodes.n), "scan_queries": int(scanned.sum()), "found": int(found.sum()), "rejected": int((scanned & ~found).sum())}
        if q.size:
            line["first_k_percentiles"] = {str(p): int(np.percentile(q, p)) for p in (50, 90, 99, 100)}
            # a tile of 64 consecutive queries is done when its LAST lane has its row
            tiles = [q[i:i + 64].max() for i in range(0, q.size, 64)]
            line["tile_last_row_percentiles"] = {str(p): int(np.percentile(tiles, p)) for p in (50, 90, 99, 100)}
        ctx.timing_reset()
        for _ in range(50):
            ctx.run(soa.STAGE_ALL)
        ctx.sync()
        line["kernel_us"] = {k: round(v[0] * 1000 / max(v[1], 1), 2) for k, v in ctx.timing().items() if v[1] > 0}
        st = ctx.stats(soa.STAGE_ALL)
        line["stats"] = st
        print(json.dumps(line), flush=True)
