"""Throughput regime, sharded: every rank's owned pods against the single batch's, at full size (cfg4 all-distinct by default).
usage: python tools/tp_shard_check.py [cfg4] [nranks=8]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nodes, fit, groups, pods, _ = bsa.synth.make(cfg, "tail")
pods = pods.copy()
pods.req[0, :] += np.arange(pods.p, dtype=np.int64)
with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
    ctx.load_nodes(nodes, fit)
    ctx.load_groups(groups)
    ctx.load_pods(pods)
    full = ctx.batch(soa.STAGE_ALL, bitmap=False)
    owned = np.zeros(pods.p, np.uint32)
    admit = np.zeros(groups.g, np.uint32)
    bad = 0
    for r in range(nr):
        ctx.set_shard(r, nr)
        part = ctx.batch(soa.STAGE_ALL, bitmap=False)
        mine = part.pf_code != 0xFF
        owned += mine
        admit += part.group_admit
        for name in ("pf_code", "pf_first_k", "fl_code", "fl_feasible"):
            a, b = getattr(part, name)[mine], getattr(full, name)[mine]
            if not np.array_equal(a, b):
                idx = np.nonzero(a != b)[0]
                bad += 1
                print(f"rank {r}: {name} differs at {idx.size} owned pods, first {idx[:5]}: {a[idx[:5]]} vs {b[idx[:5]]}")
    print("owned exactly once:", bool(np.all(owned == 1)), "admit union equal:", bool(np.array_equal(admit, full.group_admit)), "differences:", bad)
