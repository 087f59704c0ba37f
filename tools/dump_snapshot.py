#!/usr/bin/env python3
"""Dump a seeded synthetic snapshot + the request vectors of its reservation queries as JSON, for the
un-run Go benchmark bench_go/prefilter_bench_test.go.  usage: dump_snapshot.py <config> <scenario> [seed]"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
bsa = importlib.import_module("batch-scheduler_amd")
import orc  # noqa: E402  (request vectors are computed with the oracle's getPreAllocatedResource)


def main():
    cfg, sc = sys.argv[1], sys.argv[2]
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20260921
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, sc, seed=seed)
    S = nodes.lanes - 4
    leader, _, _ = orc.find_max_pg(groups)
    pre, _ = orc.pre_allocated(groups, leader, int(groups.matched[leader]), S)
    reqs = [[int(pre[j] + pods.req[j, i]) for j in range(nodes.lanes)] for i in range(min(pods.p, 2000)) if pods.group[i] not in (-1, leader)]
    lanes = ["cpu", "memory", "ephemeral-storage", "pods"] + ["nvidia.com/gpu", "example.com/extra"][:S]
    json.dump({"lanes": lanes, "alloc": nodes.allocatable.tolist(), "requested": nodes.requested.tolist(),
               "req_key": [((nodes.requested_present >> s) & 1).astype(bool).tolist() for s in range(S)],
               "unschedulable": ((nodes.flags & 7) != 0).tolist(), "requests": reqs, "percent": 0.7}, sys.stdout)


if __name__ == "__main__":
    main()
