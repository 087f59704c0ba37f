#!/usr/bin/env python3
"""Dump a seeded synthetic snapshot + the request vectors of its reservation queries as JSON, for the
un-run Go benchmark go/pkg/scheduler/core/prefilter_bench_test.go.  usage: dump_snapshot.py <config> <scenario> [seed]"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
import numpy as np  # noqa: E402


def main():
    cfg, sc = sys.argv[1], sys.argv[2]
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20260921
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, sc, seed=seed)
    S = nodes.lanes - 4
    # leader = group with the largest progress permille (core.go:716-717); warm scenarios have a unique one
    cand = ((groups.flags & 1) == 0) & ((groups.flags & 2) != 0)
    fin = np.where(cand, (groups.matched + groups.status_scheduled).astype(np.int64) * 1000 // np.maximum(groups.min_member, 1), -1)
    leader = int(np.argmax(fin))
    nf = max(0, int(groups.min_member[leader]) - int(groups.matched[leader]))       # core.go:778-779
    pre = [int(groups.min_resources[j, leader]) * nf for j in range(nodes.lanes)]     # core.go:784-788
    if pre[3] == 0:
        pre[3] = int(groups.min_member[leader]) + 1                                   # core.go:789-791
    reqs = [[int(pre[j] + pods.req[j, i]) for j in range(nodes.lanes)] for i in range(min(pods.p, 2000)) if pods.group[i] not in (-1, leader)]
    lanes = ["cpu", "memory", "ephemeral-storage", "pods"] + ["nvidia.com/gpu", "example.com/extra"][:S]
    json.dump({"lanes": lanes, "alloc": nodes.allocatable.tolist(), "requested": nodes.requested.tolist(),
               "req_key": [((nodes.requested_present >> s) & 1).astype(bool).tolist() for s in range(S)],
               "unschedulable": ((nodes.flags & 7) != 0).tolist(), "requests": reqs, "percent": 0.7}, sys.stdout)


if __name__ == "__main__":
    main()
