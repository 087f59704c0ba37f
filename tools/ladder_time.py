"""Time the leader-ladder scene (a cold cfg2 snapshot whose queue changes leader `steps - 1` times, the scene of
tests/test_gpu_epoch.py::test_leader_ladder_and_the_run_limit) on the default chain and on the general chain
(BS_NO_EPOCH=1).  GPU only; prints one line per (steps, chain).  Evidence: profiles/r04_leader_ladder_16_runs.txt"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa


def ladder_scene(steps, seed=4):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold", seed=seed)
    pods = pods.take(np.argsort(np.where(pods.group < 0, 0, pods.group), kind="stable"))      # queue in group order
    first = {int(g): int(np.nonzero(pods.group == g)[0][0]) for g in range(8)}
    for i in np.nonzero(pods.group >= 0)[0]:                                                   # eight request templates
        src = first[int(pods.group[i]) % 8]
        pods.req[:, i] = pods.req[:, src]
        pods.req_present[i] = pods.req_present[src]
    groups.min_member[:steps] = 50
    groups.matched[:steps] = np.arange(steps, dtype=np.uint32) * 3                             # strictly rising progress
    return nodes, fit, groups, pods


def main():
    for steps in (3, 9, 15, 16):
        for no_epoch in (False, True):
            if no_epoch:
                os.environ["BS_NO_EPOCH"] = "1"
            else:
                os.environ.pop("BS_NO_EPOCH", None)
            nodes, fit, groups, pods = ladder_scene(steps)
            with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
                ctx.load_nodes(nodes, fit)
                ctx.load_groups(groups)
                ctx.load_pods(pods)
                for _ in range(10):
                    ctx.run(soa.STAGE_ALL)
                ctx.sync()
                t = time.perf_counter()
                for _ in range(200):
                    ctx.run(soa.STAGE_ALL)
                ctx.sync()
                dt = (time.perf_counter() - t) / 200
                st = ctx.stats(soa.STAGE_ALL)
            label = "general chain (BS_NO_EPOCH)" if no_epoch else "default"
            print(f"ladder steps {steps:2d} {label:28s} chain {st['chain']} launches {st['launches']:2d}  {dt * 1e6:7.1f} us per batch")


if __name__ == "__main__":
    main()
