"""ms per resident step of one configuration on the library BS_AB_LIB names (default: the in-tree build) — for A/B runs of build
variants (tools/ubench/libbsched_nt.so = -DBS_NT_TABLES: launch A's table rows stored nontemporally).  GPU only."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
if os.environ.get("BS_AB_LIB"):
    bsa.capi.LIB_PATH = os.path.abspath(os.environ["BS_AB_LIB"])


def main():
    cfg, scen = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cfg3", "tail")
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, scen)
    if "--distinct" in sys.argv:                      # every pod asks for something else: no request is shared (bench.py scenarios.all_distinct_requests)
        pods = bsa.synth.all_distinct(pods, nodes, int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else 1)
    reps, steps = (1, int(sys.argv[sys.argv.index("--steps") + 1])) if "--steps" in sys.argv else (5, 300)      # --steps n: one short run (under a profiler)
    res = []
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        for rep in range(reps):
            for _ in range(20):
                ctx.run(soa.STAGE_ALL)
            ctx.sync()
            t = time.perf_counter()
            for _ in range(steps):
                ctx.run(soa.STAGE_ALL)
            ctx.sync()
            res.append((time.perf_counter() - t) / steps * 1e6)
        st = ctx.stats(soa.STAGE_ALL)
    print(cfg, scen, os.environ.get("BS_AB_LIB", "in-tree"), "us per step:", [round(x, 2) for x in res], {k: st[k] for k in ("chain", "launches", "scan_evals_executed", "filter_evals_executed", "scan_queries")})


if __name__ == "__main__":
    main()
