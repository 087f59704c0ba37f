"""Is the resident step's run-to-run spread a property of the PROCESS or of the context's allocations?  Several contexts one after the other in one
process (optionally with a dummy allocation of a different size in front of each, which moves everything the context allocates), us per step each.
usage: python tools/mode_probe.py cfg3 tail [--distinct --lanes K] [--shift]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa


def one(nodes, fit, groups, pods, steps):
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        out = []
        for _ in range(3):
            for _ in range(30):
                ctx.run(soa.STAGE_ALL)
            ctx.sync()
            t = time.perf_counter()
            for _ in range(steps):
                ctx.run(soa.STAGE_ALL)
            ctx.sync()
            out.append(round((time.perf_counter() - t) / steps * 1e6, 2))
        st = ctx.stats(soa.STAGE_ALL)
        return out, {k: st[k] for k in ("scan_evals_executed", "filter_evals_executed", "scan_queries", "filter_distinct", "filter_lane_blocks", "filter_tile_blocks") if k in st}


def main():
    cfg, scen = sys.argv[1], sys.argv[2]
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, scen)
    if "--distinct" in sys.argv:
        pods = bsa.synth.all_distinct(pods, nodes, int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else 1)
    import torch
    keep = []
    for i in range(6):
        if "--shift" in sys.argv:
            keep.append(torch.empty((1 + i) * 3 * 1024 * 1024 + 4096 * i, dtype=torch.uint8, device="cuda"))
        print(cfg, scen, "context", i, one(nodes, fit, groups, pods, 200 if cfg != "cfg4" else 60), flush=True)


if __name__ == "__main__":
    main()
