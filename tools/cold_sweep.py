"""Launch-geometry sweep of the GENERAL chain (GPU box): BS_TARGET_WAVES / BS_FILTER_WAVES vs the scan / Filter kernel times.
Cold batches take the positional chain (bs_epoch.hpp) by default; BS_NO_EPOCH=1 keeps them on the chain this tool tunes."""
import importlib, time, sys, os
os.environ["BS_NO_EPOCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd"); soa = bsa.soa
for cfg, sc in (("cfg3", "cold"), ("cfg4", "cold")):
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, sc)
    for waves, fwaves in ((8192, 8192), (4096, 4096), (2048, 2048), (1024, 1024), (2048, 4096), (1024, 2048)):
        os.environ["BS_TARGET_WAVES"] = str(waves); os.environ["BS_FILTER_WAVES"] = str(fwaves)
        with bsa.Context(scalar_lanes=1, enable_timing=2) as ctx:
            ctx.load_nodes(nodes, fit); ctx.load_groups(groups); ctx.load_pods(pods)
            for _ in range(5): ctx.run(soa.STAGE_ALL)
            ctx.sync(); ctx.timing_reset()
            for _ in range(10): ctx.run(soa.STAGE_ALL)
            tm = ctx.timing()
        with bsa.Context(scalar_lanes=1) as ctx:
            ctx.load_nodes(nodes, fit); ctx.load_groups(groups); ctx.load_pods(pods)
            for _ in range(5): ctx.run(soa.STAGE_ALL)
            ctx.sync(); t = time.perf_counter()
            for _ in range(30): ctx.run(soa.STAGE_ALL)
            ctx.sync(); tot = (time.perf_counter() - t) / 30 * 1e6
        print(cfg, sc, "waves", waves, fwaves, "scan %.1f filter %.1f total %.1f" % (tm["scan"][0] / tm["scan"][1] * 1e3, tm["filter"][0] / tm["filter"][1] * 1e3, tot), flush=True)
