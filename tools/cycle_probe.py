"""Host-side anatomy of the resident scheduling cycle (bs_groups_apply -> bs_pods_apply -> bs_batch_run (latency mode) -> bs_batch_map):
per-call host time p50 and, with BS_HOST_PROBE=1, where bs_batch_run's own time goes (printed by the library at bs_destroy).  GPU only."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
if os.environ.get("BS_AB_LIB"):                      # A/B runs of build variants (tools/ubench/*.so)
    bsa.capi.LIB_PATH = os.path.abspath(os.environ["BS_AB_LIB"])
import bench  # noqa: E402  (make_pod_deltas: the same queue deltas the bench's resident cycle uses)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, "tail")
    iters = 200
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        rng = np.random.default_rng(1)
        idx = rng.choice(groups.g, min(32, groups.g), replace=False)
        deltas = [(int(i), int(groups.matched[i]), int(groups.status_scheduled[i]), int(groups.flags[i])) for i in idx]
        darr = (soa.GroupDelta * len(deltas))(*[soa.GroupDelta(*d) for d in deltas])
        structs, keep = bench.make_pod_deltas(bsa, pods, iters + 5, max(2, pods.p // 100))
        view = soa.BatchViewStruct()
        parts = [[], [], [], [], []]
        for it in range(iters + 5):
            t0 = time.perf_counter()
            ctx.apply_group_deltas_raw(darr, len(deltas))
            t1 = time.perf_counter()
            ctx.apply_pods_raw(structs[it])
            t2 = time.perf_counter()
            ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
            t3 = time.perf_counter()
            ctx.map_raw(view)
            t4 = time.perf_counter()
            if it >= 5:
                for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
                    parts[k].append(v * 1e6)
        names = ("groups_apply", "pods_apply", "run", "map", "total")
        print(cfg, "resident cycle, us p50 / p95:", {n: (round(float(np.percentile(p, 50)), 1), round(float(np.percentile(p, 95)), 1)) for n, p in zip(names, parts)},
              "speculation", ctx.speculation_stats())


if __name__ == "__main__":
    main()
