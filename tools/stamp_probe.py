#!/usr/bin/env python3
"""Where the microseconds of the latency-bound launches go: in-kernel clock stamps of the resident cycle / the resident step.

Needs the PROBE build of the library (never the shipped one):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DBS_PROBE=1 -DBS_UNITY \
        -o tools/ubench/libbsched_probe.so batch-scheduler_amd/csrc/bsched.hip -ldl
  python tools/stamp_probe.py [cycle|step] [config=cfg3] [scenario=tail] [reps=40]

Thread 0 of the first 128 blocks of k_pods_apply (0), launch A (1), the scan / Filter blocks (2) and the final blocks (3) of
the second launch stamps s_memrealtime (100 MHz) at entry (0),
at a few points inside (after draining its outstanding memory operations) and at exit (7).  Printed per launch, in microseconds
relative to the FIRST block entry of the cycle's first launch, median over the repetitions:
  first / last block entry, last block exit, and per stamp the median / maximum over the blocks of (stamp - block entry)."""
import ctypes as C
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402

bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
NAMES = {0: "k_pods_apply", 1: "A k_fast_query_tables / pod blocks of k_fast_step_a", 2: "B producer blocks of k_fast_scan_filter_final / table blocks of k_fast_step_a", 3: "C final blocks", 4: "F Filter blocks of the whole-step launch"}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "cycle"
    config = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
    scenario = sys.argv[3] if len(sys.argv) > 3 else "tail"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    bsa.capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libbsched_probe.so")
    lib = bsa.capi.load_library()
    lib.bs_probe_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    stages = soa.STAGE_ALL if os.environ.get("PROBE_FILTER", "1") == "1" else (soa.STAGE_PREFILTER | soa.STAGE_TALLY)
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario)
    if os.environ.get("PROBE_DISTINCT") == "1":             # every pod its own request: the throughput regime (k_fast_scan_filter_t)
        pods = pods.copy()
        pods.req[0, :] += np.arange(pods.p, dtype=np.int64)
    if os.environ.get("PROBE_SHARD"):                       # "r/n": rank r of n on this one context
        shard = [int(x) for x in os.environ["PROBE_SHARD"].split("/")]
    else:
        shard = None
    buf = np.zeros((8, 128, 8), np.uint64)
    recs = []
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        if shard:
            ctx.set_shard(shard[0], shard[1])
        ctx.run(stages)
        ctx.sync()
        idx = np.random.default_rng(1).choice(groups.g, min(32, groups.g), replace=False)
        darr = (soa.GroupDelta * len(idx))(*[soa.GroupDelta(int(i), int(groups.matched[i]), int(groups.status_scheduled[i]), int(groups.flags[i])) for i in idx])
        structs, keep = bench.make_pod_deltas(bsa, pods, reps + 10, max(2, pods.p // 100))
        view = soa.BatchViewStruct()
        for it in range(reps + 10):
            lib.bs_probe_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)))      # drains the stream and clears the stamps
            if mode == "cycle":
                ctx.apply_group_deltas_raw(darr, len(idx))
                ctx.apply_pods_raw(structs[it])
                ctx.run(stages | soa.BATCH_HOST_RESULTS)
                ctx.map_raw(view)
            else:
                ctx.run(stages)
                ctx.run(stages)                                                        # the second of two back-to-back steps is the one looked at
            lib.bs_probe_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
            if it >= 10:
                recs.append(buf.copy())
    out = {}
    kernels = [k for k in range(5) if any(r[k, :, 0].any() for r in recs)]
    per = {k: [] for k in kernels}
    for r in recs:
        t0 = min(int(r[k, :, 0][r[k, :, 0] > 0].min()) for k in kernels)
        for k in kernels:
            ent = r[k, :, 0].astype(np.int64)
            live = ent > 0
            ex = r[k, :, 7].astype(np.int64)
            row = {"blocks": int(live.sum()), "first_entry": (ent[live].min() - t0) / 100.0, "last_entry": (ent[live].max() - t0) / 100.0,
                   "last_exit": (ex[live & (ex > 0)].max() - t0) / 100.0 if (live & (ex > 0)).any() else float("nan")}
            for s in range(1, 8):
                st = r[k, :, s].astype(np.int64)
                ok = live & (st > 0)
                if ok.any():
                    d = (st[ok] - ent[ok]) / 100.0
                    row[f"s{s}_med"] = float(np.median(d))
                    row[f"s{s}_max"] = float(d.max())
            per[k].append(row)
    for k in kernels:
        keys = sorted({kk for row in per[k] for kk in row})
        out[NAMES[k]] = {kk: round(float(np.median([row[kk] for row in per[k] if kk in row])), 2) for kk in keys}
    # per-block medians (relative to the cycle's first block entry): rows = blocks, columns = stamps 0..7
    blocks = {}
    for k in kernels:
        rel = []
        for r in recs:
            t0 = min(int(r[kk, :, 0][r[kk, :, 0] > 0].min()) for kk in kernels)
            a = r[k].astype(np.float64)
            a[a == 0] = np.nan
            rel.append((a - t0) / 100.0)
        med = np.nanmedian(np.array(rel), axis=0)
        blocks[NAMES[k]] = [[None if np.isnan(x) else round(float(x), 2) for x in row] for row in med if not np.isnan(row[0])]
    dump = os.environ.get("PROBE_DUMP")
    if dump:
        json.dump({"mode": mode, "workload": f"{config}/{scenario}", "blocks": blocks}, open(dump, "w"))
    print(json.dumps({"mode": mode, "workload": f"{config}/{scenario}", "stages": int(stages), "reps": reps, "unit": "us", "launches": out}, indent=1))


if __name__ == "__main__":
    main()
