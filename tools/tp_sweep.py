"""The throughput regime of the steady-state chain (every pod its own request: bench.py scenarios.all_distinct_requests), every form
of launch B behind BS_TP_FILTER (0: scan + Filter roles in one launch; 1..4: k_fast_scan + a lean Filter kernel; 5: k_fast_scan +
the transposed item, csrc/bs_filter_t.hpp; 6 / 7: one launch, Filter role by the transposed item) x BS_TP_SHARE (scan shares per tile), in ONE process: the scene is built once, every
combination gets its own context (the switches are read when a context is created).  Per combination: us per resident step (best
and median of REPS x 300 steps back to back), the step's per-kernel device times (bs_batch timing, when --kernels), and a digest of
every output array — all forms must agree with form 0 bit for bit (the line says so).  GPU only; no oracle, no test imports.

usage: python tools/tp_sweep.py cfg3|cfg4 [tail] [--forms -1,0,1,3,5  (-1 = the library's defaults)] [--shares 64,16,4] [--fwaves 8192,4096] [--kernels] [--plain] [--shard r/n[,r/n...]] [--split 1,8]
  --split m: BS_TP_SPLIT, the transposed Filter items are cut for m x the launched waves (0 / absent: the library's rule = the number of ranks)
  --shard r/n: the step of rank r of n (pod-axis shard on this one context, no collective): what a rank's launches cost
  --plain: the scene as synthesised (requests shared within a gang) instead of all-distinct
  --lanes k: the requests differ on k of the four fixed lanes (1 = rounds 3-5's scene: cpu only; 4 = cpu, memory, ephemeral storage, pods all bind)
  (BS_NO_NODEW=1 in the environment: round 5's item, without the batch's node words — the A/B switch of round 6)"""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
REPS = 3


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def digest(out):
    h = hashlib.sha256()
    for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
        h.update(np.ascontiguousarray(getattr(out, name)).tobytes())
    ev = out.fl_code == soa.FL_EVALUATED
    if out.fl_rows is not None and ev.any():                      # the rows of the evaluated pods, in pod order (slot numbering is the library's business)
        rows = np.ascontiguousarray(out.fl_rows[:, out.fl_slot[ev]])
        h.update(rows.tobytes())
    return h.hexdigest()[:16]


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--") and not a[0].isdigit() and not a[0] == "-"]
    cfg = pos[0] if pos else "cfg3"
    scen = pos[1] if len(pos) > 1 else "tail"
    forms = [int(x) for x in arg("--forms", "0,1,2,3,4,5").split(",")]
    shares = [int(x) for x in arg("--shares", "64,16,4").split(",")]
    fwaves = [int(x) for x in arg("--fwaves", "8192").split(",")]          # BS_FILTER_WAVES: waves the Filter work is cut for (8192 = the default)
    if "--lib" in sys.argv:                                      # an experiment build of the library (never the shipped one)
        bsa.capi.LIB_PATH = os.path.abspath(arg("--lib", ""))
        bsa.capi.load_library()
    nodes, fit, groups, pods, _ = bsa.synth.make(cfg, scen)
    lanes = int(arg("--lanes", "1"))                             # --lanes k: every request distinct on k of the four lanes Filter compares (synth.all_distinct)
    if "--plain" not in sys.argv:
        pods = bsa.synth.all_distinct(pods, nodes, lanes)
    shards = [[int(x) for x in sh.split("/")] for sh in arg("--shard", "").split(",")] if "--shard" in sys.argv else [None]      # "--shard 0/2,0/8,7/8"
    ref = None
    splits = [int(x) for x in arg("--split", "0").split(",")]
    for form, share, fw, shard, split in [(f, s, w, sh, sp) for f in forms for s in shares for w in fwaves for sh in shards for sp in splits]:
        for k in ("BS_TP_FILTER", "BS_TP_SHARE", "BS_FILTER_WAVES", "BS_TP_SPLIT"):
            os.environ.pop(k, None)
        if split:
            os.environ["BS_TP_SPLIT"] = str(split)
        if form >= 0:                                          # form -1: the library's defaults (no switch set)
            os.environ["BS_TP_FILTER"] = str(form)
            os.environ["BS_TP_SHARE"] = str(share)
            os.environ["BS_FILTER_WAVES"] = str(fw)
        with bsa.Context(scalar_lanes=nodes.lanes - 4, enable_timing=1 if "--kernels" in sys.argv else 0) as ctx:
            ctx.load_nodes(nodes, fit)
            ctx.load_groups(groups)
            ctx.load_pods(pods)
            if shard:                                          # rank r of n on this one context (bs_shard_set: the whole queue resident, ownership on the device)
                ctx.set_shard(shard[0], shard[1])
            out = ctx.batch(soa.STAGE_ALL, bitmap=False)
            d = digest(out)
            if ref is None or shard:                           # (a rank's outputs are its own: no comparison across shards)
                ref = d
            res = []
            for _ in range(REPS):
                for _ in range(20):
                    ctx.run(soa.STAGE_ALL)
                ctx.sync()
                t = time.perf_counter()
                for _ in range(300):
                    ctx.run(soa.STAGE_ALL)
                ctx.sync()
                res.append((time.perf_counter() - t) / 300 * 1e6)
            line = {"config": cfg, "scenario": scen, "distinct": "--plain" not in sys.argv, "lanes": lanes, "no_nodew": os.environ.get("BS_NO_NODEW", "0"), "form": form, "share": share, "filter_waves": fw, "shard": shard, "split": split,
                    "us_per_step_best": round(min(res), 2), "us_per_step_median": round(sorted(res)[len(res) // 2], 2), "digest": d,
                    "same_as_first": d == ref}
            if "--kernels" in sys.argv:
                ctx.timing_reset()
                for _ in range(50):
                    ctx.run(soa.STAGE_ALL)
                ctx.sync()
                line["kernel_us"] = {k: round(v[0] * 1000 / max(v[1], 1), 2) for k, v in ctx.timing().items() if v[1] > 0}      # mean per launch group
            st = ctx.stats(soa.STAGE_ALL)
            line.update({k: st[k] for k in ("chain", "launches", "filter_evals_executed", "scan_evals_executed", "filter_distinct")})
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
