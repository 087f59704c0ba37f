#!/usr/bin/env python3
"""bench.py — pod x node fit evaluations / s of the batched gang-feasibility path on MI355X.

One "step" = one pass of the hot path over one batch that is already resident in HBM:
PreFilter (core.go:88-167) for every pending pod against the node snapshot (running-sum table rebuilt every
step), Filter (core.go:514-564) for every (pod, node), per-group admit counts and the Permit quorum
(core.go:303).  Workload at N=1: BASELINE.json's metric configuration "10k pods x 5k nodes" (configs[2]:
10k pods / 2k groups / 5k nodes, 4 resource dims), scenario "tail" (synth.py).
N>1: the pod axis is sharded over ranks with ONE all-reduce of the per-group admit counters per step;
  --scaling weak   (default) N x 10k pods, N x 2k groups, the same 5k nodes replicated
  --scaling strong the configuration is FIXED (use --config cfg4: 50k / 5k / 20k) and split over the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement).  Beside the contract's fields:
  roofline        the longest launch of the step, priced on EXECUTED work (no figure can exceed 1); `roofline_launches`
                  has every launch; `work_avoided` says how much of the logical pods x nodes space was never evaluated
  host_cycle      SURVEY 8(d)'s host-observed scheduling cycle: group patch + pod H2D + batch + decision / Filter-row
                  D2H, p50 / p95, and the evals/s that corresponds to it (PCIe inclusive — never `value`)
  scenarios       cold / warm / busy / all-distinct requests / PreFilter-only, seeds 1..3
  cpu_baseline    the oracle (C port of the Go path) on 1 core and on all host cores
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import importlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)
SIMDS, CLOCK_HZ = 1024, 2.4e9
# VALU issue cost of the lean Filter loop per (node, compared resource lane) and 64 request slots: one v_cmp_ge_i64 into an SGPR pair + one v_addc_co_u32
# (tools/ubench/lane_loop.hip, profiles/r06_lane_loop_ubench.txt: 8.35 cycles per SIMD at eight waves per SIMD, 8.8-9.6 at three): the bound the
# throughput regime's launch B is priced against (roofline_throughput)
CYCLES_PER_NODE_LANE = 8.35
TIMED_REGIONS = 5           # the K-step timed region is run this many times (each bracketed by barrier + synchronize); ms_per_step = the median region
# the steady-state step's launches by the library's timing group -> kernel name in a rocprofv3 trace.  Two forms: the one-launch
# form of launch A + the scan / Filter roles followed by the final launch (the default where it applies, round 6), or launches A and B+C
LAUNCH_KERNELS_ONE = {"query": "k_fast_step_a", "resolve": "k_fast_final"}
LAUNCH_KERNELS_WHOLE = {"query": "k_fast_step_a"}     # round 6, the whole step in one launch (BS_STEP_A=3, the default where it applies)
LAUNCH_KERNELS_TWO = {"query": "k_fast_query_tables", "scan": "k_fast_scan_filter_final"}
LAUNCH_KERNELS = dict(LAUNCH_KERNELS_TWO)            # (set by main() once it knows which form the timed steps took)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--scenario", default="tail")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--seed", type=int, default=20260921)
    ap.add_argument("--stages", default="all", choices=["all", "prefilter"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip scenarios / seeds / host-cycle measurements")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run under rocprofv3 --pmc for HBM traffic")
    ap.add_argument("--inner-pmc", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-reps", type=int, default=2)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baseline
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    return orc


def cpu_baseline(bsa, nodes, fit, groups, pods, stages, reps):
    """The oracle (C port of the Go path) on the same inputs and stages — checker code, timed as the baseline only.
    1 core: upstream runs PreFilter on the single scheduling goroutine.  All cores: whole groups dealt over one thread per
    host core (exact here: no pod's decision depends on a pod of another group in this scenario)."""
    orc = _oracle()
    snap = orc.Snapshot(nodes, fit)
    best, iters = None, 0
    for _ in range(reps):
        sop = orc.Sop(snap, groups)
        t0 = time.perf_counter()
        sop.batch(pods, stages, bitmap=bool(stages & bsa.soa.STAGE_FILTER))
        dt = time.perf_counter() - t0
        iters = sop.iters
        best = dt if best is None else min(best, dt)
    logical = pods.p * nodes.n
    ncores = os.cpu_count() or 1
    allc = None
    try:
        # all host cores: whole groups dealt over one thread per core inside the C oracle (orc_batch_threads), best of 3
        bdist = importlib.import_module("batch-scheduler_amd.dist")
        own = bdist.owner_ranks(pods.group, groups.g, ncores)
        subsets = [pods.take(np.nonzero(own == r)[0]) for r in range(ncores)]
        wall = min(orc.batch_threads(snap, groups, subsets, stages, bitmap=bool(stages & bsa.soa.STAGE_FILTER))[0] for _ in range(3))
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:                                # pragma: no cover
            usable = ncores
        allc = {"value": logical / wall, "cores": ncores, "cores_in_affinity_mask": usable, "seconds_per_batch": wall,
                "speedup_over_one_core": best / wall,
                "how": "one pthread per host core (os.cpu_count()), whole groups per thread (exact in this scenario: no pod's decision depends on a pod "
                       "of another group); speedup_over_one_core says how many cores the box really gave the run (a container can report 256 and "
                       "schedule on a handful)"}
    except Exception as e:                                    # pragma: no cover
        allc = {"error": repr(e)}
    return {"value": logical / best, "unit": "pod x node fit evals/s", "cores": 1, "kind": "port", "faithful_cost": False,
            "sample": f"{reps} x one full batch ({pods.p} pods x {nodes.n} nodes, same seeded inputs, same stages), best of {reps}; "
                      f"reference node-loop iterations executed (core.go:604) = {iters} of {logical} logical; oracle's faithful_cost switch off "
                      f"(the reference's unconditional computeClusterResource scan at core.go:152 is NOT charged to the baseline)",
            "seconds_per_batch": best, "reference_loop_iterations": iters, "all_cores": allc}


# ------------------------------------------------------------------------------------------------ rocprofv3 passes of this very command
def profile_passes(args):
    """This command re-run (few steps, nothing else) under rocprofv3, three separate passes:
      --kernel-trace --stats          kernel-only durations per launch (what the roofline is priced with): median over the steady
                                      two thirds of the launches
      --pmc FETCH_SIZE / WRITE_SIZE   HBM bytes per launch (FETCH doubled per the gfx950 note of MI355X_MICROARCH.md)
    Returns (dict launch -> {avg_us, calls, hbm_bytes_per_launch, ...} or None, source string)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    tmp = tempfile.mkdtemp(prefix="bs_prof_", dir="/tmp")
    inner = [sys.executable, os.path.abspath(__file__), "--inner-pmc", "--config", args.config, "--scenario", args.scenario, "--seed", str(args.seed),
             "--stages", args.stages]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        d = os.path.join(tmp, "trace")
        res = subprocess.run([exe, "--kernel-trace", "--stats", "-d", d, "-o", "trace", "--", *inner], cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if res.returncode != 0 or not dbs:
            return None, f"rocprofv3 --kernel-trace failed (rc {res.returncode})"
        con = sqlite3.connect(dbs[0])
        rows = con.execute("select name, dispatch_id, duration, grid_x from kernels order by dispatch_id").fetchall()
        con.close()
        per = {}
        for name, _disp, dur, grid in rows:
            for key, kn in LAUNCH_KERNELS.items():
                if kn + "<" in name or kn + "(" in name:
                    per.setdefault(key, []).append((float(dur) / 1e3, int(grid)))
        for key, lst in per.items():
            steady = lst[len(lst) // 3:]                         # the first launches after a load size their grids on estimates (K not back yet)
            out.setdefault(key, {}).update(kernel_us=float(np.median([d for d, _ in steady])), calls=len(steady), grid_threads=int(np.median([g for _, g in steady])))
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            res = subprocess.run([exe, "--pmc", counter, "-d", d, "-o", "pmc", "--", *inner], cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if res.returncode != 0 or not dbs:
                return (out or None), f"kernel-trace ok; rocprofv3 --pmc {counter} failed (rc {res.returncode})"
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=? order by dispatch_id", (counter,)).fetchall()
            con.close()
            per = {}
            for name, _disp, v in rows:
                for key, kn in LAUNCH_KERNELS.items():
                    if kn + "<" in name or kn + "(" in name:
                        per.setdefault(key, []).append(float(v))
            for key, lst in per.items():
                out.setdefault(key, {})[counter] = float(np.median(lst[len(lst) // 3:]))
    except Exception as e:                                    # pragma: no cover
        return (out or None), f"profile pass failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for v in out.values():
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = int(v.get("FETCH_SIZE", 0.0) * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024)
    return (out or None), ("measured by this run: rocprofv3 --kernel-trace --stats (kernel-only durations) and --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                           "(separate passes, FETCH x2 per the gfx950 note)")


# ------------------------------------------------------------------------------------------------ helpers
def resident_ms(bsa, nodes, fit, groups, pods, stages, steps, warmup=10):
    """ms per step of a batch resident in HBM on a fresh single-GPU context + its work counters."""
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        for _ in range(warmup):
            ctx.run(stages)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.run(stages)
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        st = ctx.stats(stages)
    return ms, st


def launch_times(bsa, nodes, fit, groups, pods, stages, steps, device=0):
    """mean device time (us) of each launch group of a resident step, from the library's own hipEvents (enable_timing=1)"""
    with bsa.Context(scalar_lanes=nodes.lanes - 4, enable_timing=1, device=device) as ctx:
        ctx.load_nodes(nodes, fit)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        for _ in range(10):
            ctx.run(stages)
        ctx.sync()
        ctx.timing_reset()
        for _ in range(steps):
            ctx.run(stages)
        ctx.sync()
        return {k: v[0] * 1000 / v[1] for k, v in ctx.timing().items() if v[1] > 0}


def pct(xs, q):
    return float(np.percentile(xs, q)) if len(xs) else None


def drain_section(bsa, nodes, fit, groups, pods, args):
    """SURVEY 8(d)(2), both sides on the same inputs.  The queue in Compare order (core.go:368-411 keeps a gang's pods together).
      GPU   the batched scheduling cycle until nothing is ready (host/bs_drain.cpp): score the whole queue, release the first
            ready gang, assume its pods (first fit), patch nodes / groups / queue ON THE DEVICE, score again.  Per released gang: the
            duration of the cycle that decided it, and the time since the drain began.
      CPU   the reference's pod-by-pod pass (oracle/bs_oracle_seq.c, one core): per released gang the time from its first pod
            entering PreFilter to the quorum of core.go:303 turning true.
      1:1   the sequential drop-in mode (host mirror, every node loop one bs_cluster_fits round trip) on a bounded sample."""
    soa = bsa.soa
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY          # the shipped configuration: Filter not enabled
    q = pods.take(np.argsort(pods.group, kind="stable"))
    res = {"queue": "Compare order (gang by gang); stages: prefilter + tally (the shipped plugin configuration leaves Filter off)"}
    best = None
    for rep in range(2):                                # the second run is the warm one
        n2, g2 = nodes.copy(), groups.copy()
        with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
            ctx.load_nodes(nodes, fit)
            ctx.load_groups(groups)
            ctx.load_pods(q)
            ctx.run(st)
            ctx.sync()
            r = bsa.plugin.drain(ctx, n2, fit, g2, q, st | soa.BATCH_HOST_RESULTS)
            applies, rederives = ctx.apply_stats()
        best = r
    cyc = best["cycle_ns"] / 1e6
    res["gpu"] = {"gangs_released": best["n_admitted"], "pods_released": int((best["pod_node"] >= 0).sum()), "cycles": best["n_cycles"], "stuck_gangs": best["n_stuck"],
                  "total_ms": best["total_ns"] / 1e6, "gangs_per_s": best["n_admitted"] / max(best["total_ns"] * 1e-9, 1e-12),
                  "gang_admit_latency_ms_p50": pct(cyc, 50), "gang_admit_latency_ms_p95": pct(cyc, 95),
                  "time_since_drain_start_ms_p50": pct(best["admitted_ns"] / 1e6, 50), "time_since_drain_start_ms_p95": pct(best["admitted_ns"] / 1e6, 95),
                  "pods_apply_calls": applies, "pods_rederives": rederives,
                  "latency_definition": "duration of the scheduling cycle that released the gang: bs_batch_run over the whole queue + results + node choice + "
                                        "bs_nodes_assume / bs_groups_apply / bs_pods_apply"}
    orc = _oracle()
    s = orc.seq_replay(nodes, fit, groups, q, st)
    lat = (s["ready_ns"] - s["first_ns"]) / 1e6
    res["_cpu_sequential_pass"] = {"kind": "port (C restatement of the Go path, 1 core)", "gangs_released": s["n_released"],
                                   "pods_released": int((s["pod_node"] >= 0).sum()), "total_ms": s["total_ns"] / 1e6,
                                   "of_which_node_choice_ms": s["pick_ns"] / 1e6,
                                   "node_choice_note": "the first-fit node pick is UPSTREAM's work, not the plugin's: the plugin-only share of the CPU pass is total_ms - of_which_node_choice_ms "
                                                       "(bs_seq_run does the pick as well; the comparison is whole pass against whole pass)",
                                   "gangs_per_s": s["n_released"] / max(s["total_ns"] * 1e-9, 1e-12), "reference_loop_iterations": s["iters"],
                                   "gang_admit_latency_ms_p50": pct(lat, 50), "gang_admit_latency_ms_p95": pct(lat, 95),
                                   "time_since_pass_start_ms_p50": pct(s["ready_ns"] / 1e6, 50),
                                   "latency_definition": "per gang: first pod entering PreFilter (core.go:88) -> quorum of core.go:303 true"}
    res["gang_granular_drain_same_gangs_as_cpu_pass"] = sorted(best["admitted_group"].tolist()) == sorted(s["released_group"].tolist())
    res["relation"] = ("gang-granular drain (a pre-screen loop over frozen snapshots) vs pod-by-pod pass: equal on complete cold queues (tests/test_drain.py); "
                       "otherwise the pass lets partial gangs hold what they assumed and re-checks reservations pod by pod, so the two sets can differ.  "
                       "sequential_on_device IS the pod-by-pod pass (bs_seq_run): same gangs, same nodes, same codes as the CPU port's")
    # ---- the reference's own order of events on the device: bs_seq_run (PreFilter -> node choice -> assume -> Permit -> release, pod by pod)
    runs = []
    for rep in range(3):
        with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
            ctx.load_nodes(nodes, fit)
            ctx.load_groups(groups)
            ctx.load_pods(q)
            ctx.sync()
            t = time.perf_counter()
            r = ctx.seq_run(soa.STAGE_PREFILTER)
            runs.append((time.perf_counter() - t, r))
    wall, r = min(runs, key=lambda x: x[0])
    dlat = (r["ready_ns"] - r["first_ns"]) / 1e6
    identical = (r["released_group"].tolist() == s["released_group"].tolist() and r["released_pods"].tolist() == s["released_pods"].tolist()
                 and np.array_equal(r["pod_node"], s["pod_node"]) and np.array_equal(r["pf_code"], s["pf_code"]) and np.array_equal(r["pf_first_k"], s["pf_first_k"]))
    p50_dev, p50_cpu = pct(dlat, 50), pct(lat, 50)
    res["sequential_on_device"] = {
        "what": "bs_seq_run: ONE persistent workgroup walks the resident queue pod by pod with the reference's semantics (core.go:88-167, :268-309, "
                "batchscheduler.go:254-344): PreFilter against the CURRENT node requests and group counters, first-fit node choice, assume, Permit, release at the quorum",
        "same_gangs_as_cpu_pass": r["released_group"].tolist() == s["released_group"].tolist(),
        "bit_identical_to_cpu_pass": bool(identical),
        "gangs_released": r["n_released"], "pods_released": int((r["pod_node"] >= 0).sum()),
        "total_ms_host_observed": wall * 1e3, "total_ms_device": r["total_ns"] / 1e6, "us_per_pod": r["total_ns"] / 1e3 / max(q.p, 1),
        "gangs_per_s": r["n_released"] / max(wall, 1e-12),
        "gang_admit_latency_ms_p50": p50_dev, "gang_admit_latency_ms_p95": pct(dlat, 95),
        "latency_definition": "per gang, device clock: first pod entering PreFilter (core.go:88) -> quorum of core.go:303 true (the same definition as the CPU pass)",
        "speedup_vs_cpu_port": {"gang_admit_latency_p50": p50_cpu / p50_dev if p50_dev else None, "whole_pass": s["total_ns"] / 1e9 / max(wall, 1e-12)},
        "work": {"node_scans": r["node_scans"], "scan_rounds_of_1024_nodes": r["scan_rounds"], "first_fit_searches": r["node_picks"],
                 "first_fit_rounds_of_16_tiles": r["pick_rounds"], "findMaxPG_folds": r["leader_folds"], "table_summary_builds": r["table_builds"],
                 "evals_executed": r["scan_rounds"] * 1024, "evals_per_s": r["scan_rounds"] * 1024 / max(r["total_ns"] * 1e-9, 1e-12),
                 "cpu_port_reference_loop_iterations": s["iters"]},
    }
    res["same_gangs_as_cpu_pass"] = res["sequential_on_device"]["same_gangs_as_cpu_pass"]
    res["one_to_one_mode"] = one_to_one(bsa, nodes, fit, groups, q, 300)
    return res


def one_to_one(bsa, nodes, fit, groups, pods, n):
    """the 1:1 drop-in (bs_cluster_fits per PreFilter through the C++ host mirror), re-measured on the first n pods"""
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = bsa.plugin.ScheduleOperation(ctx)
        for g in range(groups.g):
            has_mr = bool(groups.flags[g] & bsa.soa.GROUP_HAS_MINRES)
            sop.add_group(int(groups.min_member[g]), int(groups.status_scheduled[g]), creation_ts=g, name_rank=g,
                          min_resources=groups.min_resources[:, g].tolist() if has_mr else None, min_resources_present=int(groups.min_resources_present[g]))
        n = min(n, pods.p)
        lat = []
        for i in range(n):
            a = time.perf_counter()
            sop.PreFilter(i + 1, i + 1, int(pods.group[i]), pods.req[:, i].tolist(), int(pods.req_present[i]), int(pods.cls[i]), int(pods.owner[i]))
            lat.append((time.perf_counter() - a) * 1e3)
        calls = sop.gpu_calls
        sop.close()
    gang = int(np.median(np.bincount(pods.group[pods.group >= 0]))) if (pods.group >= 0).any() else 1
    return {"sample_pods": n, "prefilter_latency_ms_p50": pct(lat, 50), "prefilter_latency_ms_p95": pct(lat, 95), "gpu_round_trips": calls,
            "gang_admit_latency_ms_estimate": pct(lat, 50) * gang, "evals_per_s": nodes.n / (pct(lat, 50) * 1e-3),
            "note": "every PreFilter = findMaxPG + one node scan as synchronous GPU round trips; a gang waits for all its pods"}


def make_pod_deltas(bsa, pods, n_cycles, churn, seed=2):
    """Prebuilt bs_pods_delta structs (the marshalling of ~100 pod records is the caller's and is not timed): every cycle
    removes churn/2 random pods and appends as many new ones (clones of random queue members: same gangs, same templates),
    so the queue length stays put and every cycle runs on the same amount of work."""
    soa = bsa.soa
    rng = np.random.default_rng(seed)
    half = max(1, churn // 2)
    keep, structs = [], []
    for _ in range(n_cycles):
        rem = np.sort(rng.choice(pods.p, half, replace=False)).astype(np.uint32)
        ins = pods.take(rng.integers(0, pods.p, half))
        d = soa.PodsDeltaStruct()
        d.n_remove, d.remove = half, rem.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        d.n_flags = 0
        d.insert = ins.as_struct()
        keep.append((rem, ins))
        structs.append(d)
    return structs, keep


def resident_cycle(bsa, ctx, groups, pods, nodes, stages, darr, ndeltas, out, iters, zero_copy=True):
    soa = bsa.soa
    view = soa.BatchViewStruct()
    churn = max(2, pods.p // 100)
    structs, keep = make_pod_deltas(bsa, pods, iters + 5, churn)
    ctx.load_pods(pods)
    parts = {"groups_apply": [], "pods_apply": [], "run": [], "read": [], "total": []}
    for it in range(iters + 5):
        t0 = time.perf_counter()
        ctx.apply_group_deltas_raw(darr, ndeltas)
        t1 = time.perf_counter()
        ctx.apply_pods_raw(structs[it])
        t2 = time.perf_counter()
        ctx.run(stages | soa.BATCH_HOST_RESULTS)
        t3 = time.perf_counter()
        if zero_copy:
            ctx.map_raw(view)                       # bs_batch_map: completion word, then pointers into the pinned results
        else:
            ctx.read(out=out)                       # bs_batch_read: the same wait + a host-side copy of every array
        t4 = time.perf_counter()
        if it >= 5:
            for k, v in zip(("groups_apply", "pods_apply", "run", "read", "total"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
                parts[k].append(v * 1e3)
    applies, rederives = ctx.apply_stats()
    launched, missed = ctx.speculation_stats()
    r = {k: {"p50_ms": pct(v, 50), "p95_ms": pct(v, 95)} for k, v in parts.items()}
    r["speculation"] = {"batches_launched_on_a_guess_so_far": launched, "wrong_guesses_re_run": missed,
                        "note": "the chain is launched on the previous cycle's findMaxPG answer while this cycle's is still being computed on the device; "
                                "checked when the results are asked for (bs_speculation_stats)"}
    r["queue_churn_per_cycle"] = {"removed": churn // 2, "appended": churn // 2, "of": pods.p}
    r["applies"], r["rederives"] = applies, rederives
    return r


def host_cycle(bsa, ctx, groups, pods, nodes, stages, iters=60):
    """One scheduling cycle as the Go shim would drive it, host-observed: patch the groups that changed (Permit /
    PostBind counters), hand over the pending pods, run the batch, read decisions + Filter rows back.
      plain    bs_groups_apply, bs_pods_load (the library packs the caller's arrays into its pinned buffer), bs_batch_run,
               bs_batch_read (two device-to-host copies under one stream wait)
      latency  the same cycle with the queue marshalled in place (bs_pods_map) and BS_BATCH_HOST_RESULTS (the last launch
               writes the results into pinned host memory; bs_batch_read polls a completion word)"""
    soa = bsa.soa
    rows_cap = max(ctx.filter_rows_count(), 1)
    out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=rows_cap)
    rng = np.random.default_rng(1)
    idx = rng.choice(groups.g, min(32, groups.g), replace=False)
    deltas = [(int(i), int(groups.matched[i]), int(groups.status_scheduled[i]), int(groups.flags[i])) for i in idx]   # same values: decisions stay put
    darr = (soa.GroupDelta * len(deltas))(*[soa.GroupDelta(*d) for d in deltas])
    names = ("group", "req", "req_present", "cls", "owner", "flags")
    res = {}
    for mode in ("plain", "latency"):
        parts = {"groups_apply": [], "pods_load": [], "run": [], "read": [], "total": []}
        flag = soa.BATCH_HOST_RESULTS if mode == "latency" else 0
        view = None
        for it in range(iters + 5):
            if mode == "latency":                       # marshalling: the caller writes the queue where the upload reads it
                view = ctx.map_pods(pods.p)
                for n in names:
                    getattr(view, n)[...] = getattr(pods, n)
            t0 = time.perf_counter()
            ctx.apply_group_deltas_raw(darr, len(deltas))
            t1 = time.perf_counter()
            ctx.load_pods(view if mode == "latency" else pods)
            t2 = time.perf_counter()
            ctx.run(stages | flag)
            t3 = time.perf_counter()
            ctx.read(out=out)
            t4 = time.perf_counter()
            if it >= 5:
                for k, v in zip(("groups_apply", "pods_load", "run", "read", "total"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
                    parts[k].append(v * 1e3)
        res[mode] = {k: {"p50_ms": pct(v, 50), "p95_ms": pct(v, 95)} for k, v in parts.items()}
    # ---- the queue-resident cycle: the pending queue is NOT re-uploaded; bs_pods_apply patches 1 % of it per cycle on the
    # device (half leave = a released gang's worth of pods, as many arrive), results in latency mode
    res["resident"] = resident_cycle(bsa, ctx, groups, pods, nodes, stages, darr, len(deltas), out, iters, zero_copy=True)
    res["resident_copy_out"] = resident_cycle(bsa, ctx, groups, pods, nodes, stages, darr, len(deltas), out, iters, zero_copy=False)
    ctx.load_pods(pods)
    full = []
    for it in range(iters // 2 + 5):
        t0 = time.perf_counter()
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        ctx.run(stages)
        ctx.read(out=out)
        if it >= 5:
            full.append((time.perf_counter() - t0) * 1e3)
    res["plain_with_full_groups_load"] = {"total": {"p50_ms": pct(full, 50), "p95_ms": pct(full, 95)}}
    return res, int(out.fl_rows_n[0]), out


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus:
        if world == 1 and n_gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        n_gpus = world

    import torch
    bsa = importlib.import_module("batch-scheduler_amd")
    soa, synth = bsa.soa, bsa.synth
    dist = None
    if world > 1 or bool(int(os.environ.get("BS_FORCE_DIST", "0"))):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    base = synth.CONFIGS[args.config]
    mult = world if args.scaling == "weak" else 1
    nodes, fit, groups, pods, meta = synth.make(args.config, args.scenario, seed=args.seed,
                                                pods=base["pods"] * mult, groups=base["groups"] * mult)
    stages = soa.STAGE_ALL if args.stages == "all" else (soa.STAGE_PREFILTER | soa.STAGE_TALLY)
    L = nodes.lanes

    # The timed region runs WITHOUT the library's per-launch hipEvents (they cost ~1.7 us of a 21 us step: measurement overhead, not the path);
    # the per-launch event times come from a short run of their own on a second context (launch_times) behind the timed region.  BS_TIMING=1
    # restores round 4's behaviour (events inside the timed region).
    ctx = bsa.Context(scalar_lanes=L - 4, device=local_rank, enable_timing=0 if args.inner_pmc else int(os.environ.get("BS_TIMING", "0")))
    ctx.load_nodes(nodes, fit)
    ctx.load_groups(groups)
    all_pods = pods
    partitioned = False
    if dist is not None and bool((groups.flags & soa.GROUP_HAS_POD).all()):
        # Steady state (every group has its pod): no pod's decision depends on a pod of another group, so each
        # rank loads only the pods it owns (whole groups, same ownership rule as the device) and the group
        # state is replicated.  Otherwise: whole batch on every rank + device-side ownership (bs_shard_set).
        bdist = importlib.import_module("batch-scheduler_amd.dist")
        own = bdist.owner_ranks(pods.group, groups.g, world)
        reach_hint = bdist.first_reach_thresholds(pods, groups, own, world)[rank]
        pods = pods.take(np.nonzero(own == rank)[0])
        partitioned = True
    ctx.load_pods(pods)
    if partitioned:
        ctx.first_reach_hint(reach_hint)       # where the whole queue's first pod that reaches findMaxPG stands in THIS rank's queue
    admit_t = None
    lib_stream = None
    if dist is not None:
        if partitioned:
            ctx.reduce_external(True)
        else:
            ctx.set_shard(rank, world)
        admit_t = torch.zeros(groups.g, dtype=torch.int32, device=f"cuda:{local_rank}")
        ctx.bind_admit(admit_t.data_ptr())
        # run the collective stream-ordered against the library's HIP stream: no host synchronisation per step
        lib_stream = torch.cuda.ExternalStream(ctx.stream(), device=f"cuda:{local_rank}")

    force_dist = bool(int(os.environ.get("BS_FORCE_DIST", "0")))   # exercise the collective path at world size 1
    rank_report = None
    if dist is not None:
        # self-check of the first multi-GPU run: what RCCL sees and how the pod axis was dealt
        owned = torch.tensor([pods.p if partitioned else int((importlib.import_module("batch-scheduler_amd.dist").owner_ranks(all_pods.group, groups.g, world) == rank).sum())],
                             dtype=torch.int64, device=f"cuda:{local_rank}")
        gathered = [torch.zeros_like(owned) for _ in range(world)]
        dist.all_gather(gathered, owned)
        per_rank = [int(t.item()) for t in gathered]
        rank_report = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(), "owned_pods_per_rank": per_rank,
                       "max_over_mean": max(per_rank) / max(1e-9, sum(per_rank) / len(per_rank)), "mode": "partitioned" if partitioned else "replicated"}

    def step():
        ctx.run(stages)
        if world > 1 or force_dist:
            with torch.cuda.stream(lib_stream):
                dist.all_reduce(admit_t)       # ONE RCCL all-reduce (sum) of per-group admit counts, ordered after the tally
            ctx.finish()                       # quorum bits from the reduced counters (same stream, after the collective)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if args.inner_pmc:                         # the PMC passes: a few dozen plain steps, nothing else
        for _ in range(45):
            step()
        fence()
        ctx.close()
        return None

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_reset()
    # EXACTLY K steps between a barrier + synchronize on both sides, MAX over ranks — TIMED_REGIONS times back to back, and the line reports the
    # MEDIAN region (all of them are in `timed_regions_ms`): at K = 20 one region is 0.45 ms and the first one carries the clocks' ramp-up
    # (22.4 vs 20.9 us per step between 20 and 200 steps in round 5).  `steps` stays what was asked for.
    regions = []
    for _ in range(TIMED_REGIONS):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        regions.append(el)
    elapsed = float(np.median(regions))
    timing = ctx.timing()
    if rank == 0 and not any(v[1] for v in timing.values()):
        # (single context on this rank's device; in a sharded run rank 0's own part of the queue)
        timing = {k: (us / 1e3, 1) for k, us in launch_times(bsa, nodes, fit, groups, pods, stages, 40, device=local_rank).items()}

    # executed-work counters from one extra, untimed, instrumented batch
    ctx.stats_arm()
    step()
    ctx.sync()
    stats = ctx.stats_read()
    out = ctx.read(bitmap=False, rows=False)

    logical = all_pods.p * nodes.n                    # whole job: every pending pod (all ranks) against every node
    ms_per_step = elapsed / args.steps * 1e3
    value = logical * args.steps / elapsed

    result = None
    if rank == 0:
        single = world == 1 and not force_dist
        # ---------------- roofline: every launch of the step, priced on its ALGORITHMIC (compulsory) bytes with KERNEL-ONLY times
        # PreFilter-path eval reads 16 L + 2 bytes, Filter-path eval 16*4 + 1 in + 1/8 out — per pod x node of the
        # reference's loops.  The batch evaluates each distinct request once and keeps tables / node lanes on chip, so what a
        # launch HAS to move is the compulsory traffic of 8(d): every input once, every output once:
        #   A  pods P (8 L + 21) + nodes N (16 L + 6) in, table rows M 8 LP + per-pod scratch P 10 out
        #   B  table rows M 8 LP + getLeftResource lanes N 33 + slots in, first rows + Filter rows (rows x ceil(N/64) x 8) out
        #   C  per-pod scratch P 14 + group counters G 16 in, decisions P 18 + admit / ready G 5 out
        # frac = those bytes / kernel time / 8 TB/s: a figure that cannot exceed 1 and that says what it is — the step is two
        # latency-bound launches over a working set that lives in L2 / MALL, nowhere near a bandwidth roofline.
        LP = 4 if L == 4 else (8 if L <= 8 else 16)
        W = (nodes.n + 63) // 64
        rows = max(1, stats["filter_distinct"])
        # which form did the timed steps take?  (the one-launch form has no "scan" group: its first launch is timed as "query", k_fast_final as "resolve")
        no_scan = bool(stats["fast_path"]) and not (timing.get("scan", (0.0, 0))[1]) and bool(timing.get("query", (0.0, 0))[1])
        one_launch = no_scan and bool(timing.get("resolve", (0.0, 0))[1])
        whole_step = no_scan and not one_launch                 # no k_fast_final either: the pod blocks of k_fast_step_a finish their own pods
        LAUNCH_KERNELS.clear()
        LAUNCH_KERNELS.update(LAUNCH_KERNELS_WHOLE if whole_step else LAUNCH_KERNELS_ONE if one_launch else LAUNCH_KERNELS_TWO)
        if whole_step:
            # everything once: pods + nodes + the class directory in; decisions, first rows, Filter rows, admit / ready out (neither table rows nor per-pod
            # scratch have to leave the chip: the block that derives a pod's scratch finishes the pod)
            alg = {"query": pods.p * (8 * L + 21) + nodes.n * (16 * L + 6) + nodes.n * 33 + stats["scan_queries"] * (8 * L + 4) + stats["scan_queries"] * 8
                            + rows * (W * 8 + 4) + pods.p * 18 + groups.g * 21}
        elif one_launch:
            # A' = pods + nodes + the class directory in; per-pod scratch, first rows, Filter rows out — the table rows are never written (the block that
            # builds a chunk scans it from its registers);  C = per-pod scratch + group counters in, decisions + admit / ready out
            alg = {
                "query": pods.p * (8 * L + 21) + nodes.n * (16 * L + 6) + nodes.n * 33 + stats["scan_queries"] * (8 * L + 4) + pods.p * 10
                         + stats["scan_queries"] * 4 + rows * (W * 8 + 4),
                "resolve": pods.p * (14 + 18) + groups.g * 21 + rows * 4,
            }
        else:
            alg = {
                "query": pods.p * (8 * L + 21) + nodes.n * (16 * L + 6) + nodes.n * 8 * LP + pods.p * 10,
                # launches B and C share ONE launch (k_fast_scan_filter_final): scan / Filter blocks + final blocks
                "scan": nodes.n * 8 * LP + nodes.n * 33 + stats["scan_queries"] * (8 * LP + 12) + rows * (64 + W * 8 + 4) + pods.p * (14 + 18) + groups.g * 21,
            }
        prof, prof_src = (None, "not collected")
        if single and not args.no_pmc and stats["fast_path"]:
            prof, prof_src = profile_passes(args)
        launches = []
        for key in (("query",) if whole_step else ("query", "resolve") if one_launch else ("query", "scan")):
            ms, n = timing.get(key, (0.0, 0))
            pk = (prof or {}).get(key, {})
            kernel_us = pk.get("kernel_us")
            event_us = ms / n * 1e3 if n else None
            t_us = kernel_us if kernel_us else event_us
            if not t_us:
                continue
            tb = pk.get("hbm_bytes_per_launch")
            e = {"kernel": LAUNCH_KERNELS[key] if stats["fast_path"] else key, "avg_launch_us": t_us,
                 "time_source": "rocprofv3 --kernel-trace (kernel only)" if kernel_us else "hipEvents on the library stream (spans the dispatch gap)",
                 "hip_event_us": event_us, "algorithmic_bytes_per_launch": alg[key], "achieved": alg[key] / (t_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": alg[key] / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": tb,
                 "physical_frac": (tb / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tb else None}
            if key == ("query" if (one_launch or whole_step) else "scan"):
                e["evals_executed_per_launch"] = stats["scan_evals_executed"] + stats["filter_evals_executed"]
            launches.append(e)
        roofline = None
        if launches:
            dom = max(launches, key=lambda x: x["avg_launch_us"])
            roofline = dict(dom)
            ksum = sum(x["avg_launch_us"] for x in launches)
            bpe = (16 * L + 2) + ((16 * 4 + 1 + 0.125) if args.stages == "all" else 0.0)       # bytes per logical pod x node eval
            ev_exec = stats["scan_evals_executed"] + stats["filter_evals_executed"]
            roofline.update({"bound": "latency", "nominal_bound": "hbm",
                             "frac_per_eval_logical": value * bpe / (HBM_PEAK_GBS * 1e9),
                             "frac_per_eval_executed": (stats["scan_evals_executed"] * (16 * L + 2) + stats["filter_evals_executed"] * 65.125) / (dom["avg_launch_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9),
                             "bytes_per_eval": bpe,
                             "frac_note": "SURVEY 8(d)'s formula on the LOGICAL rate: evals/s x (PreFilter + Filter bytes per eval) / 8 TB/s (> 1: the step does not do "
                                          "per-eval work - request classes and pruning, see work_avoided); on the EXECUTED work: (scan evals x (16 L + 2) + Filter evals x "
                                          "65.125 bytes) / the dominant kernel's time / 8 TB/s; `frac` above is compulsory bytes / kernel time / 8 TB/s",
                             "limiter": "dependent-load chains and in-launch hand-overs (one small launch, three dependency levels; see frac)" if whole_step else
                                        "launch latency and dependent-load chains (the step is two small launches, three dependency levels; see frac)",
                             "step_form": "the whole step in ONE launch (k_fast_step_a<S, true>: pod blocks | class-slot block | table blocks | Filter blocks; the pod blocks "
                                          "finish their own pods after an in-launch hand-over)" if whole_step
                                          else "one-launch form of launch A + scan / Filter roles (k_fast_step_a, class slots from the class directory), then k_fast_final" if one_launch
                                          else "launch A (k_fast_query_tables), then scan / Filter / final blocks in one launch (k_fast_scan_filter_final)",
                             "source": prof_src, "sum_of_launch_us": ksum, "ms_per_step_us": ms_per_step * 1e3,
                             "note": "the step's longest launch (by kernel-only time).  achieved = compulsory algorithmic bytes of the launch (every input once, "
                                     "every output once; DESIGN.md section 7) / its mean kernel duration; traffic = HBM bytes per launch from PMC; "
                                     "physical_frac = traffic / time / 8 TB/s.  HBM is the nominal roofline of this scan / compare work (SURVEY 8(d)) and no "
                                     "launch is anywhere near it: the working set lives in L2 / MALL and every launch is a chain of dependent loads.  "
                                     "The utilisation figure for real per-pair work is roofline_throughput (every pod its own request, VALU-issue bound)."})
        work_avoided = {"logical_evals_per_step": logical,
                        "prefilter_evals_executed": stats["scan_evals_executed"], "filter_evals_executed": stats["filter_evals_executed"],
                        "scan_queries": stats["scan_queries_logical"], "scan_queries_distinct": stats["scan_queries"],
                        "filter_distinct_requests": stats["filter_distinct"],
                        "executed_fraction_of_logical": (stats["scan_evals_executed"] + stats["filter_evals_executed"]) / max(1, 2 * logical),
                        "how": "pods with equal derived requests share one evaluated row (request classes), 64-row table groups whose largest running sum "
                               "cannot reach the smallest request are skipped, the expanded pods x nodes bitmap is not materialised"}

        cycle, extras, cpu, drain, roofline_tp = None, None, None, None, None
        if single and not args.no_extras:
            cyc, nrows, _ = host_cycle(bsa, ctx, groups, pods, nodes, stages)
            p50, p95 = cyc["resident"]["total"]["p50_ms"], cyc["resident"]["total"]["p95_ms"]
            cycle = {"definition": "host-observed scheduling cycle, the way a shim drives it.  'resident' (the headline): the pending queue STAYS on the device — "
                                   "bs_groups_apply (32 groups) + bs_pods_apply (1 % of the queue leaves, as many pods arrive; the delta is read from pinned memory) "
                                   "+ bs_batch_run in latency mode (results written to pinned host memory by the last launch) + bs_batch_map (completion word "
                                   "polled, results read in place: no copy of any kind).  'resident_copy_out': the same with bs_batch_read copying every "
                                   "array out of the pinned memory.  'latency': the whole queue re-marshalled into the pinned upload buffer every cycle (bs_pods_map + bs_pods_load) + "
                                   "latency-mode results.  'plain': bs_pods_load packs the caller's arrays, results copied back under one stream wait.",
                     "modes": cyc,
                     "gang_admit_latency_ms_p50": p50, "gang_admit_latency_ms_p95": p95,
                     "gang_admit_latency_reload_ms_p50": cyc["latency"]["total"]["p50_ms"], "gang_admit_latency_plain_ms_p50": cyc["plain"]["total"]["p50_ms"],
                     "evals_per_s_at_p50": logical / (p50 * 1e-3), "filter_rows": nrows,
                     "filter_result_bytes": nrows * ((nodes.n + 63) // 64) * 8 + pods.p * 4,
                     "expanded_bitmap_bytes_avoided": pods.p * ((nodes.n + 63) // 64) * 8}
            extras = {}
            for sc in ("cold", "warm", "busy"):
                n2, f2, g2, p2, _ = synth.make(args.config, sc, seed=args.seed)
                ms, st = resident_ms(bsa, n2, f2, g2, p2, stages, 100)
                extras[sc] = {"ms_per_step": ms, "evals_per_s": p2.p * n2.n / (ms * 1e-3), "fast_path": st["fast_path"], "chain": st["chain"],
                              "launches": st["launches"], "class_mode": st["class_mode"]}
            # ---- the throughput regime: every pod its own request, on k = 1, 2 and 4 of the four lanes Filter compares (synth.all_distinct).  One
            # utilisation figure per launch B: executed pod x node evals / kernel time against the VALU-issue bound of the lean Filter loop for the
            # k the launch REALLY compared (device counters: filter_lane_blocks / filter_tile_blocks), CYCLES_PER_NODE_LANE per node, lane and 64 slots.
            def tp_entry(cfg_name, n_, f_, g_, p_, k_lanes, steps, lsteps):
                pk = synth.all_distinct(p_, n_, k_lanes)
                ms_k, st_k = resident_ms(bsa, n_, f_, g_, pk, stages, steps, warmup=40)   # (the first launches after a load size their grids on an estimate of the class count)
                b_us = launch_times(bsa, n_, f_, g_, pk, stages, lsteps).get("scan")
                ev = st_k["scan_evals_executed"] + st_k["filter_evals_executed"]
                counted = st_k["filter_tile_blocks"] > 0         # (below ~1024 distinct requests launch B runs the latency-regime item, which carries no counter)
                k_meas = st_k["filter_lane_blocks"] / st_k["filter_tile_blocks"] if counted else float(k_lanes)
                peak = SIMDS * CLOCK_HZ / (CYCLES_PER_NODE_LANE * max(k_meas, 1e-9)) * 64
                rows_b = st_k["filter_distinct"] * ((n_.n + 63) // 64) * 8
                return {"workload": f"{cfg_name}/{args.scenario}, every pod its own request on {k_lanes} of the 4 lanes Filter compares "
                                    f"({st_k['filter_distinct']} distinct Filter requests, {st_k['scan_queries']} distinct scan queries)",
                        "k_lanes_perturbed": k_lanes, "k_compared_lanes": k_meas, "k_source": "device counters of an instrumented step (bs_batch_stats: filter_lane_blocks / filter_tile_blocks)" if counted else
                                    "nominal: the batch is below the throughput regime (<= 16 tiles of class slots), launch B ran the latency-regime Filter item",
                        "kernel_us": b_us, "whole_step_ms": ms_k, "evals_executed_per_launch": ev,
                        "achieved_evals_per_s": (ev / (b_us * 1e-6)) if b_us else None, "peak_evals_per_s": peak,
                        "frac": (ev / (b_us * 1e-6) / peak) if b_us else None,
                        "output_bytes_per_launch": rows_b, "output_GBps": (rows_b / (b_us * 1e-6) / 1e9) if b_us else None}
            tp_here = {f"k{k}": tp_entry(args.config, nodes, fit, groups, all_pods, k, 60, 40) for k in (1, 2, 4)}
            e1 = tp_here["k1"]
            extras["all_distinct_requests"] = {"ms_per_step": e1["whole_step_ms"], "evals_per_s": logical / (e1["whole_step_ms"] * 1e-3),
                                               "evals_executed_per_step": e1["evals_executed_per_launch"], "k_compared_lanes": e1["k_compared_lanes"]}
            extras["all_distinct_k4"] = {"ms_per_step": tp_here["k4"]["whole_step_ms"], "evals_per_s": logical / (tp_here["k4"]["whole_step_ms"] * 1e-3),
                                         "evals_executed_per_step": tp_here["k4"]["evals_executed_per_launch"], "k_compared_lanes": tp_here["k4"]["k_compared_lanes"]}
            roofline_tp = {"kernel": "k_fast_scan_filter_t<S> (launch B: scan role + transposed Filter role, csrc/bs_filter_t.hpp)",
                           "bound": "valu-issue", "cycles_per_node_and_lane_at_the_bound": CYCLES_PER_NODE_LANE,
                           "bound_source": "tools/ubench/lane_loop.hip (profiles/r06_lane_loop_ubench.txt): v_cmp_ge_i64 -> SGPR pair + v_addc_co_u32 per node, lane and 64 request slots, "
                                           "8.35 cycles per SIMD at 8 waves; peak = 1024 SIMDs x 2.4 GHz / (8.35 x k) x 64 slots",
                           "time_source": "hipEvents around the launch on the library stream (bs_timing_get), mean of the timed steps; rocprofv3 kernel-only times and PMC under profiles/r06_*",
                           "hbm_note": "SURVEY 8(d)'s 65.125 bytes per Filter eval would be hundreds of TB/s here: the operands live in SGPRs / the scalar cache and VGPRs; the launch's "
                                       "HBM-side duty is its OUTPUT, the Filter rows (distinct requests x nodes / 8 bytes): output_GBps",
                           "here": tp_here}
            # the headline entry (the figure a reader looks for first): k = 1 at this config, as in rounds 4-5
            roofline_tp.update({k: e1[k] for k in ("workload", "kernel_us", "evals_executed_per_launch", "achieved_evals_per_s", "k_compared_lanes", "peak_evals_per_s", "frac",
                                                   "whole_step_ms", "output_bytes_per_launch", "output_GBps")})
            if args.config == "cfg3":
                # ... and the same at BASELINE configs[3]'s size (50k pods x 20k nodes: ~1e9 pairs really evaluated per step), where the launch is
                # long enough for the compares to show: the regime's figure at cfg3 is mostly the launch's own latency
                n4, f4, g4, p4, _ = synth.make("cfg4", args.scenario, seed=args.seed)
                roofline_tp["at_cfg4"] = {f"k{k}": tp_entry("cfg4", n4, f4, g4, p4, k, 30, 20) for k in (1, 2, 4)}
            ms, st = resident_ms(bsa, nodes, fit, groups, all_pods, soa.STAGE_PREFILTER | soa.STAGE_TALLY, 100)
            extras["prefilter_only"] = {"ms_per_step": ms, "evals_per_s": logical / (ms * 1e-3)}
            extras["filter_increment_ms"] = ms_per_step - ms if args.stages == "all" else None
            if args.stages == "all":
                # Filter's deny entry (core.go:183-185) replayed inside the batch: the reference's behaviour with the Filter extension point enabled
                ms_fd, st = resident_ms(bsa, nodes, fit, groups, all_pods, stages | soa.BATCH_FILTER_DENY, 100)
                fd_launch = launch_times(bsa, nodes, fit, groups, all_pods, stages | soa.BATCH_FILTER_DENY, 40)      # device time per launch group (hipEvents)
                plain_launch = launch_times(bsa, nodes, fit, groups, all_pods, stages, 40)
                with bsa.Context(scalar_lanes=nodes.lanes - 4) as c2:
                    c2.load_nodes(nodes, fit)
                    c2.load_groups(groups)
                    c2.load_pods(all_pods)
                    fd = c2.batch(stages | soa.BATCH_FILTER_DENY, bitmap=False, rows=False)
                    reruns = c2.filter_deny_reruns()
                extras["filter_deny_on_device"] = {"ms_per_step": ms_fd, "extra_ms_over_the_what_if_filter": ms_fd - ms_per_step, "launches": st["launches"], "chain": st["chain"],
                                                   "device_us_per_launch_group": fd_launch, "device_us_per_launch_group_without_the_flag": plain_launch,
                                                   "extra_device_us": sum(fd_launch.values()) - sum(plain_launch.values()),
                                                   "ms_per_step_note": "steps run back to back WITHOUT a read in between: a BS_BATCH_FILTER_DENY batch waits for the stream before it resets the pinned verdict words "
                                                                       "an unread predecessor may still store into (bs_batch_run, fd_unsynced), so ms_per_step here is a host-serialised figure; a caller that "
                                                                       "reads every batch has waited anyway.  extra_device_us is what the flag adds on the device (one apply launch behind the chain)",
                                                   "pods_turned_away_by_filters_entry": int(((fd.pf_code == soa.PF_ERR_DENIED) & (out.pf_code != soa.PF_ERR_DENIED)).sum()),
                                                   "groups_ready": int(fd.group_ready.sum()), "fixed_point_reruns": reruns,
                                                   "note": "BS_BATCH_FILTER_DENY: results == PreFilter + Filter-on-every-node pod by pod (tests/test_gpu_filter_deny.py)"}
            seeds = {}
            for sd in (1, 2, 3):
                n2, f2, g2, p2, _ = synth.make(args.config, args.scenario, seed=sd)
                ms, _st = resident_ms(bsa, n2, f2, g2, p2, stages, 60)
                seeds[str(sd)] = ms
            extras["ms_per_step_by_seed"] = seeds
            drain = drain_section(bsa, nodes, fit, groups, all_pods, args)
        if single and not args.no_cpu_baseline:
            cpu = cpu_baseline(bsa, nodes, fit, groups, pods, stages, args.cpu_reps)
            if drain is not None:
                seq = drain.pop("_cpu_sequential_pass", None)
                if seq:
                    cpu["sequential_pass"] = seq
                    cpu["gang_admit_latency_ms_p50"] = seq["gang_admit_latency_ms_p50"]
                    cpu["gang_admit_latency_ms_p95"] = seq["gang_admit_latency_ms_p95"]
        if drain is not None:
            drain.pop("_cpu_sequential_pass", None)
        codes = np.bincount(out.pf_code, minlength=256)
        result = {
            "metric": "pod x node fit evals/sec", "value": value, "unit": "evals/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"{args.config}/{args.scenario}: {all_pods.p} pods / {groups.g} groups / {nodes.n} nodes, "
                                   f"{L} int64 lanes (cpu, mem, eph, pods{', gpu' if L > 4 else ''}), seed {args.seed}",
                       "stages": "prefilter+filter+tally+ready" if args.stages == "all" else "prefilter+tally+ready",
                       "parallelism": (f"pod-axis shard x{world}, " + ("pods partitioned by owning rank" if partitioned else "replicated batch, device-side ownership")
                                       + ", 1 all-reduce of admit[G]") if dist is not None else "single GPU",
                       "value_definition": "`value` = LOGICAL pods x nodes per step / step time with every input already resident in HBM (the contract's definition); beside it "
                                           "`value_host_observed` = the same logical evals / the host-observed resident-queue cycle (SURVEY 8(d)'s own definition, PCIe inclusive) and "
                                           "`value_executed` = the pod x node compares the step REALLY executed / step time (request classes + pruning: a fraction of a percent of the logical space).  "
                                           "A step re-runs the whole path (table build, PreFilter, Filter, tally, quorum).  What a host "
                                           "observes per scheduling cycle — group patch, queue delta, batch, decisions back — is `value_host_observed` "
                                           "(= logical evals / host_cycle resident p50): that is SURVEY 8(d)'s own definition of the metric (logical P x N per batch / "
                                           "wall time, host-observed, including the per-batch inputs' way in and the decisions' way out, node SoA resident).  Both are "
                                           "LOGICAL rates: `work_avoided` says how little of the pods x nodes space is actually evaluated (`value_executed`), and the "
                                           "figure for real per-pair work is `roofline_throughput` (every pod its own request)",
                       "logical_evals_per_step": logical, "fast_path": stats["fast_path"], "chain": stats["chain"], "launches_per_step": stats["launches"],
                       "tables_built": stats["tables_built"],
                       "decisions": {soa.PF_NAMES.get(i, str(i)): int(c) for i, c in enumerate(codes) if c},
                       "groups_ready": int(out.group_ready.sum())},
            "timed_regions_ms": [r * 1e3 for r in regions],
            "timed_regions_note": f"{TIMED_REGIONS} regions of exactly {args.steps} steps each, every one between barrier + synchronize (max over ranks); ms_per_step and value are the MEDIAN region's",
            "value_resident": value,
            "value_executed": (stats["scan_evals_executed"] + stats["filter_evals_executed"]) / (ms_per_step * 1e-3),
            "value_note": "`value` is a LOGICAL rate (pods x nodes / step time); `value_executed` = pod x node compares the step really executed / step time",
            "value_host_observed": cycle["evals_per_s_at_p50"] if cycle else None,
            "roofline": roofline,
            "roofline_throughput": roofline_tp,
            "roofline_launches": launches,
            "work_avoided": work_avoided,
            "host_cycle": cycle,
            # the reference's semantics (pod-by-pod pass on the device, bit-identical to the CPU port's pass) — NOT the batched cycle
            "gang_admit_latency_ms_p50": drain["sequential_on_device"]["gang_admit_latency_ms_p50"] if drain else None,
            "gang_admit_latency_ms_p95": drain["sequential_on_device"]["gang_admit_latency_ms_p95"] if drain else None,
            "gang_admit_latency_definition": "drain.sequential_on_device (bs_seq_run): first pod of the gang entering PreFilter -> quorum, reference semantics; "
                                             "compare with cpu_baseline.gang_admit_latency_ms_p50 (the CPU port's pass on the same queue)",
            "batched_cycle_latency_ms_p50": cycle["gang_admit_latency_ms_p50"] if cycle else None,
            "batched_cycle_latency_ms_p95": cycle["gang_admit_latency_ms_p95"] if cycle else None,
            "drain": drain,
            "scenarios": extras,
            "kernel_ms_per_step": {k: v[0] / v[1] for k, v in timing.items() if v[1]},
            "cpu_baseline": cpu,
        }
        if dist is not None:
            result["ranks"] = rank_report
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        # the JSON line goes out LAST: anything native libraries (RCCL) left in the C stdio buffer is flushed first
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)
    return result


if __name__ == "__main__":
    main()
