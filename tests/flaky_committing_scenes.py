"""Round 6 debugging aid (test infrastructure: it uses the oracle, so it lives under tests/): tests/test_gpu_filter_deny.py::test_committing_batches[steady], a few seeds
many times over, optionally against a build variant (BS_AB_LIB); prints which rounds differ from the oracle and in what.  Driven by tools/r06_flaky2.sh on the GPU box."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import orc
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
if os.environ.get("BS_AB_LIB"):            # a build variant instead of the shipped library
    bsa.capi.LIB_PATH = os.path.abspath(os.environ["BS_AB_LIB"])
from test_gpu_filter_deny import scene, flags
from test_gpu_parity import load_ctx
seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [7042]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0
for rep in range(reps):
    for seed in seeds:
        nodes, fit, groups, pods = scene(seed, True)
        sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
        exps = [sop.batch(pods, flags(soa), bitmap=False) for _ in range(2)]
        with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
            log = []
            for rnd in range(2):
                exp = exps[rnd]
                ctx.run(flags(soa) | soa.BATCH_COMMIT)
                got = ctx.read(bitmap=False, rows=False)
                diffs = [n for n in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready") if not np.array_equal(getattr(got, n), getattr(exp, n))]
                st = ctx.stats_read()
                log.append((rnd, diffs, {k: st[k] for k in ("chain", "launches", "fast_path")}, ctx.filter_deny_reruns()))
            if rep == 0 and seed in (7042, 7043):
                print("info", seed, log, flush=True)
            if any(l[1] for l in log):
                bad += 1
                print("rep", rep, "seed", seed, log, flush=True)
print("bad", bad, "of", reps * len(seeds))
