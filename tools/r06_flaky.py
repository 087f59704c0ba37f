"""Round 6 debugging aid: the hand-over time-out test's scenario many times over, with what each batch did (launch counts per kernel group, speculation)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BS_TEST_HANDOVER_TIMEOUT"] = "2"
bsa = importlib.import_module("batch-scheduler_amd")
soa = bsa.soa
nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    with bsa.Context(scalar_lanes=nodes.lanes - 4, enable_timing=int(os.environ.get("T","0"))) as ctx:
        ctx.load_nodes(nodes, fit); ctx.load_groups(groups); ctx.load_pods(pods)
        log = []
        for k in range(4):
            ctx.timing_reset()
            try:
                ctx.run(soa.STAGE_ALL)
                t = {n: v[1] for n, v in ctx.timing().items() if v[1]}
                ctx.read(bitmap=False, rows=False)
                log.append((k, "ok", t, ctx.speculation_stats()))
            except bsa.capi.BsError as e:
                log.append((k, "ERR %d" % e.status, {n: v[1] for n, v in ctx.timing().items() if v[1]}, ctx.speculation_stats()))
        if not (log[0][1] == "ok" and log[1][1] == "ok" and log[2][1].startswith("ERR") and log[3][1] == "ok"):
            bad += 1
            print("rep", rep, log)
print("bad", bad)
