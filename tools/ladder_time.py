import importlib, sys, os, time
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tests')]
import numpy as np
bsa=importlib.import_module("batch-scheduler_amd"); soa=bsa.soa
import test_gpu_epoch as T
for steps in (3, 9, 15):
    for env in (None, "1"):
        if env: os.environ["BS_NO_EPOCH"]=env
        else: os.environ.pop("BS_NO_EPOCH", None)
        for cfgname in ("cfg2",):
            nodes, fit, groups, pods = T._ladder_scene(bsa, soa, steps)
            with bsa.Context(scalar_lanes=nodes.lanes-4) as ctx:
                ctx.load_nodes(nodes, fit); ctx.load_groups(groups); ctx.load_pods(pods)
                for _ in range(10): ctx.run(soa.STAGE_ALL)
                ctx.sync(); t=time.perf_counter()
                for _ in range(200): ctx.run(soa.STAGE_ALL)
                ctx.sync(); dt=(time.perf_counter()-t)/200
                st=ctx.stats(soa.STAGE_ALL)
            print(f"ladder steps {steps:2d} {'general chain (BS_NO_EPOCH)' if env else 'default':28s} chain {st['chain']} launches {st['launches']:2d}  {dt*1e6:7.1f} us per batch")
