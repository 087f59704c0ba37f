import importlib, sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
bsa = importlib.import_module("batch-scheduler_amd"); soa = bsa.soa
import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import _force_class_mode, load_ctx
orc = importlib.import_module("orc")
for seed in (8057, 8058):
    rng = np.random.default_rng(seed)
    sc = random_objects(seed, n_nodes=60 + seed % 100, n_groups=9, n_pods=180, n_scalars=seed % 3, n_classes=3)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
    _force_class_mode(groups, rng, 3)
    groups.matched[:] = rng.integers(0, 4, groups.g)
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    exp_a = sop.batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT)
        g2 = ctx.read_groups()
        new_matched = rng.integers(0, 6, groups.g).astype(np.uint32)
        for gs in (g2, sop.groups):
            gs.flags &= ~np.uint8(soa.GROUP_DENIED)
            gs.matched[:] = new_matched
        print(seed, "flags", g2.flags.tolist(), "matched", g2.matched.tolist(), "P", pods.p, "K?", ctx.filter_rows_count())
        ctx.load_groups(g2)
        ctx.batch(soa.STAGE_ALL)
        print(seed, ctx.stats(soa.STAGE_ALL))
