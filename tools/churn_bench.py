"""BASELINE config 5: cfg3 + a stream of node events (40 % requested-update, 30 % append, 30 % stable
remove); after every 100 events the whole batch is re-scored.  Times, per round, bs_nodes_apply (host
mirror edit + upload of the changed suffix + re-derivation) and the re-score batch (PreFilter + tally,
decisions read back).  Parity under churn is asserted in tests/test_gpu_parity.py (test_churn_*); nothing under
oracle/ is used here.
Usage (GPU box): python tools/churn_bench.py [events_total=10000] [events_per_round=100]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np

bsa = importlib.import_module("batch-scheduler_amd")
soa, capi, synth = bsa.soa, bsa.capi, bsa.synth


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    nodes, fit, groups, pods, _ = synth.make("cfg3", "tail")
    rng = np.random.default_rng(5)
    L = nodes.lanes
    alloc, req = nodes.allocatable.copy(), nodes.requested.copy()
    ap, rp, fl = nodes.allocatable_present.copy(), nodes.requested_present.copy(), nodes.flags.copy()
    fitb = fit.to_bool()
    ctx = capi.Context(scalar_lanes=L - 4)
    ctx.load_nodes(nodes); ctx.load_fit(fit); ctx.load_groups(groups); ctx.load_pods(pods)
    stages = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n + total, bitmap=False)
    ctx.batch(stages, bitmap=False)
    t_apply, t_score = [], []
    for _ in range(total // per):
        deltas = []
        for _e in range(per):
            kind = int(rng.choice([capi.DELTA_UPDATE, capi.DELTA_APPEND, capi.DELTA_REMOVE], p=[0.4, 0.3, 0.3]))
            n = alloc.shape[1]
            d = capi.NodeDelta()
            d.kind = kind
            if kind == capi.DELTA_REMOVE:
                idx = int(rng.integers(0, n)); d.index = idx
                alloc, req = np.delete(alloc, idx, 1), np.delete(req, idx, 1)
                ap, rp, fl = np.delete(ap, idx), np.delete(rp, idx), np.delete(fl, idx)
                fitb = np.delete(fitb, idx, 1)
            else:
                src = int(rng.integers(0, n))
                col_a, col_r = alloc[:, src].copy(), req[:, src].copy()
                col_r[0] = int(col_a[0] * rng.random()); col_r[1] = int(col_a[1] * rng.random())
                for j in range(L):
                    d.allocatable[j], d.requested[j] = int(col_a[j]), int(col_r[j])
                d.allocatable_present, d.requested_present = int(ap[src]), int(rp[src])
                d.fit_default, d.n_fit_exceptions = 1, 0
                fcol = np.ones(fitb.shape[0], bool)
                if kind == capi.DELTA_UPDATE:
                    idx = int(rng.integers(0, n)); d.index = idx
                    alloc[:, idx], req[:, idx], ap[idx], rp[idx], fl[idx] = col_a, col_r, ap[src], rp[src], 0
                    fitb[:, idx] = fcol
                else:
                    alloc, req = np.concatenate([alloc, col_a[:, None]], 1), np.concatenate([req, col_r[:, None]], 1)
                    ap, rp = np.append(ap, ap[src]).astype(np.uint32), np.append(rp, rp[src]).astype(np.uint32)
                    fl = np.append(fl, 0).astype(np.uint8)
                    fitb = np.concatenate([fitb, fcol[:, None]], 1)
            deltas.append(d)
        a = time.perf_counter()
        ctx.apply_node_deltas(deltas)
        b = time.perf_counter()
        ctx.run(stages)
        got = ctx.read(bitmap=False)
        c = time.perf_counter()
        t_apply.append(b - a); t_score.append(c - b)
    print(json.dumps({"workload": f"cfg3/tail + {total} node events, re-score every {per}", "rounds": len(t_apply),
                      "apply_ms_p50": round(float(np.median(t_apply)) * 1e3, 3), "rescore_ms_p50": round(float(np.median(t_score)) * 1e3, 3),
                      "events_per_s": round(total / (sum(t_apply) + sum(t_score))),
                      "nodes_end": int(alloc.shape[1]), "groups_ready_last_round": int(got.group_ready.sum()),
                      "note": "apply = bs_nodes_apply of the round's deltas (host mirror edit, upload and re-derive from the first changed index); "
                              "rescore = PreFilter + tally batch + decisions D2H, host-observed"}))


if __name__ == "__main__":
    main()
