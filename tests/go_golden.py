"""Golden vectors from the REFERENCE's own Go functions — input builder, oracle-side evaluation, comparison.

tools/dump_golden_input.py writes tests/golden/go_reference_input.json (small, seeded).  A maintainer with Go runs
go/pkg/scheduler/core/golden_dump_test.go next to the reference's core.go; it writes tests/golden/go_reference_dump.json.
tests/test_go_reference_dump.py compares the oracle with that dump entry by entry (skipped while the dump is absent).
`oracle_dump` produces the same document from the oracle, so the format, the loader and the comparison are exercised
in every CPU run (oracle vs its own dump must compare clean; a perturbed dump must not).
"""
import importlib

import numpy as np

soa = importlib.import_module("batch-scheduler_amd.soa")
NS = "ns/"


def build_input(seed: int = 20260921, n_nodes: int = 48, n_groups: int = 9, n_pods: int = 60, classes: int = 3) -> dict:
    """A warm scene with a unique leader (no findMaxPG tie, so Go's random map order cannot matter), a few groups that
    have not seen a pod yet, unschedulable nodes, scalar keys present / absent on both sides, non-f32-representable memory."""
    rng = np.random.default_rng(seed)
    lanes = ["cpu", "memory", "ephemeral-storage", "pods", "example.com/gpu"]
    nodes = []
    for k in range(n_nodes):
        gpu = bool(rng.random() < 0.4)
        alloc = [int(rng.choice([8000, 16000, 32000])) - int(rng.integers(0, 300)), int(rng.choice([32, 64, 128])) * (1 << 30) - (int(rng.integers(1, 4000)) | 1) * 1024,
                 int(rng.integers(100, 500)) * (1 << 30), int(rng.choice([40, 110])), 8 if gpu else 0]
        util = float(rng.choice([0.5, 0.66, 0.72, 0.9]))
        has_req_gpu = gpu and bool(rng.random() < 0.7)
        nodes.append({"name": f"n{k}", "alloc": alloc, "alloc_keys": [gpu],
                      "requested": [int(alloc[0] * util), int(alloc[1] * util) | 1, int(alloc[2] * util * 0.5), 0, int(rng.integers(0, 8)) if has_req_gpu else 0],
                      "req_keys": [has_req_gpu], "pod_count": int(rng.integers(1, 9)), "unschedulable": bool(rng.random() < 0.06),
                      "fit": [bool(rng.random() < 0.9) for _ in range(classes)]})
    groups = []
    for g in range(n_groups):
        mm = int(rng.integers(3, 9))
        seen = g < n_groups - 2                                   # the last two groups: controller-created, no pod seen yet
        gpu = int(rng.choice([0, 0, 1]))
        groups.append({"name": f"{NS}g{g}", "min_member": mm, "scheduled": int(rng.integers(0, 2)) if seen else 0,
                       "matched": 0, "latch": bool(seen and rng.random() < 0.1), "has_pod": seen, "cls": int(rng.integers(0, classes)),
                       "min_resources": [int(rng.choice([500, 1000, 2000])), int(rng.choice([1, 2, 4])) * (1 << 30), 0, 0, gpu] if seen else None,
                       "min_res_keys": [gpu > 0]})
    groups[0].update(min_member=24, scheduled=0, latch=False, min_resources=[2000, 4 * (1 << 30), 0, 0, 0], min_res_keys=[False])
    # distinct progress per candidate: matched chosen so that (matched + scheduled) * 1000 / min_member never ties at the top
    seenp = set()
    for g in groups:
        if not g["has_pod"] or g["latch"]:
            continue
        for m in rng.permutation(g["min_member"]):
            p = (int(m) + g["scheduled"]) * 1000 // g["min_member"]
            if p not in seenp and int(m) + g["scheduled"] < g["min_member"]:
                g["matched"] = int(m)
                seenp.add(p)
                break
    pods = []
    for i in range(n_pods):
        r = rng.random()
        if r < 0.05:
            grp, mr = "", None
        elif r < 0.08:
            grp, mr = "ghost", None
        else:
            gi = int(rng.integers(0, n_groups))
            grp, mr = f"g{gi}", groups[gi]["min_resources"]
        gpu = int(mr[4]) if mr else int(rng.choice([0, 1]))
        k = int(rng.choice([1, 1, 2, 6]))                          # heterogeneous gangs: some members ask for a multiple
        req = [(int(mr[0]) if mr else 1000) * k, (int(mr[1]) if mr else (1 << 30)) * k, 0, 0, gpu]
        cls = groups[int(grp[1:])]["cls"] if grp.startswith("g") and grp != "ghost" else int(rng.integers(0, classes))
        pods.append({"uid": f"u{i}", "group": grp, "req": req, "req_keys": [gpu > 0], "cls": cls})
    queries = []
    for q in range(40):
        gpu = int(rng.choice([0, 0, 1, 4, 60]))
        queries.append({"cls": int(rng.integers(0, classes)), "percent": float(np.float32(rng.choice([1.0, 0.7]))),
                        "req": [int(rng.integers(0, 200000)), int(rng.integers(0, 600)) * (1 << 30), int(rng.integers(0, 2)) * (1 << 30), int(rng.integers(0, 60)), gpu],
                        "req_keys": [bool(gpu > 0 or rng.random() < 0.2)]})
    pairs = [[int(rng.integers(0, n_pods)), int(rng.integers(0, n_nodes))] for _ in range(120)]
    return {"lanes": lanes, "classes": classes, "nodes": nodes, "groups": groups, "pods": pods, "queries": queries, "filter_pairs": pairs}


def to_soa(inp: dict):
    L, S = len(inp["lanes"]), len(inp["lanes"]) - 4
    n = len(inp["nodes"])
    alloc, req = np.zeros((L, n), np.int64), np.zeros((L, n), np.int64)
    ap, rp, fl = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
    fit = np.zeros((inp["classes"], n), bool)
    for k, nd in enumerate(inp["nodes"]):
        alloc[:, k], req[:, k] = nd["alloc"], nd["requested"]
        req[3, k] = nd["pod_count"]                                # core.go:650-653: AllowedPodNumber of requested is 0 -> len(Pods())
        for s in range(S):
            ap[k] |= int(nd["alloc_keys"][s]) << s
            rp[k] |= int(nd["req_keys"][s]) << s
            if not nd["alloc_keys"][s]:
                alloc[4 + s, k] = 0
            if not nd["req_keys"][s]:
                req[4 + s, k] = 0
        if nd["unschedulable"]:
            fl[k] |= soa.NODE_UNSCHEDULABLE
        fit[:, k] = nd["fit"]
    nodes = soa.Nodes(alloc, req, ap, rp, fl)
    G = len(inp["groups"])
    groups = soa.Groups.empty(G, L)
    gidx = {}
    for g, gr in enumerate(inp["groups"]):
        gidx[gr["name"]] = g
        groups.min_member[g], groups.status_scheduled[g], groups.matched[g] = gr["min_member"], gr["scheduled"], gr["matched"]
        f = (soa.GROUP_SCHEDULED_LATCH if gr["latch"] else 0) | (soa.GROUP_HAS_POD if gr["has_pod"] else 0)
        if gr["min_resources"] is not None:
            f |= soa.GROUP_HAS_MINRES
            groups.min_resources[:, g] = gr["min_resources"]
            for s in range(S):
                groups.min_resources_present[g] |= int(gr["min_res_keys"][s]) << s
                if not gr["min_res_keys"][s]:
                    groups.min_resources[4 + s, g] = 0
        groups.flags[g] = f
        groups.cls[g] = gr["cls"]
    P = len(inp["pods"])
    pg, preq, ppres, pcls = np.zeros(P, np.int32), np.zeros((L, P), np.int64), np.zeros(P, np.uint32), np.zeros(P, np.uint32)
    for i, p in enumerate(inp["pods"]):
        pg[i] = soa.POD_NOT_GROUPED if p["group"] == "" else gidx.get(NS + p["group"], soa.POD_GROUP_MISSING)
        preq[:, i] = p["req"]
        for s in range(S):
            ppres[i] |= int(p["req_keys"][s]) << s
            if not p["req_keys"][s]:
                preq[4 + s, i] = 0
        pcls[i] = p["cls"]
    pods = soa.Pods(pg, preq, ppres, pcls, np.zeros(P, np.uint64), np.zeros(P, np.uint8))
    return nodes, soa.FitMasks.from_bool(fit), groups, pods, gidx


def _keys(present: int, S: int):
    return [bool((present >> s) & 1) for s in range(S)]


def _pf_message(code: int, pods, i, inp) -> str:
    if code < 16:
        return ""
    name = NS + inp["pods"][i]["group"]
    return {soa.PF_ERR_PG_NOT_FOUND: f"can not found pod group: {name}", soa.PF_ERR_DENIED: f"pod with pgName: {name} last failed in 20s, deny",
            soa.PF_ERR_OCCUPIED: "pod group has been occupied by", soa.PF_REJECT_FIRST: "cluster resource not enough",
            soa.PF_REJECT_RESERVE: "cluster resource not enough"}.get(code, f"code {code}")


def oracle_dump(inp: dict, orc) -> dict:
    """the document golden_dump_test.go writes, computed by the oracle"""
    nodes, fit, groups, pods, gidx = to_soa(inp)
    S = nodes.lanes - 4
    names = [g["name"] for g in inp["groups"]]
    snap = orc.Snapshot(nodes, fit)
    out = {}
    leader, _, panic = orc.find_max_pg(groups)
    out["find_max_pg"] = {"leaders_seen": [names[leader] if leader >= 0 else ""]}
    pre = []
    for g, gr in enumerate(inp["groups"]):
        for m in (gr["matched"], 0):
            lanes, present = orc.pre_allocated(groups, g, m, S)
            pre.append({"group": g, "matched": m, "lanes": lanes, "keys": _keys(present, S)})
    out["pre_allocated"] = pre
    single = []
    for c in range(inp["classes"]):
        for k in range(nodes.n):
            for pct in (1.0, float(np.float32(0.7))):
                lanes, present = snap.single_node_resource(c, k, pct)
                single.append({"cls": c, "node": k, "percent": pct, "lanes": lanes, "keys": _keys(present, S)})
    out["single_node"] = single
    fits = []
    for qi, q in enumerate(inp["queries"]):
        present = sum(int(b) << s for s, b in enumerate(q["req_keys"]))
        req = list(q["req"])
        for s in range(S):
            if not q["req_keys"][s]:
                req[4 + s] = 0
        ok, fk, _ = snap.compare_cluster(q["cls"], req, present, q["percent"])
        fits.append({"query": qi, "fits": ok, "first_k": fk if ok else -1})
    out["cluster_fits"] = fits
    left = []
    for k in range(nodes.n):
        r = snap.left_resource(k)
        left.append({"node": k, "lanes": r[0] if r else None, "keys": _keys(r[1], S) if r else None})
    out["left_resource"] = left
    sop = orc.Sop(snap, groups)
    filt = []
    if leader >= 0:
        for pi, ni in inp["filter_pairs"]:
            if pods.group[pi] < 0:
                continue
            flc, fn = sop.filter_node(pods, pi, leader, ni)
            msg = ""
            if flc == soa.FL_EVALUATED and fn == soa.FN_ERR_NOT_ENOUGH:
                msg = "resource not enough"
            elif flc == soa.FL_EVALUATED and fn == soa.FN_ERR_SNAPSHOT:
                msg = "SnapShot not initialized"
            filt.append({"pod": pi, "node": ni, "leader": names[leader], "err": msg})
    out["filter"] = filt
    seq = []
    for i in range(pods.p):
        code, _ = sop.prefilter(pods, i)
        seq.append({"pod": i, "err": _pf_message(code, pods, i, inp), "leader_after": names[sop.leader] if sop.leader >= 0 else ""})
    out["prefilter_sequence"] = seq
    return out


def _err_class(msg: str) -> str:
    for key in ("can not found pod group", "last failed in 20s", "occupied by", "cluster resource not enough", "resource not enough", "SnapShot not initialized"):
        if key in msg:
            return key
    return msg


def compare(ref: dict, mine: dict) -> list:
    """differences between the reference's dump and the oracle's (empty list == the oracle is pinned on every entry)"""
    bad = []
    rl, ml = ref["find_max_pg"]["leaders_seen"], mine["find_max_pg"]["leaders_seen"]
    if ml[0] not in rl:
        bad.append(("find_max_pg", rl, ml))
    unique_leader = len(rl) == 1
    for sec, keyf in (("pre_allocated", lambda e: (e["group"], e["matched"])), ("single_node", lambda e: (e["cls"], e["node"], round(float(e["percent"]), 4))),
                      ("cluster_fits", lambda e: e["query"]), ("left_resource", lambda e: e["node"]), ("filter", lambda e: (e["pod"], e["node"]))):
        if sec == "filter" and not unique_leader:
            continue
        r = {keyf(e): e for e in ref[sec]}
        m = {keyf(e): e for e in mine[sec]}
        if set(r) != set(m):
            bad.append((sec, "key sets differ", sorted(set(r) ^ set(m))[:5]))
            continue
        for k in r:
            a, b = dict(r[k]), dict(m[k])
            if "err" in a:
                a["err"], b["err"] = _err_class(a["err"]), _err_class(b["err"])
            a.pop("percent", None), b.pop("percent", None)
            if a != b:
                bad.append((sec, k, a, b))
    for a, b in zip(ref["prefilter_sequence"], mine["prefilter_sequence"]):
        if _err_class(a["err"]) != _err_class(b["err"]) or (unique_leader and a["leader_after"] != b["leader_after"]):
            bad.append(("prefilter_sequence", a, b))
    if len(ref["prefilter_sequence"]) != len(mine["prefilter_sequence"]):
        bad.append(("prefilter_sequence", "length"))
    return bad
