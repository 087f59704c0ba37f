"""Small random object-level scenarios (Go-shaped objects of oracle/naive_ref.py) used to cross-check
the C oracle against the naive restatement and, on the GPU, the HIP path against the oracle."""
import numpy as np

import naive_ref as nv

SCALARS = ["nvidia.com/gpu", "tencent.cr/tencentip"]


def random_objects(seed, n_nodes=None, n_groups=None, n_pods=None, n_scalars=None, n_classes=None, edge=True):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(0, 3)) if n_scalars is None else n_scalars
    names = SCALARS[:S]
    N = int(rng.integers(0, 12)) if n_nodes is None else n_nodes
    G = int(rng.integers(1, 6)) if n_groups is None else n_groups
    P = int(rng.integers(1, 24)) if n_pods is None else n_pods
    C = int(rng.integers(1, 4)) if n_classes is None else n_classes

    def qty(hi, lo=0):
        return int(rng.integers(lo, hi + 1))

    nodes = []
    for _ in range(N):
        a, r = nv.Resource(), nv.Resource()
        al = {"cpu": qty(64000, 1000), "memory": qty(2 ** 38, 2 ** 30) | 1, "pods": qty(250, 1)}
        if rng.random() < 0.7:
            al["ephemeral-storage"] = qty(2 ** 40)
        for nm in names:
            if rng.random() < 0.7:
                al[nm] = qty(16)
        a.Add(al)
        util = rng.random() * 1.1
        rq = {"cpu": int(al["cpu"] * util), "memory": int(al["memory"] * util)}
        if "ephemeral-storage" in al and rng.random() < 0.8:
            rq["ephemeral-storage"] = int(al["ephemeral-storage"] * util * 0.5)
        for nm in names:
            if rng.random() < 0.6:
                rq[nm] = qty(8)
        r.Add(rq)
        info = nv.NodeInfo(a, r, qty(60))
        if edge:
            x = rng.random()
            if x < 0.05:
                info.nil = True
            elif x < 0.10:
                info.has_node = False
            elif x < 0.18:
                info.unschedulable = True
            elif x < 0.24:
                info.taint_err = True
            if rng.random() < 0.1:
                r.AllowedPodNumber = qty(5, 1)     # requested.AllowedPodNumber != 0 path of core.go:650-653
        for c in range(C):
            if rng.random() < 0.15:
                info.labels_fit[c] = False
        nodes.append(info)

    def pod_requests():
        rq = {"cpu": int(rng.choice([100, 500, 1000, 2000, 4000]))}
        if rng.random() < 0.8:
            rq["memory"] = int(rng.choice([1, 2, 4, 8])) * 2 ** 30
        if rng.random() < 0.3:
            rq["ephemeral-storage"] = 10 * 2 ** 30
        for nm in names:
            x = rng.random()
            if x < 0.3:
                rq[nm] = int(rng.choice([0, 1, 2, 8]))
            elif edge and x < 0.33:
                rq[nm] = -1
        if edge and rng.random() < 0.1:
            rq["pods"] = 1
        return rq

    cache = {}
    gnames = [f"ns/g{i}" for i in range(G)]
    for nm in gnames:
        mm = qty(6, 1)
        if edge and rng.random() < 0.06:
            mm = 0
        pg = nv.PodGroup(nm, mm)
        x = rng.random()
        if x < 0.25:
            pg.status_scheduled = qty(mm)
        elif edge and x < 0.30:
            pg.status_scheduled = mm + qty(2, 1)
        pgs = nv.PGS(pg)
        if rng.random() < 0.5:
            pgs.matched = qty(max(mm, 1))
        if rng.random() < 0.55:
            pgs.pod = nv.Pod(nm + "-rep", nm, pod_requests(), cls=qty(C - 1))
            if rng.random() < 0.85:
                pg.min_resources = nv.pod_resource_require(pgs.pod).ResourceList()
        elif edge and rng.random() < 0.2:
            pg.min_resources = pod_requests()           # MinResources set by the user, no pod seen yet
        if rng.random() < 0.15:
            pgs.scheduled = True
        if rng.random() < 0.08:
            pg.occupied_by = "owner-a"
        cache[nm] = pgs
    denied = {nm for nm in gnames if rng.random() < 0.1}

    pods, permitted = [], set()
    for i in range(P):
        x = rng.random()
        if x < 0.08:
            grp = None
        elif x < 0.12:
            grp = "ns/missing"
        else:
            grp = gnames[qty(G - 1)]
        refs = ()
        y = rng.random()
        if y < 0.2:
            refs = ("owner-a",)
        elif y < 0.24:
            refs = ("owner-b",)
        elif y < 0.27:
            refs = ("owner-b", "owner-a")
        pod = nv.Pod(f"uid{i}", grp, pod_requests(), cls=qty(C - 1), owner_refs=refs)
        if rng.random() < 0.06:
            permitted.add(pod.uid)
        pods.append(pod)
    return dict(nodes=nodes, cache=cache, pods=pods, names=names, n_classes=C, denied=denied, permitted=permitted)


def naive_batch(sc, run_filter=True):
    """Sequential naive replay of one batch -> dict of per-pod results (same shape as BatchOut)."""
    sop = nv.ScheduleOperation(sc["nodes"], sc["cache"])
    sop.denied = set(sc["denied"])
    sop.permitted = set(sc["permitted"])
    gnames = list(sc["cache"].keys())
    codes, fks, leaders, flc, feas, bits = [], [], [], [], [], []
    admit = {nm: 0 for nm in gnames}
    N = len(sc["nodes"])
    for pod in sc["pods"]:
        code, fk = sop.prefilter(pod)
        codes.append(code)
        fks.append(fk)
        leaders.append(gnames.index(sop.max_finished_pg) if sop.max_pg_status is not None else -1)
        fl, f, row = nv.soa.FL_NOT_RUN, 0, []
        if run_filter and code < 16:
            fl = nv.soa.FL_PASS_NOT_GROUPED
            for k in range(N):
                fl, fn = sop.filter_node(pod, k)
                ok = fl < 16 and (fl != nv.soa.FL_EVALUATED or fn < 16)
                row.append(ok)
                f += ok
            if N == 0:
                fl, _ = sop.filter_node(pod, 0)
        flc.append(fl)
        feas.append(f)
        bits.append(row)
        if pod.group in admit and code < 16 and (not run_filter or f > 0):
            admit[pod.group] += 1
    ready = []
    for nm in gnames:
        pgs = sc["cache"][nm]
        ready.append(int(nv.u32(pgs.matched + admit[nm]) >= nv.u32(pgs.pod_group.min_member - pgs.pod_group.status_scheduled)))
    return dict(pf_code=codes, pf_first_k=fks, pf_leader=leaders, fl_code=flc, fl_feasible=feas, bits=bits,
                group_admit=[admit[nm] for nm in gnames], group_ready=ready, denied=set(sop.denied))
