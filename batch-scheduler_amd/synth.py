"""Seeded synthetic cluster snapshots for the configurations of BASELINE.json.

Counter-based SplitMix64 (vectorised with numpy) so that any element is a pure function of
(seed, stream, index): the same snapshot is produced on the build container and on the GPU box.

Configs (BASELINE.json `configs`):
  cfg2  1k pods / 200 groups / 500 nodes, cpu+mem (S=0)
  cfg3  10k pods / 2k groups / 5k nodes, cpu/mem/eph/gpu (S=1)     <- the bench workload
  cfg4  50k pods / 5k groups / 20k nodes (S=1), pod axis sharded over GPUs
Scenarios:
  cold  nothing seen yet: matched = 0, no representative pods -> core.go:136-147 for every group
  warm  steady state: every group has its pod/MinResources; one leader one pod short of quorum
        (core.go:157-166 for everybody else)
  busy  warm on a saturated cluster with a leader that still needs most of its gang: most
        reservations fail after a full scan (REJECT + deny-cache)
  tail  saturated cluster whose free capacity sits in freshly appended nodes at the END of the node
        list (what a cluster autoscaler produces): every pod has to walk almost the whole list
        before the running sum covers it.  Worst case for the reference's early exit, and the
        workload bench.py quotes.
"""
from __future__ import annotations

import numpy as np

from . import soa

GI = 1 << 30
KI = 1 << 10
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

CONFIGS = {
    "cfg2": dict(pods=1000, groups=200, nodes=500, scalars=0, classes=8),
    "cfg3": dict(pods=10000, groups=2000, nodes=5000, scalars=1, classes=32),
    "cfg4": dict(pods=50000, groups=5000, nodes=20000, scalars=1, classes=64),
    "tiny": dict(pods=96, groups=12, nodes=150, scalars=1, classes=3),
}


def _mix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class Stream:
    """SplitMix64 in counter mode: element i of stream s under seed."""

    def __init__(self, seed: int, stream: int):
        with np.errstate(over="ignore"):
            self.base = _mix(np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF)]) + _GAMMA * np.uint64(stream + 1))[0]

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            return _mix(self.base + idx * _GAMMA)

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def integers(self, n: int, lo: int, hi: int) -> np.ndarray:
        """inclusive range"""
        return (lo + (self.u64(n) % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int64)

    def choice(self, n: int, values, probs=None) -> np.ndarray:
        values = np.asarray(values)
        if probs is None:
            return values[self.integers(n, 0, len(values) - 1)]
        cdf = np.cumsum(probs)
        return values[np.minimum(np.searchsorted(cdf, self.uniform(n), side="right"), len(values) - 1)]


def make_nodes(seed: int, n: int, scalars: int, scenario: str):
    L = 4 + scalars
    sku = Stream(seed, 1).choice(n, [0, 1, 2], [0.5, 0.3, 0.2])
    cap_cpu = np.array([32000, 64000, 96000])[sku]
    cap_mem = np.array([128, 256, 512])[sku] * GI
    cap_eph = np.array([500, 1024, 2048])[sku] * GI
    cap_pods = np.array([110, 110, 250])[sku]
    is_gpu = sku == 2
    alloc = np.zeros((L, n), np.int64)
    alloc[0] = cap_cpu - Stream(seed, 2).integers(n, 100, 500)
    # reserved memory in odd-KiB steps: allocatable is NOT float32-representable
    alloc[1] = cap_mem - (Stream(seed, 3).integers(n, GI // KI, 4 * GI // KI) | 1) * KI
    alloc[2] = cap_eph - Stream(seed, 4).integers(n, 0, 64) * GI
    alloc[3] = cap_pods
    u = Stream(seed, 5).uniform(n)
    v = Stream(seed, 6).uniform(n)
    if scenario in ("cold", "warm"):
        util = np.where(u < 0.6, 0.1 + 0.5 * v, np.where(u < 0.9, 0.6 + 0.3 * v, 0.9 + 0.1 * v))
    elif scenario == "busy":
        util = np.where(u < 0.85, 0.72 + 0.26 * v, 0.05 + 0.4 * v)
    elif scenario == "tail":
        tail0 = n - max(1, n // 12)
        util = np.where(np.arange(n) >= tail0, 0.02 + 0.18 * v, 0.70 + 0.03 * v)
    else:
        raise ValueError(scenario)
    req = np.zeros((L, n), np.int64)
    jitter = Stream(seed, 7).uniform(3 * n).reshape(3, n)
    req[0] = np.floor(alloc[0] * np.clip(util + 0.04 * (jitter[0] - 0.5), 0, 1.05)).astype(np.int64)
    req[1] = np.floor(alloc[1] * np.clip(util + 0.04 * (jitter[1] - 0.5), 0, 1.05)).astype(np.int64)
    req[2] = np.floor(alloc[2] * np.clip(0.5 * util + 0.04 * (jitter[2] - 0.5), 0, 1.0)).astype(np.int64)
    req[3] = Stream(seed, 8).integers(n, 5, 60)            # podCount
    ap = np.zeros(n, np.uint32)
    rp = np.zeros(n, np.uint32)
    if scalars >= 1:
        alloc[4] = np.where(is_gpu, 8, 0)
        ap |= np.where(is_gpu, 1, 0).astype(np.uint32)
        has_req = is_gpu & (Stream(seed, 9).uniform(n) < 0.7)      # Q3: key present on 70 % of GPU nodes
        rp |= np.where(has_req, 1, 0).astype(np.uint32)
        gpu_used = np.minimum(8, np.floor(8 * util + Stream(seed, 10).uniform(n)).astype(np.int64))
        req[4] = np.where(has_req, gpu_used, 0)
    for s in range(1, scalars):                                     # further extended resources
        present = Stream(seed, 20 + s).uniform(n) < 0.5
        alloc[4 + s] = np.where(present, 16, 0)
        ap |= (np.where(present, 1, 0) << s).astype(np.uint32)
        hr = present & (Stream(seed, 40 + s).uniform(n) < 0.6)
        rp |= (np.where(hr, 1, 0) << s).astype(np.uint32)
        req[4 + s] = np.where(hr, Stream(seed, 60 + s).integers(n, 0, 12), 0)
    flags = np.zeros(n, np.uint8)
    f = Stream(seed, 11).uniform(n)
    flags[f < 0.02] = soa.NODE_UNSCHEDULABLE
    flags[(f >= 0.02) & (f < 0.025)] = soa.NODE_TAINT_ERR
    flags[(f >= 0.025) & (f < 0.027)] = soa.NODE_NO_NODE
    return soa.Nodes(alloc, req, ap, rp, flags)


def make_fit(seed: int, n: int, classes: int) -> soa.FitMasks:
    fit = Stream(seed, 12).uniform(classes * n).reshape(classes, n) < 0.95
    return soa.FitMasks.from_bool(fit)


def make_groups_and_pods(seed: int, p: int, g: int, scalars: int, classes: int, scenario: str):
    L = 4 + scalars
    size = max(1, p // g)
    # group templates
    t_cpu = Stream(seed, 13).choice(g, [500, 1000, 2000, 4000])
    t_mem = Stream(seed, 14).choice(g, [1, 2, 4, 8]) * GI
    t_eph = Stream(seed, 15).choice(g, [0, 10 * GI])
    t_gpu = Stream(seed, 16).choice(g, [0, 1, 8], [0.8, 0.12, 0.08])
    t_cls = Stream(seed, 17).integers(g, 0, classes - 1)
    hetero = Stream(seed, 18).uniform(g) < 0.10

    groups = soa.Groups.empty(g, L)
    groups.min_member[:] = size
    pgroup = np.minimum(np.arange(p) // size, g - 1).astype(np.int32)
    # interleave the queue a little: pods of neighbouring groups arrive mixed, as a real queue does
    order = np.argsort(np.arange(p) // (4 * size) * (4 * size) + Stream(seed, 19).integers(p, 0, 4 * size - 1), kind="stable")
    pgroup = pgroup[order]
    preq = np.zeros((L, p), np.int64)
    het_scale = np.where(hetero[pgroup], Stream(seed, 21).choice(p, [1, 2]), 1)
    preq[0] = t_cpu[pgroup] * het_scale
    preq[1] = t_mem[pgroup] * het_scale
    preq[2] = t_eph[pgroup]
    ppres = np.zeros(p, np.uint32)
    if scalars >= 1:
        preq[4] = t_gpu[pgroup]
        ppres |= np.where(t_gpu[pgroup] > 0, 1, 0).astype(np.uint32)
    pcls = t_cls[pgroup].astype(np.uint32)
    powner = (pgroup.astype(np.uint64) + np.uint64(1))            # one owner (job) per gang
    pflags = np.zeros(p, np.uint8)
    stray = Stream(seed, 22).uniform(p)
    pgroup = np.where(stray < 0.01, soa.POD_NOT_GROUPED, pgroup).astype(np.int32)    # 1 % ordinary pods
    pods = soa.Pods(pgroup, preq, ppres, pcls, powner, pflags)

    if scenario != "cold":
        groups.flags[:] = soa.GROUP_HAS_POD | soa.GROUP_HAS_MINRES
        groups.cls[:] = t_cls
        groups.min_resources[0] = t_cpu
        groups.min_resources[1] = t_mem
        groups.min_resources[2] = t_eph
        if scalars >= 1:
            groups.min_resources[4] = t_gpu
            groups.min_resources_present[:] = np.where(t_gpu > 0, 1, 0)
        groups.occupied_by[:] = np.arange(g, dtype=np.uint64) + np.uint64(1)
        groups.matched[:] = Stream(seed, 23).integers(g, 0, max(0, size - 2))
        leader = int(Stream(seed, 24).integers(1, 0, g - 1)[0])
        if scenario == "warm":
            groups.matched[leader] = max(1, size - 1)
        elif scenario == "busy":
            groups.min_member[leader] = 64
            groups.matched[leader] = 8
            groups.min_resources[0, leader] = 4000
            groups.min_resources[1, leader] = 8 * GI
        elif scenario == "tail":
            groups.min_member[leader] = 48
            groups.matched[leader] = 40
            groups.min_resources[0, leader] = 4000
            groups.min_resources[1, leader] = 8 * GI
            if scalars >= 1:
                groups.min_resources[4, leader] = 0
                groups.min_resources_present[leader] = 0
        done = Stream(seed, 25).uniform(g) < 0.05                  # a few gangs already latched
        done[leader] = False
        groups.flags[done] |= soa.GROUP_SCHEDULED_LATCH
    return groups, pods


def make(config: str = "cfg3", scenario: str = "tail", seed: int = 20260921, **overrides):
    """-> (Nodes, FitMasks, Groups, Pods, meta)"""
    cfg = dict(CONFIGS[config])
    cfg.update(overrides)
    nodes = make_nodes(seed, cfg["nodes"], cfg["scalars"], scenario)
    fit = make_fit(seed, cfg["nodes"], cfg["classes"])
    groups, pods = make_groups_and_pods(seed, cfg["pods"], cfg["groups"], cfg["scalars"], cfg["classes"], scenario)
    meta = dict(config=config, scenario=scenario, seed=seed, **cfg)
    return nodes, fit, groups, pods, meta


# ---------------------------------------------------------------------------------------------------
# checkFit inputs (core.go:741-759): node labels / taints and pod templates in the object form of
# fitspec.py.  `quirks` mixes in the corner cases of the upstream matchers (requirements that fail to
# convert, empty terms, non-integer Gt/Lt operands, empty keys and values, unknown operators/effects).
def make_fit_scene(seed: int, n: int, classes: int, quirks: bool = True):
    import random
    rnd = random.Random(seed * 1000003 + n * 131 + classes)
    zones = ["az-a", "az-b", "az-c", "az-d"]
    itypes = ["c32", "c64", "g96"]
    teams = ["ml", "web", "batch"]
    odd_values = ["12a", "+5", "-3", "", "007", "9223372036854775807", "9223372036854775808", "x_y.z-1"]
    nodes = []
    for i in range(n):
        it = rnd.choices(itypes, [5, 3, 2])[0]
        labels = {"kubernetes.io/hostname": f"node-{i}", "topology.kubernetes.io/zone": rnd.choice(zones),
                  "node.kubernetes.io/instance-type": it, "pool": f"p{rnd.randrange(8)}", "rack": str(rnd.randrange(40))}
        if it == "g96":
            labels["gpu-count"] = "8"
        if rnd.random() < 0.8:
            labels["disk"] = rnd.choice(["ssd", "hdd"])
        if quirks and rnd.random() < 0.3:
            labels["weird"] = rnd.choice(odd_values)
        taints = []
        if rnd.random() < 0.15:
            taints.append(("dedicated", rnd.choice(teams), "NoSchedule"))
        if it == "g96" and rnd.random() < 0.5:
            taints.append(("nvidia.com/gpu", "present", "NoSchedule"))
        if rnd.random() < 0.03:
            taints.append(("node.kubernetes.io/unreachable", "", "NoExecute"))
        if rnd.random() < 0.10:
            taints.append(("prefer", "x", "PreferNoSchedule"))
        if quirks and rnd.random() < 0.01:
            taints.append(("odd", "y", "SomethingElse"))
        nodes.append({"name": f"node-{i}", "labels": labels, "taints": taints})

    def rand_expr():
        kind = rnd.randrange(12 if quirks else 7)
        if kind == 0:
            return ("topology.kubernetes.io/zone", "In", rnd.sample(zones, rnd.randrange(1, 4)))
        if kind == 1:
            return ("node.kubernetes.io/instance-type", "NotIn", [rnd.choice(itypes)])
        if kind == 2:
            return (rnd.choice(["gpu-count", "disk", "weird"]), "Exists", [])
        if kind == 3:
            return (rnd.choice(["gpu-count", "disk", "missing"]), "DoesNotExist", [])
        if kind == 4:
            return ("rack", "Gt", [str(rnd.randrange(-2, 40))])
        if kind == 5:
            return ("rack", "Lt", [str(rnd.randrange(0, 45))])
        if kind == 6:
            return ("pool", "In", [f"p{rnd.randrange(8)}", f"p{rnd.randrange(8)}"])
        if kind == 7:
            return ("weird", rnd.choice(["Gt", "Lt", "In", "NotIn"]), [rnd.choice(["4", "-3", "007", "12a"])])
        if kind == 8:   # conversion errors: wrong value counts / non-integer operand
            return rnd.choice([("pool", "In", []), ("disk", "Exists", ["ssd"]), ("rack", "Gt", ["abc"]), ("rack", "Lt", ["1", "2"]),
                               ("rack", "Gt", [])])
        if kind == 9:   # unknown operator, invalid key, invalid value
            return rnd.choice([("pool", "Foo", ["p1"]), ("bad key!", "Exists", []), ("pool", "In", ["not valid!"]),
                               ("a/b/c", "DoesNotExist", []), ("pool", "NotIn", ["x" * 64])])
        if kind == 10:
            return ("weird", "In", [""])                       # the empty string is a legal label value
        return ("rack", rnd.choice(["Gt", "Lt"]), [rnd.choice(["+5", "-0", "9223372036854775807", "9223372036854775808"])])

    def rand_field():
        kind = rnd.randrange(6 if quirks else 2)
        if kind == 0:
            return ("metadata.name", "In", [f"node-{rnd.randrange(max(n, 1))}"])
        if kind == 1:
            return ("metadata.name", "NotIn", [f"node-{rnd.randrange(max(n, 1))}"])
        if kind == 2:
            return ("metadata.namespace", rnd.choice(["In", "NotIn"]), [rnd.choice(["", "default"])])
        if kind == 3:
            return ("metadata.name", "In", ["node-0", "node-1"])   # field selectors take exactly one value
        if kind == 4:
            return ("metadata.name", "Exists", [])
        return ("metadata.name", "Bogus", ["node-0"])

    tol_pool = [("dedicated", "Equal", "ml", "NoSchedule"), ("dedicated", "Equal", "web", ""), ("dedicated", "Exists", "", ""),
                ("nvidia.com/gpu", "", "present", "NoSchedule"), ("nvidia.com/gpu", "Exists", "", "NoExecute"),
                ("node.kubernetes.io/unreachable", "Exists", "", "NoExecute"), ("", "Exists", "", ""),
                ("", "Exists", "", "NoSchedule"), ("dedicated", "Equal", "batch", "PreferNoSchedule")]
    if quirks:
        tol_pool += [("dedicated", "Weird", "ml", ""), ("", "Equal", "present", ""), ("", "", "", "NoExecute"),
                     ("dedicated", "Equal", "ml", "SomethingElse")]
    templates = []
    for _ in range(classes):
        tp = {"node_selector": {}, "required": None, "tolerations": []}
        r = rnd.random()
        if r < 0.35:
            tp["node_selector"]["topology.kubernetes.io/zone"] = rnd.choice(zones)
        if 0.25 < r < 0.45:
            tp["node_selector"]["disk"] = rnd.choice(["ssd", "hdd"])
        if quirks and rnd.random() < 0.06:
            tp["node_selector"][rnd.choice(["bad key!", "pool"])] = rnd.choice(["p1", "not valid!", ""])
        if rnd.random() < 0.5:
            terms = []
            for _t in range(rnd.choice([0, 1, 1, 1, 2, 3]) if quirks else rnd.choice([1, 1, 2])):
                term = {"expressions": [rand_expr() for _e in range(rnd.choice([0, 1, 1, 2, 3]) if quirks else rnd.choice([1, 2]))],
                        "fields": [rand_field() for _f in range(rnd.choice([0, 0, 0, 1, 2]) if quirks else 0)]}
                terms.append(term)
            tp["required"] = terms
        for _o in range(rnd.choice([0, 0, 1, 1, 2, 3])):
            tp["tolerations"].append(rnd.choice(tol_pool))
        templates.append(tp)
    return nodes, templates


def all_distinct(pods, nodes, k_lanes: int = 1):
    """The throughput regime's scenes (bench.py scenarios.all_distinct_*, tools/tp_sweep.py): every pod asks for something nobody else asks
    for, on k_lanes of the four fixed lanes Filter compares (getLeftResource, core.go:436-475: cpu, memory, ephemeral storage, pods).
    k_lanes = 1 is rounds 3-5's scene (cpu + queue index: memory / ephemeral / pods of every request stay below the cluster's smallest
    left, so the Filter item's lane mask leaves ONE lane to compare).  For k_lanes > 1 the further lanes start just above the smallest left
    of that lane over the nodes Filter can evaluate and rise with the queue index (1 MiB per pod; 1 per pod modulo 64 on the pods lane):
    every tile of 64 requests then has to compare those lanes on every node, and no request exceeds the largest left (no tile fails outright)."""
    p = pods.copy()
    idx = np.arange(p.p, dtype=np.int64)
    p.req[0, :] += idx
    if k_lanes > 1:
        ok = (nodes.flags & (soa.NODE_NIL | soa.NODE_NO_NODE)) == 0
        left = (nodes.allocatable[:4] - nodes.requested[:4])[:, ok]
        step = {1: 1 << 20, 2: 1 << 20, 3: 1}
        for j in (1, 2, 3)[: k_lanes - 1]:
            lo, hi = int(left[j].min()), int(left[j].max())
            span = max(1, min(hi - lo - 1, step[j] * (p.p if j != 3 else 64)))
            p.req[j, :] = lo + 1 + (idx * step[j]) % span
    return p
