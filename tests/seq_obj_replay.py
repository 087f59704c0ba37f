"""Object-level sequential pass (TEST INFRASTRUCTURE): oracle/naive_seq.SeqOperation — the reference's ScheduleOperation with real
go-cache TTL maps, Permit (core.go:268-309), StartBatchSchedule (batchscheduler.go:254-344) and PostBind (core.go:312-362) on Go-shaped
objects — driven pod by pod through the loop oracle/bs_oracle_seq.c states (PreFilter -> [Filter on every node] -> first-fit node
choice -> assume -> Permit -> release).  It is the second, independent statement the C pass is pinned against
(tests/test_seq_oracle_pin.py): groups enter with waiting pods of EARLIER cycles (synthetic uids in MatchedPodNodes), so a release
has to allow, delete and PostBind every entry, not only the pods of this pass."""
import copy

import naive_ref as nv
import naive_seq as ns

CLOSED_PHASE = 4          # any phase that is none of Pending / PreScheduling / Scheduling (Scheduled = 3, Running, Failed, ...)


def build_operation(sc, closed=()):
    """SeqOperation over deep copies of the scene's objects.  sc["cache"][name].matched waiting pods of earlier cycles become
    MatchedPodNodes entries with synthetic uids; `closed`: names of groups whose phase lets StartBatchSchedule release nobody."""
    nodes = copy.deepcopy(sc["nodes"])
    cache = {}
    for nm, pgs in sc["cache"].items():
        g = ns.SeqGroup(copy.deepcopy(pgs.pod_group))
        g.pod = copy.deepcopy(pgs.pod)
        g.scheduled = pgs.scheduled
        g.phase = CLOSED_PHASE if nm in closed else (ns.PRESCHEDULING if pgs.matched else ns.PENDING)
        for k in range(pgs.matched):
            g.matched_pod_nodes.set(f"waiting/{nm}/{k}", -1, 0, 60 * ns.SECOND)
            g.pod_name_uids.set(f"waiting/{nm}/{k}", f"waiting/{nm}/{k}", 0, 60 * ns.SECOND)
        g.matched = pgs.matched
        cache[nm] = g
    op = ns.SeqOperation(nodes, cache)
    for nm in sc["denied"]:
        op.last_denied.add(nm, "", 0, 20 * ns.SECOND)
    for uid in sc["permitted"]:
        op.last_permitted.add(uid, "", 0, 2 * ns.SECOND)
    return op


def _lane_req(info):
    return info.requested.AllowedPodNumber or info.pod_count        # core.go:650-653


def holds(info, req: nv.Resource):
    """the stated node-choice rule (bs_oracle_seq.c `holds`): cpu / mem / eph bind when asked for; pods lane: requested + 1 <= allocatable;
    a requested scalar needs the allocatable key"""
    a, r = info.allocatable, info.requested
    for want, have, used in ((req.MilliCPU, a.MilliCPU, r.MilliCPU), (req.Memory, a.Memory, r.Memory), (req.EphemeralStorage, a.EphemeralStorage, r.EphemeralStorage)):
        if want > 0 and want > nv.wrap64(have - used):
            return False
    if _lane_req(info) + 1 > a.AllowedPodNumber:
        return False
    for name, want in (req.ScalarResources or {}).items():
        if want <= 0:
            continue
        if name not in (a.ScalarResources or {}):
            return False
        if want > a.ScalarResources[name] - (r.ScalarResources or {}).get(name, 0):
            return False
    return True


def assume(info, req: nv.Resource):
    r = info.requested
    r.MilliCPU += req.MilliCPU
    r.Memory += req.Memory
    r.EphemeralStorage += req.EphemeralStorage
    if r.AllowedPodNumber:
        r.AllowedPodNumber += 1
    else:
        info.pod_count += 1
    for name, want in (req.ScalarResources or {}).items():
        if r.ScalarResources is None:
            r.ScalarResources = {}
        r.ScalarResources[name] = r.ScalarResources.get(name, 0) + want


def replay(sc, closed=(), run_filter=False, filter_deny=False, scalar_names=()):
    """-> dict(pf_code, pf_first_k, pf_leader, pod_node, released_group, released_pods, matched, status_scheduled, latch, closed, denied)
    run_filter: the plugin's Filter gates the node choice (a what-if: no TTL writes).  filter_deny: Filter is CALLED on every node as the
    framework would with every node offered (core.go:170-191): a failing call deny-lists the group (:183-185), a passing one enters
    lastPermittedPod (:188)."""
    op = build_operation(sc, closed)
    gnames = list(op.cache.keys())
    uid_index = {pod.uid: i for i, pod in enumerate(sc["pods"])}
    P = len(sc["pods"])
    out = dict(pf_code=[], pf_first_k=[], pf_leader=[], pod_node=[-1] * P, released_group=[], released_pods=[], last_permitted=[0] * P)
    slot_of = {}
    for i, pod in enumerate(sc["pods"]):
        code, fk = op.prefilter(pod)
        out["pf_code"].append(code)
        out["pf_first_k"].append(fk)
        out["pf_leader"].append(gnames.index(op.max_finished_pg) if op.max_pg_status is not None else -1)
        if code >= 16:
            continue
        if pod.group is not None and pod.group not in op.cache:
            assert op.permit(pod, pod.uid, 0) == (False, 3)     # core.go:275-278 -> Unschedulable: the framework forgets the assumed pod
            continue
        req = nv.pod_resource_require(pod, True)
        req.ScalarResources = {k: v for k, v in (req.ScalarResources or {}).items() if k in scalar_names} or None
        passes = None
        if run_filter:
            passes = []
            for k in range(len(op.nodes)):
                if filter_deny:
                    fl, fn = op.filter(pod, k)
                else:
                    inner = nv.ScheduleOperation(op.nodes, op.cache)
                    inner.max_finished_pg, inner.max_pg_status = op.max_finished_pg, op.max_pg_status
                    fl, fn = inner.filter_node(pod, k)
                passes.append(fl < 16 and (fl != nv.soa.FL_EVALUATED or fn < 16))
            if filter_deny and pod.group is not None:
                out["last_permitted"][i] = int(op.last_permitted.get(pod.uid, op.now) is not None and any(passes))
        at = -1
        for k, info in enumerate(op.nodes):
            if info.nil or not info.has_node or info.unschedulable or info.taint_err or not nv.check_fit(pod, info):
                continue
            if passes is not None and not passes[k]:
                continue
            if holds(info, req):
                at = k
                break
        if at < 0:
            continue
        assume(op.nodes[at], req)
        if pod.group is None:
            assert op.permit(pod, pod.uid, at) == (True, 2)     # core.go:269-272
            out["pod_node"][i] = at
            continue
        ready, pcode = op.permit(pod, pod.uid, at)
        if not ready:
            continue
        allowed = op.start_batch(pod.group)
        if not allowed:
            continue
        for uid, node in allowed:
            op.postbind(pod.group)
            if uid in uid_index:
                out["pod_node"][uid_index[uid]] = node
        g = gnames.index(pod.group)
        if g in slot_of:
            out["released_pods"][slot_of[g]] += len(allowed)
        else:
            slot_of[g] = len(out["released_group"])
            out["released_group"].append(g)
            out["released_pods"].append(len(allowed))
    op._refresh()
    out["matched"] = [op.cache[nm].matched for nm in gnames]
    out["status_scheduled"] = [nv.u32(op.cache[nm].pod_group.status_scheduled) for nm in gnames]
    out["latch"] = [bool(op.cache[nm].scheduled) for nm in gnames]
    out["closed"] = [op.cache[nm].phase not in (ns.PENDING, ns.PRESCHEDULING, ns.SCHEDULING) for nm in gnames]
    out["denied"] = [op.last_denied.get(nm, op.now) is not None for nm in gnames]
    out["op"] = op
    return out
