"""Sequential, stateful restatement of the reference plugin loop (TEST INFRASTRUCTURE ONLY).

core.go ScheduleOperation with its go-cache TTL maps on a virtual clock, plus the in-memory part of
the batch release (batchscheduler.go:254-344) and PostBind (core.go:312-362).  Built on the object-level
arithmetic of naive_ref.py.  Used to check the C++ host mirror (batch-scheduler_amd/host/bs_host.cpp),
which runs the same calls with the node arithmetic on the GPU.
"""
from __future__ import annotations

import naive_ref as nv

SECOND = 1_000_000_000
PENDING, PRESCHEDULING, SCHEDULING, SCHEDULED = 0, 1, 2, 3


class TTL:
    """patrickmn/go-cache v2.1.0 semantics"""

    def __init__(self):
        self.items = {}

    def set(self, k, v, now, ttl):
        self.items[k] = (v, now + ttl if ttl > 0 else 0)

    def get(self, k, now):
        it = self.items.get(k)
        if it is None or (it[1] > 0 and now > it[1]):
            return None
        return it

    def add(self, k, v, now, ttl):
        if self.get(k, now) is not None:
            return False
        self.set(k, v, now, ttl)
        return True

    def delete(self, k):
        self.items.pop(k, None)

    def live(self, now):
        return {k: v for k, v in self.items.items() if not (v[1] > 0 and now > v[1])}


class SeqGroup(nv.PGS):
    def __init__(self, pod_group, max_schedule_time_s=None, creation_ts=0):
        super().__init__(pod_group)
        self.matched_pod_nodes = TTL()     # uid -> node
        self.pod_name_uids = TTL()         # pod name -> uid
        self.phase = PENDING
        self.max_schedule_time_s = max_schedule_time_s
        self.creation_ts = creation_ts


class SeqOperation:
    def __init__(self, nodes, cache, max_schedule_time_s=60.0):
        self.nodes, self.cache = nodes, cache     # cache: ordered dict name -> SeqGroup
        self.now = 0
        self.last_denied = TTL()
        self.last_permitted = TTL()
        self.max_schedule_time = int(max_schedule_time_s * SECOND)
        self.max_finished_pg, self.max_pg_status = "", None

    def set_time(self, seconds):
        self.now = int(round(seconds * SECOND))

    def _refresh(self):
        for pgs in self.cache.values():
            pgs.matched = len(pgs.matched_pod_nodes.live(self.now))

    def wait_time(self, pgs):
        if pgs.max_schedule_time_s is not None:
            return int(pgs.max_schedule_time_s * SECOND)
        return self.max_schedule_time

    def prefilter(self, pod):
        if pod.group is None:
            return nv.soa.PF_PASS_NOT_GROUPED, nv.soa.K_NOT_SCANNED
        if self.last_permitted.get(pod.uid, self.now) is not None:
            return nv.soa.PF_PASS_LAST_PERMITTED, nv.soa.K_NOT_SCANNED
        pgs = self.cache.get(pod.group)
        if pgs is None:
            return nv.soa.PF_ERR_PG_NOT_FOUND, nv.soa.K_NOT_SCANNED
        if self.last_denied.get(pod.group, self.now) is not None:
            return nv.soa.PF_ERR_DENIED, nv.soa.K_NOT_SCANNED
        inner = nv.ScheduleOperation(self.nodes, self.cache)
        if inner.fill_occupied_obj(pgs, pod) is not None:
            return nv.soa.PF_ERR_OCCUPIED, nv.soa.K_NOT_SCANNED
        self._refresh()
        try:
            name, mx, _ = nv.find_max_pg(self.cache)
        except nv.GoPanic:
            return nv.soa.PF_PANIC_DIV0, nv.soa.K_NOT_SCANNED
        self.max_finished_pg, self.max_pg_status = name, mx
        if name == "" or mx is None:
            return nv.soa.PF_PASS_NO_MAX, nv.soa.K_NOT_SCANNED
        matched = mx.matched
        if matched == 0:
            pre = nv.get_pre_allocated(pgs, matched)
            ok, k = nv.compare_cluster(self.nodes, pgs.pod, pre, 1.0)
            if not ok:
                self.last_denied.add(pod.group, "", self.now, 20 * SECOND)
                return nv.soa.PF_REJECT_FIRST, nv.soa.K_NONE
            return nv.soa.PF_PASS_FIRST_FITS, k
        if self.max_finished_pg == pod.group:
            return nv.soa.PF_PASS_IS_MAX, nv.soa.K_NOT_SCANNED
        pre = nv.get_pre_allocated(mx, matched)
        pre.Add(nv.pod_resource_require(pod).ResourceList())
        ok, k = nv.compare_cluster(self.nodes, mx.pod, pre, 0.7)
        if not ok:
            self.last_denied.add(pod.group, "", self.now, 20 * SECOND)
            return nv.soa.PF_REJECT_RESERVE, nv.soa.K_NONE
        return nv.soa.PF_PASS_RESERVE_FITS, k

    def filter(self, pod, node_idx):
        inner = nv.ScheduleOperation(self.nodes, self.cache)
        inner.max_finished_pg, inner.max_pg_status = self.max_finished_pg, self.max_pg_status
        fl, fn = inner.filter_node(pod, node_idx)
        if pod.group is not None and pod.group in self.cache and fl != nv.soa.FL_PANIC_NIL_MAX:
            err = fl >= 16 or (fl == nv.soa.FL_EVALUATED and fn >= 16)
            if err:
                self.last_denied.add(pod.group, "", self.now, 20 * SECOND)
            else:
                self.last_permitted.add(pod.uid, "", self.now, 2 * SECOND)
        return fl, fn

    def permit(self, pod, pod_name, node):
        """-> (ready, code) codes: 0 ready, 1 waiting, 2 not matched, 3 not found"""
        if pod.group is None:
            return True, 2
        pgs = self.cache.get(pod.group)
        if pgs is None:
            return False, 3
        if pgs.phase == PENDING:
            pgs.phase = PRESCHEDULING
        wait = self.wait_time(pgs)
        pgs.matched_pod_nodes.set(pod.uid, node, self.now, wait)
        old = pgs.pod_name_uids.get(pod_name, self.now)
        if old is not None:
            pgs.matched_pod_nodes.delete(old[0])
        pgs.pod_name_uids.set(pod_name, pod.uid, self.now, wait)
        have = len(pgs.matched_pod_nodes.live(self.now))
        if nv.u32(have) >= nv.u32(pgs.pod_group.min_member - pgs.pod_group.status_scheduled):
            pgs.scheduled = True
            return True, 0
        return False, 1

    def postbind(self, group):
        pgs = self.cache.get(group)
        if pgs is None:
            return
        pgs.pod_group.status_scheduled += 1
        pgs.phase = SCHEDULED if pgs.pod_group.status_scheduled >= pgs.pod_group.min_member else SCHEDULING

    def start_batch(self, group):
        pgs = self.cache.get(group)
        if pgs is None or pgs.phase not in (PRESCHEDULING, SCHEDULING):
            return []
        live = pgs.matched_pod_nodes.live(self.now)
        if nv.u32(len(live)) < nv.u32(pgs.pod_group.min_member - pgs.pod_group.status_scheduled):
            return []
        out = sorted((uid, v[0]) for uid, v in live.items())
        for uid, _ in out:
            pgs.matched_pod_nodes.delete(uid)
        return out

    def less(self, a, b):
        """a, b = (group name or None, priority, queue ts)"""
        (g1, p1, t1), (g2, p2, t2) = a, b
        if p1 > p2:
            return True
        if p1 == p2:
            if g1 is None and g2 is None:
                return t1 < t2
            if g1 is None:
                return True
            if g2 is None:
                return False
        pg1, pg2 = self.cache.get(g1), self.cache.get(g2)
        if pg1 is None or pg2 is None:
            return False
        if p1 == p2 and pg1.creation_ts < pg2.creation_ts:
            return True
        if p1 == p2 and pg1.creation_ts == pg2.creation_ts and g1 > g2:
            return True
        return p1 == p2 and pg1.creation_ts == pg2.creation_ts and g1 == g2 and t1 < t2
