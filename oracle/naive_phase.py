"""Independent Python restatement of the PodGroup phase machine and of CreateMergePatch — TEST INFRASTRUCTURE ONLY (second opinion for
batch-scheduler_amd/host/bs_phase.cpp; nothing in the product imports it).

Written from /root/reference/pkg/scheduler/controller/controller.go:111-130,179-311, pkg/scheduler/core/core.go:279-281,325-360,
pkg/scheduler/batch/batchscheduler.go:258-285 and pkg/util/k8s.go:34-48 on Go-shaped objects (a status dict with the json tags of
pkg/apis/podgroup/v1/types.go:104-130; the Succeed / Failed maps of cache.go:52-67 as Python sets).  The merge patch follows
evanphx/json-patch v4.5.0 merge.go (getDiff / matchesValue), which is not vendored in the reference tree: recalled, and pinned on the two
expected strings of pkg/util/k8s_test.go:31-78."""
from __future__ import annotations

import copy
import json

H48 = 48 * 3600 * 10 ** 9
OPEN_FOR_RELEASE = ("PreScheduling", "Scheduling")


# ---------------------------------------------------------------------------------------------- merge patch
def _kind(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "bool"
    if isinstance(v, (int, float)):
        return "float64"
    if isinstance(v, str):
        return "string"
    if isinstance(v, list):
        return "array"
    return "map"


def _matches(a, b) -> bool:
    if _kind(a) != _kind(b):
        return False
    k = _kind(a)
    if k == "map":
        return all(_matches(a.get(x), b.get(x)) for x in set(a) | set(b))
    if k == "array":
        return len(a) == len(b) and all(_matches(x, y) for x, y in zip(a, b))
    if k == "float64":
        return float(a) == float(b)
    return a == b


def get_diff(a: dict, b: dict) -> dict:
    into = {}
    for key, bv in b.items():
        if key not in a:
            into[key] = bv
            continue
        av = a[key]
        if _kind(av) != _kind(bv):
            into[key] = bv
        elif _kind(av) == "map":
            d = get_diff(av, bv)
            if d:
                into[key] = d
        elif _kind(av) == "nil":
            pass
        elif not _matches(av, bv):
            into[key] = bv
    for key in a:
        if key not in b:
            into[key] = None
    return into


def _go_string(s: str) -> str:
    out = ['"']
    for ch in s:
        o = ord(ch)
        if ch in '"\\':
            out.append("\\" + ch)
        elif ch == "\n":
            out.append("\\n")
        elif ch == "\r":
            out.append("\\r")
        elif ch == "\t":
            out.append("\\t")
        elif o < 0x20 or ch in "<>&":
            out.append("\\u%04x" % o)
        elif o in (0x2028, 0x2029):
            out.append("\\u%04x" % o)
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def _go_number(v) -> str:
    f = float(v)
    if f == int(f) and abs(f) < 1e21:
        return ("-" if (f < 0 or (f == 0 and str(f).startswith("-"))) else "") + str(abs(int(f)))
    r = repr(f)
    if "e" in r:                                      # Python: 1e-07, 1.5e+22 — Go: 1e-7, 1.5e+22
        m, e = r.split("e")
        sign, digits = e[0], e[1:].lstrip("0") or "0"
        if abs(f) >= 1e21:
            return f"{m.rstrip('0').rstrip('.') if '.' in m else m}e+{digits.zfill(2)}"
        if abs(f) < 1e-6:
            return f"{m.rstrip('0').rstrip('.') if '.' in m else m}e-{digits if len(digits) > 1 else digits}"
        return format(f, "f").rstrip("0").rstrip(".")
    if abs(f) < 1e-6 and f != 0:
        mant = f"{f:e}"
        m, e = mant.split("e")
        return f"{m.rstrip('0').rstrip('.')}e-{e[1:].lstrip('0')}"
    return r


def go_marshal(v) -> str:
    k = _kind(v)
    if k == "nil":
        return "null"
    if k == "bool":
        return "true" if v else "false"
    if k == "float64":
        return _go_number(v)
    if k == "string":
        return _go_string(v)
    if k == "array":
        return "[" + ",".join(go_marshal(x) for x in v) + "]"
    return "{" + ",".join(_go_string(key) + ":" + go_marshal(v[key]) for key in sorted(v, key=lambda s: s.encode("utf-8"))) + "}"


def create_merge_patch(original: str, new: str) -> str:
    a, b = json.loads(original), json.loads(new)
    if not isinstance(a, dict) or not isinstance(b, dict):
        raise ValueError("not a JSON object")
    return go_marshal(get_diff(a, b))


# ---------------------------------------------------------------------------------------------- PodGroupStatus
def rfc3339(ns: int) -> str:
    import datetime
    return (datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=ns // 10 ** 9)).strftime("%Y-%m-%dT%H:%M:%SZ")


def status_doc(st: dict) -> dict:
    d = {"phase": st["phase"], "scheduled": st["scheduled"], "running": st["running"], "succeeded": st["succeeded"], "failed": st["failed"],
         "scheduleStartTime": rfc3339(st["scheduleStartTime"]) if st["scheduleStartTime"] else None}
    if st.get("occupiedBy"):
        d["occupiedBy"] = st["occupiedBy"]
    return d


def status_patch(frm: dict, to: dict) -> str:
    return create_merge_patch(json.dumps({"status": status_doc(frm)}), json.dumps({"status": status_doc(to)}))


def new_status(**kw) -> dict:
    st = {"phase": "", "scheduled": 0, "running": 0, "succeeded": 0, "failed": 0, "scheduleStartTime": 0, "occupiedBy": ""}
    st.update(kw)
    return st


def u32(x: int) -> int:
    return x & 0xFFFFFFFF


class Controller:
    """the cache entry's Succeed / Failed maps + syncHandler"""

    def __init__(self):
        self.succeed, self.failed = set(), set()

    def sync_handler(self, min_member: int, creation_ns: int, pg_status: dict, pods: list):
        """-> (status patched at :211-220 or None, pgCopy.Status at :293, actions: set of strings)"""
        pg = copy.deepcopy(pg_status)
        cp = copy.deepcopy(pg_status)
        actions, recovered = set(), None
        if cp["phase"] == "":
            cp["phase"] = "Pending"
        elif cp["phase"] == "Pending" and cp["scheduleStartTime"] != 0:
            actions.add("listed")
            cp["scheduled"] = u32(len(pods))
            if cp["scheduled"] > 0 and pg != cp:
                actions.add("patch_recover")
                pg = copy.deepcopy(cp)
                recovered = copy.deepcopy(cp)
        if cp["scheduled"] == min_member and cp["running"] == 0 and cp["scheduleStartTime"] != 0 and cp["scheduleStartTime"] - creation_ns > H48:
            actions.add("no_requeue")
            return recovered, cp, actions
        if cp["phase"] in ("Scheduled", "Running", "Scheduling"):
            actions.add("listed")
            not_pending = running = 0
            for uid, phase in pods:
                if phase == "Running":
                    running += 1
                elif phase == "Succeeded":
                    self.succeed.add(uid)
                elif phase == "Failed":
                    self.failed.add(uid)
                if phase != "Pending":
                    not_pending += 1
            cp["failed"], cp["succeeded"], cp["running"] = len(self.failed), len(self.succeed), running
            if not_pending > cp["scheduled"]:
                cp["scheduled"] = not_pending
            if not_pending < min_member and not_pending != 0:
                cp["scheduled"] = not_pending
                cp["phase"] = "Scheduling"
            if u32(cp["succeeded"] + cp["running"]) >= min_member:
                cp["phase"] = "Running"
            if cp["failed"] != 0 and u32(cp["failed"] + cp["running"] + cp["succeeded"]) >= min_member:
                cp["phase"] = "Failed"
            if cp["succeeded"] >= min_member:
                cp["phase"] = "Finished"
        if pg != cp:
            actions.add("patch")
            if cp["phase"] in ("Finished", "Failed"):
                actions.add("cache_delete")
        return recovered, cp, actions


def enqueue(min_member: int, creation_ns: int, st: dict) -> bool:
    if st["phase"] in ("Finished", "Failed"):
        return False
    if st["scheduled"] == min_member and st["running"] == 0 and st["scheduleStartTime"] != 0 and st["scheduleStartTime"] - creation_ns > H48:
        return False
    return True


def post_bind(min_member: int, st: dict, now_ns: int):
    cp = copy.deepcopy(st)
    cp["scheduled"] = u32(cp["scheduled"] + 1)
    if cp["scheduled"] >= min_member:
        cp["phase"] = "Scheduled"
    else:
        cp["phase"] = "Scheduling"
        if cp["scheduleStartTime"] == 0:
            cp["scheduleStartTime"] = now_ns
    return cp, cp["phase"] != st["phase"]


def start_gate(min_member: int, st: dict):
    open_ = st["phase"] in OPEN_FOR_RELEASE
    return open_, open_ and st["scheduled"] >= min_member
