"""Full-size parity cases (every size BASELINE.json lists) shared by the golden generator and the GPU tests.

The oracle finishes a cfg3 batch in seconds but needs ~2 minutes for a cfg4 batch with Filter and ~100 s for the
10 000-event churn stream, so its outputs on the seeded synthetic inputs are committed as digests
(tests/golden/fullsize_digests.json, written by tests/golden/make_fullsize_golden.py from the oracle alone) and the
GPU results must reproduce them byte for byte; one case per family is also re-computed live on the GPU box.
"""
import hashlib
import importlib

import numpy as np

ARRAYS = ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready")

# (config, scenario, seed): full batches, all stages.  cfg3: BASELINE configs[2]; cfg4: configs[3].
BATCH_CASES = [("cfg3", "cold", 1), ("cfg3", "warm", 2), ("cfg3", "busy", 3), ("cfg3", "tail", 20260921),
               ("cfg4", "cold", 5), ("cfg4", "warm", 6), ("cfg4", "tail", 4)]
# BASELINE configs[4]: 10k pods / 5k nodes, 10 000 node events, a re-score every 100
CHURN = dict(config="cfg3", scenario="tail", seed=7, rounds=100, events=100)


# the pending-queue counterpart: 10 000 pod events (40 % stable remove, 30 % append, 10 % insert in mid-queue, 20 % flag flip),
# a re-score every 100 through bs_pods_apply — the queue itself is never re-uploaded
POD_CHURN = dict(config="cfg3", scenario="tail", seed=9, rounds=100, events=100)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def digest(out, bitmap: bool) -> dict:
    d = {name: sha(getattr(out, name)) for name in ARRAYS}
    if bitmap:
        d["fl_bitmap"] = sha(out.fl_bitmap)
    d["histogram"] = {str(i): int(c) for i, c in enumerate(np.bincount(out.pf_code, minlength=64)) if c}
    d["groups_ready"] = int(out.group_ready.sum())
    return d


def case_key(config, scenario, seed) -> str:
    return f"{config}/{scenario}/{seed}"


class ChurnStream:
    """The node event stream of BASELINE config 5: 40 % requested-update, 30 % append, 30 % stable remove, seeded.
    Keeps the host mirror of the node list (what a full reload would upload) next to the deltas it hands out."""

    def __init__(self, nodes, fit, seed: int):
        self.capi = importlib.import_module("batch-scheduler_amd.capi")
        self.soa = importlib.import_module("batch-scheduler_amd.soa")
        self.rng = np.random.default_rng(seed + 11)
        self.L = nodes.lanes
        self.alloc, self.req = nodes.allocatable.copy(), nodes.requested.copy()
        self.ap, self.rp, self.fl = nodes.allocatable_present.copy(), nodes.requested_present.copy(), nodes.flags.copy()
        self.fitb = fit.to_bool()

    def next_deltas(self, events: int) -> list:
        capi, rng = self.capi, self.rng
        deltas = []
        for _ in range(events):
            kind = int(rng.choice([capi.DELTA_UPDATE, capi.DELTA_APPEND, capi.DELTA_REMOVE], p=[0.4, 0.3, 0.3]))
            n = self.alloc.shape[1]
            d = capi.NodeDelta()
            d.kind = kind
            if kind == capi.DELTA_REMOVE:
                idx = int(rng.integers(0, n))
                d.index = idx
                self.alloc, self.req = np.delete(self.alloc, idx, 1), np.delete(self.req, idx, 1)
                self.ap, self.rp, self.fl = np.delete(self.ap, idx), np.delete(self.rp, idx), np.delete(self.fl, idx)
                self.fitb = np.delete(self.fitb, idx, 1)
            else:
                src = int(rng.integers(0, n))
                col_a, col_r = self.alloc[:, src].copy(), self.req[:, src].copy()
                col_r[0] = int(col_a[0] * rng.random())
                col_r[1] = int(col_a[1] * rng.random())
                a_p, r_p = int(self.ap[src]), int(self.rp[src])
                for j in range(self.L):
                    d.allocatable[j], d.requested[j] = int(col_a[j]), int(col_r[j])
                d.allocatable_present, d.requested_present = a_p, r_p
                d.fit_default, d.n_fit_exceptions = 1, 1
                exc = int(rng.integers(0, self.fitb.shape[0]))
                d.fit_exceptions[0] = exc
                fcol = np.ones(self.fitb.shape[0], bool)
                fcol[exc] = False
                if kind == capi.DELTA_UPDATE:
                    idx = int(rng.integers(0, n))
                    d.index = idx
                    self.alloc[:, idx], self.req[:, idx], self.ap[idx], self.rp[idx], self.fl[idx] = col_a, col_r, a_p, r_p, 0
                    self.fitb[:, idx] = fcol
                else:
                    self.alloc, self.req = np.concatenate([self.alloc, col_a[:, None]], 1), np.concatenate([self.req, col_r[:, None]], 1)
                    self.ap, self.rp = np.append(self.ap, a_p).astype(np.uint32), np.append(self.rp, r_p).astype(np.uint32)
                    self.fl = np.append(self.fl, 0).astype(np.uint8)
                    self.fitb = np.concatenate([self.fitb, fcol[:, None]], 1)
            deltas.append(d)
        return deltas

    def current(self):
        return self.soa.Nodes(self.alloc, self.req, self.ap, self.rp, self.fl), self.soa.FitMasks.from_bool(self.fitb)


class PodChurnStream:
    """The pod event stream of POD_CHURN, seeded.  Every call draws one delta of `events` events against the current queue
    (removals, flag flips, appended pods, pods inserted in mid-queue; one in ten new pods asks for something no pod asked
    before: a new request class) and keeps the host mirror of the queue — what a full reload would upload."""

    def __init__(self, pods, seed: int):
        self.soa = importlib.import_module("batch-scheduler_amd.soa")
        self.rng = np.random.default_rng(seed + 23)
        self.pods = pods.copy()
        self.novel = 0

    def next_delta(self, events: int) -> dict:
        rng, soa, cur = self.rng, self.soa, self.pods
        kinds = rng.choice(4, events, p=[0.4, 0.3, 0.1, 0.2])
        n_rem = min(int((kinds == 0).sum()), cur.p)
        n_app, n_mid, n_flag = int((kinds == 1).sum()), int((kinds == 2).sum()), int((kinds == 3).sum())
        remove = np.sort(rng.choice(cur.p, n_rem, replace=False)).astype(np.uint32)
        stay = np.setdiff1d(np.arange(cur.p, dtype=np.uint32), remove)
        flag_index = np.sort(rng.choice(stay, min(n_flag, len(stay)), replace=False)).astype(np.uint32)
        flag_value = (cur.flags[flag_index] ^ np.uint8(soa.POD_LAST_PERMITTED)).astype(np.uint8)
        ni = n_app + n_mid
        src = rng.integers(0, cur.p, ni)
        ins = cur.take(src)                                                # new pods of existing templates, in existing gangs
        ins.flags[:] = 0
        for k in np.nonzero(rng.random(ni) < 0.1)[0]:
            self.novel += 1
            ins.req[0, k] = 100 + 7 * self.novel                         # nobody asked for this before: a new request class
        pn = cur.p - n_rem + ni
        kept = cur.p - n_rem
        mid = np.sort(rng.choice(max(kept + n_mid, 1), n_mid, replace=False)) if n_mid else np.zeros(0, np.int64)   # positions among kept + mid pods
        insert_at = np.concatenate([mid, np.arange(pn - n_app, pn)]).astype(np.uint32)
        delta = dict(remove=remove, flag_index=flag_index, flag_value=flag_value, insert=ins, insert_at=insert_at)
        self.pods = cur.patched(**delta)
        assert self.pods.p == pn
        return delta
