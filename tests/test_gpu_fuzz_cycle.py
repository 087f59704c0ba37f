"""Random SEQUENCES of everything a scheduling cycle can do to one context — queue patches (bs_pods_apply), group patches
(bs_groups_apply), assumed pods (bs_nodes_assume), node list surgery (bs_nodes_apply), full reloads, batches in every mode (what-if / committing, Filter off / on / with its deny
entry, results copied out / written to pinned memory / read in place) and whole pod-by-pod passes (bs_seq_run) — against the oracle
stepping through the same sequence on host copies of the state.  The single-feature tests pin each entry point; this one pins
their INTERACTIONS: what a batch leaves behind for the next patch, what a pass leaves behind for the next batch, the leader
(sop.maxFinishedPG, core.go:58-59) carried from a committing batch or a pass into whatever comes next."""
import numpy as np
import pytest

from fullsize import ChurnStream
from test_gpu_parity import assert_batch_equal, load_ctx
import naive_ref as nv
from scenarios import random_objects
from test_gpu_parity import _force_class_mode
from test_gpu_queue import random_delta, scene
from test_gpu_seq import assert_groups_equal

pytestmark = pytest.mark.gpu
EXTRA = int(__import__("os").environ.get("BS_FUZZ_EXTRA", "0"))      # a longer one-off hunt: BS_FUZZ_EXTRA=3000 pytest tests/test_gpu_fuzz_cycle.py
NAMES = ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready")


def big_scene(seed, soa, steady):
    """several blocks of pods, several tiles of nodes, more groups than a wave: the hand-overs between blocks are in play"""
    rng = np.random.default_rng(seed)
    sc = random_objects(seed, n_nodes=200 + seed % 150, n_groups=70, n_pods=900, n_scalars=seed % 3, n_classes=3)
    nodes, fit, groups, pods, _ = nv.to_soa(sc["nodes"], sc["cache"], sc["pods"], sc["names"], sc["n_classes"], denied=sc["denied"], permitted=sc["permitted"])
    if steady:
        _force_class_mode(groups, rng, 3)
        groups.matched[:] = rng.integers(1, 4, groups.g)
    return rng, nodes, fit, groups, pods


def run_sequence(seed, steady, bsa, soa, orc, rounds=16, big=False):
    rng, nodes, fit, groups, pods = big_scene(seed, soa, steady) if big else scene(seed, bsa, soa, steady)
    nodes, groups, cur = nodes.copy(), groups.copy(), pods
    leader = -1                                                # sop.maxFinishedPG as the context carries it
    log = []
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd in range(rounds):
            op = rng.choice(["batch", "batch", "batch", "batch", "pods", "pods", "groups", "groups", "assume", "nodes", "reload", "seq"]) if rnd else "batch"
            where = f"seed {seed} steady {steady} round {rnd} {op} after {log}"
            log.append(op)
            if op == "pods":
                d = random_delta(rng, cur, soa, novel_base=40 * rnd)
                ctx.apply_pods(**d)
                cur = cur.patched(**d)
                assert ctx.read_pods().equal(cur), where
            elif op == "groups":
                deltas = []
                for i in rng.choice(groups.g, min(4, groups.g), replace=False):
                    groups.matched[i] = rng.integers(0, groups.min_member[i] + 2)
                    groups.status_scheduled[i] = rng.integers(0, 3)
                    groups.flags[i] = (groups.flags[i] & 0x6) | int(rng.integers(0, 2)) | (8 * int(rng.integers(0, 2))) | (soa.GROUP_PHASE_CLOSED if rng.random() < 0.15 else 0)
                    deltas.append((i, groups.matched[i], groups.status_scheduled[i], groups.flags[i]))
                ctx.apply_group_deltas(deltas)
            elif op == "assume":
                reqs = []
                for n in rng.choice(nodes.n, min(5, nodes.n), replace=False):
                    nodes.requested[:3, n] += rng.integers(0, 1 + np.maximum(nodes.allocatable[:3, n], 0) // 8)
                    nodes.requested[3, n] += 1
                    if nodes.lanes > 4 and rng.random() < 0.5:
                        nodes.requested[4, n] += 1
                        nodes.requested_present[n] |= 1
                    reqs.append((int(n), nodes.requested[:, n].tolist(), int(nodes.requested_present[n])))
                ctx.assume_nodes(reqs)
            elif op == "nodes":                                # list surgery: requested-update / append / stable remove (BASELINE config 5's events)
                stream = ChurnStream(nodes, fit, seed * 100 + rnd)
                ctx.apply_node_deltas(stream.next_deltas(int(rng.integers(1, 4))))
                nodes, fit = stream.current()
            elif op == "reload":                               # a full reload of one input in mid-sequence
                what = rng.choice(["pods", "groups", "nodes"])
                if what == "pods":
                    ctx.load_pods(cur)
                elif what == "groups":
                    ctx.load_groups(groups)
                else:
                    ctx.load_nodes(nodes, fit)
            elif op == "seq":
                st = soa.STAGE_PREFILTER | (soa.STAGE_FILTER if rng.random() < 0.5 else 0)
                if (st & soa.STAGE_FILTER) and rng.random() < 0.5:
                    st |= soa.BATCH_FILTER_DENY                # Filter's TTL writes inside the pass (core.go:183-188)
                s = orc.seq_replay(nodes, fit, groups, cur, st, leader=leader)
                r = ctx.seq_run(st)
                for name in ("pf_code", "pf_first_k", "pf_leader", "pod_node") + (("last_permitted",) if st & soa.BATCH_FILTER_DENY else ()):
                    assert np.array_equal(r[name], s[name]), f"{where}: {name}"
                assert r["released_group"].tolist() == s["released_group"].tolist() and r["released_pods"].tolist() == s["released_pods"].tolist(), where
                nodes, groups, leader = s["nodes"], s["groups"], s["leader"]
                req, pres = ctx.read_node_requests()
                assert np.array_equal(req, nodes.requested) and np.array_equal(pres, nodes.requested_present), f"{where}: node requests after the pass"
                assert_groups_equal(ctx.read_groups(), groups, soa, where)
            else:
                st = soa.STAGE_ALL if rng.random() < 0.75 else (soa.STAGE_PREFILTER | soa.STAGE_TALLY)
                if (st & soa.STAGE_FILTER) and rng.random() < 0.4:
                    st |= soa.BATCH_FILTER_DENY
                commit = rng.random() < 0.3
                host = not commit and rng.random() < 0.5
                sop = orc.Sop(orc.Snapshot(nodes, fit), groups).carry(leader)
                exp = sop.batch(cur, st, bitmap=False)
                ctx.run(st | (soa.BATCH_COMMIT if commit else 0) | (soa.BATCH_HOST_RESULTS if host else 0))
                if not commit and rng.random() < 0.2:          # a patch BETWEEN run and read: the batch is settled first, against the state it ran on
                    i = int(rng.integers(0, groups.g))
                    groups.matched[i] = rng.integers(0, groups.min_member[i] + 2)
                    ctx.apply_group_deltas([(i, groups.matched[i], groups.status_scheduled[i], groups.flags[i])])
                view = None
                if host and rng.random() < 0.6:
                    try:
                        view = ctx.map_results()
                    except bsa.BsError as e:                   # general chain / a re-run batch: no host results to map
                        assert e.status == -4, where
                if view is not None:
                    for name in NAMES:
                        assert np.array_equal(view[name], getattr(exp, name)), f"{where}: {name} (mapped)"
                else:
                    assert_batch_equal(ctx.read(bitmap=False, rows=False), exp, where, bitmap=False)
                if commit:
                    groups, leader = sop.groups, sop.leader
                    assert_groups_equal(ctx.read_groups(), groups, soa, where + " (committed state)")
    return log


@pytest.mark.parametrize("steady", [True, False], ids=["steady", "positional"])
@pytest.mark.parametrize("seed", range(9500, 9600 + EXTRA))
def test_random_cycle_sequences(seed, steady, bsa, soa, orc):
    run_sequence(seed, steady, bsa, soa, orc, rounds=20)


@pytest.mark.parametrize("steady", [True, False], ids=["steady", "positional"])
@pytest.mark.parametrize("seed", range(9700 + 100000, 9720 + 100000 + EXTRA // 10))
def test_random_cycle_sequences_on_larger_scenes(seed, steady, bsa, soa, orc):
    run_sequence(seed, steady, bsa, soa, orc, rounds=12, big=True)


def test_two_contexts_on_two_threads(bsa, soa, orc):
    """include/bsched.h: calls on ONE context are serialised by the caller — two contexts are independent.  Two threads, each driving
    its own context through batches, queue patches and passes at the same time (ctypes drops the GIL inside every call), must each see
    exactly what a lone context sees: nothing in the library is shared between contexts."""
    import threading
    plans = []
    for seed in (9801, 9802):
        rng, nodes, fit, groups, pods = big_scene(seed, soa, steady=(seed % 2 == 1))
        steps, cur = [], pods
        for rnd in range(10):
            d = random_delta(rng, cur, soa, novel_base=30 * rnd)
            cur = cur.patched(**d)
            exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(cur, soa.STAGE_ALL, bitmap=False)
            seq = orc.seq_replay(nodes, fit, groups, cur, soa.STAGE_PREFILTER) if rnd == 9 else None
            steps.append((d, exp, seq))
        plans.append((nodes, fit, groups, pods, steps))
    errors = []

    def drive(plan, tag):
        try:
            nodes, fit, groups, pods, steps = plan
            with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
                for rnd, (d, exp, seq) in enumerate(steps):
                    ctx.apply_pods(**d)
                    ctx.run(soa.STAGE_ALL | (soa.BATCH_HOST_RESULTS if rnd % 2 else 0))
                    assert_batch_equal(ctx.read(bitmap=False, rows=False), exp, f"thread {tag} round {rnd}", bitmap=False)
                    if seq is not None:
                        r = ctx.seq_run(soa.STAGE_PREFILTER)
                        for name in ("pf_code", "pf_first_k", "pf_leader", "pod_node"):
                            assert np.array_equal(r[name], seq[name]), f"thread {tag}: pass: {name}"
        except BaseException as e:                           # noqa: BLE001 — handed to the main thread
            errors.append((tag, e))

    threads = [threading.Thread(target=drive, args=(p, k)) for k, p in enumerate(plans)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    if errors:
        raise errors[0][1]
