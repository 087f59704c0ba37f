"""Speculative launch of the steady-state chain (csrc/bsched.hip, bs_ctx 'speculation'; bs_speculation_stats).

After a group patch findMaxPG runs again on the device and bs_batch_run would have to WAIT for its answer (the running-sum table the
batch uses) before launching anything.  In a steady state it launches on the previous cycle's answer instead and checks the guess
when the results are first asked for; a wrong guess re-runs the batch there.  Results must equal the oracle's either way."""
import numpy as np
import pytest

from test_gpu_parity import load_ctx
from test_gpu_queue import random_delta

pytestmark = pytest.mark.gpu
NAMES = ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready")


def expect(orc, soa, nodes, fit, groups, pods):
    return orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL, bitmap=False)


def same(got, exp, what):
    for name in NAMES:
        a, b = (got[name] if isinstance(got, dict) else getattr(got, name)), getattr(exp, name)
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{what}: {name} differs at {bad[:8].tolist()} ({len(bad)} total): got {a[bad[:8]].tolist()} exp {b[bad[:8]].tolist()}")


def leader_of(orc, groups):
    return int(orc.find_max_pg(groups)[0])


@pytest.mark.parametrize("mode", ["read", "map", "sync"])
def test_cycles_with_right_and_wrong_guesses(mode, bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    rng = np.random.default_rng(3)
    cur, g = pods, groups.copy()
    with load_ctx(bsa, nodes, fit, g, pods) as ctx:
        same(ctx.batch(soa.STAGE_ALL, bitmap=False), expect(orc, soa, nodes, fit, g, cur), "first batch")
        lead0 = leader_of(orc, g)
        wrong = 0
        for it in range(40):
            # the counters of a few groups move (Permit / PostBind) ...
            idx = rng.choice(g.g, 8, replace=False)
            if it % 5 == 4:
                # ... and every fifth cycle ANOTHER group, of another fit class, takes the lead: the old leader is let through
                # (its latch is set, findMaxPG skips it) and a group of the next class gets all its pods but one matched
                others = np.nonzero((g.cls != g.cls[lead0]) & ((g.flags & soa.GROUP_SCHEDULED_LATCH) == 0) & (g.min_member > g.status_scheduled + 1))[0]
                cand = int(others[it % len(others)])
                g.flags[lead0] |= soa.GROUP_SCHEDULED_LATCH
                g.matched[cand] = g.min_member[cand] - g.status_scheduled[cand] - 1
                idx = np.unique(np.append(idx, [cand, lead0]))
            else:
                sel = idx[idx != lead0]
                g.matched[sel] = np.minimum(g.matched[sel], 1)
            d = random_delta(rng, cur, soa, max_events=6, novel_base=1000 * (it + 1))
            cur = cur.patched(**d)
            exp = expect(orc, soa, nodes, fit, g, cur)
            lead = leader_of(orc, g)
            wrong += int(lead != lead0 and (lead < 0 or lead0 < 0 or g.cls[lead] != g.cls[lead0] or (g.matched[lead] == 0) != (g.matched[lead0] == 0)))
            lead0 = lead
            deltas = [(int(i), int(g.matched[i]), int(g.status_scheduled[i]), int(g.flags[i])) for i in idx]
            # the cycle, back to back (what a shim does): group patch, queue patch, batch
            ctx.apply_group_deltas(deltas)
            ctx.apply_pods(**d)
            if mode == "map":
                ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
                same(ctx.map_results(), exp, f"cycle {it}")
            elif mode == "read":
                ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
                same(ctx.read(bitmap=False, rows=False), exp, f"cycle {it}")
            else:
                ctx.run(soa.STAGE_ALL)
                ctx.sync()
                same(ctx.read(bitmap=False, rows=False), exp, f"cycle {it}")
        launched, missed = ctx.speculation_stats()
        print(f"speculation [{mode}]: {launched} of 40 batches launched on a guess, {missed} wrong guesses re-run; {wrong} cycles changed the table")
        # (whether the answer has landed by the time the batch is launched is a race the host usually loses: that is the point)
        assert launched >= 1 and missed <= launched and missed <= wrong, (launched, missed, wrong)
        assert wrong > 0, "the scene has to contain leader changes across fit classes"


@pytest.mark.parametrize("speculate", [True, False], ids=["speculating", "BS_NO_SPECULATE"])
def test_latency_mode_results_are_complete_when_the_word_arrives(speculate, bsa, soa, orc, monkeypatch):
    """bs_batch_read in latency mode copies the results out of the pinned memory the moment the completion word is there.  Round 3
    published the word behind a bare s_waitcnt: another XCD's mirror writes could still be under way, and a read that followed at
    once returned the PREVIOUS cycle's values for a few pods about one run in three (final_tail, csrc/bs_fast.hpp: now a
    system-scope release fence per block).  120 cycles, every output compared at once."""
    if not speculate:
        monkeypatch.setenv("BS_NO_SPECULATE", "1")
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    rng = np.random.default_rng(11)
    cur = pods
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.batch(soa.STAGE_ALL, bitmap=False)
        for it in range(120):
            d = random_delta(rng, cur, soa, max_events=8, novel_base=1000 * (it + 1))
            cur = cur.patched(**d)
            exp = expect(orc, soa, nodes, fit, groups, cur)
            ctx.apply_group_deltas([(3, int(groups.matched[3]), int(groups.status_scheduled[3]), int(groups.flags[3]))])
            ctx.apply_pods(**d)
            ctx.run(soa.STAGE_ALL | soa.BATCH_HOST_RESULTS)
            got = ctx.read(bitmap=False, rows=False)
            same(got, exp, f"cycle {it}")


def test_speculation_off(bsa, soa, orc, monkeypatch):
    monkeypatch.setenv("BS_NO_SPECULATE", "1")
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for it in range(5):
            ctx.apply_group_deltas([(3, int(groups.matched[3]), int(groups.status_scheduled[3]), int(groups.flags[3]))])
            same(ctx.batch(soa.STAGE_ALL, bitmap=False), expect(orc, soa, nodes, fit, groups, pods), f"cycle {it}")
        assert ctx.speculation_stats() == (0, 0)


def test_committing_and_positional_batches_never_guess(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")        # groups without their pod: captures possible
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.apply_group_deltas([(3, int(groups.matched[3]), int(groups.status_scheduled[3]), int(groups.flags[3]))])
        same(ctx.batch(soa.STAGE_ALL, bitmap=False), expect(orc, soa, nodes, fit, groups, pods), "cold")
        assert ctx.speculation_stats()[0] == 0
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.batch(soa.STAGE_ALL, bitmap=False)
        ctx.apply_group_deltas([(3, int(groups.matched[3]), int(groups.status_scheduled[3]), int(groups.flags[3]))])
        exp = sop.batch(pods, soa.STAGE_ALL, bitmap=False)
        same(ctx.batch(soa.STAGE_ALL | soa.BATCH_COMMIT, bitmap=False), exp, "commit")
        assert ctx.speculation_stats()[0] == 0
        assert ctx.read_groups().state_equal(sop.groups)
