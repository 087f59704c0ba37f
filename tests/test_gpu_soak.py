"""Soak across the two wrapping counters of a context (csrc/bsched.hip, bs_ctx): the 16-bit slot stamp (1 + stamp_ctr, comes
round every 65 535 batches — 3 s of a live scheduler) and the key sequence of the 64-bit atomicMin keys (~key_seq in the high
word, runs out after 2^32 - 2 batches).  Batches before, at and after each crossing have to equal the oracle's, on the
steady-state chain and on the positional chain, with bs_pods_apply churn in between so that slots, classes and pairs go stale."""
import numpy as np
import pytest

from test_gpu_parity import load_ctx
from test_gpu_queue import random_delta

pytestmark = pytest.mark.gpu
NAMES = ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready")


def check(ctx, orc, soa, nodes, fit, groups, cur, where):
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(cur, soa.STAGE_ALL, bitmap=False)
    got = ctx.read(bitmap=False, rows=False)
    for name in NAMES:
        assert np.array_equal(getattr(got, name), getattr(exp, name)), f"{where}: {name}"


def scene(bsa, scenario):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", scenario)
    return nodes, fit, groups, pods


@pytest.mark.parametrize("scenario,chain", [("tail", 1), ("cold", 2)], ids=["steady", "positional"])
def test_70000_batches_across_the_stamp_wrap(scenario, chain, bsa, soa, orc):
    """a real soak: >= 70 000 back-to-back batches, queue churn every 500 batches; compared with the oracle at the start, around the
    wrap of the slot stamp (batch 65 534 of the context) and at the end"""
    nodes, fit, groups, pods = scene(bsa, scenario)
    rng = np.random.default_rng(5)
    cur = pods
    checkpoints = {0, 1, 2, 499, 500, 501, 30000, 65530, 65531, 65532, 65533, 65534, 65535, 65536, 65537, 65540, 66000, 69999}
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert ctx.stats(soa.STAGE_ALL)["chain"] == chain        # (one instrumented batch: the chain this scene takes)
        for it in range(70000):
            if it and it % 500 == 0:
                d = random_delta(rng, cur, soa, max_events=10, novel_base=it)
                ctx.apply_pods(**d)
                cur = cur.patched(**d)
            ctx.run(soa.STAGE_ALL)
            if it in checkpoints:
                check(ctx, orc, soa, nodes, fit, groups, cur, f"batch {it}")
        ctx.sync()


@pytest.mark.parametrize("scenario", ["tail", "cold", "warm"])
@pytest.mark.parametrize("hook,start", [("BS_STAMP_START", 65520), ("BS_KEYSEQ_START", 0xFFFFFFF0)], ids=["stamp", "key-seq"])
def test_every_batch_across_a_wrap_with_churn(scenario, hook, start, bsa, soa, orc, monkeypatch):
    """the counters started just below their wrap points (test hooks read at bs_create): 40 batches, each behind a random queue
    delta, each compared with the oracle — the crossing falls in the middle"""
    monkeypatch.setenv(hook, str(start))
    nodes, fit, groups, pods = scene(bsa, scenario)
    rng = np.random.default_rng(9)
    cur = pods
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for it in range(40):
            if it:
                d = random_delta(rng, cur, soa, max_events=8, novel_base=100 * it)
                ctx.apply_pods(**d)
                cur = cur.patched(**d)
            ctx.run(soa.STAGE_ALL)
            check(ctx, orc, soa, nodes, fit, groups, cur, f"{hook} batch {it}")


def test_commit_batches_across_the_key_wrap(bsa, soa, orc, monkeypatch):
    """BS_BATCH_COMMIT persists the deny entries the keyed minima find: across the re-key the committed group state stays the oracle's"""
    monkeypatch.setenv("BS_KEYSEQ_START", str(0xFFFFFFFA))
    nodes, fit, groups, pods = scene(bsa, "cold")
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for it in range(10):
            exp = sop.batch(pods, soa.STAGE_ALL, bitmap=False)            # the oracle's Sop mutates its groups: the committed state
            ctx.run(soa.STAGE_ALL | soa.BATCH_COMMIT)
            got = ctx.read(bitmap=False, rows=False)
            for name in NAMES:
                assert np.array_equal(getattr(got, name), getattr(exp, name)), f"commit batch {it}: {name}"
            assert ctx.read_groups().state_equal(sop.groups), f"commit batch {it}: committed group state"
