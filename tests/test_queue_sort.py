"""Batched queue ordering: bs_queue_sort against the reference's Compare (core.go:368-411)."""
import numpy as np
import pytest


def _scene(rng, p, g, ties):
    names = [f"g{int(rng.integers(0, max(2, g // (3 if ties else 1))))}" if ties else f"g{i:05d}" for i in range(g)]
    creation = rng.integers(0, 4 if ties else 10 ** 6, g)
    prio = rng.choice([0, 0, 0, 5, -3, 2 ** 31 - 1, -(2 ** 31)], p).astype(np.int32) if ties else rng.integers(-100, 100, p).astype(np.int32)
    group = rng.integers(0, g, p).astype(np.int32)
    r = rng.random(p)
    group[r < 0.15] = -1                       # no label
    group[(r >= 0.15) & (r < 0.20)] = -2       # labelled, lister error
    ts = rng.integers(0, 50 if ties else 10 ** 12, p).astype(np.int64) - (25 if ties else 0)
    return names, creation, prio, group, ts


def _objs(names, creation, prio, group, ts):
    def gobj(gi):
        return None if gi == -1 else ("missing" if gi < 0 else {"creation": int(creation[gi]), "name": names[gi]})
    return [(int(prio[i]), gobj(int(group[i])), int(ts[i])) for i in range(len(prio))]


@pytest.mark.parametrize("seed,p,g,ties", [(1, 200, 12, True), (2, 300, 40, False), (3, 64, 5, True), (4, 1, 1, False)])
def test_oracle_order_is_sorted_under_compare(seed, p, g, ties, orc):
    rng = np.random.default_rng(seed)
    names, creation, prio, group, ts = _scene(rng, p, g, ties)
    ranks = orc.queue_order_ranks(creation, names)
    for i in range(g):
        for j in range(g):                     # the rank IS (creation asc, name desc), equal pairs share it
            a, b = (creation[i], names[i]), (creation[j], names[j])
            before = a[0] < b[0] or (a[0] == b[0] and a[1] > b[1])
            assert (ranks[i] < ranks[j]) == before and (ranks[i] == ranks[j]) == (a == b)
    perm = orc.queue_order(prio, group, ts, ranks)
    assert sorted(perm.tolist()) == list(range(p))
    o = _objs(names, creation, prio, group, ts)
    for a in range(p):                         # nothing later in the queue is Less than something earlier — for EVERY pair
        for b in range(a + 1, p):
            assert not orc.queue_less(o[perm[b]], o[perm[a]]), (a, b, o[perm[a]], o[perm[b]])
    # and whenever the reference's Compare orders a pair strictly, the queue agrees
    pos = np.empty(p, np.int64)
    pos[perm] = np.arange(p)
    for _ in range(2000):
        i, j = (int(x) for x in rng.integers(0, p, 2))
        if orc.queue_less(o[i], o[j]):
            assert pos[i] < pos[j]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,p,g,ties", [(1, 1, 1, False), (2, 63, 7, True), (3, 1000, 50, True), (4, 1025, 200, False), (5, 10000, 2000, False),
                                            (6, 10000, 2000, True), (7, 50000, 5000, False), (8, 2048, 3, True)])
def test_gpu_queue_sort_equals_oracle(seed, p, g, ties, bsa, orc):
    rng = np.random.default_rng(seed)
    names, creation, prio, group, ts = _scene(rng, p, g, ties)
    ranks = orc.queue_order_ranks(creation, names)
    exp = orc.queue_order(prio, group, ts, ranks)
    with bsa.Context(scalar_lanes=0) as ctx:
        ctx.load_queue_order(ranks)
        got = ctx.queue_sort(prio, group, ts)
        assert np.array_equal(got, exp)
        assert np.array_equal(ctx.queue_sort(prio, group, ts), exp)          # buffers are reused
        # a shorter list afterwards, and the degenerate case of one key for everybody (stable: identity)
        assert np.array_equal(ctx.queue_sort(prio[: p // 2 + 1], group[: p // 2 + 1], ts[: p // 2 + 1]), orc.queue_order(prio[: p // 2 + 1], group[: p // 2 + 1], ts[: p // 2 + 1], ranks))
        z = np.zeros(p, np.int32)
        assert np.array_equal(ctx.queue_sort(z, z - 1, np.zeros(p, np.int64)), np.arange(p))


@pytest.mark.gpu
def test_gpu_queue_sort_agrees_with_the_host_mirror_less(bsa, soa, orc):
    """the C++ mirror of Compare (bs_host.cpp, row a13) and the device's order agree pair by pair"""
    rng = np.random.default_rng(11)
    p, g = 400, 9
    names = [f"n{i}" for i in range(g)]
    creation = rng.integers(0, 3, g)
    prio = rng.integers(0, 3, p).astype(np.int32)
    group = rng.integers(-1, g, p).astype(np.int32)
    ts = rng.integers(0, 1000, p).astype(np.int64)
    ranks = orc.queue_order_ranks(creation, names)
    nodes, fit, *_ = bsa.synth.make("tiny", "warm")
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as ctx:
        ctx.load_nodes(nodes, fit)
        sop = bsa.plugin.ScheduleOperation(ctx)
        # name_rank is order-isomorphic to the name string (bigger string <-> bigger rank)
        order = sorted(range(g), key=lambda i: names[i])
        name_rank = {i: r for r, i in enumerate(order)}
        for i in range(g):
            assert sop.add_group(3, creation_ts=int(creation[i]), name_rank=name_rank[i]) == i
        ctx.load_queue_order(ranks)
        perm = ctx.queue_sort(prio, group, ts)
        for a in range(0, p - 1):
            x, y = int(perm[a]), int(perm[a + 1])
            assert not sop.Less((int(group[y]), int(prio[y]), int(ts[y])), (int(group[x]), int(prio[x]), int(ts[x])))
        sop.close()
