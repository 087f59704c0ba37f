"""Full-size parity at every size BASELINE.json lists (GPU): cfg3 all four scenarios and cfg4 cold / warm / tail with
every stage on, the 10 000-event churn stream of config 5.  The oracle's answers are committed as digests
(tests/golden/fullsize_digests.json, see tests/fullsize.py); cfg3 and the churn stream are also re-computed live.
BS_SKIP_SLOW_LIVE=1 skips the live cfg4 re-computation (~2 minutes of one host core)."""
import json
import os

import numpy as np
import pytest

import fullsize
from test_gpu_parity import assert_batch_equal, load_ctx

pytestmark = pytest.mark.gpu
DIGESTS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")))


def _check_digest(got, want, what, bitmap=True):
    have = fullsize.digest(got, bitmap)
    bad = [k for k in want if have.get(k) != want[k]]
    assert not bad, f"{what}: {bad} differ from the oracle's committed digests (have {[have.get(k) for k in bad]}, want {[want[k] for k in bad]})"


@pytest.mark.parametrize("config,scenario,seed", fullsize.BATCH_CASES)
def test_full_size_batch_equals_oracle_digest(config, scenario, seed, bsa, soa):
    """PreFilter codes, early-exit indices, leaders, Filter codes / feasible counts / the whole pods x nodes bitmap,
    admit counters and quorum bits, bit for bit, at 10k x 5k and 50k x 20k."""
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario, seed=seed)
    want = DIGESTS[fullsize.case_key(config, scenario, seed)]
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        got = ctx.batch(soa.STAGE_ALL)
        _check_digest(got, want, f"{config}/{scenario}")
        assert fullsize.sha(got.bitmap_from_rows()) == want["fl_bitmap"], "slot rows (what the Go Filter bit-tests) vs the oracle bitmap"
        st = ctx.stats(soa.STAGE_ALL)
        assert st["fast_path"] == (0 if scenario == "cold" else 1)
        # pod-axis sharding at full size: rank 1 of 8 (replicated batch, device-side ownership)
        if config == "cfg4":
            ctx.set_shard(1, 8)
            part = ctx.batch(soa.STAGE_ALL, bitmap=False)
            mine = part.pf_code != 0xFF
            assert 0 < mine.sum() < pods.p
            assert np.array_equal(part.pf_code[mine], got.pf_code[mine]) and np.array_equal(part.fl_feasible[mine], got.fl_feasible[mine])
            ctx.set_shard(0, 1)


@pytest.mark.parametrize("scenario,seed", [("cold", 1), ("warm", 2), ("busy", 3)])
def test_batch_cfg3_all_stages_live_oracle(scenario, seed, bsa, soa, orc):
    """BASELINE configs[2] with Filter on, against the oracle run here and now (the digests cannot go stale unnoticed)."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg3", scenario, seed=seed)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    _check_digest(exp, DIGESTS[fullsize.case_key("cfg3", scenario, seed)], f"oracle vs its own committed digest, cfg3/{scenario}")
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, f"cfg3/{scenario}")


@pytest.mark.skipif(os.environ.get("BS_SKIP_SLOW_LIVE") == "1", reason="BS_SKIP_SLOW_LIVE=1")
def test_batch_cfg4_tail_live_oracle(bsa, soa, orc):
    """BASELINE configs[3] (50k pods / 5k groups / 20k nodes), all stages, against the oracle run live."""
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg4", "tail", seed=4)
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp, "cfg4/tail")


def test_churn_stream_cfg5_10000_events(bsa, soa, orc):
    """BASELINE configs[4] as SURVEY 8(d) defines it: cfg3 + 10 000 node events (40 % requested-update, 30 % append,
    30 % stable remove), a re-score after every 100 — 100 incremental re-scores, each equal to a full oracle
    recompute (digest of every output array; every 10th round re-computed live)."""
    c = DIGESTS["churn"]["params"]
    assert c == fullsize.CHURN
    nodes, fit, groups, pods, _ = bsa.synth.make(c["config"], c["scenario"], seed=c["seed"])
    stream = fullsize.ChurnStream(nodes, fit, c["seed"])
    st = soa.STAGE_PREFILTER | soa.STAGE_TALLY
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        for rnd, want in enumerate(DIGESTS["churn"]["rounds"]):
            ctx.apply_node_deltas(stream.next_deltas(c["events"]))
            assert ctx.n == want["n"]
            got = ctx.batch(st, bitmap=False)
            have = fullsize.sha(np.concatenate([getattr(got, a).view(np.uint8).ravel() for a in fullsize.ARRAYS]))
            if have != want["all"] or rnd % 10 == 9:
                cur_nodes, cur_fit = stream.current()
                exp = orc.Sop(orc.Snapshot(cur_nodes, cur_fit), groups).batch(pods, st, bitmap=False)
                assert_batch_equal(got, exp, f"churn round {rnd}", bitmap=False)
            assert have == want["all"], f"churn round {rnd}: digest differs although the live oracle agrees -> stale golden file"
            assert int(got.group_ready.sum()) == want["groups_ready"]
