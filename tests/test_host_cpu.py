"""CPU-side checks that need no GPU: ABI surface, host containers, generator, sharding rule, and the
N>1 collective path under gloo with world_size 2."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol(bsa):
    path = bsa.build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "bsched.h")).read()
    declared = set(re.findall(r"^\s*(?:int|uint32_t|const char\*)\s+(bs_[a-z_0-9]+)\s*\(", header, re.M))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libbsched.so does not export {name}"
    assert declared == set(bsa.capi.ABI_SYMBOLS), declared ^ set(bsa.capi.ABI_SYMBOLS)
    assert lib.bs_abi_version() == 7
    lib.bs_strerror.restype = ctypes.c_char_p
    assert lib.bs_strerror(-2) == b"no usable gfx950 device"


def test_code_object_is_gfx950_only(bsa):
    path = bsa.build.build()
    blob = open(path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_"):
        assert other not in blob


def test_no_gpu_raises_instead_of_falling_back(bsa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bsa.capi.BsError):
        bsa.Context(scalar_lanes=0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "batch-scheduler_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "bs_oracle" not in text and "import orc" not in text and "naive_ref" not in text and "naive_fit" not in text, f


def test_fit_masks_roundtrip(soa):
    rng = np.random.default_rng(0)
    for n in (0, 1, 31, 32, 33, 100):
        fit = rng.random((3, n)) < 0.5
        fm = soa.FitMasks.from_bool(fit)
        assert fm.bits.shape == (3, (n + 31) // 32)
        assert np.array_equal(fm.to_bool(), fit)


def test_synth_is_deterministic_and_shaped(bsa, soa):
    a = bsa.synth.make("cfg2", "tail", seed=7)
    b = bsa.synth.make("cfg2", "tail", seed=7)
    c = bsa.synth.make("cfg2", "tail", seed=8)
    assert np.array_equal(a[0].allocatable, b[0].allocatable) and np.array_equal(a[3].req, b[3].req)
    assert not np.array_equal(a[0].requested, c[0].requested)
    nodes, fit, groups, pods, meta = a
    assert (nodes.n, groups.g, pods.p, nodes.lanes) == (500, 200, 1000, 4)
    # allocatable memory must not be float32-representable (exercises the rounding of core.go:658)
    mem = nodes.allocatable[1]
    assert np.mean(mem.astype(np.float32).astype(np.int64) != mem) > 0.9
    for cfg, shape in (("cfg3", (5000, 2000, 10000, 5)), ("cfg4", (20000, 5000, 50000, 5))):
        m = bsa.synth.CONFIGS[cfg]
        assert (m["nodes"], m["groups"], m["pods"], 4 + m["scalars"]) == shape


def test_synth_tail_is_the_hard_case_for_the_reference(bsa, soa, orc):
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "tail")
    sop = orc.Sop(orc.Snapshot(nodes, fit), groups)
    out = sop.batch(pods, soa.STAGE_PREFILTER)
    assert sop.iters > 0.8 * pods.p * nodes.n          # the early exit of core.go:623 fires late
    assert (out.pf_code == soa.PF_PASS_RESERVE_FITS).sum() > 0.9 * pods.p


def test_owner_ranks_partition(bsa):
    dist = importlib.import_module("batch-scheduler_amd.dist")
    _, _, groups, pods, _ = bsa.synth.make("cfg2", "warm")
    for nranks in (1, 2, 3, 8):
        own = dist.owner_ranks(pods.group, groups.g, nranks)
        assert own.min() >= 0 and own.max() < nranks
        for g in range(groups.g):
            assert len(set(own[pods.group == g].tolist())) <= 1, "a group never straddles ranks"
        counts = np.bincount(own, minlength=nranks)
        assert counts.min() > 0.5 * pods.p / nranks


def test_owner_ranks_balance_on_any_queue_order(bsa):
    """Ownership is balanced by pod count: max / mean <= 1.2 on the cfg4 queue — gang-sorted as generated, shuffled (every
    gang's first pod early in the queue: cutting queue positions gave rank 0 nearly everything) and reversed — groups intact."""
    dist = importlib.import_module("batch-scheduler_amd.dist")
    _, _, groups, pods, _ = bsa.synth.make("cfg4", "tail")
    rng = np.random.default_rng(4)
    for order in (np.arange(pods.p), rng.permutation(pods.p), np.arange(pods.p)[::-1]):
        g = pods.group[order]
        for nranks in (2, 4, 8):
            own = dist.owner_ranks(g, groups.g, nranks)
            counts = np.bincount(own, minlength=nranks)
            assert counts.max() / counts.mean() <= 1.2, (nranks, counts)
            first = np.full(groups.g, -1)
            first[g[::-1][g[::-1] >= 0]] = own[::-1][g[::-1] >= 0]          # owner of each group's first pod
            assert np.array_equal(own[g >= 0], first[g[g >= 0]]), "a group never straddles ranks"


WORKER = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BS_ROOT"], "oracle"))
bsa = importlib.import_module("batch-scheduler_amd"); bdist = importlib.import_module("batch-scheduler_amd.dist")
import orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nodes, fit, groups, pods, _ = bsa.synth.make("tiny", "busy", seed=5, pods=192, groups=24)
full = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, bsa.soa.STAGE_ALL)
own = bdist.owner_ranks(pods.group, groups.g, world)
# this rank's share: admit counters of the pods it owns (the oracle stands in for the per-rank GPU)
mine = own == rank
passed = mine & (full.pf_code < 16) & (full.fl_feasible > 0) & (pods.group >= 0)
part = np.bincount(pods.group[passed], minlength=groups.g).astype(np.int32)
t = torch.from_numpy(part.copy())
bdist.all_reduce_admit(t, dist)
assert np.array_equal(t.numpy().astype(np.uint32), full.group_admit), (rank, t, full.group_admit)
ready = (groups.matched + t.numpy().astype(np.uint32)) >= (groups.min_member - groups.status_scheduled)
assert np.array_equal(ready.astype(np.uint8), full.group_ready)
# max-over-ranks timing reduction as bench.py does it
tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
assert tt.item() == world
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_allreduce_of_admit_counters(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, BS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("ok") == 2


def test_tools_and_product_do_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    for d in ("tools", "batch-scheduler_amd"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".sh", ".hip", ".hpp", ".cpp", ".h")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "import orc" not in text and "libbs_oracle" not in text and "naive_" not in text, os.path.join(dirpath, f)
