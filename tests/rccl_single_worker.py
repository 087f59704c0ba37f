"""bs_comm_init + the library's own ncclAllReduce at world size 1, in its own process (tests/test_gpu_parity.py gives it a
deadline: on some boxes RCCL's bootstrap takes minutes)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402


def main():
    bsa = importlib.import_module("batch-scheduler_amd")
    import orc
    from test_gpu_parity import assert_batch_equal, load_ctx
    soa = bsa.soa
    nodes, fit, groups, pods, _ = bsa.synth.make("tiny", "warm")
    exp = orc.Sop(orc.Snapshot(nodes, fit), groups).batch(pods, soa.STAGE_ALL)
    with load_ctx(bsa, nodes, fit, groups, pods) as ctx:
        ctx.comm_init(bsa.capi.comm_unique_id(), 0, 1)
        for _ in range(3):
            assert_batch_equal(ctx.batch(soa.STAGE_ALL), exp)
    print("RCCL_SINGLE_OK")


if __name__ == "__main__":
    main()
