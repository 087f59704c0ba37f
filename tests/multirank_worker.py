"""One rank of the multi-process GPU tests (tests/test_gpu_multirank.py): several processes share ONE MI355X, each with
its own bs_ctx, and meet in a real collective.
  mode native-replicated   whole queue on every rank, bs_comm_init + the library's own ncclAllReduce (RCCL)
  mode native-partitioned  each rank loads the pods of the groups it owns, bs_comm_init + the library's ncclAllReduce
  mode gloo-partitioned    partitioned, the admit counters are summed with torch.distributed (gloo) through host memory,
                           written back to the device buffer and bs_batch_finish computes the quorum bits
usage: multirank_worker.py <mode> <rank> <world> <workdir> <config> <scenario> <seed>"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_early_returners(pods, soa, seed):
    """pods that return from PreFilter BEFORE findMaxPG (core.go:89-98: no label / a lastPermittedPod entry), sprinkled over the queue —
    front included: they leave the stale sop.maxFinishedPG of whoever reached :118 in front of them on the WHOLE queue, which is what a
    rank of a partitioned batch cannot see (bs_first_reach_hint)"""
    rng = np.random.default_rng(seed + 4711)
    pods = pods.copy()
    pods.flags[rng.random(pods.p) < 0.06] |= soa.POD_LAST_PERMITTED
    pods.group[rng.random(pods.p) < 0.03] = soa.POD_NOT_GROUPED
    pods.flags[:3] |= soa.POD_LAST_PERMITTED                  # the head of the queue does not reach
    return pods


def main():
    mode, rank, world, work, config, scenario, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6], int(sys.argv[7])
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    import torch                                           # before libbsched.so: both must end up on ONE HIP runtime (as in bench.py)
    torch.cuda.init()
    bsa = importlib.import_module("batch-scheduler_amd")
    bdist = importlib.import_module("batch-scheduler_amd.dist")
    soa = bsa.soa
    nodes, fit, groups, pods, _ = bsa.synth.make(config, scenario, seed=seed)
    pods = with_early_returners(pods, soa, seed)
    idx = np.arange(pods.p)
    hint = None
    if mode.endswith("partitioned"):
        own = bdist.owner_ranks(pods.group, groups.g, world)
        idx = np.nonzero(own == rank)[0]
        hint = bdist.first_reach_thresholds(pods, groups, own, world)[rank]
    mine = pods.take(idx)
    device = 0
    if mode.startswith("native"):
        device = rank % max(1, torch.cuda.device_count())   # one GPU per rank where there are several
    ctx = bsa.Context(scalar_lanes=nodes.lanes - 4, device=device)
    ctx.load_nodes(nodes, fit)
    ctx.load_groups(groups)
    ctx.load_pods(mine)
    if hint is not None:
        ctx.first_reach_hint(hint)                          # the one thing a rank cannot know from its own pods (bsched.h)
    status = "ok"
    try:
        if mode.startswith("native"):
            uid_path = os.path.join(work, "uid.bin")
            if rank == 0:
                uid = bsa.capi.comm_unique_id()
                with open(uid_path + ".tmp", "wb") as f:
                    f.write(uid)
                os.rename(uid_path + ".tmp", uid_path)
            t0 = time.time()
            while not os.path.exists(uid_path):
                if time.time() - t0 > 60:
                    raise RuntimeError("no unique id from rank 0")
                time.sleep(0.01)
            uid = open(uid_path, "rb").read()
            if mode == "native-partitioned":
                ctx.reduce_external(True)
            ctx.comm_init(uid, rank, world)                 # ncclCommInitRank: ranks of one communicator on ONE device
            outs = []
            for _ in range(3):                              # back-to-back batches: the collective is stream-ordered
                ctx.run(soa.STAGE_ALL)
                outs.append(ctx.read())
        else:
            import torch.distributed as dist
            dist.init_process_group("gloo", init_method=f"file://{os.path.join(work, 'rdzv')}", rank=rank, world_size=world)
            ctx.reduce_external(True)
            admit_t = torch.zeros(groups.g, dtype=torch.int32, device="cuda:0")
            ctx.bind_admit(admit_t.data_ptr())             # the counters live in caller-owned device memory
            outs = []
            for _ in range(2):
                ctx.run(soa.STAGE_ALL)                      # device: per-rank admit counters, no quorum pass yet
                ctx.sync()
                t = admit_t.cpu()
                dist.all_reduce(t)                          # the ONE collective of the batch
                admit_t.copy_(t)
                torch.cuda.synchronize()
                ctx.finish()                                # device: quorum bits from the reduced counters
                outs.append(ctx.read())
            dist.destroy_process_group()
    except Exception as e:                                   # reported to the parent, which decides what it means
        status = f"{type(e).__name__}: {e}"
        outs = []
    res = {"status": np.array(status), "idx": idx}
    if outs:
        o = outs[-1]
        same = all(np.array_equal(getattr(o, a), getattr(p, a)) for p in outs[:-1] for a in ("pf_code", "group_admit", "group_ready", "fl_feasible"))
        res.update(pf_code=o.pf_code, pf_first_k=o.pf_first_k, pf_leader=o.pf_leader, fl_code=o.fl_code, fl_feasible=o.fl_feasible, fl_bitmap=o.fl_bitmap,
                   group_admit=o.group_admit, group_ready=o.group_ready, repeatable=np.array(same))
    np.savez(os.path.join(work, f"rank{rank}.npz"), **res)
    ctx.close()


if __name__ == "__main__":
    main()
