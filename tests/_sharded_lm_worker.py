"""Worker of tests/test_host_gpu.py::test_sharded_lm_*: one rank of the cell-sharded LM (gloo on a shared GPU
in the test; RCCL with one rank per GPU in production).  Rank 0 prints one JSON line."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
parallel = importlib.import_module("nid-pose-estimation_amd.parallel")


def main():
    cfg, nb, backend = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    devid = local % torch.cuda.device_count()
    torch.cuda.set_device(devid)
    pair = synth.make_pair(cfg)
    ncell = pair.cell * pair.cell
    lo, hi = parallel.cell_range(rank, world, ncell)
    ctx = capi.from_pair(pair, nb, device=devid, cell_begin=lo, cell_end=hi)
    ctx.compute_href(pair.pose_init)
    prob = parallel.ShardedProblem(ctx, hostlib, capi, torch.device("cuda", devid), float(np.sqrt(0.95)))
    pose, recs = prob.lm(pair.pose_init, 10)
    if rank == 0:
        print(json.dumps({"pose": pose.tolist(), "lm_trials": [r["lm_trials"] for r in recs],
                          "chi2": [r["chi2"] for r in recs], "world": world}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
