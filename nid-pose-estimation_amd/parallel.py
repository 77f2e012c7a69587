"""Cell sharding across the GPUs of one node.

Given the pose, every cell's (Hc, Hj, err, J) depends only on that cell's
pixels, so rank r of R owns the contiguous cell range
[r*cells/R, (r+1)*cells/R).  The only exchange per evaluation is the sum of the
per-rank partial blocks [chi2, b(6), H upper triangle (21), n_active] = 32
doubles (RCCL all-reduce over xGMI; gloo in the CPU tests).  The reference has
no multi-GPU path; this is new work.

torch is plumbing here (process group, device tensors), not the compute path.
"""
from __future__ import annotations

import numpy as np

REDUCED_LEN = 32


def cell_range(rank: int, world: int, ncell: int):
    """Contiguous, exhaustive, order-preserving partition of cell ids."""
    if not (0 <= rank < world) or world > ncell:
        raise ValueError(f"bad shard {rank}/{world} of {ncell} cells")
    return rank * ncell // world, (rank + 1) * ncell // world


def all_ranges(world: int, ncell: int):
    return [cell_range(r, world, ncell) for r in range(world)]


def allreduce_reduced(block, group=None):
    """In-place sum of a 32-double partial block over the process group."""
    import torch.distributed as dist
    dist.all_reduce(block, op=dist.ReduceOp.SUM, group=group)
    return block


def allgather_cells(local_cells, world: int, group=None):
    """Gather per-cell blocks (cells_local x NID_CELL_OUT) from every rank, in rank (= cell id)
    order -- the form needed when each host-side g2o edge must receive its own set_h/set_j."""
    import torch
    import torch.distributed as dist
    outs = [torch.empty_like(local_cells) for _ in range(world)]
    dist.all_gather(outs, local_cells, group=group)
    return torch.cat(outs, dim=0)


def unpack_reduced_np(r):
    """numpy twin of nid_unpack_reduced (no library needed: used by the CPU gloo tests)."""
    r = np.asarray(r, dtype=np.float64)
    H = np.zeros((6, 6))
    idx = 7
    for a in range(6):
        for b in range(a, 6):
            H[a, b] = H[b, a] = r[idx]
            idx += 1
    return H, r[1:7].copy(), float(r[0]), int(r[28])


def pack_reduced_np(H, b, chi2, n_active):
    r = np.zeros(REDUCED_LEN)
    r[0] = chi2
    r[1:7] = b
    idx = 7
    for a in range(6):
        for c in range(a, 6):
            r[idx] = H[a, c]
            idx += 1
    r[28] = n_active
    return r
