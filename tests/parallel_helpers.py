"""Numpy / torch.distributed twins of the cell-sharding arithmetic for the CPU tests.

The product's multi-GPU path is C++ behind the C-ABI (include/nid/nid_multi.h, csrc/nid_multi.inc: shard
contexts, RCCL all-reduce from C++, the LM on shards through the host library).  What stays here is what the
world_size-2 gloo test on CPU needs: the cell partition (same formula as nid_multi_cell_range), the packing
of the 32-double partial block and the collectives on it.
"""
from __future__ import annotations

import numpy as np

REDUCED_LEN = 32


def cell_range(rank: int, world: int, ncell: int):
    """Contiguous, exhaustive, order-preserving partition of cell ids."""
    if not (0 <= rank < world) or world > ncell:
        raise ValueError(f"bad shard {rank}/{world} of {ncell} cells")
    return rank * ncell // world, (rank + 1) * ncell // world


def cell_set(rank: int, world: int, ncell: int, interleaved: bool = False):
    """The cell ids a rank owns: the contiguous range above, or -- NID_PARTITION_INTERLEAVED of nid_multi.h -- every
    world-th cell starting at `rank` (nid_create_strided)."""
    if not interleaved:
        lo, hi = cell_range(rank, world, ncell)
        return np.arange(lo, hi)
    if not (0 <= rank < world) or world > ncell:
        raise ValueError(f"bad shard {rank}/{world} of {ncell} cells")
    return np.arange(rank, ncell, world)


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
