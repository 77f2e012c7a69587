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


# ---------------------------------------------------------------------------------------------------
# The optimisation itself on cell shards (BASELINE configs[3]): every rank runs the same LM loop; each
# evaluation is one kernel launch over the rank's cells for all candidate poses + ONE all-reduce of the
# partial [n, 32] blocks.  The loop is the fused / batched-trials form of the reference's
# OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-225) as
# host/g2o_min.cpp::solveFused restates it -- same lambda schedule, same accept/reject rule -- with the
# 6x6 solve and SE(3) update done by the host library's routines, so all ranks take identical decisions
# (they see bit-identical reduced sums after the all-reduce).
class ShardedProblem:
    """One frame pair, this rank's cell range.  `ctx` is a capi.Context created with cell_begin/cell_end
    = cell_range(rank, world, cells), reference / target / href state set.  torch is plumbing: the device
    tensor the kernel writes its partial blocks into, and the process group."""

    def __init__(self, ctx, hostlib, capi, device, delta, group=None):
        import torch
        self.ctx, self.hostlib, self.capi, self.group, self.delta = ctx, hostlib, capi, group, float(delta)
        self.dev = device
        self.blocks = torch.zeros((capi.NID_MAX_BATCH, REDUCED_LEN), dtype=torch.float64, device=device)
        self.stream = torch.cuda.Stream(device=device)
        ctx.set_stream(self.stream.cuda_stream)
        ctx.set_block_threads(256)   # blocking, small launches: the latency shape (see host/legacy_ops.cpp)

    def evaluate(self, poses7, want_jac):
        """[(H, b, chi2, n_active)] of the WHOLE image for up to NID_MAX_BATCH poses: one launch + one all-reduce."""
        import torch
        import torch.distributed as dist
        poses7 = np.asarray(poses7, dtype=np.float64).reshape(-1, 7)
        n = poses7.shape[0]
        with torch.cuda.stream(self.stream):
            self.ctx.launch_batch(0, poses7, self.delta, want_jac, reduced_dev=self.blocks.data_ptr())
            if dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(self.blocks[:n], group=self.group)
            host = self.blocks[:n].cpu()
        for k in range(n):
            self.ctx.wait(k)
        return [unpack_reduced_np(host[k].numpy()) for k in range(n)]

    def lm(self, pose7, iterations=10, tau=1e-5, max_trials=10):
        """Returns (pose7, records); records[i] = dict(chi2, lambda_, lm_trials, rho, pose7)."""
        hl = self.hostlib
        lo_scale, hi_scale = 1.0 / 3.0, 2.0 / 3.0
        pose = np.asarray(pose7, dtype=np.float64).copy()
        lam, ni, n_bad = -1.0, 2.0, 0
        x_prev = np.zeros(6)
        recs = []
        for it in range(iterations):
            H, b, chi, _ = self.evaluate([pose], True)[0]
            ini_chi = cur_chi = chi
            if it == 0:
                lam, ni, n_bad = tau * max(abs(H[j, j]) for j in range(6)), 2.0, 0
            rho, q, accepted = 0.0, 0, False
            while q < max_trials:
                nb = min(16, self.capi.NID_MAX_BATCH, max_trials - q)
                lams, nis, xs, oks, cands = [], [], [], [], []
                l, n_i = lam, ni
                for k in range(nb):
                    lams.append(l); nis.append(n_i)
                    ok, x = hl.ldlt6_solve(H + l * np.eye(6), b)
                    if not ok:
                        x = xs[-1] if xs else x_prev        # a failed solve leaves x untouched
                    xs.append(x); oks.append(ok)
                    cands.append(hl.se3_mul(hl.se3_exp(x), pose))
                    l *= n_i; n_i *= 2
                res = self.evaluate(cands, False)
                stop = False
                for k in range(nb):
                    temp = res[k][2] if oks[k] else float(np.finfo(np.float64).max)
                    scale = 0.0
                    for j in range(6):
                        scale += xs[k][j] * (lams[k] * xs[k][j] + b[j])
                    scale += 1e-3
                    rho = (cur_chi - temp) / scale
                    q += 1
                    x_prev = xs[k]
                    if rho > 0 and np.isfinite(temp):
                        alpha = min(1.0 - (2 * rho - 1) ** 3, hi_scale)
                        lam, ni = lams[k] * max(lo_scale, alpha), 2.0
                        cur_chi, pose, accepted = temp, cands[k], True
                        break
                    lam, ni = lams[k] * nis[k], nis[k] * 2
                    if not (rho < 0):
                        stop = True
                        break
                if accepted or stop:
                    break
            recs.append(dict(iteration=it, chi2=cur_chi, lambda_=lam, lm_trials=q, rho=rho, pose7=pose.copy()))
            if q == max_trials or rho == 0:
                break
            n_bad = n_bad + 1 if (ini_chi - cur_chi) * 1e3 < ini_chi else 0
            if n_bad >= 3:
                break
        return pose, recs
