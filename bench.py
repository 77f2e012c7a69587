#!/usr/bin/env python3
"""bench.py -- NID Gauss-Newton iterations/sec on MI355X (BASELINE.json metric).

One "step" = one NID cost + Jacobian evaluation of every active cell at one
SE(3) candidate plus the Huber-weighted reduction to the 6x6 normal equations,
delivered to host memory.  Workload at N=1 is
BASELINE.json configs[1]: the 640x480 pair, 16x16 cells, 8-bin B-spline
histogram (synthetic pair: the ETH-CVG data is not available offline).

N = 1: the timed region is the library's own host pipeline (nid_run_sequence):
64 poses per kernel launch, consecutive launches alternating between two
streams, every pose's 6x6 system collected from pinned host memory.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), the cells
of the SAME pair are partitioned over the ranks (strong scaling); the per-rank
partial [chi2, b(6), H upper (21), n_active] blocks (32 doubles per pose) of a
group of launches are summed by one all-reduce over xGMI on a comm stream while
the compute streams work on the next group.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400000)
    ap.add_argument("--warmup", type=int, default=40000)
    ap.add_argument("--config", default="A", choices=["A", "B", "S"])
    ap.add_argument("--bins", type=int, default=8)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--batch", type=int, default=64,
                    help="candidate poses per kernel launch at N=1 (1 = one launch per step)")
    ap.add_argument("--compute-streams", type=int, default=0, help="N>1: 1, 2 or 4 (0 = 2 for N<=2, else 4)")
    ap.add_argument("--group", type=int, default=4,
                    help="N>1: kernel launches per all-reduce (group of GROUP*BATCH poses)")
    ap.add_argument("--cost-only", action="store_true",
                    help="time cost-only evaluations (what LM trial poses need) instead of cost+Jacobian; not the metric")
    ap.add_argument("--strict", action="store_true",
                    help="NID_MATH_STRICT (every rounding of the reference path) instead of the default FAST math; not the metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) is the product path; gloo lets the multi-rank code path be "
                         "exercised on a box with fewer GPUs than ranks (ranks then share devices)")
    return ap.parse_args()


def pose_trajectory(synth, pair, n):
    """LM-like sequence of candidates around the disturbed start pose."""
    rng = np.random.default_rng(20211003)
    out = []
    for _ in range(n):
        out.append(synth.perturb_pose7(pair.pose_init, rng.normal(0, 1.5e-3, 3), rng.normal(0, 2e-3, 3)))
    return out


def cpu_baseline(pair, bins, seconds):
    """The oracle (CPU restatement of the reference's CPU edge) timed on this
    host, single thread like the reference (OpenMP off, g2o/CMakeLists.txt:58).
    Rebuilt with -march=native here (g2o/CMakeLists.txt:67); bounded sample."""
    from oracle import oracle_py
    lib = None
    try:
        out = os.path.join("/tmp", f"libnid_oracle_native_{os.getpid()}.so")
        oracle_py.build(march="-march=native", out=out)
        lib = oracle_py.load(out)
    except Exception:
        lib = oracle_py.load()
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    o = oracle_py.Oracle(pair.rows, pair.cols, pair.cell, bins, pair.fx, pair.fy, pair.cx, pair.cy, lib=lib)
    pts = oracle_py.backproject(pair.depth_m, synth.matrix_colmajor16(pair.T_wc0), pair.fx, pair.fy, pair.cx, pair.cy)
    o.set_reference(pts, pair.im0)
    o.set_target(pair.im1)
    o.compute_href(pair.pose_init)
    poses = pose_trajectory(synth, pair, 64)
    delta = float(np.sqrt(0.95))
    n = 0
    t0 = time.perf_counter()
    while True:
        _, _, err, J = o.evaluate(poses[n % len(poses)], True)
        oracle_py.normal_equations(err, J, delta)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 400:
            break
    return {"value": n / el, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{n} cost+Jacobian evaluations + 6x6 reduction of the same {pair.cols}x{pair.rows} pair, "
                      f"{el:.1f} s on 1 host core (oracle rebuilt -O3 -march=native -ffp-contract=off)"}


def pose_error_vs_ref(pair, bins):
    """The second half of the BASELINE metric: the reference driver's optimisation (10 LM iterations
    from the disturbed start, NID_pose_estimation.cpp:163-366) on the C++ host stack + HIP kernels,
    against the CPU oracle's LM on the same pair; also the wall time of both."""
    from oracle import oracle_py
    hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    t0 = time.perf_counter()
    pose_gpu, recs, _ = hostlib.run_lm(pair, bins, pair.pose_init, 10)
    t_gpu = time.perf_counter() - t0
    o_gpu = hostlib.last_optimize_seconds()
    t0 = time.perf_counter()
    pose_fused, recs_f, _ = hostlib.run_lm(pair, bins, pair.pose_init, 10, fused=1)
    t_fused = time.perf_counter() - t0
    o_fused = hostlib.last_optimize_seconds()
    t0 = time.perf_counter()
    pose_spec, recs_s, _ = hostlib.run_lm(pair, bins, pair.pose_init, 10, fused=2)
    t_spec = time.perf_counter() - t0
    o_spec = hostlib.last_optimize_seconds()
    o = oracle_py.from_pair(pair, bins, jac_bound="cpu", xform="matrix")
    o.compute_href(pair.pose_init)
    t0 = time.perf_counter()
    pose_cpu, recs_o = o.lm(pair.pose_init, 10)
    t_cpu = time.perf_counter() - t0
    mv = synth.pose7_minimal
    # BASELINE configs[4]: 3-level pyramid, 10 iterations per level (own coarse-to-fine definition)
    t0 = time.perf_counter()
    pose_pyr, per_pyr, _ = hostlib.run_pyramid_lm(pair, bins, pair.pose_init, levels=3, iterations=10, fused=2)
    t_pyr = time.perf_counter() - t0
    t0 = time.perf_counter()
    pose_pyr_o, per_pyr_o = oracle_py.pyramid_lm(pair, bins, pair.pose_init, levels=3, iterations=10)
    t_pyr_o = time.perf_counter() - t0
    pyramid = {
        "levels": 3, "iterations_per_level": 10,
        "max_abs_minimal_vector_diff": float(np.abs(mv(pose_pyr) - mv(pose_pyr_o)).max()),
        "same_lm_trace": [[r["lm_trials"] for r in lv] for lv in per_pyr] == [[r["lm_trials"] for r in lv] for lv in per_pyr_o],
        "lm_outer_iterations": [len(lv) for lv in per_pyr],
        "error_vs_truth_end": float(np.linalg.norm(mv(pair.pose_true) - mv(pose_pyr))),
        "wall_s": {"hip_fused_batched_trials": t_pyr, "cpu_oracle_1core": t_pyr_o},
    }
    return {
        "pyramid_3_levels": pyramid,
        "max_abs_minimal_vector_diff": float(np.abs(mv(pose_gpu) - mv(pose_cpu)).max()),
        "fused_path_diff": float(np.abs(mv(pose_fused) - mv(pose_cpu)).max()),
        "tolerance": 1e-6,
        "same_lm_trace": [r["lm_trials"] for r in recs] == [r["lm_trials"] for r in recs_o],
        "lm_outer_iterations": len(recs),
        "error_vs_truth_start": float(np.linalg.norm(mv(pair.pose_true) - mv(pair.pose_init))),
        "error_vs_truth_end": float(np.linalg.norm(mv(pair.pose_true) - mv(pose_gpu))),
        "speculative_path_diff": float(np.abs(mv(pose_spec) - mv(pose_cpu)).max()),
        "lm_wall_s": {"hip_reference_flow": t_gpu, "hip_fused": t_fused, "hip_fused_batched_trials": t_spec,
                      "cpu_oracle_1core": t_cpu},
        "optimize_only_s": {"hip_reference_flow": o_gpu, "hip_fused": o_fused, "hip_fused_batched_trials": o_spec},
        "note": "lm_wall_s includes the per-pair setup (upload, back-projection, reference weights); "
                "reference schedule = 1 Jacobian + k cost-only + 1 verbose evaluation per outer iteration",
    }


def measured_traffic(config, bins, poses_per_launch):
    """HBM bytes per launch of the evaluation kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written from tools/summarize_profile.py output); None if that
    configuration / launch shape was not profiled.  bench.py cannot run the profiler on itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            e = json.load(fh)[f"{config}:{bins}"]
        return float(e["hbm_bytes_per_launch"]) if int(e["poses_per_launch"]) == poses_per_launch else None
    except Exception:
        return None


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback of the product path)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    capi = importlib.import_module("nid-pose-estimation_amd.capi")
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    parallel = importlib.import_module("nid-pose-estimation_amd.parallel")
    pair = synth.make_pair(args.config)
    ncell = pair.cell * pair.cell
    lo, hi = parallel.cell_range(rank, world, ncell)
    ctx = capi.from_pair(pair, args.bins, device=local_rank, cell_begin=lo, cell_end=hi)
    if args.block_threads:
        ctx.set_block_threads(args.block_threads)
    if args.strict:
        ctx.set_math_mode(capi.MATH_STRICT)
    sides = None
    if world > 1:
        # Explicit streams (torch's default stream has the null handle, which nid_set_stream() reads as "use
        # the context's own stream"): consecutive kernel launches rotate over 2 (N <= 2) or 4 COMPUTE
        # streams, so that many launches are resident at once (at N = 8 a shard of 32 cells x 16 poses is
        # only 2 workgroups per CU, and the kernel wants 5); a COMM stream waits for a group's launches,
        # runs its RCCL all-reduce and the D2H copy, while the compute streams already work on the next group.
        ncomp = args.compute_streams if args.compute_streams in (1, 2, 4) else (2 if world <= 2 else 4)
        ncomp = max(1, min(ncomp, capi.NID_SLOTS // max(1, min(args.batch, capi.NID_MAX_BATCH))))
        sides = [torch.cuda.Stream(device=dev) for _ in range(ncomp)]
        comm = torch.cuda.Stream(device=dev)
        ctx.set_stream(sides[0].cuda_stream)
    # world == 1: the library's own in-order stream; torch.cuda.synchronize() below fences the whole device
    cnt, href = ctx.compute_href(pair.pose_init)
    delta = float(np.sqrt(0.95))
    K, W = args.steps, args.warmup
    poses = pose_trajectory(synth, pair, 256)

    # device-side result ring (world > 1): a group = G launches of B poses on one stream, summed by ONE
    # all-reduce of [G*B, 32] doubles (the collective is latency-bound: fewer, larger ones)
    B = max(1, min(args.batch, capi.NID_MAX_BATCH))
    assert capi.NID_SLOTS % B == 0
    G = max(1, args.group)
    ngroups = 2
    ring = torch.zeros((ngroups, G * B, capi.NID_REDUCED_LEN), dtype=torch.float64, device=dev)
    host_ring = torch.zeros((ngroups, G * B, capi.NID_REDUCED_LEN), dtype=torch.float64).pin_memory()
    done = [torch.cuda.Event() for _ in range(ngroups)] if world > 1 else None
    pose_arr = np.stack(poses)

    def launch_group(j, n):
        """world > 1: poses j*G*B .. +n-1 as up to G kernel launches per rank, alternating between the two
        compute streams (each with its own B result slots, reused launch after launch: a stream runs its
        launches in order); then, on the comm stream, the RCCL sum of the [G*B, 32] partial blocks over xGMI
        and the copy to pinned host memory."""
        gidx = j % ngroups
        if j >= ngroups:
            done[gidx].synchronize()             # the group that last used this ring entry has been delivered
        nl = (n + B - 1) // B
        for l in range(nl):
            m = min(B, n - l * B)
            idx = [(j * G * B + l * B + k) % len(poses) for k in range(m)]
            st = sides[l % len(sides)]
            ctx.set_stream(st.cuda_stream)
            with torch.cuda.stream(st):
                ctx.launch_batch((l % len(sides)) * B, pose_arr[idx], delta, True,
                                 reduced_dev=ring[gidx, l * B].data_ptr())
        for st in sides[:min(nl, len(sides))]:
            comm.wait_stream(st)
        with torch.cuda.stream(comm):
            dist.all_reduce(ring[gidx])
            host_ring[gidx].copy_(ring[gidx], non_blocking=True)
            done[gidx].record(comm)

    def run(n):
        if world == 1:
            # the C host loop of the library drives the pipeline (B poses per launch, 16/B launches in
            # flight); every pose's 6x6 system is collected from pinned host memory
            seq = pose_arr[np.arange(n) % len(poses)]
            return ctx.run_sequence(seq, delta, batch=B, want_jac=not args.cost_only)
        ngr = (n + G * B - 1) // (G * B)
        for j in range(ngr):
            launch_group(j, min(G * B, n - j * G * B))
        for j in range(max(0, ngr - ngroups), ngr):
            done[j % ngroups].synchronize()
        for k in range(capi.NID_SLOTS):
            try:
                ctx.wait(k)                      # clear the slots' pending marks (their kernels are long done)
            except capi.NidError:
                pass
        return None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run(W)
    barrier()
    t0 = time.perf_counter()
    results = run(K)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the last result is finite and every rank agrees after the all-reduce
    ablation = bool(os.environ.get("NID_HIP_LIB"))  # kernel-ablation builds (exp/) produce meaningless numbers
    if world == 1:
        assert ablation or (results.shape == (K, capi.NID_REDUCED_LEN) and np.all(np.isfinite(results)))
        ctx.launch(0, poses[(K - 1) % len(poses)], delta, not args.cost_only)
        H, b, chi2, na = ctx.wait(0)
        assert ablation or (np.array_equal(capi.unpack_reduced(results[K - 1])[0], H)
                            and capi.unpack_reduced(results[K - 1])[2] == chi2), \
            "pipelined result differs from a single launch"
    else:
        # the pipelined result of the last step must equal a synchronous evaluation of the same pose
        jl, kl = (K - 1) // (G * B), (K - 1) % (G * B)
        piped = host_ring[jl % ngroups, kl].clone().numpy()
        torch.cuda.synchronize(dev)
        ctx.set_stream(sides[0].cuda_stream)
        with torch.cuda.stream(sides[0]):
            ctx.launch(0, poses[(K - 1) % len(poses)], delta, True, reduced_dev=ring[0, 0].data_ptr())
            dist.all_reduce(ring[0, 0])
        torch.cuda.synchronize(dev)
        ctx.wait(0)
        sync = ring[0, 0].cpu().numpy()
        # the per-rank partial blocks are bitwise reproducible; the collective may sum the ranks in a different
        # order for a [G*B,32] tensor than for a [32] one (ring / tree by message size), so from 3 ranks on the
        # comparison is to rounding
        same = np.array_equal(piped, sync) if world <= 2 else np.allclose(piped, sync, rtol=1e-12, atol=1e-300)
        assert same, "pipelined multi-rank result differs from the synchronous one"
        H, b, chi2, na = capi.unpack_reduced(sync)
    assert ablation or (np.isfinite(chi2) and np.all(np.isfinite(H)) and na > 0)
    if args.cost_only and rank == 0:
        print("[bench] --cost-only: cost evaluations without the Jacobian phase; not the BASELINE metric", file=sys.stderr)

    # dominant-kernel duration: groups of 10 identical launches (B poses each, this rank's cells) back to
    # back on the launch stream between ONE pair of HIP events -- the per-launch duration a kernel trace
    # reports; launches of a stream are serialised, so nothing else overlaps them
    ev_ms = []
    for i in range(min(max(K // (10 * B), 5), 40)):
        idx = [(i * B + k) % len(poses) for k in range(B)]
        ev_ms.append(ctx.time_launches(pose_arr[idx], delta, repeats=10, want_jac=not args.cost_only))

    if rank == 0:
        eval_ms = float(np.median(ev_ms))
        contract = ctx.contract_bytes() * B  # this rank's cells x poses per launch
        achieved = contract / (eval_ms * 1e-3) / 1e9
        out = {
            "metric": "NID GN iterations/sec (640x480 dense pair)" if args.config == "A" else
                      f"NID GN iterations/sec ({pair.cols}x{pair.rows} dense pair)",
            "value": K / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{pair.cols}x{pair.rows} synthetic RGB-D pair, {pair.cell}x{pair.cell} cells of "
                            f"{pair.rows // pair.cell}x{pair.cols // pair.cell} px, {args.bins}-bin cubic B-spline "
                            f"histograms, cost+Jacobian+Huber 6x6 reduction per step, "
                            f"{int((cnt[lo:hi] >= 300).sum())} active cells on rank 0",
                "cells": ncell, "bins": args.bins,
                "parallelism": f"cells/{world}" + ("" if world == 1 else
                                                   f" + {'RCCL' if args.backend == 'nccl' else 'gloo'} all-reduce([{G * B},32] f64)"),
                "pipelining": f"{B} candidate poses per kernel launch, 2 launches in flight on 2 streams, "
                              + ("each pose's 6x6 system lands in pinned host memory" if world == 1 else
                                 f"launches rotate over {len(sides) if sides else 2} compute streams; one all-reduce of [{G * B},32] f64 per "
                                 f"{G} launches + D2H to pinned memory on a comm stream, 2 groups in flight"),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.config, args.bins, B) if world == 1 else None,
                "poses_per_launch": B,
                "kernel": "nid::k_eval2<128, JAC=true, FAST, NB=8|10|generic, DBG=false, EXT=(poses per launch > 16)>",
                "kernel_ms": eval_ms,
                "algorithmic_bytes_per_launch": contract,
                "achieved_pipelined": (contract / B) * (K / elapsed) / 1e9,
                "note": "achieved = contract bytes (68 B/px + 64 B/cell, SURVEY 8d) / median per-launch duration of "
                        "evaluation launches running one at a time (10 back to back per HIP event pair; what "
                        "rocprofv3 reports per kernel); "
                        "achieved_pipelined = the same bytes / (timed region / launches): the timed pipeline keeps "
                        "launches on two streams in flight, so the next launch fills the tail of the previous one; "
                        "the tile is L2/MALL-resident after the first launch",
            },
        }
        out["check"] = {"chi2": chi2, "n_active": int(na), "H00": float(H[0, 0]), "b0": float(b[0])}
        if not args.no_cpu_baseline and world == 1:   # CPU legs: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(pair, args.bins, args.cpu_seconds)
            out["pose_error_vs_ref"] = pose_error_vs_ref(pair, args.bins)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
