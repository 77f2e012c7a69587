"""The N>1 path on CPU: world_size-2 gloo processes.  Each rank produces the
partial 6x6 block of ITS cells -- which cells those are comes from the PRODUCT
(capi.cell_range / capi.cell_set = nid_multi_cell_range / nid_multi_cell_partition
of libnid_hip.so, which need no device); the oracle stands in for the GPU kernel
(test infrastructure), the blocks are unpacked by the product's nid_unpack_reduced;
the result must equal the unsharded reduction."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parallel_helpers as parallel   # tests/parallel_helpers.py: numpy / torch.distributed helpers of this test
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    from oracle import oracle_py
    capi = importlib.import_module("nid-pose-estimation_amd.capi")
    pair = synth.make_pair("S")
    o = oracle_py.from_pair(pair, 8)
    o.compute_href(pair.pose_init)
    _, _, err, J = o.evaluate(pair.pose_init, True)
    ncell = pair.cell * pair.cell
    lo, hi = capi.cell_range(rank, world, ncell)                    # the product's partition (GPU-free)
    assert (lo, hi) == parallel.cell_range(rank, world, ncell)      # ... which the test's numpy twin restates
    delta = float(np.sqrt(0.95))
    # this rank's partial block over its own cells only
    e_loc = np.full(ncell, np.nan); e_loc[lo:hi] = err[lo:hi]
    H, b, chi2, na = oracle_py.normal_equations(e_loc, J, delta)
    block = torch.from_numpy(parallel.pack_reduced_np(H, b, chi2, na))
    parallel.allreduce_reduced(block)
    # the interleaved partition (NID_PARTITION_INTERLEAVED): other cell sets, the same sum
    own = capi.cell_set(rank, world, ncell, capi.PARTITION_INTERLEAVED)
    assert np.array_equal(own, parallel.cell_set(rank, world, ncell, interleaved=True))
    e_int = np.full(ncell, np.nan); e_int[own] = err[own]
    Hi, bi, chi2i, nai = oracle_py.normal_equations(e_int, J, delta)
    block_i = torch.from_numpy(parallel.pack_reduced_np(Hi, bi, chi2i, nai))
    parallel.allreduce_reduced(block_i)
    # per-cell form
    cells = torch.from_numpy(np.concatenate([err[lo:hi, None], J[lo:hi]], axis=1))
    gathered = parallel.allgather_cells(cells, world) if (hi - lo) * world == ncell else None
    Hf, bf, cf, nf = oracle_py.normal_equations(err, J, delta)
    Hs, bs, cs, ns = capi.unpack_reduced(block.numpy())
    ok = (ns == nf and np.allclose(cs, cf, rtol=1e-13) and np.allclose(Hs, Hf, rtol=1e-12, atol=1e-13)
          and np.allclose(bs, bf, rtol=1e-12, atol=1e-13))
    Hq, bq, cq, nq = capi.unpack_reduced(block_i.numpy())
    ok = ok and nq == nf and np.allclose(cq, cf, rtol=1e-13) and np.allclose(Hq, Hf, rtol=1e-12, atol=1e-13) \
        and np.allclose(bq, bf, rtol=1e-12, atol=1e-13)
    if gathered is not None:
        ok = ok and np.array_equal(np.isnan(gathered[:, 0].numpy()), np.isnan(err)) \
            and np.array_equal(np.nan_to_num(gathered[:, 1:].numpy()), np.nan_to_num(J))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cell_sharding_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_cell_ranges_partition():
    import parallel_helpers as parallel   # tests/parallel_helpers.py: numpy / torch.distributed helpers of this test
    for ncell in (16, 256, 1024, 250):
        for world in (1, 2, 3, 4, 8):
            rs = parallel.all_ranges(world, ncell)
            assert rs[0][0] == 0 and rs[-1][1] == ncell
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.cell_range(2, 2, 16)


def test_pack_unpack_roundtrip(capi):
    import parallel_helpers as parallel   # tests/parallel_helpers.py: numpy / torch.distributed helpers of this test
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)); H = A + A.T; b = rng.normal(size=6)
    r = parallel.pack_reduced_np(H, b, 1.25, 7)
    H2, b2, c2, n2 = parallel.unpack_reduced_np(r)
    assert np.array_equal(H, H2) and np.array_equal(b, b2) and c2 == 1.25 and n2 == 7
    H3, b3, c3, n3 = capi.unpack_reduced(r)     # the C-ABI's layout is the same
    assert np.array_equal(H, H3) and np.array_equal(b, b3) and c3 == 1.25 and n3 == 7


def test_interleaved_cell_sets_partition():
    import parallel_helpers as parallel   # tests/parallel_helpers.py: numpy / torch.distributed helpers of this test
    for ncell in (16, 256, 1024, 250):
        for world in (1, 2, 3, 4, 8):
            sets = [parallel.cell_set(r, world, ncell, interleaved=True) for r in range(world)]
            allc = np.sort(np.concatenate(sets))
            assert np.array_equal(allc, np.arange(ncell))                    # exhaustive, disjoint
            assert max(len(x) for x in sets) - min(len(x) for x in sets) <= 1  # balanced
            assert all(np.all(np.diff(x) == world) for x in sets if len(x) > 1)
    with pytest.raises(ValueError):
        parallel.cell_set(0, 17, 16, interleaved=True)
