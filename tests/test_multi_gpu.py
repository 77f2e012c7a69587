"""include/nid/nid_multi.h on a GPU: the cells of one frame pair sharded over several shard contexts in C++
(BASELINE configs[3] and [4]).  One GPU is enough: a device list may repeat a device
({0,0} = two shards on GPU 0, summed on the host), and RCCL runs with a communicator of one rank, which proves
that librccl loads, that the all-reduce is ordered behind the kernel on the shard's stream and that the result
comes home.  More than one rank per device is refused by RCCL itself; the 2/4/8-GPU runs are the driver's."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DELTA = float(np.sqrt(0.95))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def hostlib():
    h = importlib.import_module("nid-pose-estimation_amd.hostlib")
    h.load()
    return h


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_shards_equal_single_context(capi, synth, pair_A, nshards):
    """Per-cell outputs of the shards are the single context's bit for bit (a cell does not know how the image
    was split); the summed 6x6 systems agree to rounding; the host sum is bitwise reproducible."""
    pair, nb = pair_A, 8
    one = capi.from_pair(pair, nb)
    cnt, href = one.compute_href(pair.pose_init)
    m = capi.multi_from_pair(pair, nb, devices=[0] * nshards)
    assert m.shards() == nshards
    cnt_m, href_m = m.compute_href(pair.pose_init)
    assert np.array_equal(cnt, cnt_m) and np.array_equal(_bits(href), _bits(href_m))
    act = cnt >= 300
    poses = [pair.pose_init, pair.pose_true, synth.perturb_pose7(pair.pose_init, [1e-3, 0, 2e-3], [0, 3e-3, 0])]
    for pose in poses:
        a, b = one.evaluate(pose, True), m.evaluate(pose, True)
        for x, y in zip(a, b):
            assert np.array_equal(_bits(x[act]), _bits(y[act]))
            assert np.isnan(y[~act]).all()
        H, bb, chi2, na = one.normal_equations(pose, DELTA)
        Hm, bm, chi2m, nam = m.normal_equations(pose, DELTA)
        assert na == nam == int(act.sum())
        np.testing.assert_allclose(chi2m, chi2, rtol=1e-13)
        np.testing.assert_allclose(Hm, H, rtol=0, atol=1e-12 * np.abs(H).max())
        np.testing.assert_allclose(bm, bb, rtol=0, atol=1e-12 * np.abs(bb).max())
        Hm2, bm2, chi2m2, _ = m.normal_equations(pose, DELTA)
        assert np.array_equal(_bits(Hm), _bits(Hm2)) and np.array_equal(_bits(bm), _bits(bm2)) and chi2m == chi2m2
    # a batch of trial poses in one launch per shard; a pending slot is refused until collected
    m.launch_batch(4, poses, DELTA, want_jac=False)
    with pytest.raises(capi.NidError):
        m.launch_batch(5, poses[:1], DELTA)
    for k, pose in enumerate(poses):
        _, _, chi2k, nak = m.wait(4 + k)
        _, _, chi2s, _ = m.normal_equations(pose, DELTA, want_jac=False)
        assert chi2k == chi2s and nak == int(act.sum())
    with pytest.raises(capi.NidError):
        m.wait(4)                                   # nothing pending any more
    # nid_multi_launch_chain: the rejection chain of an LM iteration, first trial(s) with the Jacobian phase
    m.launch_chain(0, poses, 1, DELTA)
    for k, pose in enumerate(poses):
        Hk, bk, chi2k, _ = m.wait(k)
        Hs, bs, chi2s, _ = m.normal_equations(pose, DELTA, want_jac=(k == 0))
        assert chi2k == chi2s and np.array_equal(_bits(Hk), _bits(Hs)) and np.array_equal(_bits(bk), _bits(bs))
    with pytest.raises(capi.NidError):
        capi.multi_from_pair(synth.make_pair("S"), nb, devices=[0] * 17)   # more shards than cells


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_interleaved_partition_equals_single_context(capi, synth, pair_A, nshards):
    """NID_PARTITION_INTERLEAVED (shard k owns cells k, k + K, ...): per-cell outputs are the single context's bit for
    bit, the summed 6x6 systems agree to rounding, the reference-stage outputs (counts, Href, per-pixel weights and bin
    indices) land in the right cells, and the Href state handed back in (the legacy operator's path) gives the same
    evaluation -- same checks as for contiguous ranges."""
    pair, nb = pair_A, 8
    one = capi.from_pair(pair, nb)
    cnt, href, bsv1, bsi1 = one.compute_href(pair.pose_init, dump=True)
    m = capi.multi_from_pair(pair, nb, devices=[0] * nshards, partition=capi.PARTITION_INTERLEAVED)
    assert m.shards() == nshards
    cnt_m, href_m, bsv, bsi = m.compute_href(pair.pose_init, dump=True)
    assert np.array_equal(cnt, cnt_m) and np.array_equal(_bits(href), _bits(href_m))
    assert np.array_equal(_bits(bsv), _bits(bsv1)) and np.array_equal(bsi, bsi1)
    act = cnt >= 300
    poses = [pair.pose_init, pair.pose_true, synth.perturb_pose7(pair.pose_init, [1e-3, 0, 2e-3], [0, 3e-3, 0])]
    for pose in poses:
        a, b = one.evaluate(pose, True), m.evaluate(pose, True)
        for x, y in zip(a, b):
            assert np.array_equal(_bits(x[act]), _bits(y[act])) and np.isnan(y[~act]).all()
        H, bb, chi2, na = one.normal_equations(pose, DELTA)
        Hm, bm, chi2m, nam = m.normal_equations(pose, DELTA)
        assert na == nam == int(act.sum())
        np.testing.assert_allclose(chi2m, chi2, rtol=1e-13)
        np.testing.assert_allclose(Hm, H, rtol=0, atol=1e-12 * np.abs(H).max())
        np.testing.assert_allclose(bm, bb, rtol=0, atol=1e-12 * np.abs(bb).max())
    m2 = capi.multi_from_pair(pair, nb, devices=[0] * nshards, partition=capi.PARTITION_INTERLEAVED)
    m2.set_href_state(cnt, href, bsv1, bsi1)
    for x, y in zip(one.evaluate(poses[2], True), m2.evaluate(poses[2], True)):
        assert np.array_equal(_bits(x[act]), _bits(y[act]))
    seq = np.stack([poses[i % 3] for i in range(200)])
    ra = m.run_sequence(seq, DELTA, batch=64, group=2)
    for i in (0, 100, 199):
        Hs, bs, chis, _ = m.normal_equations(seq[i], DELTA)
        assert np.array_equal(_bits(capi.unpack_reduced(ra[i])[0]), _bits(Hs))
    with pytest.raises(capi.NidError):
        capi.multi_from_pair(pair, nb, devices=[0, 0], rank=0, world=2, partition=capi.PARTITION_INTERLEAVED)   # one shard per process
    with pytest.raises(capi.NidError):
        capi.multi_from_pair(pair, nb, devices=[0], partition=7)


def test_rccl_one_rank_communicator(capi, synth, pair_A):
    """RCCL from C++: ncclCommInitRank (1 rank) through the library, ncclAllReduce(ncclDouble) in-stream behind the
    evaluation kernel, result copied to pinned host memory -- equal to the host-summed path bit for bit (a sum
    over one rank is the identity), for single evaluations, batches and the pipelined loop."""
    pair, nb = pair_A, 8
    host = capi.multi_from_pair(pair, nb, devices=[0])
    host.compute_href(pair.pose_init)
    m = capi.multi_from_pair(pair, nb, devices=[0], rank=0, world=1)
    m.compute_href(pair.pose_init)
    m.comm_init(capi.rccl_unique_id())
    assert m.comm_ranks() == 1
    poses = [synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 2e-4 * k, 1e-4]) for k in range(40)]
    for pose in poses[:3]:
        a, b = host.normal_equations(pose, DELTA), m.normal_equations(pose, DELTA)
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1])) and a[2:] == b[2:]
    m.launch_batch(10, poses[:20], DELTA)
    host.launch_batch(10, poses[:20], DELTA)
    for k in range(20):
        a, b = host.wait(10 + k), m.wait(10 + k)
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and a[2] == b[2]
    # an LM rejection chain (first pose with the Jacobian phase, the rest cost only) through the all-reduce
    for n_jac in (0, 1, 3):
        m.launch_chain(40, poses[:7], n_jac, DELTA)
        host.launch_chain(40, poses[:7], n_jac, DELTA)
        for k in range(7):
            a, b = host.wait(40 + k), m.wait(40 + k)
            assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1])) and a[2:] == b[2:]
            assert (np.abs(a[0]).max() > 0) == (k < n_jac)
    seq = np.stack([poses[i % len(poses)] for i in range(300)])
    ra = host.run_sequence(seq, DELTA, batch=16, group=4)
    rb = m.run_sequence(seq, DELTA, batch=16, group=4)
    assert np.array_equal(_bits(ra), _bits(rb))
    H, b, chi2, na = host.normal_equations(seq[299], DELTA)
    assert np.array_equal(_bits(capi.unpack_reduced(rb[299])[0]), _bits(H))
    # RCCL refuses two ranks on one device: the library says so instead of hanging
    two = capi.multi_from_pair(pair, nb, devices=[0, 0])
    with pytest.raises(capi.NidError):
        two.comm_init_local()
    with pytest.raises(capi.NidError):
        two.set_reduce_mode(capi.REDUCE_RCCL)


def test_two_physical_devices_over_rccl(capi, synth, pair_A):
    """Two shards on two DISTINCT devices of one process: nid_multi_comm_init_local (ncclCommInitAll: RCCL over xGMI, no
    launcher) and the pipelined sequence with one all-reduce per group -- every pose's summed block equals the
    single-device evaluation's up to the summation tree (chi2, b, H to 1e-12 relative), the active count exactly.
    Needs two GPUs: skipped on a one-GPU box (RCCL refuses two ranks on one device, see the test above)."""
    if capi.load().nid_device_count() < 2:
        pytest.skip("needs two GPUs")
    pair, nb = pair_A, 8
    host = capi.from_pair(pair, nb)
    host.compute_href(pair.pose_init)
    m = capi.multi_from_pair(pair, nb, devices=[0, 1])
    m.compute_href(pair.pose_init)
    m.comm_init_local()
    assert m.comm_ranks() == 2
    m.set_reduce_mode(capi.REDUCE_RCCL)
    rng = np.random.default_rng(11)
    poses = np.stack([synth.perturb_pose7(pair.pose_init, rng.normal(0, 1e-3, 3), rng.normal(0, 2e-3, 3)) for _ in range(24)])
    seq = poses[np.arange(2 * 16 * 4 + 19) % len(poses)]
    out = m.run_sequence(seq, DELTA, batch=16, group=4)
    ref = host.run_sequence(seq, DELTA, batch=16)
    assert out.shape == ref.shape
    assert np.array_equal(out[:, 28], ref[:, 28])
    np.testing.assert_allclose(out[:, :28], ref[:, :28], rtol=1e-12, atol=1e-12 * np.abs(ref[:, :28]).max())
    for i in (0, 17, len(seq) - 1):
        H, b, chi2, na = m.normal_equations(seq[i], DELTA)
        Hs, bs, chi2s, nas = capi.unpack_reduced(out[i])
        assert np.array_equal(_bits(H), _bits(Hs)) and np.array_equal(_bits(b), _bits(bs)) and chi2 == chi2s and na == nas
    m.close(); host.close()


@pytest.mark.parametrize("batch,group", [(64, 1), (16, 4), (8, 3)])
def test_pipelined_sequence_on_shards(capi, synth, pair_A, batch, group):
    """nid_multi_run_sequence (the bench's timed region at N > 1): groups of launches on two streams per shard, one
    exchange per group, two groups in flight -- every pose's summed block equals its synchronous evaluation."""
    pair, nb = pair_A, 8
    m = capi.multi_from_pair(pair, nb, devices=[0, 0, 0])
    m.compute_href(pair.pose_init)
    rng = np.random.default_rng(5)
    poses = np.stack([synth.perturb_pose7(pair.pose_init, rng.normal(0, 1e-3, 3), rng.normal(0, 2e-3, 3)) for _ in range(40)])
    n = 2 * batch * group * 2 + batch + 3          # several full groups, a partial group, a partial launch
    seq = poses[np.arange(n) % len(poses)]
    out = m.run_sequence(seq, DELTA, batch=batch, group=group)
    assert out.shape == (n, capi.NID_REDUCED_LEN) and np.all(np.isfinite(out))
    for i in (0, 1, batch, n // 2, n - 1):
        H, b, chi2, na = m.normal_equations(seq[i], DELTA)
        Hs, bs, chi2s, nas = capi.unpack_reduced(out[i])
        assert np.array_equal(_bits(H), _bits(Hs)) and np.array_equal(_bits(b), _bits(bs)) and chi2 == chi2s and na == nas


def test_lm_on_shards_through_the_host_library(hostlib, synth, pair_A):
    """BASELINE configs[3]: the reference driver's optimisation with the cells of the pair on 1 / 2 / 4 shards
    (libnid_host.so: same g2o-shaped stack, same LM, the operators and the fused path run on a nid_multi).
    Per-edge flow: per-cell outputs are the same bits -> the whole optimisation is the same bits.  Fused flow:
    the 6x6 sums differ by their summation tree only."""
    nb = 8
    try:
        hostlib.set_devices([0])
        ref_edge = hostlib.run_lm(pair_A, nb, pair_A.pose_init, 10, fused=0)
        ref_fused = hostlib.run_lm(pair_A, nb, pair_A.pose_init, 10, fused=2)
        for devs in ([0, 0], [0, 0, 0, 0]):
            hostlib.set_devices(devs)
            pose, recs, _ = hostlib.run_lm(pair_A, nb, pair_A.pose_init, 10, fused=0)
            assert np.array_equal(_bits(pose), _bits(ref_edge[0]))
            assert [r["chi2"] for r in recs] == [r["chi2"] for r in ref_edge[1]]
            pose, recs, _ = hostlib.run_lm(pair_A, nb, pair_A.pose_init, 10, fused=2)
            assert [r["lm_trials"] for r in recs] == [r["lm_trials"] for r in ref_fused[1]]
            np.testing.assert_allclose([r["chi2"] for r in recs], [r["chi2"] for r in ref_fused[1]], rtol=1e-11)
            np.testing.assert_allclose(synth.pose7_minimal(pose), synth.pose7_minimal(ref_fused[0]), rtol=0, atol=1e-9)
    finally:
        hostlib.set_devices([0])


def test_pyramid_on_shards(hostlib, oracle, synth, pair_A):
    """BASELINE configs[4]: 3-level coarse-to-fine schedule x 10 LM iterations per level with every level's cells
    sharded (256 / 64 / 16 cells over 4 and 8 shards: two cells per shard on the coarsest level), against the
    oracle's restatement of the schedule and against the unsharded run."""
    nb = 8
    pose_o, per_o = oracle.pyramid_lm(pair_A, nb, pair_A.pose_init, levels=3, iterations=10)
    try:
        hostlib.set_devices([0])
        pose_1, per_1, _ = hostlib.run_pyramid_lm(pair_A, nb, pair_A.pose_init, levels=3, iterations=10, fused=2)
        for devs in ([0] * 4, [0] * 8):
            hostlib.set_devices(devs)
            pose, per, log = hostlib.run_pyramid_lm(pair_A, nb, pair_A.pose_init, levels=3, iterations=10, fused=2)
            assert [[r["lm_trials"] for r in lv] for lv in per] == [[r["lm_trials"] for r in lv] for lv in per_o]
            d = np.abs(synth.pose7_minimal(pose) - synth.pose7_minimal(pose_o)).max()
            print(f"pyramid on {len(devs)} shards: max |pose - pose_oracle| = {d:.3e}")
            assert d < 1e-8
            np.testing.assert_allclose(synth.pose7_minimal(pose), synth.pose7_minimal(pose_1), rtol=0, atol=1e-9)
        hostlib.set_devices([0] * 17)               # more shards than the coarsest level has cells: refused, not wrong
        with pytest.raises(RuntimeError):
            hostlib.run_pyramid_lm(pair_A, nb, pair_A.pose_init, levels=3, iterations=2, fused=2)
        with pytest.raises(RuntimeError):
            hostlib.run_lm(synth.make_pair("S"), nb, pair_A.pose_init, 2)
    finally:
        hostlib.set_devices([0])


def test_legacy_operators_follow_buffer_contents(hostlib, oracle, synth, pair_S, pair_S_edge):
    """A caller that REFILLS its buffers in place for the next frame pair (same addresses: what malloc hands back
    after a free, NID_pose_estimation.cpp:229-251, 385-392) gets the new pair's results: the resident state is
    keyed on content, and CudaComputeHref always uploads."""
    lib = hostlib.load()
    lib.nid_legacy_reset()
    nb = 8
    N, ncell = pair_S.rows * pair_S.cols, pair_S.cell ** 2
    dp = lambda a: a.ctypes.data_as(hostlib.c_dp)
    ip = lambda a: a.ctypes.data_as(hostlib.c_ip)
    depth = np.zeros(N); T = np.zeros(16); intr = pair_S.intr.copy(); pts = np.zeros(3 * N)
    im0 = np.zeros(N); im1 = np.zeros(N); bsv = np.zeros(4 * N); bsi = np.zeros(N, dtype=np.int32)
    cnt = np.zeros(ncell, dtype=np.int32); href = np.zeros(ncell)
    for pair in (pair_S, pair_S_edge, pair_S):
        depth[:] = pair.depth_m.reshape(-1); T[:] = synth.matrix_colmajor16(pair.T_wc0)
        im0[:] = pair.im0.reshape(-1); im1[:] = pair.im1.reshape(-1); href[:] = 0.0
        M0 = oracle.se3_to_matrix16(pair.pose_init)
        lib.nid_legacy_call_Calculate3Dpoint(dp(depth), dp(T), dp(pts), dp(intr), pair.rows, pair.cols)
        lib.nid_legacy_call_CudaComputeHref(dp(im0), dp(pts), dp(M0), dp(intr), nb, 3, pair.cell, pair.rows, pair.cols,
                                            dp(bsv), ip(bsi), ip(cnt), dp(href))
        o = oracle.from_pair(pair, nb, jac_bound="cpu", xform="matrix")
        cnt_o, href_o = o.compute_href(pair.pose_init)
        assert np.array_equal(cnt, cnt_o)
        act = cnt_o >= 300
        Ht = np.zeros(ncell); Hj = np.zeros(ncell); der = np.zeros(6 * ncell)
        M = oracle.se3_to_matrix16(pair.pose_true)
        lib.nid_legacy_call_CudaComputeH(1, dp(im0), dp(im1), dp(pts), ip(cnt), dp(bsv), ip(bsi), dp(M), dp(intr),
                                         nb, 3, pair.cell, pair.rows, pair.cols, dp(href), dp(Ht), dp(Hj), dp(der))
        Hc_o, Hj_o, err_o, J_o = o.evaluate(pair.pose_true, True)
        np.testing.assert_allclose(Ht[act], Hc_o[act], rtol=0, atol=1e-11)
        np.testing.assert_allclose(Hj[act], Hj_o[act], rtol=0, atol=1e-11)
        # only the target changes in place (a new second frame against the same reference): followed as well
        im1[:] = np.roll(pair.im1, 3, axis=1).reshape(-1)
        o.set_target(np.roll(pair.im1, 3, axis=1))
        Ht2 = np.zeros(ncell); Hj2 = np.zeros(ncell)
        lib.nid_legacy_call_CudaComputeH(0, dp(im0), dp(im1), dp(pts), ip(cnt), dp(bsv), ip(bsi), dp(M), dp(intr),
                                         nb, 3, pair.cell, pair.rows, pair.cols, dp(href), dp(Ht2), dp(Hj2), dp(der))
        Hc_o2, Hj_o2, _, _ = o.evaluate(pair.pose_true, False)
        np.testing.assert_allclose(Ht2[act], Hc_o2[act], rtol=0, atol=1e-11)
        assert not np.allclose(Ht2[act], Ht[act])
    # A change IN PLACE that touches none of the 64 sampled elements of the quick fingerprint (one pixel in the middle of
    # a cell).  NID_LEGACY_VERIFY_EVERY_CALL: every call reads the caller's buffers in full (include/nid/legacy_ops.h), so an
    # UNDECLARED change is followed on the NEXT call -- of any of the four big buffers -- and nothing is uploaded while nothing changes.
    pair = pair_S
    lib.nid_legacy_invalidate.argtypes = [hostlib.C.c_uint]
    lib.nid_legacy_set_trust_buffers.argtypes = [hostlib.C.c_int]
    lib.nid_legacy_set_verify_mode.argtypes = [hostlib.C.c_int]
    lib.nid_legacy_stale_detections.restype = hostlib.C.c_long
    VERIFY_ROTATING, VERIFY_EVERY_CALL = 0, 1
    lib.nid_legacy_set_verify_mode(VERIFY_EVERY_CALL)   # (the strongest of the three modes first; the default is further down)
    samples = {int(k * (N - 1) // 63) for k in range(64)}
    cellpx = (pair.rows // pair.cell // 2) * pair.cols + pair.cols // pair.cell // 2      # inside cell 0
    assert cellpx not in samples

    def evaluate():   # (Htarget | Hjoint per cell: the target's entropy does not see the reference side, the joint one does)
        a, b = np.zeros(ncell), np.zeros(ncell)
        lib.nid_legacy_call_CudaComputeH(0, dp(im0), dp(im1), dp(pts), ip(cnt), dp(bsv), ip(bsi), dp(M), dp(intr),
                                         nb, 3, pair.cell, pair.rows, pair.cols, dp(href), dp(a), dp(b), dp(der))
        return np.stack([a, b], axis=1)

    base = evaluate()
    u0 = lib.nid_legacy_upload_count()
    for k in range(5):
        assert np.array_equal(evaluate(), base)
    assert lib.nid_legacy_upload_count() == u0                     # verified on every call, uploaded on none
    im1[cellpx] = 255.0 - im1[cellpx]                              # undeclared
    changed = evaluate()
    assert lib.nid_legacy_upload_count() == u0 + 1 and changed[0, 0] != base[0, 0] and np.array_equal(changed[1:], base[1:])
    im1[cellpx] = 255.0 - im1[cellpx]                              # back, undeclared again
    assert np.array_equal(evaluate(), base) and lib.nid_legacy_upload_count() == u0 + 2
    # the reference side: one reference weight of one pixel (bs_ref), one reference pixel (im0 + its point)
    wpx = 4 * cellpx
    assert all(wpx + q not in {int(k * (4 * N - 1) // 63) for k in range(64)} for q in range(4))
    keep = bsv[wpx:wpx + 4].copy()
    bsv[wpx:wpx + 4] = keep[::-1]
    c2 = evaluate()
    assert lib.nid_legacy_upload_count() == u0 + 3 and c2[0, 1] != base[0, 1] and np.array_equal(c2[1:], base[1:])
    bsv[wpx:wpx + 4] = keep
    assert np.array_equal(evaluate(), base) and lib.nid_legacy_upload_count() == u0 + 4
    pk = pts[3 * cellpx:3 * cellpx + 3].copy()
    assert all(3 * cellpx + q not in {int(k * (3 * N - 1) // 63) for k in range(64)} for q in range(3))
    pts[3 * cellpx:3 * cellpx + 3] = np.nan                        # the pixel loses its depth
    c3 = evaluate()
    # (a new reference invalidates the device's reference weights: the caller's bs_ref / counters / Href go up again too)
    assert lib.nid_legacy_upload_count() == u0 + 6 and np.array_equal(c3[1:], base[1:])
    pts[3 * cellpx:3 * cellpx + 3] = pk
    assert np.array_equal(evaluate(), base)
    u1 = lib.nid_legacy_upload_count()
    assert np.array_equal(evaluate(), base) and lib.nid_legacy_upload_count() == u1

    # The DEFAULT, NID_LEGACY_VERIFY_ROTATING (round 6): every call checks the cheap keys and a few of the 128 slices of each big
    # buffer -- hashed by the pool's workers while the device evaluates, joined before the call returns -- so an undeclared
    # change in place is found within 43 calls (35 in the LM's pattern), by a call that says so (stderr, nid_legacy_stale_detections), uploads the
    # new content and evaluates it before it returns.  Nothing of the caller's is read between calls.
    lib.nid_legacy_set_verify_mode(VERIFY_ROTATING)
    assert np.array_equal(evaluate(), base)
    d0, u0 = lib.nid_legacy_stale_detections(), lib.nid_legacy_upload_count()
    im1[cellpx] = 255.0 - im1[cellpx]                              # undeclared
    got, calls = None, 0
    SLICES = 43                                                    # ceil(NID_LEGACY_SLICES / (3/4 NID_LEGACY_SLICES_PER_CALL)): cost-only calls (include/nid/legacy_ops.h)
    for k in range(SLICES):                                        # (up to SLICES - 1 calls may still see the old content)
        got = evaluate(); calls += 1
        if not np.array_equal(got, base):
            break
    assert got[0, 0] != base[0, 0] and np.array_equal(got[1:], base[1:]), "the change was not followed within NID_LEGACY_SLICES calls"
    assert lib.nid_legacy_stale_detections() == d0 + 1 and lib.nid_legacy_upload_count() == u0 + 1
    assert np.array_equal(evaluate(), got)                         # ... and stays followed
    im1[cellpx] = 255.0 - im1[cellpx]
    lib.nid_legacy_invalidate(2)                                   # declared: at once, and not counted as a detection
    assert np.array_equal(evaluate(), base) and lib.nid_legacy_stale_detections() == d0 + 1
    for k in range(40):                                            # nothing changes: nothing is reported or uploaded
        assert np.array_equal(evaluate(), base)
    assert lib.nid_legacy_stale_detections() == d0 + 1 and lib.nid_legacy_upload_count() == u0 + 2
    # a change in the LAST slice of the largest buffer (bs_ref's final row: not a sampled index either), 4 slices per call
    lib.nid_legacy_set_verify_slices.argtypes = [hostlib.C.c_int]
    lib.nid_legacy_set_verify_slices(16)                           # four times the default
    lastpx = N - 2
    keep = bsv[4 * lastpx:4 * lastpx + 4].copy()
    assert all(4 * lastpx + q not in {int(k * (4 * N - 1) // 63) for k in range(64)} for q in range(4))
    bsv[4 * lastpx:4 * lastpx + 4] = np.where(np.isnan(keep), 0.0, keep + 0.25)
    for k in range(11):                                            # 128 slices, 12 per cost-only call
        got = evaluate()
        if lib.nid_legacy_stale_detections() == d0 + 2:
            break
    assert lib.nid_legacy_stale_detections() == d0 + 2, "a change in the last slice was not found within 11 calls"
    bsv[4 * lastpx:4 * lastpx + 4] = keep
    lib.nid_legacy_invalidate(4)
    assert np.array_equal(evaluate(), base)
    lib.nid_legacy_set_verify_slices(0)

    # TRUSTED buffers (opt-in, round 4's default): address + length + 64 samples per call; a change in place is declared
    # with nid_legacy_invalidate and followed at once, or undeclared and followed within 128 calls of the pair.
    lib.nid_legacy_set_trust_buffers(1)
    try:
        assert np.array_equal(evaluate(), base)
        u0 = lib.nid_legacy_upload_count()
        im1[cellpx] = 255.0 - im1[cellpx]
        lib.nid_legacy_invalidate(2)                      # NID_LEGACY_TARGET
        changed = evaluate()
        assert lib.nid_legacy_upload_count() == u0 + 1 and changed[0, 0] != base[0, 0] and np.array_equal(changed[1:], base[1:])
        im1[cellpx] = 255.0 - im1[cellpx]                 # back, undeclared this time
        assert not np.array_equal(evaluate(), base)       # ... and not noticed by the next call: that is the trade
        seen_after = None
        for k in range(130):
            if np.array_equal(evaluate(), base):
                seen_after = k
                break
        assert seen_after is not None and seen_after <= 128 and lib.nid_legacy_upload_count() == u0 + 2
        for k in range(130):                               # the next periodic check finds every key complete
            assert np.array_equal(evaluate(), base)
        assert lib.nid_legacy_upload_count() == u0 + 2
    finally:
        lib.nid_legacy_set_trust_buffers(0)
    lib.nid_legacy_set_verify_mode(VERIFY_EVERY_CALL)
    # the small per-cell arrays are fully hashed on every call: a changed count is followed at once
    cnt_keep = cnt.copy()
    cnt[0] = 0
    assert np.isnan(evaluate()[0, 0])
    cnt[:] = cnt_keep
    assert np.array_equal(evaluate(), base)
    lib.nid_legacy_set_verify_mode(VERIFY_ROTATING)
    lib.nid_legacy_reset()


def test_bench_two_ranks_with_an_rccl_that_never_returns(tmp_path):
    """bench.py --gpus 2 with NID_RCCL_LIBRARY pointing at a librccl whose ncclCommInitRank never returns (a missing
    rank, a fabric that is down): instead of hanging until the driver's timeout with no record, the ranks' deadline
    ends the job with ONE JSON error line on stdout and a non-zero exit code.  (Both ranks share the one GPU of the box:
    nothing gets as far as a collective.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", stub, os.path.join(root, "tests", "cpp", "rccl_stub.c")])
    env = dict(os.environ, NID_RCCL_LIBRARY=stub, NID_BENCH_COMM_DEADLINE="5", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--quick",
           "--no-cpu-baseline", "--preheat-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode != 0, "a job whose communicator never comes up must not report success"
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and "ncclCommInitRank did not return" in lines[0]["error"] and lines[0]["value"] is None, r.stdout + r.stderr[-2000:]
