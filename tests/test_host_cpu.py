"""Host logic above the C-ABI (libnid_host.so) that needs no GPU: the Eigen-free
SE(3) algebra, dense LDLT and Huber kernel of include/g2o_min/g2o_min.h against
the oracle's restatement of the same reference formulas."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def hostlib():
    h = importlib.import_module("nid-pose-estimation_amd.hostlib")
    h.load()
    return h


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_se3_matches_oracle_bitwise(hostlib, oracle):
    rng = np.random.default_rng(11)
    p = oracle.se3_exp(rng.normal(0, 0.05, 6))
    for _ in range(50):
        upd = rng.normal(0, 0.02, 6)
        a, b = hostlib.se3_exp(upd), oracle.se3_exp(upd)
        assert np.array_equal(_bits(a), _bits(b))
        pa, pb = hostlib.se3_mul(a, p), oracle.se3_mul(b, p)
        assert np.array_equal(_bits(pa), _bits(pb))
        assert np.array_equal(_bits(hostlib.se3_to_matrix16(pa)), _bits(oracle.se3_to_matrix16(pb)))
        p = pa
    tiny = np.array([1e-7, -2e-7, 3e-7, 0.01, 0.02, -0.01])   # theta < 1e-5 branch (se3quat.h:238-243)
    assert np.array_equal(_bits(hostlib.se3_exp(tiny)), _bits(oracle.se3_exp(tiny)))


def test_minimal_vector(hostlib):
    p = np.array([0.1, -0.2, 0.3, 0.9, 1.0, 2.0, 3.0])
    np.testing.assert_array_equal(hostlib.minimal_vector(p), [1.0, 2.0, 3.0, 0.1, -0.2, 0.3])


def test_ldlt_matches_oracle_and_numpy(hostlib, oracle):
    rng = np.random.default_rng(12)
    for k in range(20):
        A = rng.normal(size=(6, 6))
        H = A @ A.T + 10.0 ** rng.integers(-3, 3) * np.eye(6)
        b = rng.normal(size=6)
        ok1, x1 = hostlib.ldlt6_solve(H, b)
        ok2, x2 = oracle.ldlt6_solve(H, b)
        assert ok1 and ok2
        assert np.array_equal(_bits(x1), _bits(x2))
        np.testing.assert_allclose(x1, np.linalg.solve(H, b), rtol=1e-8)
    ok, _ = hostlib.ldlt6_solve(-np.eye(6), np.ones(6))
    assert not ok


def test_huber_float_threshold(hostlib):
    delta = float(np.sqrt(0.95))
    dsqr = float(np.float32(delta * delta))      # robust_kernel_impl.h:84
    assert dsqr != delta * delta
    for e2 in (0.5, dsqr, np.nextafter(dsqr, 2.0), 0.97, 4.0):
        rho = hostlib.huber(e2, delta)
        if e2 <= dsqr:
            assert rho[0] == e2 and rho[1] == 1.0 and rho[2] == 0.0
        else:
            s = np.sqrt(e2)
            assert rho[0] == 2 * s * delta - dsqr and rho[1] == delta / s


def test_host_library_exports(hostlib):
    nm = subprocess.run(["nm", "-D", "--defined-only", hostlib.LIB_PATH], capture_output=True, text=True).stdout
    for sym in ("nid_host_run_lm", "nid_host_run_pyramid_lm", "nid_pyr_down_u8", "nid_pyr_down_depth_u16",
                "nid_host_standard_property", "nid_png_info", "nid_png_read_gray_u8", "nid_png_read_u16", "nid_legacy_reset", "nid_legacy_context", "nid_legacy_upload_count",
                "nid_legacy_set_jacobian_bound", "nid_legacy_set_trust_buffers", "nid_legacy_set_verify_mode", "nid_legacy_set_verify_slices", "nid_legacy_stale_detections", "nid_legacy_invalidate", "nid_legacy_set_devices", "nid_legacy_set_rank", "nid_legacy_multi",
                "nid_host_set_devices", "nid_host_set_rank"):
        assert re.search(rf" T {sym}\b", nm), sym
    # the three legacy operators keep their C++ linkage (mangled), as in the reference
    assert "_Z16Calculate3DpointPdS_S_S_ii" in nm
    assert "_Z15CudaComputeHrefPdS_S_S_iiiiiS_PiS0_S_" in nm
    assert re.search(r"_ZN3g2o12CudaComputeHEb", nm)
    # and the host library must not contain or link the oracle
    ldd = subprocess.run(["ldd", hostlib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "libnid_hip.so" in ldd


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_with_stub(tmp_path, caller, name, sanitizer="address"):
    """host/legacy_ops.cpp + a caller + tests/cpp/nid_hip_stub.cpp (the entry points of libnid_hip.so the operators
    call, without a GPU) as ONE sanitizer build (AddressSanitizer unless told otherwise)."""
    exe = tmp_path / name
    subprocess.check_call(["g++", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17", "-pthread",
                           "-I", os.path.join(ROOT, "include", "nid", "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", caller),
                           os.path.join(ROOT, "nid-pose-estimation_amd", "host", "legacy_ops.cpp"),
                           os.path.join(ROOT, "tests", "cpp", "nid_hip_stub.cpp"), "-o", str(exe)])
    return exe


def _write_lm_caller_input(tmp_path):
    """in.bin of tests/cpp/legacy_lm_caller.cpp: 640x480, 16x16 cells, 64 CudaComputeH calls, one undeclared in-place
    change of the target (a pixel of cell 0 off the fingerprint's 64 sampled indices) in front of call 13"""
    rows, cols, cell, nb, ncalls, change_at, pause_every = 480, 640, 16, 8, 64, 13, 8
    N, ncell = rows * cols, cell * cell
    rng = np.random.default_rng(3)
    change_px = (rows // cell // 2) * cols + cols // cell // 2          # inside cell 0, off the 64 sampled indices
    assert change_px not in {int(k * (N - 1) // 63) for k in range(64)}
    im1 = rng.integers(0, 256, N).astype(np.float64)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([rows, cols, cell, nb, ncalls, change_at, change_px, pause_every], dtype=np.int32).tofile(f)
        np.array([481.2, -480.0, 319.5, 239.5, 1 / 5000.0]).tofile(f)
        np.eye(4).reshape(-1).tofile(f)
        np.eye(4).reshape(-1).tofile(f)
        poses = np.tile(np.eye(4).reshape(-1), (ncalls, 1))
        poses[:, 12] = np.arange(ncalls) * 1e-3
        poses.tofile(f)
        (2.0 + rng.random(N)).tofile(f)
        rng.integers(0, 256, N).astype(np.float64).tofile(f)
        im1.tofile(f)
    return rows, cols, cell, nb, ncalls, change_at, change_px, im1


@pytest.mark.parametrize("threads", ["3", "0"])
def test_legacy_operators_read_caller_buffers_only_inside_a_call_asan(tmp_path, threads):
    """VERDICT r05 item 1 / ADVICE r05 (high), without a GPU: the reference's call pattern (tests/cpp/legacy_lm_caller.cpp:
    CudaComputeHref, 64 CudaComputeH calls with host work in between, one undeclared in-place change, then the frees of
    NID_pose_estimation.cpp:388-395 at once) on host/legacy_ops.cpp under AddressSanitizer, the HIP library stubbed.  A
    worker thread that reads a caller buffer after its call has returned is a heap-use-after-free here (round 5's
    default mode fails this test with exactly that report).  Default verification mode; with and without pool threads."""
    exe = _build_with_stub(tmp_path, "legacy_lm_caller.cpp", "legacy_lm_caller_asan")
    hdr = open(os.path.join(ROOT, "include", "nid", "legacy_ops.h")).read()
    per_call = int(re.search(r"#define NID_LEGACY_SLICES_PER_CALL (\d+)", hdr).group(1))
    SLICES = -(-int(re.search(r"#define NID_LEGACY_SLICES (\d+)", hdr).group(1)) // (per_call - per_call // 4))   # calls until every slice was checked (cost-only calls check 3/4 of per_call)
    rows, cols, cell, nb, ncalls, change_at, change_px, im1 = _write_lm_caller_input(tmp_path)
    ncell = cell * cell
    env = dict(os.environ, NID_LEGACY_HASH_THREADS=threads, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", MALLOC_PERTURB_="165")
    for drop in ("NID_LEGACY_TRUST_BUFFERS", "NID_LEGACY_VERIFY_EVERY_CALL", "NID_LEGACY_VERIFY_SLICES", "NID_LEGACY_ALWAYS_UPLOAD", "LD_PRELOAD"):
        env.pop(drop, None)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert "AddressSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-4000:]
    assert r.stderr.count("rewritten IN PLACE") == 1, r.stderr[-2000:]
    rec = np.fromfile(tmp_path / "out.bin", dtype=np.float64, offset=4 * ncell).reshape(ncalls, 8, ncell)
    # the stub's "evaluation" returns per-cell sums of the RESIDENT target: which content each call evaluated
    rb, cb = rows // cell, cols // cell
    old0 = -im1.reshape(rows, cols)[:rb, :cb].sum()
    new0 = old0 - (255.0 - 2 * im1[change_px])
    ht0 = rec[:, 0, 0]
    followed = int(np.argmax(ht0 == new0))
    assert np.all(ht0[:change_at] == old0) and change_at <= followed < change_at + SLICES, (followed, ht0)
    assert np.all(ht0[followed:] == new0) and np.all(ht0[:followed] == old0)
    assert np.all(rec[:, 0, 1:] == rec[0, 0, 1:])                       # no other cell's target changed


def test_legacy_operators_hash_pool_under_thread_sanitizer(tmp_path):
    """The same compiled caller under ThreadSanitizer (three pool workers): the pool claims parts on the job's own
    counter, wakes parked workers through a generation word and joins them inside the call -- no report of a data race
    between a worker and the caller (the results written by the workers are read behind finish()'s acquire)."""
    exe = _build_with_stub(tmp_path, "legacy_lm_caller.cpp", "legacy_lm_caller_tsan", sanitizer="thread")
    _write_lm_caller_input(tmp_path)
    env = dict(os.environ, NID_LEGACY_HASH_THREADS="3", TSAN_OPTIONS="halt_on_error=0")
    for drop in ("NID_LEGACY_TRUST_BUFFERS", "NID_LEGACY_VERIFY_EVERY_CALL", "NID_LEGACY_VERIFY_SLICES", "NID_LEGACY_ALWAYS_UPLOAD", "LD_PRELOAD"):
        env.pop(drop, None)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    if "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this environment")
    assert "ThreadSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-4000:]
    assert r.stderr.count("rewritten IN PLACE") == 1, r.stderr[-2000:]


def test_fork_between_legacy_calls_asan(tmp_path):
    """fork() after a CudaComputeH.  The hash pool's job lives inside one call, its fork handlers
    find the locks free; the child (no worker threads) hashes alone and still follows an in-place change.  The caller
    (tests/cpp/legacy_fork_caller.cpp) ends itself with SIGALRM if anybody hangs."""
    exe = _build_with_stub(tmp_path, "legacy_fork_caller.cpp", "legacy_fork_caller_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "fork ok" in r.stdout and "AddressSanitizer" not in r.stderr, (r.returncode, r.stderr[-3000:])


def test_pyramid_downsampling_matches_oracle(hostlib, oracle):
    """Own pyramid definition (host/nid_pyramid.cpp) against its numpy restatement: 2x2 box mean rounded half
    up for the images; valid-mean with 0 = invalid for the depth (holes, out-of-range values, mixed blocks)."""
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    im[:4] = 255
    im[4:8] = 0
    assert np.array_equal(hostlib.pyr_down_u8(im), oracle.pyr_down_u8(im))
    dep = rng.integers(40, 20000, (48, 64)).astype(np.uint16)
    dep[rng.random((48, 64)) < 0.3] = 0          # holes: blocks with 0..4 valid samples
    dep[10:14, 10:14] = 49                        # 0.0098 m: below the validity bound
    dep[20:22, 20:22] = 65535                     # 13.1 m: valid at 1/5000
    got, ref = hostlib.pyr_down_depth_u16(dep), oracle.pyr_down_depth_u16(dep)
    assert np.array_equal(got, ref)
    assert (got == 0).any() and (got > 0).any()
    # a block whose samples are all invalid stays invalid, a mixed block averages the valid ones only
    blk = np.array([[0, 10000], [0, 20000]], dtype=np.uint16)
    assert hostlib.pyr_down_depth_u16(blk)[0, 0] == 15000
    assert hostlib.pyr_down_depth_u16(np.zeros((2, 2), dtype=np.uint16))[0, 0] == 0
    # three levels of the synthetic pair keep 30x40-pixel cells
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    lv = oracle.pyramid_levels(synth.make_pair("A"), 3)
    assert [(p.rows, p.cols, p.cell) for p in lv] == [(480, 640, 16), (240, 320, 8), (120, 160, 4)]
    assert lv[1].fx == lv[0].fx / 2 and lv[1].cx == (lv[0].cx - 0.5) / 2
    assert np.array_equal(hostlib.pyr_down_u8(lv[0].im1), lv[1].im1)


def _png_bytes(img, depth, colour, filters, palette=None, split=1, interlace=0):
    """Reference PNG encoder for the reader test: every scanline filtered with the type the caller asks for."""
    import struct
    import zlib
    img = np.ascontiguousarray(img)
    raw = img.astype(">u2").tobytes() if depth == 16 else img.astype(np.uint8).tobytes()
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    bpp = ch * depth // 8
    stride = img.shape[1] * bpp
    rows = [bytearray(raw[r * stride:(r + 1) * stride]) for r in range(img.shape[0])]
    out = bytearray()
    for r, cur in enumerate(rows):
        ft = filters[r % len(filters)]
        up = rows[r - 1] if r else bytearray(stride)
        line = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = up[i]
            c = up[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (cur[i] - pred) & 0xFF
        out += bytes([ft]) + line

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    z = zlib.compress(bytes(out), 6)
    parts = [z[i * len(z) // split:(i + 1) * len(z) // split] for i in range(split)]
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], depth, colour, 0, 0, interlace))
    if palette is not None:
        body += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    body += chunk(b"tEXt", b"Comment\x00ancillary chunks are skipped")
    for part in parts:
        body += chunk(b"IDAT", part)
    return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IEND", b"")


def _cv_gray(rgb, swap_rb):
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    c0, c2 = (r, b) if swap_rb else (b, r)
    return ((c0 * 4899 + g * 9617 + c2 * 1868 + 8192) >> 14).astype(np.uint8)


def test_png_reader_and_grey_conversion(hostlib, tmp_path):
    """PNG decoding on zlib (all five scanline filters, 8/16 bit, grey / RGB / RGBA / palette,
    split IDAT) and the driver's colour -> grey conversion (OpenCV fixed point with the reference's channel
    order quirk).  OpenCV itself is not available: the conversion is checked against its published formula."""
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    rgb[0, 0] = (255, 255, 255); rgb[0, 1] = (0, 0, 0); rgb[0, 2] = (255, 0, 0); rgb[0, 3] = (0, 0, 255)
    for filters, split in (([0], 1), ([1], 1), ([2], 2), ([3], 1), ([4], 3), ([0, 1, 2, 3, 4], 2)):
        p = tmp_path / "rgb.png"
        p.write_bytes(_png_bytes(rgb, 8, 2, filters, split=split))
        assert np.array_equal(hostlib.png_read_gray_u8(str(p)), _cv_gray(rgb, False)), filters
        assert np.array_equal(hostlib.png_read_gray_u8(str(p), swap_rb=True), _cv_gray(rgb, True)), filters
    assert _cv_gray(rgb, False)[0, 2] == 29 and _cv_gray(rgb, True)[0, 2] == 76   # pure red: quirk vs luma
    rgba = np.concatenate([rgb, rng.integers(0, 256, (13, 17, 1), dtype=np.uint8)], axis=2)
    (tmp_path / "rgba.png").write_bytes(_png_bytes(rgba, 8, 6, [4, 3]))
    assert np.array_equal(hostlib.png_read_gray_u8(str(tmp_path / "rgba.png")), _cv_gray(rgb, False))
    grey = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    (tmp_path / "g8.png").write_bytes(_png_bytes(grey, 8, 0, [1, 4]))
    assert np.array_equal(hostlib.png_read_gray_u8(str(tmp_path / "g8.png")), grey)
    assert np.array_equal(hostlib.png_read_u16(str(tmp_path / "g8.png")), grey.astype(np.uint16))
    d16 = rng.integers(0, 65536, (9, 11)).astype(np.uint16)
    for filters in ([0], [2, 4, 1, 3]):
        (tmp_path / "d16.png").write_bytes(_png_bytes(d16, 16, 0, filters))
        assert np.array_equal(hostlib.png_read_u16(str(tmp_path / "d16.png")), d16)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (7, 5), dtype=np.uint8)
    (tmp_path / "pal.png").write_bytes(_png_bytes(idx, 8, 3, [0, 1], palette=pal))
    assert np.array_equal(hostlib.png_read_gray_u8(str(tmp_path / "pal.png")), _cv_gray(pal[idx], False))
    # interop with a real encoder, when one is around
    try:
        from PIL import Image
    except Exception:
        Image = None
    if Image is not None:
        Image.fromarray(rgb, "RGB").save(tmp_path / "pil_rgb.png")
        assert np.array_equal(hostlib.png_read_gray_u8(str(tmp_path / "pil_rgb.png")), _cv_gray(rgb, False))
        Image.fromarray(d16.astype(np.uint16)).save(tmp_path / "pil_d16.png")
        assert np.array_equal(hostlib.png_read_u16(str(tmp_path / "pil_d16.png")), d16)
    # refusals
    lib = hostlib.load()
    bad = tmp_path / "bad.png"
    bad.write_bytes(_png_bytes(rgb, 8, 2, [0])[:60])
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) in (-2, -4)
    bad.write_bytes(_png_bytes(rgb, 8, 2, [0], interlace=1))
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) == -3
    bad.write_bytes(b"P5\n2 2\n255\n1234")
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) == -2
    assert lib.nid_png_info(os.fsencode(str(tmp_path / "missing.png")), None, None, None, None) == -1
    with pytest.raises(RuntimeError):
        hostlib.png_read_u16(str(tmp_path / "rgb.png"))      # depth must be single-channel
    # hostile headers: sizes that would need gigabytes are refused before anything is allocated; nothing throws
    import struct
    import zlib
    good = _png_bytes(rgb, 8, 2, [0])

    def with_ihdr(cols, rows):
        ihdr = struct.pack(">IIBBBBB", cols, rows, 8, 2, 0, 0, 0)
        return good[:16] + ihdr + struct.pack(">I", zlib.crc32(b"IHDR" + ihdr) & 0xffffffff) + good[33:]
    bad.write_bytes(with_ihdr(1 << 20, 1 << 20))            # 2^40 pixels
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) == -3
    bad.write_bytes(with_ihdr(8000, 8000))                   # under the pixel cap, but far more than 60 bytes of IDAT inflate to
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) == -2
    corrupt = bytearray(good)
    corrupt[-20] ^= 0x40                                      # a flipped bit in the IDAT payload: the chunk CRC catches it
    bad.write_bytes(bytes(corrupt))
    assert lib.nid_png_info(os.fsencode(str(bad)), None, None, None, None) == -2


def test_png_reader_survives_mutated_files(tmp_path):
    """host/nid_png.cpp reads files a driver is pointed at: 4 000 mutants of five valid PNGs (tests/cpp/png_fuzz.cpp: bit flips,
    truncations, rewritten header fields and chunk lengths, inserted runs -- chunk CRCs re-made for most, so the parser behind
    the CRC check sees them) under AddressSanitizer + UndefinedBehaviorSanitizer: refused or decoded, never a report."""
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (13, 17, 1), dtype=np.uint8)], axis=2)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    seeds = {"rgb": _png_bytes(rgb, 8, 2, [0, 1, 2, 3, 4], split=2), "rgba": _png_bytes(rgba, 8, 6, [4, 3]),
             "d16": _png_bytes(rng.integers(0, 65536, (9, 11)).astype(np.uint16), 16, 0, [2, 4, 1, 3]),
             "pal": _png_bytes(rng.integers(0, 16, (7, 5), dtype=np.uint8), 8, 3, [0, 1], palette=pal),
             "g8": _png_bytes(rng.integers(0, 256, (9, 11), dtype=np.uint8), 8, 0, [1, 4])}
    paths = []
    for name, data in seeds.items():
        (tmp_path / f"{name}.png").write_bytes(data)
        paths.append(str(tmp_path / f"{name}.png"))
    exe = tmp_path / "png_fuzz"
    subprocess.check_call(["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-O1", "-g", "-std=c++17",
                           "-I", os.path.join(ROOT, "nid-pose-estimation_amd", "host"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "png_fuzz.cpp"),
                           os.path.join(ROOT, "nid-pose-estimation_amd", "host", "nid_png.cpp"), "-lz", "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), "4000", str(tmp_path)] + paths, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    m = re.search(r"4000 mutants: (\d+) decoded, (\d+) refused", r.stdout)
    assert m and int(m.group(1)) > 20 and int(m.group(2)) > 2000, r.stdout


def test_driver_refuses_cpu_mode(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "nid-pose-estimation_amd", "nid_pose_estimation")
    cfg = tmp_path / "c.yaml"
    cfg.write_text("%YAML:1.0\nimage0_id: '0000'\nimage1_id: '0001'\nimage0_type: rgb\nimage1_type: rgb\n"
                   "dataset: eth_cvg\nim_address: /nonexistent/\ndepth_factor: 5000.0\nfx: 1\nfy: 1\ncx: 0\ncy: 0\nuse_gpu: 0\n")
    r = subprocess.run([exe, str(cfg)], capture_output=True, text=True)
    assert r.returncode == 2 and "use_gpu" in r.stderr


@pytest.mark.parametrize("compiler", ["g++", "/opt/rocm/lib/llvm/bin/clang++"])
def test_direct_results_avx512_sums_equal_the_scalar_loop(tmp_path, compiler):
    """csrc/nid_hostsum.cpp (the host's part of a DIRECT launch on a CPU with AVX-512: one load per record -- arrival
    test, the 27 products of the normal equations, re-arming) against the scalar loop of csrc/nid_capi.hip, bit for bit
    on 200 000 random records (zeros, signed zeros, 600 binades of magnitudes, inactive cells, records with a word still
    missing); built WITHOUT -ffp-contract=off on purpose: the file must not depend on the library's build flags.
    No GPU involved; tests/test_parity_gpu.py::test_direct_results_equal_in_launch_reduction is the end-to-end check."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which(compiler) or os.path.exists(compiler)):
        pytest.skip(f"{compiler} not available")
    exe = str(tmp_path / "hostsum_check")
    subprocess.check_call([compiler, "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "hostsum_check.cpp"),
                           os.path.join(root, "nid-pose-estimation_amd", "csrc", "nid_hostsum.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if r.returncode == 2:
        pytest.skip("no AVX-512 on this CPU: the library takes the scalar loop here")
    assert r.returncode == 0, r.stdout[-2000:]
