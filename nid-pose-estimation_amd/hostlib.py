"""ctypes binding of libnid_host.so: the C++ host stack above the C-ABI (g2o-shaped
API, legacy operator wrappers, the reference driver's optimisation).  Plumbing for
tests and tools; fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnid_host.so")
_lib = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class PoseProblem(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("cell_num", C.c_int32), ("bin_num", C.c_int32),
                ("iterations", C.c_int32), ("jac_bound_cuda", C.c_int32), ("fused", C.c_int32),
                ("strict_math", C.c_int32), ("legacy_setup", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("depth_factor", C.c_double), ("huber_delta", C.c_double),
                ("im0", C.POINTER(C.c_uint8)), ("im1", C.POINTER(C.c_uint8)),
                ("depth_u16", C.POINTER(C.c_uint16)), ("T_wc0_colmajor", c_dp)]


class LmRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("lm_trials", C.c_int32), ("chi2", C.c_double),
                ("lambda_", C.c_double), ("rho", C.c_double), ("pose7", C.c_double * 7), ("time_s", C.c_double)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    lib = C.CDLL(LIB_PATH)
    lib.nid_host_run_lm.restype = C.c_int
    lib.nid_host_run_lm.argtypes = [C.POINTER(PoseProblem), c_dp, C.POINTER(LmRecord), C.c_int, C.c_char_p, C.c_int]
    lib.nid_host_last_optimize_seconds.restype = C.c_double
    lib.nid_host_last_optimize_seconds.argtypes = []
    lib.nid_host_se3_exp.argtypes = [c_dp, c_dp]
    lib.nid_host_se3_mul.argtypes = [c_dp, c_dp, c_dp]
    lib.nid_host_se3_to_matrix.argtypes = [c_dp, c_dp]
    lib.nid_host_ldlt6_solve.restype = C.c_int
    lib.nid_host_ldlt6_solve.argtypes = [c_dp, c_dp, c_dp]
    lib.nid_host_minimal_vector.argtypes = [c_dp, c_dp]
    lib.nid_host_huber.argtypes = [C.c_double, C.c_double, c_dp]
    lib.nid_legacy_call_Calculate3Dpoint.argtypes = [c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int]
    lib.nid_legacy_call_CudaComputeHref.argtypes = [c_dp] * 4 + [C.c_int] * 5 + [c_dp, c_ip, c_ip, c_dp]
    lib.nid_legacy_call_CudaComputeH.argtypes = ([C.c_int, c_dp, c_dp, c_dp, c_ip, c_dp, c_ip, c_dp, c_dp]
                                                 + [C.c_int] * 5 + [c_dp] * 4)
    lib.nid_legacy_reset.restype = None
    lib.nid_legacy_set_jacobian_bound.argtypes = [C.c_int]
    lib.nid_legacy_upload_count.restype = C.c_long
    lib.nid_legacy_set_math_mode.argtypes = [C.c_int]
    lib.nid_legacy_set_launch_shape.restype = None
    lib.nid_legacy_set_launch_shape.argtypes = [C.c_int, C.c_int]
    lib.nid_legacy_set_resident.restype = None
    lib.nid_legacy_set_resident.argtypes = [C.c_int]
    lib.nid_host_set_devices.restype = None
    lib.nid_host_set_devices.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int]
    lib.nid_host_set_rank.restype = None
    lib.nid_host_set_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    _lib = lib
    return lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(c_dp)


def se3_exp(upd6):
    out = np.zeros(7)
    load().nid_host_se3_exp(_dp(_d(upd6)), _dp(out))
    return out


def se3_mul(a7, b7):
    out = np.zeros(7)
    load().nid_host_se3_mul(_dp(_d(a7)), _dp(_d(b7)), _dp(out))
    return out


def se3_to_matrix16(p7):
    out = np.zeros(16)
    load().nid_host_se3_to_matrix(_dp(_d(p7)), _dp(out))
    return out


def ldlt6_solve(H, b):
    x = np.zeros(6)
    ok = load().nid_host_ldlt6_solve(_dp(_d(H).reshape(36)), _dp(_d(b)), _dp(x))
    return bool(ok), x


def minimal_vector(p7):
    out = np.zeros(6)
    load().nid_host_minimal_vector(_dp(_d(p7)), _dp(out))
    return out


def huber(e2, delta):
    rho = np.zeros(3)
    load().nid_host_huber(float(e2), float(delta), _dp(rho))
    return rho


def run_lm(pair, bin_num, pose7, iterations=10, jac_bound_cuda=False, fused=False, huber_delta=None, synth=None,
           strict=False, legacy_setup=False):
    """The reference driver's optimisation (NID_pose_estimation.cpp:163-366) on the C++ host stack.  legacy_setup: the
    fused flows set the pair up through Calculate3Dpoint / CudaComputeHref (round 5's route) instead of natively."""
    import importlib
    synth = synth or importlib.import_module("nid-pose-estimation_amd.synth")
    im0 = np.ascontiguousarray(pair.im0, dtype=np.uint8)
    im1 = np.ascontiguousarray(pair.im1, dtype=np.uint8)
    dep = np.ascontiguousarray(pair.depth_u16, dtype=np.uint16)
    T = _d(synth.matrix_colmajor16(pair.T_wc0))
    pb = PoseProblem(pair.rows, pair.cols, pair.cell, bin_num, iterations, 1 if jac_bound_cuda else 0,
                     int(fused), 1 if strict else 0, 1 if legacy_setup else 0, pair.fx, pair.fy, pair.cx, pair.cy, 1.0 / 5000,
                     float(huber_delta) if huber_delta else 0.0,
                     im0.ctypes.data_as(C.POINTER(C.c_uint8)), im1.ctypes.data_as(C.POINTER(C.c_uint8)),
                     dep.ctypes.data_as(C.POINTER(C.c_uint16)), _dp(T))
    p = _d(pose7).copy()
    trace = (LmRecord * iterations)()
    log = C.create_string_buffer(16384)
    n = load().nid_host_run_lm(C.byref(pb), _dp(p), trace, iterations, log, len(log))
    if n < 0:
        raise RuntimeError("nid_host_run_lm failed: " + log.value.decode(errors="replace"))
    recs = [dict(iteration=t.iteration, chi2=t.chi2, lambda_=t.lambda_, lm_trials=t.lm_trials, rho=t.rho,
                 pose7=np.array(list(t.pose7)), time_s=t.time_s) for t in trace[:n]]
    return p, recs, log.value.decode(errors="replace")


def _problem(pair, bin_num, iterations, jac_bound_cuda, fused, huber_delta, strict, synth, legacy_setup=False):
    im0 = np.ascontiguousarray(pair.im0, dtype=np.uint8)
    im1 = np.ascontiguousarray(pair.im1, dtype=np.uint8)
    dep = np.ascontiguousarray(pair.depth_u16, dtype=np.uint16)
    T = _d(synth.matrix_colmajor16(pair.T_wc0))
    pb = PoseProblem(pair.rows, pair.cols, pair.cell, bin_num, iterations, 1 if jac_bound_cuda else 0,
                     int(fused), 1 if strict else 0, 1 if legacy_setup else 0, pair.fx, pair.fy, pair.cx, pair.cy, 1.0 / 5000,
                     float(huber_delta) if huber_delta else 0.0,
                     im0.ctypes.data_as(C.POINTER(C.c_uint8)), im1.ctypes.data_as(C.POINTER(C.c_uint8)),
                     dep.ctypes.data_as(C.POINTER(C.c_uint16)), _dp(T))
    return pb, (im0, im1, dep, T)   # keep the arrays alive while the struct is in use


def run_pyramid_lm(pair, bin_num, pose7, levels=3, iterations=10, jac_bound_cuda=False, fused=False,
                   huber_delta=None, synth=None, strict=False, legacy_setup=False):
    """Coarse-to-fine schedule (own definition, host/nid_pyramid.cpp): `iterations` LM iterations per level,
    coarsest level first.  Returns (pose7, [records per level, coarsest first], log)."""
    import importlib
    synth = synth or importlib.import_module("nid-pose-estimation_amd.synth")
    pb, keep = _problem(pair, bin_num, iterations, jac_bound_cuda, fused, huber_delta, strict, synth, legacy_setup)
    lib = load()
    lib.nid_host_run_pyramid_lm.restype = C.c_int
    lib.nid_host_run_pyramid_lm.argtypes = [C.POINTER(PoseProblem), C.c_int, c_dp, C.POINTER(LmRecord), C.c_int,
                                            C.POINTER(C.c_int), C.c_char_p, C.c_int]
    p = _d(pose7).copy()
    trace = (LmRecord * (levels * iterations))()
    done = (C.c_int * levels)()
    log = C.create_string_buffer(65536)
    n = lib.nid_host_run_pyramid_lm(C.byref(pb), levels, _dp(p), trace, iterations, done, log, len(log))
    if n < 0:
        raise RuntimeError(f"nid_host_run_pyramid_lm failed ({n}): " + log.value.decode(errors="replace"))
    per_level = []
    for l in range(levels):
        per_level.append([dict(iteration=t.iteration, chi2=t.chi2, lambda_=t.lambda_, lm_trials=t.lm_trials, rho=t.rho,
                               pose7=np.array(list(t.pose7)), time_s=t.time_s)
                          for t in trace[l * iterations:l * iterations + done[l]]])
    del keep
    return p, per_level, log.value.decode(errors="replace")


def pyr_down_u8(im):
    im = np.ascontiguousarray(im, dtype=np.uint8)
    out = np.zeros((im.shape[0] // 2, im.shape[1] // 2), dtype=np.uint8)
    lib = load()
    lib.nid_pyr_down_u8.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    lib.nid_pyr_down_u8.restype = None
    lib.nid_pyr_down_u8(im.ctypes.data_as(C.POINTER(C.c_uint8)), im.shape[0], im.shape[1],
                        out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def pyr_down_depth_u16(dep, depth_factor=1.0 / 5000):
    dep = np.ascontiguousarray(dep, dtype=np.uint16)
    out = np.zeros((dep.shape[0] // 2, dep.shape[1] // 2), dtype=np.uint16)
    lib = load()
    lib.nid_pyr_down_depth_u16.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_uint16)]
    lib.nid_pyr_down_depth_u16.restype = None
    lib.nid_pyr_down_depth_u16(dep.ctypes.data_as(C.POINTER(C.c_uint16)), dep.shape[0], dep.shape[1], float(depth_factor),
                               out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def png_read_gray_u8(path, swap_rb=False):
    """Grey u8 image from a PNG the way the reference's driver gets it (imread UNCHANGED + CV_RGB2GRAY)."""
    lib = load()
    lib.nid_png_read_gray_u8.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_size_t]
    r, c = C.c_int(0), C.c_int(0)
    rc = lib.nid_png_read_gray_u8(os.fsencode(path), int(swap_rb), C.byref(r), C.byref(c), None, 0)
    if rc:
        raise RuntimeError(f"nid_png_read_gray_u8({path}) -> {rc}")
    out = np.zeros((r.value, c.value), dtype=np.uint8)
    rc = lib.nid_png_read_gray_u8(os.fsencode(path), int(swap_rb), C.byref(r), C.byref(c),
                                  out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size)
    if rc:
        raise RuntimeError(f"nid_png_read_gray_u8({path}) -> {rc}")
    return out


def png_read_u16(path):
    lib = load()
    lib.nid_png_read_u16.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint16), C.c_size_t]
    r, c = C.c_int(0), C.c_int(0)
    rc = lib.nid_png_read_u16(os.fsencode(path), C.byref(r), C.byref(c), None, 0)
    if rc:
        raise RuntimeError(f"nid_png_read_u16({path}) -> {rc}")
    out = np.zeros((r.value, c.value), dtype=np.uint16)
    rc = lib.nid_png_read_u16(os.fsencode(path), C.byref(r), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_uint16)), out.size)
    if rc:
        raise RuntimeError(f"nid_png_read_u16({path}) -> {rc}")
    return out


def set_resident(on):
    """The legacy operators' / the host LM's single-pose evaluations through the resident evaluator (nid_set_resident)."""
    load().nid_legacy_set_resident(1 if on else 0)


def set_launch_shape(jac_threads, cost_threads):
    """Threads per workgroup of the legacy operators' / the host LM's launches (nid_set_launch_shape)."""
    load().nid_legacy_set_launch_shape(int(jac_threads), int(cost_threads))


def set_devices(devices=(0,), reduce_rccl=False):
    """Every later run_lm / run_pyramid_lm shards the cells of each pair over these devices of this process."""
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    load().nid_host_set_devices(dv.ctypes.data_as(C.POINTER(C.c_int32)), dv.size, 1 if reduce_rccl else 0)


def set_rank(device, rank, world, rccl_id):
    """This process is rank `rank` of `world` (one per GPU); rccl_id = capi.rccl_unique_id() of rank 0."""
    buf = (C.c_uint8 * 128).from_buffer_copy(rccl_id) if rccl_id is not None else None
    load().nid_host_set_rank(int(device), int(rank), int(world), buf)


def last_optimize_seconds():
    """Wall time of optimize() inside the last run_lm (per-pair setup excluded)."""
    return float(load().nid_host_last_optimize_seconds())
