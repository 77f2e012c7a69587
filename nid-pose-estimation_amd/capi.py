"""ctypes binding of the C-ABI (include/nid/nid_c.h) exported by libnid_hip.so.

This is plumbing for tests and bench.py; the product is the shared library.
Loading fails loudly when the HIP library has not been built -- there is no
CPU fallback of any kind.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NID_HIP_LIB") or os.path.join(_HERE, "libnid_hip.so")  # NID_HIP_LIB: experiment builds of the same C-ABI

NID_OK = 0
NID_SLOTS = 1024
NID_MAX_BATCH = 256
NID_REDUCED_LEN = 32
NID_CELL_OUT = 10
JACBOUND_CPU, JACBOUND_CUDA = 0, 1
XFORM_QUAT, XFORM_MATRIX = 0, 1
MATH_FAST, MATH_STRICT = 0, 1

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_fp = C.POINTER(C.c_float)


class NidConfig(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("cell_num", C.c_int32),
                ("bin_num", C.c_int32), ("bs_degree", C.c_int32), ("device", C.c_int32),
                ("cell_begin", C.c_int32), ("cell_end", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


# every symbol include/nid/nid_c.h declares (tests/test_abi.py checks the export table)
SYMBOLS = [
    "nid_abi_version", "nid_status_string", "nid_last_error", "nid_device_count", "nid_create", "nid_create_strided",
    "nid_destroy", "nid_set_options", "nid_set_math_mode", "nid_set_href_nan_markers", "nid_set_stream", "nid_set_block_threads", "nid_set_launch_shape",
    "nid_set_reference_depth", "nid_set_pair_u16", "nid_set_reference_points", "nid_backproject", "nid_backproject_release", "nid_get_points3d", "nid_set_target_u8",
    "nid_set_target_f64", "nid_set_reference_image_f64", "nid_compute_href",
    "nid_compute_href_matrix", "nid_set_href_state", "nid_plain_nid", "nid_evaluate", "nid_evaluate_matrix",
    "nid_normal_equations", "nid_launch", "nid_launch_batch", "nid_launch_chain", "nid_launch_batch_to", "nid_run_sequence", "nid_run_chain", "nid_set_loop_form", "nid_set_direct_results", "nid_set_resident", "nid_resident_pause", "nid_resident_stats", "nid_wait", "nid_slot_buffers", "nid_debug_read_device", "nid_launch_to",
    "nid_unpack_reduced", "nid_debug_enable_pixel_dump", "nid_debug_get_pixel_dump",
    "nid_debug_enable_stamps", "nid_debug_get_stamps", "nid_bspline4_host", "nid_bspline4_poly_host", "nid_log2_fast_host", "nid_div_small_host", "nid_last_kernel_ms", "nid_enable_timing", "nid_time_launches", "nid_time_kernel",
    "nid_contract_bytes", "nid_debug_repair_count", "nid_set_short_sequence_policy",
]

_lib = None


class NidError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NidError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                       " (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.nid_abi_version.restype = C.c_int
    lib.nid_status_string.restype = C.c_char_p
    lib.nid_status_string.argtypes = [C.c_int]
    lib.nid_last_error.restype = C.c_char_p
    lib.nid_last_error.argtypes = [vp]
    lib.nid_device_count.restype = C.c_int
    lib.nid_create.argtypes = [C.POINTER(NidConfig), C.POINTER(vp)]
    lib.nid_create_strided.argtypes = [C.POINTER(NidConfig), C.c_int32, C.POINTER(vp)]
    lib.nid_destroy.argtypes = [vp]
    lib.nid_set_options.argtypes = [vp, C.c_int, C.c_int]
    lib.nid_set_math_mode.argtypes = [vp, C.c_int]
    lib.nid_set_stream.argtypes = [vp, vp]
    lib.nid_set_block_threads.argtypes = [vp, C.c_int]
    lib.nid_set_launch_shape.argtypes = [vp, C.c_int, C.c_int]
    lib.nid_set_reference_depth.argtypes = [vp, c_dp, c_u8p, c_dp]
    lib.nid_set_pair_u16.argtypes = [vp, C.POINTER(C.c_uint16), C.c_double, c_u8p, c_u8p, c_dp, c_dp, c_dp, c_ip, c_dp]
    lib.nid_set_reference_points.argtypes = [vp, c_dp, c_u8p]
    lib.nid_get_points3d.argtypes = [vp, c_dp]
    lib.nid_backproject.argtypes = [c_dp, c_dp] + [C.c_double] * 4 + [C.c_int32] * 3 + [c_dp]
    lib.nid_set_target_u8.argtypes = [vp, c_u8p]
    lib.nid_set_target_f64.argtypes = [vp, c_dp]
    lib.nid_set_reference_image_f64.argtypes = [c_dp, C.c_int64, c_u8p]
    lib.nid_compute_href.argtypes = [vp, c_dp, c_ip, c_dp, c_dp, c_ip]
    lib.nid_compute_href_matrix.argtypes = [vp, c_dp, c_ip, c_dp, c_dp, c_ip]
    lib.nid_set_href_state.argtypes = [vp, c_ip, c_dp, c_dp, c_ip]
    lib.nid_evaluate.argtypes = [vp, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp]
    lib.nid_evaluate_matrix.argtypes = [vp, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp]
    lib.nid_normal_equations.argtypes = [vp, c_dp, C.c_int, C.c_double, c_dp, c_dp, c_dp, c_ip]
    lib.nid_launch.argtypes = [vp, C.c_int, c_dp, C.c_int, C.c_double]
    lib.nid_launch_batch.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double]
    lib.nid_launch_batch_to.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double, vp]
    lib.nid_launch_chain.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double]
    lib.nid_run_sequence.argtypes = [vp, c_dp, C.c_int, C.c_int, C.c_int, C.c_double, c_dp]
    lib.nid_launch_to.argtypes = [vp, C.c_int, c_dp, C.c_int, C.c_double, vp]
    lib.nid_run_chain.argtypes = [vp, c_dp, C.c_int, C.c_int, C.c_double, c_dp, c_dp]
    lib.nid_wait.argtypes = [vp, C.c_int, c_dp, c_dp, c_dp, c_ip]
    lib.nid_slot_buffers.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    if hasattr(lib, "nid_debug_read_device"):  # (tools/build_variant.py builds of earlier trees lack the newest entry points)
        lib.nid_debug_read_device.argtypes = [vp, vp, C.POINTER(C.c_double), C.c_size_t]
    lib.nid_unpack_reduced.argtypes = [c_dp, c_dp, c_dp, c_dp, c_ip]
    lib.nid_debug_enable_pixel_dump.argtypes = [vp, C.c_int]
    lib.nid_debug_get_pixel_dump.argtypes = [vp, c_dp, c_dp, c_dp, c_ip, c_dp]
    lib.nid_debug_enable_stamps.argtypes = [vp, C.c_int]
    lib.nid_debug_get_stamps.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.nid_bspline4_host.restype = None
    lib.nid_bspline4_host.argtypes = [C.c_double, C.c_int, c_dp, c_dp]
    lib.nid_bspline4_poly_host.restype = None
    lib.nid_bspline4_poly_host.argtypes = [C.c_double, C.c_int, c_dp, c_dp]
    lib.nid_div_small_host.restype = C.c_double
    lib.nid_div_small_host.argtypes = [C.c_double, C.c_double]
    lib.nid_last_kernel_ms.argtypes = [vp, C.c_int, c_fp, c_fp]
    lib.nid_enable_timing.argtypes = [vp, C.c_int]
    lib.nid_set_loop_form.argtypes = [vp, C.c_int]
    lib.nid_set_direct_results.argtypes = [vp, C.c_int]
    lib.nid_set_resident.argtypes = [vp, C.c_int]
    if hasattr(lib, "nid_resident_pause"):
        lib.nid_resident_pause.argtypes = [vp]
    lib.nid_resident_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.nid_time_launches.argtypes = [vp, C.c_int, c_dp, C.c_int, C.c_double, C.c_int, c_fp]
    if hasattr(lib, "nid_time_kernel"):   # (an older experiment build, NID_HIP_LIB, may lack it)
        lib.nid_time_kernel.argtypes = [vp, C.c_int, c_dp, C.c_int, C.c_double, C.c_int, c_fp]
    if hasattr(lib, "nid_debug_repair_count"):   # (an older experiment build, NID_HIP_LIB, may lack the newest diagnostics)
        lib.nid_debug_repair_count.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    if hasattr(lib, "nid_set_short_sequence_policy"):
        lib.nid_set_short_sequence_policy.argtypes = [vp, C.c_int, C.c_int]
    lib.nid_contract_bytes.restype = C.c_int64
    lib.nid_contract_bytes.argtypes = [vp]
    _lib = lib
    return lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class Context:
    """One frame pair on one MI355X (or one cell shard of it)."""

    def __init__(self, rows, cols, cell_num, bin_num, fx, fy, cx, cy, device=0, cell_begin=0,
                 cell_end=0, jac_bound=JACBOUND_CPU, xform=XFORM_QUAT, cell_stride=1):
        self.lib = load()
        cfg = NidConfig(rows, cols, cell_num, bin_num, 3, device, cell_begin, cell_end, fx, fy, cx, cy)
        h = C.c_void_p()
        rc = self.lib.nid_create_strided(C.byref(cfg), int(cell_stride), C.byref(h))
        if rc != NID_OK:
            raise NidError(f"nid_create: {self.lib.nid_status_string(rc).decode()} ({rc})")
        self.h = h
        self.rows, self.cols, self.cell_num, self.bin_num = rows, cols, cell_num, bin_num
        self.ncell = cell_num * cell_num
        self.cell_begin = cell_begin
        self.cell_end = cell_end if cell_end else self.ncell
        self._check(self.lib.nid_set_options(self.h, jac_bound, xform), "nid_set_options")

    @classmethod
    def borrow(cls, multi, k):
        """Shard k of a Multi as a Context (not owned: closing it does nothing)."""
        self = cls.__new__(cls)
        self.lib = load()
        self.h = C.c_void_p(multi.lib.nid_multi_shard(multi.h, int(k)))
        if not self.h:
            raise NidError(f"nid_multi_shard({k}): no such shard")
        self._borrowed = True
        self._owner = multi   # keeps the Multi (and with it this shard) alive as long as the borrowed handle is
        self.rows, self.cols, self.cell_num, self.bin_num = multi.rows, multi.cols, multi.cell_num, multi.bin_num
        self.ncell = multi.ncell
        self.cell_begin, self.cell_end = 0, self.ncell
        return self

    def _check(self, rc, what):
        if rc != NID_OK:
            msg = self.lib.nid_last_error(self.h).decode() if self.h else ""
            raise NidError(f"{what}: {self.lib.nid_status_string(rc).decode()} ({rc}) {msg}")

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.nid_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup --------------------------------------------------------------
    def set_reference_depth(self, depth_m, im0, T_wc0_colmajor16):
        d, im, T = _d(depth_m).reshape(-1), _u8(im0).reshape(-1), _d(T_wc0_colmajor16)
        assert d.size == self.rows * self.cols and im.size == d.size and T.size == 16
        self._check(self.lib.nid_set_reference_depth(self.h, _dp(d), im.ctypes.data_as(c_u8p), _dp(T)),
                    "nid_set_reference_depth")

    def set_pair_u16(self, depth_u16, depth_factor, im0, im1, T_wc0_colmajor16, pose0, matrix=False):
        """nid_set_pair_u16: the whole pair in the driver's formats + the reference stage at pose0 (pose7, or a
        column-major 4x4 with matrix=True); returns (bs_counter, Href)."""
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16).reshape(-1)
        a, b, T, p = _u8(im0).reshape(-1), _u8(im1).reshape(-1), _d(T_wc0_colmajor16), _d(pose0)
        assert d.size == self.rows * self.cols and a.size == d.size and b.size == d.size and T.size == 16 and p.size == (16 if matrix else 7)
        cnt = np.zeros(self.ncell, dtype=np.int32)
        href = np.full(self.ncell, np.nan)
        self._check(self.lib.nid_set_pair_u16(self.h, d.ctypes.data_as(C.POINTER(C.c_uint16)), float(depth_factor), a.ctypes.data_as(c_u8p),
                                              b.ctypes.data_as(c_u8p), _dp(T), None if matrix else _dp(p), _dp(p) if matrix else None,
                                              _ip(cnt), _dp(href)), "nid_set_pair_u16")
        return cnt, href

    def set_reference_points(self, points3d, im0):
        p, im = _d(points3d).reshape(-1), _u8(im0).reshape(-1)
        assert p.size == 3 * self.rows * self.cols and im.size == self.rows * self.cols
        self._check(self.lib.nid_set_reference_points(self.h, _dp(p), im.ctypes.data_as(c_u8p)),
                    "nid_set_reference_points")

    def get_points3d(self):
        out = np.empty(3 * self.rows * self.cols)
        self._check(self.lib.nid_get_points3d(self.h, _dp(out)), "nid_get_points3d")
        return out

    def set_target(self, im1):
        im = _u8(im1).reshape(-1)
        assert im.size == self.rows * self.cols
        self._check(self.lib.nid_set_target_u8(self.h, im.ctypes.data_as(c_u8p)), "nid_set_target_u8")

    def compute_href(self, pose7, dump=False):
        cnt = np.zeros(self.ncell, dtype=np.int32)
        href = np.full(self.ncell, np.nan)
        N = self.rows * self.cols
        bsv = np.zeros((N, 4)) if dump else None
        bsi = np.zeros(N, dtype=np.int32) if dump else None
        self._check(self.lib.nid_compute_href(self.h, _dp(_d(pose7)), _ip(cnt), _dp(href), _dp(bsv), _ip(bsi)),
                    "nid_compute_href")
        return (cnt, href, bsv, bsi) if dump else (cnt, href)

    def compute_href_matrix(self, pose16):
        cnt = np.zeros(self.ncell, dtype=np.int32)
        href = np.full(self.ncell, np.nan)
        self._check(self.lib.nid_compute_href_matrix(self.h, _dp(_d(pose16)), _ip(cnt), _dp(href), None, None), "nid_compute_href_matrix")
        return cnt, href

    # ---- per iteration --------------------------------------------------------
    def plain_nid(self, pose7, bins=8):
        """Plain-histogram NID per cell (NID_standard_property.cpp): dict of per-cell arrays + 'total'."""
        n = self.ncell
        out = {k: np.full(n, np.nan) for k in ("Href", "Hcur", "Hjoint", "nid", "mi")}
        n_in = np.zeros(n, dtype=np.int32)
        total = C.c_double(0)
        self.lib.nid_plain_nid.argtypes = [C.c_void_p, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, C.POINTER(C.c_double)]
        self._check(self.lib.nid_plain_nid(self.h, _dp(_d(pose7)), int(bins), _dp(out["Href"]), _dp(out["Hcur"]),
                                           _dp(out["Hjoint"]), _dp(out["nid"]), _dp(out["mi"]), _ip(n_in),
                                           C.byref(total)), "nid_plain_nid")
        out["n_in"] = n_in
        out["total"] = float(total.value)
        return out

    def evaluate(self, pose7, want_jac=True):
        Hc = np.full(self.ncell, np.nan)
        Hj = np.full(self.ncell, np.nan)
        err = np.full(self.ncell, np.nan)
        J = np.full((self.ncell, 6), np.nan) if want_jac else None
        self._check(self.lib.nid_evaluate(self.h, _dp(_d(pose7)), 1 if want_jac else 0, _dp(Hc), _dp(Hj),
                                          _dp(err), _dp(J)), "nid_evaluate")
        return Hc, Hj, err, J

    def evaluate_matrix(self, pose16, want_jac=True):
        Hc = np.full(self.ncell, np.nan)
        Hj = np.full(self.ncell, np.nan)
        err = np.full(self.ncell, np.nan)
        J = np.full((self.ncell, 6), np.nan) if want_jac else None
        self._check(self.lib.nid_evaluate_matrix(self.h, _dp(_d(pose16)), 1 if want_jac else 0, _dp(Hc),
                                                 _dp(Hj), _dp(err), _dp(J)), "nid_evaluate_matrix")
        return Hc, Hj, err, J

    def normal_equations(self, pose7, delta, want_jac=True):
        H = np.zeros(36)
        b = np.zeros(6)
        chi2 = C.c_double(0)
        na = C.c_int32(0)
        self._check(self.lib.nid_normal_equations(self.h, _dp(_d(pose7)), 1 if want_jac else 0, float(delta),
                                                  _dp(H), _dp(b), C.byref(chi2), C.byref(na)),
                    "nid_normal_equations")
        return H.reshape(6, 6), b, chi2.value, na.value

    def launch(self, slot, pose7, delta, want_jac=True, reduced_dev=None):
        p = _d(pose7)
        if reduced_dev is None:
            rc = self.lib.nid_launch(self.h, slot, _dp(p), 1 if want_jac else 0, float(delta))
        else:
            rc = self.lib.nid_launch_to(self.h, slot, _dp(p), 1 if want_jac else 0, float(delta),
                                        C.c_void_p(reduced_dev))
        self._check(rc, "nid_launch")

    def launch_batch(self, first_slot, poses7, delta, want_jac=True, reduced_dev=None):
        p = _d(np.asarray(poses7).reshape(-1, 7))
        if reduced_dev is None:
            rc = self.lib.nid_launch_batch(self.h, first_slot, p.shape[0], _dp(p), 1 if want_jac else 0, float(delta))
        else:
            rc = self.lib.nid_launch_batch_to(self.h, first_slot, p.shape[0], _dp(p), 1 if want_jac else 0,
                                              float(delta), C.c_void_p(reduced_dev))
        self._check(rc, "nid_launch_batch")

    def launch_chain(self, first_slot, poses7, n_jac, delta):
        """an LM rejection chain: the first n_jac poses with the Jacobian phase, the rest cost only, concurrently"""
        p = _d(np.asarray(poses7).reshape(-1, 7))
        self._check(self.lib.nid_launch_chain(self.h, first_slot, p.shape[0], _dp(p), int(n_jac), float(delta)),
                    "nid_launch_chain")

    def run_sequence(self, poses7, delta, batch=8, want_jac=True, collect=True):
        p = _d(np.asarray(poses7).reshape(-1, 7))
        out = np.zeros((p.shape[0], NID_REDUCED_LEN)) if collect else None
        self._check(self.lib.nid_run_sequence(self.h, _dp(p), p.shape[0], int(batch), 1 if want_jac else 0,
                                              float(delta), _dp(out)), "nid_run_sequence")
        return out

    def run_chain(self, poses7, delta, want_jac=True, collect=True):
        """A dependent chain: one pose per launch, each result awaited on the host before the next launch.
        Returns (reduced blocks or None, seconds)."""
        p = _d(np.asarray(poses7).reshape(-1, 7))
        out = np.zeros((p.shape[0], NID_REDUCED_LEN)) if collect else None
        sec = C.c_double(0)
        self._check(self.lib.nid_run_chain(self.h, _dp(p), p.shape[0], 1 if want_jac else 0, float(delta), _dp(out),
                                           C.byref(sec)), "nid_run_chain")
        return out, float(sec.value)

    def wait(self, slot):
        H = np.zeros(36)
        b = np.zeros(6)
        chi2 = C.c_double(0)
        na = C.c_int32(0)
        self._check(self.lib.nid_wait(self.h, slot, _dp(H), _dp(b), C.byref(chi2), C.byref(na)), "nid_wait")
        return H.reshape(6, 6), b, chi2.value, na.value

    def slot_buffers(self, slot):
        """(reduced_dev, cellout_dev): device addresses of the slot's blocks (nid_slot_buffers)"""
        r, c = C.c_void_p(0), C.c_void_p(0)
        self._check(self.lib.nid_slot_buffers(self.h, int(slot), C.byref(r), C.byref(c)), "nid_slot_buffers")
        return r.value, c.value

    def read_device(self, dev, shape):
        out = np.zeros(shape)
        self._check(self.lib.nid_debug_read_device(self.h, C.c_void_p(dev), _dp(out), out.nbytes), "nid_debug_read_device")
        return out

    def set_math_mode(self, mode):
        self._check(self.lib.nid_set_math_mode(self.h, int(mode)), "nid_set_math_mode")

    def set_stream(self, stream_handle):
        self._check(self.lib.nid_set_stream(self.h, C.c_void_p(stream_handle)), "nid_set_stream")

    def set_block_threads(self, n):
        self._check(self.lib.nid_set_block_threads(self.h, int(n)), "nid_set_block_threads")

    def set_launch_shape(self, jac_threads=0, cost_threads=0):
        self._check(self.lib.nid_set_launch_shape(self.h, int(jac_threads), int(cost_threads)), "nid_set_launch_shape")

    def set_loop_form(self, on=True):
        self._check(self.lib.nid_set_loop_form(self.h, 1 if on else 0), "nid_set_loop_form")

    def set_direct_results(self, on=True):
        self._check(self.lib.nid_set_direct_results(self.h, int(on)), "nid_set_direct_results")

    def set_resident(self, on=True):
        """False / 0: off; True / 1: single-pose requests are answered by the resident kernel."""
        self._check(self.lib.nid_set_resident(self.h, int(on)), "nid_set_resident")

    def resident_pause(self):
        self._check(self.lib.nid_resident_pause(self.h), "nid_resident_pause")

    def resident_stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self.lib.nid_resident_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "nid_resident_stats")
        return dict(served=a.value, fallbacks=b.value, starts=c.value)

    def enable_timing(self, on=True):
        self._check(self.lib.nid_enable_timing(self.h, 1 if on else 0), "nid_enable_timing")

    def last_kernel_ms(self, slot):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self.lib.nid_last_kernel_ms(self.h, slot, C.byref(a), C.byref(b)), "nid_last_kernel_ms")
        return a.value, b.value

    def time_launches(self, poses7, delta, repeats=10, want_jac=True):
        """ms per launch of `repeats` back-to-back launches of these poses (one HIP event pair)."""
        ps = np.ascontiguousarray(np.asarray(poses7, dtype=np.float64).reshape(-1, 7))
        ms = C.c_float(0)
        self._check(self.lib.nid_time_launches(self.h, ps.shape[0], _dp(ps), int(want_jac), float(delta), int(repeats),
                                               C.byref(ms)), "nid_time_launches")
        return float(ms.value)

    def time_kernel(self, poses7, delta, repeats=10, want_jac=True):
        """ms of the evaluation kernel ALONE per launch of these poses (events right around k_eval2, one launch at a time)."""
        ps = np.ascontiguousarray(np.asarray(poses7, dtype=np.float64).reshape(-1, 7))
        ms = C.c_float(0)
        self._check(self.lib.nid_time_kernel(self.h, ps.shape[0], _dp(ps), int(want_jac), float(delta), int(repeats),
                                             C.byref(ms)), "nid_time_kernel")
        return float(ms.value)

    def contract_bytes(self):
        return int(self.lib.nid_contract_bytes(self.h))

    def set_short_sequence_policy(self, poses_per_launch=0, streams=0):
        """How nid_launch_batch / nid_run_sequence split a short sequence (0, 0: the library's table)."""
        self._check(self.lib.nid_set_short_sequence_policy(self.h, int(poses_per_launch), int(streams)), "nid_set_short_sequence_policy")

    def repair_count(self, reset=False):
        """(cell, pose) evaluations that ran the fold's repair pass (kLinFlagW in csrc/nid_kernels.hip.h)."""
        if not hasattr(self.lib, "nid_debug_repair_count"):
            return -1
        n = C.c_int64(0)
        self._check(self.lib.nid_debug_repair_count(self.h, C.byref(n), 1 if reset else 0), "nid_debug_repair_count")
        return int(n.value)

    # ---- debug ----------------------------------------------------------------
    def enable_pixel_dump(self, on=True):
        """True / 1: cost-phase dump (u, v, ic, jc, wc); 2: Jacobian-phase dump (gx, gy, pc, jc, dw in the same arrays)."""
        self._check(self.lib.nid_debug_enable_pixel_dump(self.h, int(on)), "nid_debug_enable_pixel_dump")

    def enable_stamps(self, on=True):
        self._check(self.lib.nid_debug_enable_stamps(self.h, 1 if on else 0), "nid_debug_enable_stamps")

    def stamps(self):
        n = self.cell_end - self.cell_begin
        out = np.zeros((n, 10), dtype=np.int64)
        self._check(self.lib.nid_debug_get_stamps(self.h, out.ctypes.data_as(C.POINTER(C.c_int64))), "nid_debug_get_stamps")
        return out

    def pixel_dump(self):
        N = self.rows * self.cols
        u = np.zeros(N); v = np.zeros(N); ic = np.zeros(N)
        jc = np.zeros(N, dtype=np.int32); wc = np.zeros((N, 4))
        self._check(self.lib.nid_debug_get_pixel_dump(self.h, _dp(u), _dp(v), _dp(ic), _ip(jc), _dp(wc)),
                    "nid_debug_get_pixel_dump")
        return dict(u=u, v=v, ic=ic, jc=jc, wc=wc)


def unpack_reduced(r):
    lib = load()
    r = _d(r)
    H = np.zeros(36)
    b = np.zeros(6)
    chi2 = C.c_double(0)
    na = C.c_int32(0)
    lib.nid_unpack_reduced(_dp(r), _dp(H), _dp(b), C.byref(chi2), C.byref(na))
    return H.reshape(6, 6), b, chi2.value, na.value


def log2_fast_host(x):
    lib = load()
    lib.nid_log2_fast_host.restype = C.c_double
    lib.nid_log2_fast_host.argtypes = [C.c_double]
    return float(lib.nid_log2_fast_host(float(x)))


def bspline4_poly_host(u, bin_num):
    B = np.zeros(4)
    D = np.zeros(4)
    load().nid_bspline4_poly_host(float(u), int(bin_num), _dp(B), _dp(D))
    return B, D


def bspline4_host(u, bin_num):
    B = np.zeros(4)
    D = np.zeros(4)
    load().nid_bspline4_host(float(u), int(bin_num), _dp(B), _dp(D))
    return B, D


def from_pair(pair, bin_num, device=0, cell_begin=0, cell_end=0, jac_bound=JACBOUND_CPU, xform=XFORM_QUAT,
              math=MATH_FAST, cell_stride=1):
    """Context set up like the reference's main() (NID_pose_estimation.cpp:253-257):
    back-projection (on the device) + target image; the caller runs compute_href."""
    import importlib
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    ctx = Context(pair.rows, pair.cols, pair.cell, bin_num, pair.fx, pair.fy, pair.cx, pair.cy, device=device,
                  cell_begin=cell_begin, cell_end=cell_end, jac_bound=jac_bound, xform=xform, cell_stride=cell_stride)
    ctx.set_math_mode(math)
    ctx.set_reference_depth(pair.depth_m, pair.im0, synth.matrix_colmajor16(pair.T_wc0))
    ctx.set_target(pair.im1)
    return ctx


# ---------------------------------------------------------------------------------------------------------------
# include/nid/nid_multi.h: cell shards over several GPUs (one process, or one process per GPU), in C++
REDUCE_HOST, REDUCE_RCCL, REDUCE_HOOK = 0, 1, 2
PARTITION_CONTIGUOUS, PARTITION_INTERLEAVED = 0, 1
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, c_dp, C.c_int64, C.c_void_p)
RCCL_ID_BYTES = 128
MULTI_SYMBOLS = [
    "nid_multi_cell_range", "nid_multi_cell_partition", "nid_multi_resident_pause", "nid_multi_create", "nid_multi_create_rank", "nid_multi_create_partitioned", "nid_multi_destroy", "nid_multi_last_error",
    "nid_multi_shards", "nid_multi_shard", "nid_multi_world", "nid_multi_comm_unique_id", "nid_comm_create_rank",
    "nid_comm_create_local", "nid_comm_destroy", "nid_comm_ranks", "nid_multi_attach_comm", "nid_multi_comm_init",
    "nid_multi_comm_init_local", "nid_multi_comm_ranks", "nid_multi_time_exchange", "nid_multi_set_exchange_hook", "nid_multi_set_reduce_mode",
    "nid_multi_set_options",
    "nid_multi_set_math_mode", "nid_multi_set_href_nan_markers", "nid_multi_set_block_threads", "nid_multi_set_launch_shape", "nid_multi_set_resident", "nid_multi_set_reference_depth",
    "nid_multi_set_reference_points", "nid_multi_set_target_u8", "nid_multi_set_pair_u16", "nid_multi_compute_href",
    "nid_multi_compute_href_matrix", "nid_multi_set_href_state", "nid_multi_evaluate", "nid_multi_evaluate_matrix",
    "nid_multi_normal_equations", "nid_multi_launch_batch", "nid_multi_launch_chain", "nid_multi_wait", "nid_multi_run_sequence",
    "nid_multi_contract_bytes",
]


def _load_multi():
    lib = load()
    if getattr(lib, "_multi_ready", False):
        return lib
    vp = C.c_void_p
    lib.nid_multi_cell_range.argtypes = [C.c_int32] * 3 + [c_ip, c_ip]
    if hasattr(lib, "nid_multi_cell_partition"):
        lib.nid_multi_cell_partition.argtypes = [C.c_int32] * 4 + [c_ip, c_ip, c_ip]
    lib.nid_multi_create.argtypes = [C.POINTER(NidConfig), c_ip, C.c_int32, C.POINTER(vp)]
    lib.nid_multi_create_rank.argtypes = [C.POINTER(NidConfig), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.nid_multi_create_partitioned.argtypes = [C.POINTER(NidConfig), c_ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.nid_multi_destroy.argtypes = [vp]
    lib.nid_multi_last_error.restype = C.c_char_p
    lib.nid_multi_last_error.argtypes = [vp]
    lib.nid_multi_shards.argtypes = [vp]
    lib.nid_multi_shard.restype = vp
    lib.nid_multi_shard.argtypes = [vp, C.c_int32]
    lib.nid_multi_comm_unique_id.argtypes = [c_u8p]
    lib.nid_multi_comm_init.argtypes = [vp, c_u8p]
    lib.nid_multi_comm_init_local.argtypes = [vp]
    lib.nid_multi_comm_ranks.argtypes = [vp, c_ip]
    lib.nid_multi_time_exchange.argtypes = [vp, C.c_int, C.c_int, c_fp]
    lib.nid_multi_set_reduce_mode.argtypes = [vp, C.c_int]
    lib.nid_multi_set_exchange_hook.argtypes = [vp, EXCHANGE_FN, vp]
    lib.nid_multi_set_options.argtypes = [vp, C.c_int, C.c_int]
    lib.nid_multi_set_math_mode.argtypes = [vp, C.c_int]
    lib.nid_multi_set_block_threads.argtypes = [vp, C.c_int]
    lib.nid_multi_set_reference_depth.argtypes = [vp, c_dp, c_u8p, c_dp]
    lib.nid_multi_set_target_u8.argtypes = [vp, c_u8p]
    lib.nid_multi_set_pair_u16.argtypes = [vp, C.POINTER(C.c_uint16), C.c_double, c_u8p, c_u8p, c_dp, c_dp, c_dp, c_ip, c_dp]
    lib.nid_multi_compute_href.argtypes = [vp, c_dp, c_ip, c_dp, c_dp, c_ip]
    lib.nid_multi_set_href_state.argtypes = [vp, c_ip, c_dp, c_dp, c_ip]
    lib.nid_multi_evaluate.argtypes = [vp, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp]
    lib.nid_multi_normal_equations.argtypes = [vp, c_dp, C.c_int, C.c_double, c_dp, c_dp, c_dp, c_ip]
    lib.nid_multi_launch_batch.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double]
    lib.nid_multi_launch_chain.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double]
    lib.nid_multi_wait.argtypes = [vp, C.c_int, c_dp, c_dp, c_dp, c_ip]
    lib.nid_multi_run_sequence.argtypes = [vp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_dp]
    lib.nid_multi_contract_bytes.restype = C.c_int64
    lib.nid_multi_contract_bytes.argtypes = [vp]
    lib._multi_ready = True
    return lib


def cell_range(k, n, ncell):
    lib = _load_multi()
    lo, hi = C.c_int32(0), C.c_int32(0)
    rc = lib.nid_multi_cell_range(k, n, ncell, C.byref(lo), C.byref(hi))
    if rc != NID_OK:
        raise NidError(f"nid_multi_cell_range({k}, {n}, {ncell}): {lib.nid_status_string(rc).decode()}")
    return lo.value, hi.value


def cell_set(k, n, ncell, partition=0):
    """The cell ids shard k of n owns under `partition` (PARTITION_CONTIGUOUS / PARTITION_INTERLEAVED): the library's
    own arithmetic (nid_multi_cell_partition), no device needed."""
    lib = _load_multi()
    b, e, st = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib.nid_multi_cell_partition(k, n, ncell, int(partition), C.byref(b), C.byref(e), C.byref(st))
    if rc != NID_OK:
        raise NidError(f"nid_multi_cell_partition({k}, {n}, {ncell}, {partition}): {lib.nid_status_string(rc).decode()}")
    return np.arange(b.value, e.value, st.value)


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0 of a multi-process job; 128 bytes)."""
    lib = _load_multi()
    buf = (C.c_uint8 * RCCL_ID_BYTES)()
    rc = lib.nid_multi_comm_unique_id(buf)
    if rc != NID_OK:
        raise NidError(f"nid_multi_comm_unique_id: {lib.nid_status_string(rc).decode()} ({rc})")
    return bytes(buf)


class Multi:
    """One frame pair on the cell shards of several GPUs: `devices` = one shard per entry in this process, or
    rank/world = this process's shard of a one-process-per-GPU job (then comm_init(id) before evaluating)."""

    def __init__(self, rows, cols, cell_num, bin_num, fx, fy, cx, cy, devices=(0,), rank=None, world=None,
                 jac_bound=JACBOUND_CPU, xform=XFORM_QUAT, partition=None):
        self.lib = _load_multi()
        cfg = NidConfig(rows, cols, cell_num, bin_num, 3, 0, 0, 0, fx, fy, cx, cy)
        h = C.c_void_p()
        if partition is not None:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self.lib.nid_multi_create_partitioned(C.byref(cfg), _ip(dv), dv.size, int(rank or 0), int(world or 1),
                                                       int(partition), C.byref(h))
        elif world is not None:
            rc = self.lib.nid_multi_create_rank(C.byref(cfg), int(devices[0]), int(rank), int(world), C.byref(h))
        else:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self.lib.nid_multi_create(C.byref(cfg), _ip(dv), dv.size, C.byref(h))
        if rc != NID_OK:
            raise NidError(f"nid_multi_create: {self.lib.nid_status_string(rc).decode()} ({rc})")
        self.h = h
        self.rows, self.cols, self.cell_num, self.bin_num = rows, cols, cell_num, bin_num
        self.ncell = cell_num * cell_num
        self._check(self.lib.nid_multi_set_options(self.h, jac_bound, xform), "nid_multi_set_options")

    def _check(self, rc, what):
        if rc != NID_OK:
            msg = self.lib.nid_multi_last_error(self.h).decode() if self.h else ""
            raise NidError(f"{what}: {self.lib.nid_status_string(rc).decode()} ({rc}) {msg}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.nid_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shards(self):
        return int(self.lib.nid_multi_shards(self.h))

    def comm_init(self, id_bytes):
        buf = (C.c_uint8 * RCCL_ID_BYTES).from_buffer_copy(id_bytes)
        self._check(self.lib.nid_multi_comm_init(self.h, buf), "nid_multi_comm_init")

    def comm_init_local(self):
        self._check(self.lib.nid_multi_comm_init_local(self.h), "nid_multi_comm_init_local")

    def time_exchange(self, blocks, repeats=20):
        ms = C.c_float(0)
        self._check(self.lib.nid_multi_time_exchange(self.h, int(blocks), int(repeats), C.byref(ms)), "nid_multi_time_exchange")
        return float(ms.value)

    def comm_ranks(self):
        n = C.c_int32(0)
        self._check(self.lib.nid_multi_comm_ranks(self.h, C.byref(n)), "nid_multi_comm_ranks")
        return n.value

    def set_exchange_hook(self, fn):
        """fn(array) must replace the float64 array by its sum over all processes, in place (e.g. a gloo all_reduce)."""
        def _cb(ptr, count, _user):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:
                return 1
        self._hook = EXCHANGE_FN(_cb)       # keep the trampoline alive
        self._check(self.lib.nid_multi_set_exchange_hook(self.h, self._hook, None), "nid_multi_set_exchange_hook")

    def set_reduce_mode(self, mode):
        self._check(self.lib.nid_multi_set_reduce_mode(self.h, int(mode)), "nid_multi_set_reduce_mode")

    def set_math_mode(self, mode):
        self._check(self.lib.nid_multi_set_math_mode(self.h, int(mode)), "nid_multi_set_math_mode")

    def set_block_threads(self, n):
        self._check(self.lib.nid_multi_set_block_threads(self.h, int(n)), "nid_multi_set_block_threads")

    def set_reference_depth(self, depth_m, im0, T_wc0_colmajor16):
        d, im, T = _d(depth_m).reshape(-1), _u8(im0).reshape(-1), _d(T_wc0_colmajor16)
        self._check(self.lib.nid_multi_set_reference_depth(self.h, _dp(d), im.ctypes.data_as(c_u8p), _dp(T)),
                    "nid_multi_set_reference_depth")

    def set_target(self, im1):
        im = _u8(im1).reshape(-1)
        self._check(self.lib.nid_multi_set_target_u8(self.h, im.ctypes.data_as(c_u8p)), "nid_multi_set_target_u8")

    def set_pair_u16(self, depth_u16, depth_factor, im0, im1, T_wc0_colmajor16, pose0, matrix=False):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16).reshape(-1)
        a, b, T, p = _u8(im0).reshape(-1), _u8(im1).reshape(-1), _d(T_wc0_colmajor16), _d(pose0)
        cnt = np.zeros(self.ncell, dtype=np.int32)
        href = np.full(self.ncell, np.nan)
        self._check(self.lib.nid_multi_set_pair_u16(self.h, d.ctypes.data_as(C.POINTER(C.c_uint16)), float(depth_factor), a.ctypes.data_as(c_u8p),
                                                    b.ctypes.data_as(c_u8p), _dp(T), None if matrix else _dp(p), _dp(p) if matrix else None,
                                                    _ip(cnt), _dp(href)), "nid_multi_set_pair_u16")
        return cnt, href

    def compute_href(self, pose7, dump=False):
        cnt = np.zeros(self.ncell, dtype=np.int32)
        href = np.full(self.ncell, np.nan)
        N = self.rows * self.cols
        bsv = np.zeros((N, 4)) if dump else None
        bsi = np.zeros(N, dtype=np.int32) if dump else None
        self._check(self.lib.nid_multi_compute_href(self.h, _dp(_d(pose7)), _ip(cnt), _dp(href), _dp(bsv), _ip(bsi)),
                    "nid_multi_compute_href")
        return (cnt, href, bsv, bsi) if dump else (cnt, href)

    def set_href_state(self, cnt, href, bsv, bsi):
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        bsi = np.ascontiguousarray(bsi, dtype=np.int32)
        self._check(self.lib.nid_multi_set_href_state(self.h, _ip(cnt), _dp(_d(href)), _dp(_d(bsv)), _ip(bsi)),
                    "nid_multi_set_href_state")

    def evaluate(self, pose7, want_jac=True):
        Hc = np.full(self.ncell, np.nan)
        Hj = np.full(self.ncell, np.nan)
        err = np.full(self.ncell, np.nan)
        J = np.full((self.ncell, 6), np.nan) if want_jac else None
        self._check(self.lib.nid_multi_evaluate(self.h, _dp(_d(pose7)), 1 if want_jac else 0, _dp(Hc), _dp(Hj),
                                                _dp(err), _dp(J)), "nid_multi_evaluate")
        return Hc, Hj, err, J

    def normal_equations(self, pose7, delta, want_jac=True):
        H = np.zeros(36)
        b = np.zeros(6)
        chi2 = C.c_double(0)
        na = C.c_int32(0)
        self._check(self.lib.nid_multi_normal_equations(self.h, _dp(_d(pose7)), 1 if want_jac else 0, float(delta),
                                                        _dp(H), _dp(b), C.byref(chi2), C.byref(na)),
                    "nid_multi_normal_equations")
        return H.reshape(6, 6), b, chi2.value, na.value

    def launch_chain(self, first_slot, poses7, n_jac, delta):
        p = _d(np.asarray(poses7).reshape(-1, 7))
        self._check(self.lib.nid_multi_launch_chain(self.h, first_slot, p.shape[0], _dp(p), int(n_jac), float(delta)),
                    "nid_multi_launch_chain")

    def launch_batch(self, first_slot, poses7, delta, want_jac=True):
        p = _d(np.asarray(poses7).reshape(-1, 7))
        self._check(self.lib.nid_multi_launch_batch(self.h, first_slot, p.shape[0], _dp(p), 1 if want_jac else 0,
                                                    float(delta)), "nid_multi_launch_batch")

    def wait(self, slot):
        H = np.zeros(36)
        b = np.zeros(6)
        chi2 = C.c_double(0)
        na = C.c_int32(0)
        self._check(self.lib.nid_multi_wait(self.h, slot, _dp(H), _dp(b), C.byref(chi2), C.byref(na)), "nid_multi_wait")
        return H.reshape(6, 6), b, chi2.value, na.value

    def run_sequence(self, poses7, delta, batch=64, group=1, want_jac=True, collect=True):
        p = _d(np.asarray(poses7).reshape(-1, 7))
        out = np.zeros((p.shape[0], NID_REDUCED_LEN)) if collect else None
        self._check(self.lib.nid_multi_run_sequence(self.h, _dp(p), p.shape[0], int(batch), int(group),
                                                    1 if want_jac else 0, float(delta), _dp(out)),
                    "nid_multi_run_sequence")
        return out

    def contract_bytes(self):
        return int(self.lib.nid_multi_contract_bytes(self.h))


def multi_from_pair(pair, bin_num, devices=(0,), rank=None, world=None, jac_bound=JACBOUND_CPU, xform=XFORM_QUAT,
                    math=MATH_FAST, partition=None):
    import importlib
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    m = Multi(pair.rows, pair.cols, pair.cell, bin_num, pair.fx, pair.fy, pair.cx, pair.cy, devices=devices,
              rank=rank, world=world, jac_bound=jac_bound, xform=xform, partition=partition)
    m.set_math_mode(math)
    m.set_reference_depth(pair.depth_m, pair.im0, synth.matrix_colmajor16(pair.T_wc0))
    m.set_target(pair.im1)
    return m
