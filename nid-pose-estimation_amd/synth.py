"""Deterministic synthetic RGB-D pairs for the NID path.

The ETH-CVG dataset the reference's config points at
(config_eth_cvg.yaml:11) is not available offline, so every test and bench
input is generated here: a multi-scale blocky texture (64-bit LCG, fixed seed
20211003), a smooth depth map quantised to u16 at 5000/m
(NID_pose_estimation.cpp:73,106), a second view rendered under a known
relative motion followed by a global illumination change (the case NID is
designed for), and the reference's own pose disturbance
(NID_pose_estimation.cpp:186-212).

Pure numpy; no oracle, no GPU.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

LCG_SEED = 20211003
LCG_A = 6364136223846793005
LCG_C = 1442695040888963407
MASK64 = (1 << 64) - 1


def _lcg_bytes(n: int, state: int):
    out = np.empty(n, dtype=np.uint8)
    for i in range(n):
        state = (state * LCG_A + LCG_C) & MASK64
        out[i] = (state >> 33) & 0xFF
    return out, state


# --------------------------------------------------------------------------
# SE(3) helpers (numpy; conventions of g2o::SE3Quat, se3quat.h)
# pose7 = [qx, qy, qz, qw, tx, ty, tz]
# --------------------------------------------------------------------------
def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def quat_from_R(R):
    """Shepperd form used by Eigen's Quaterniond(Matrix3d); returns xyzw, w >= 0, unit."""
    m = np.asarray(R, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_R(q):
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([
        [1 - (tyy + tzz), txy - twz, txz + twy],
        [txy + twz, 1 - (txx + tzz), tyz - twx],
        [txz - twy, tyz + twx, 1 - (txx + tyy)],
    ])


def pose7_from_Rt(R, t):
    return np.concatenate([quat_from_R(R), np.asarray(t, dtype=np.float64)])


def pose7_to_matrix(p):
    """4x4 homogeneous matrix (row-major numpy array)."""
    M = np.eye(4)
    M[:3, :3] = quat_to_R(p[:4])
    M[:3, 3] = p[4:7]
    return M


def matrix_colmajor16(M):
    """Eigen .data() layout of a 4x4 (column-major 16 doubles)."""
    return np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(16).copy()


def pose7_minimal(p):
    """SE3Quat::toMinimalVector (se3quat.h:155-164): (t, q.xyz)."""
    return np.array([p[4], p[5], p[6], p[0], p[1], p[2]])


# --------------------------------------------------------------------------
@dataclasses.dataclass
class Pair:
    rows: int
    cols: int
    cell: int
    fx: float
    fy: float
    cx: float
    cy: float
    im0: np.ndarray        # u8 [rows, cols]
    im1: np.ndarray        # u8 [rows, cols]
    depth_u16: np.ndarray  # u16 [rows, cols], 5000 counts / metre
    T_wc0: np.ndarray      # 4x4 camera-0-to-world
    pose_true: np.ndarray  # pose7 world->camera-1 (T_cw1)
    pose_init: np.ndarray  # pose7 disturbed start (NID_pose_estimation.cpp:186-212)

    @property
    def depth_m(self):
        # NID_pose_estimation.cpp:73,106: convertTo(CV_64F, 1.0/5000)
        return self.depth_u16.astype(np.float64) * (1.0 / 5000)

    @property
    def intr(self):
        return np.array([self.fx, self.fy, self.cx, self.cy, 1.0 / 5000])


def blocky_texture(rows, cols, seed=LCG_SEED):
    """Multi-scale blocky texture (block sizes 64,32,16,8), 5x5 box blur, u8."""
    acc = np.zeros((rows, cols), dtype=np.float64)
    state = seed
    for s in (64, 32, 16, 8):
        gr, gc = -(-rows // s), -(-cols // s)
        vals, state = _lcg_bytes(gr * gc, state)
        grid = vals.reshape(gr, gc).astype(np.float64)
        up = np.repeat(np.repeat(grid, s, axis=0), s, axis=1)[:rows, :cols]
        acc += s * up
    lo, hi = acc.min(), acc.max()
    acc = 5.0 + (acc - lo) * (245.0 / (hi - lo))
    # 5x5 box blur with edge replication
    pad = np.pad(acc, 2, mode="edge")
    blur = np.zeros_like(acc)
    for dr in range(5):
        for dc in range(5):
            blur += pad[dr:dr + rows, dc:dc + cols]
    blur /= 25.0
    return np.clip(np.rint(blur), 0, 255).astype(np.uint8)


def analytic_texture(rows, cols):
    r = np.arange(rows, dtype=np.float64)[:, None]
    c = np.arange(cols, dtype=np.float64)[None, :]
    img = (127 + 60 * np.sin(0.05 * c) * np.cos(0.07 * r) + 40 * np.sin(0.013 * (r + c))
           + 20 * np.sin(0.31 * c + 0.23 * r))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def smooth_depth_u16(rows, cols, scale=1.0):
    r = np.arange(rows, dtype=np.float64)[:, None] / scale
    c = np.arange(cols, dtype=np.float64)[None, :] / scale
    d = 2.0 + 0.5 * np.sin(0.01 * r) + 0.3 * np.cos(0.008 * c)
    return np.rint(d * 5000.0).astype(np.uint16)


def render_second_view(im0, depth_m, fx, fy, cx, cy, T_wc0, T_cw1):
    """Forward bilinear splat of frame 0 into camera 1 (normalised); holes are
    filled from frame 0 at the same pixel."""
    rows, cols = im0.shape
    r = np.arange(rows, dtype=np.float64)[:, None]
    c = np.arange(cols, dtype=np.float64)[None, :]
    z = depth_m
    x = z * (c - cx) / fx
    y = z * (r - cy) / fy
    P = np.stack([x, y, z, np.ones_like(z)], axis=-1).reshape(-1, 4)
    M = T_cw1 @ T_wc0
    Q = P @ M.T
    u = fx * Q[:, 0] / Q[:, 2] + cx
    v = fy * Q[:, 1] / Q[:, 2] + cy
    val = im0.reshape(-1).astype(np.float64)
    ok = (z.reshape(-1) > 0.01) & (Q[:, 2] > 0)
    iu = np.floor(u).astype(np.int64)
    iv = np.floor(v).astype(np.int64)
    du = u - iu
    dv = v - iv
    num = np.zeros(rows * cols)
    den = np.zeros(rows * cols)
    for oy, ox, w in ((0, 0, (1 - du) * (1 - dv)), (0, 1, du * (1 - dv)),
                      (1, 0, (1 - du) * dv), (1, 1, du * dv)):
        xx = iu + ox
        yy = iv + oy
        m = ok & (xx >= 0) & (xx < cols) & (yy >= 0) & (yy < rows)
        idx = yy[m] * cols + xx[m]
        np.add.at(num, idx, w[m] * val[m])
        np.add.at(den, idx, w[m])
    out = im0.reshape(-1).astype(np.float64).copy()
    m = den > 1e-3
    out[m] = num[m] / den[m]
    return out.reshape(rows, cols)


def illumination_change(img_f):
    """u8(min(255, 0.8*255*(I/255)^0.6 + 10)): global gain/gamma/offset."""
    out = 0.8 * 255.0 * np.power(np.clip(img_f, 0, 255) / 255.0, 0.6) + 10.0
    return np.clip(np.rint(np.minimum(out, 255.0)), 0, 255).astype(np.uint8)


def _upsample2(img_f):
    """2x bilinear upsample with pixel-centre alignment: out[2i+a] samples i + (a-0.5)/2."""
    rows, cols = img_f.shape

    def axis_up(a, n, axis):
        pos = (np.arange(2 * n, dtype=np.float64) - 0.5) / 2.0
        i0 = np.clip(np.floor(pos).astype(np.int64), 0, n - 1)
        i1 = np.clip(i0 + 1, 0, n - 1)
        f = np.clip(pos - np.floor(pos), 0, 1)
        a0 = np.take(a, i0, axis=axis)
        a1 = np.take(a, i1, axis=axis)
        shape = [1, 1]
        shape[axis] = 2 * n
        f = f.reshape(shape)
        return a0 * (1 - f) + a1 * f

    return axis_up(axis_up(img_f, rows, 0), cols, 1)


def disturb_pose(R_cw, t_cw, r_offset=0.005, t_offset=0.02):
    """NID_pose_estimation.cpp:186-212."""
    a = r_offset * math.pi
    rot = rot_x(a) @ rot_y(a) @ rot_z(a)
    t_dist = np.array([0.5 * t_offset, -t_offset, -t_offset])
    return rot @ R_cw, t_cw + t_dist


def flash_term(rows, cols, amplitude=300.0, sigma_frac=0.12):
    """Radial "flash" of the second exposure (BASELINE configs[0] is the ETH-CVG
    real_flash pair): a Gaussian hot spot slightly off the image centre, strong enough to drive a disc of
    about 13 % of the pixels into saturation (255) after the illumination change."""
    r = np.arange(rows, dtype=np.float64)[:, None]
    c = np.arange(cols, dtype=np.float64)[None, :]
    r0, c0 = 0.45 * rows, 0.55 * cols
    sig = sigma_frac * cols
    return amplitude * np.exp(-((r - r0) ** 2 + (c - c0) ** 2) / (2.0 * sig * sig))


def make_pair(config="A", texture="blocky", edge_cases=False, rows=None, cols=None, cell=None,
              r_offset=0.005, t_offset=0.02, flash=False):
    """config 'A' = 640x480 / 16x16 cells, 'B' = 1280x960 / 32x32 cells (A upsampled 2x),
    'S' = 160x120 / 4x4 cells (small parity case; cells stay 30x40 px).  flash=True adds the saturating
    hot spot of flash_term() to the second image; edge_cases=True adds 5 % zero-depth holes, saturated and
    black patches and an inactive cell."""
    base = dict(fx=481.20, fy=-480.0, cx=319.5, cy=239.5)
    if config == "S":
        R_, C_, G_ = 120, 160, 4
        fx, fy, cx, cy = 120.3, -120.0, 79.5, 59.5
    else:
        R_, C_, G_ = 480, 640, 16
        fx, fy, cx, cy = base["fx"], base["fy"], base["cx"], base["cy"]
    if rows is not None:
        R_, C_, G_ = rows, cols, cell
    tex = blocky_texture(R_, C_) if texture == "blocky" else analytic_texture(R_, C_)
    depth_u16 = smooth_depth_u16(R_, C_, scale=1.0 if config != "S" else 0.25)
    if edge_cases:
        st = LCG_SEED ^ 0x5DEECE66D
        holes, st = _lcg_bytes(R_ * C_, st)
        depth_u16 = depth_u16.copy()
        depth_u16[holes.reshape(R_, C_) < 13] = 0            # ~5 % zero-depth holes
        tex = tex.copy()
        tex[R_ // 8: R_ // 8 + 24, C_ // 8: C_ // 8 + 36] = 255   # saturated patch
        tex[R_ // 2: R_ // 2 + 24, C_ // 2: C_ // 2 + 36] = 0     # black patch
        # one cell with almost no valid depth -> N_c < 300 -> inactive
        rb, cb = R_ // G_, C_ // G_
        depth_u16[rb * (G_ - 2): rb * (G_ - 1), cb * 1: cb * 2][:, : cb - 6] = 0
    depth_m = depth_u16.astype(np.float64) * (1.0 / 5000)

    # camera 0 in the world: mild rotation + offset (exercises the back-projection)
    T_wc0 = np.eye(4)
    T_wc0[:3, :3] = rot_z(0.02) @ rot_y(-0.03) @ rot_x(0.01)
    T_wc0[:3, 3] = [0.10, -0.05, 0.20]
    # true relative motion cam0 -> cam1
    deg = math.pi / 180.0
    T_01 = np.eye(4)
    T_01[:3, :3] = rot_z(0.4 * deg) @ rot_y(0.8 * deg) @ rot_x(-0.3 * deg)
    T_01[:3, 3] = [0.03, 0.01, -0.02]
    T_wc1 = T_wc0 @ T_01
    T_cw1 = np.linalg.inv(T_wc1)

    im1_f = render_second_view(tex, depth_m, fx, fy, cx, cy, T_wc0, T_cw1)
    im1 = illumination_change(im1_f)
    if flash:   # saturating hot spot on top of the global gain / gamma / offset change
        im1 = np.clip(np.rint(np.minimum(im1.astype(np.float64) + flash_term(R_, C_), 255.0)), 0, 255).astype(np.uint8)
    if edge_cases:
        im1 = im1.copy()
        im1[R_ // 3: R_ // 3 + 20, C_ // 3: C_ // 3 + 30] = 255
        im1[2 * R_ // 3: 2 * R_ // 3 + 20, C_ // 4: C_ // 4 + 30] = 0

    if config == "B":
        tex = np.clip(np.rint(_upsample2(tex.astype(np.float64))), 0, 255).astype(np.uint8)
        im1 = np.clip(np.rint(_upsample2(im1.astype(np.float64))), 0, 255).astype(np.uint8)
        d_up = _upsample2(depth_u16.astype(np.float64))
        if edge_cases:
            d_up = np.repeat(np.repeat(depth_u16, 2, 0), 2, 1).astype(np.float64)
        depth_u16 = np.rint(d_up).astype(np.uint16)
        R_, C_, G_ = 2 * R_, 2 * C_, 2 * G_
        fx, fy = 2 * fx, 2 * fy
        cx, cy = 2 * cx + 0.5, 2 * cy + 0.5

    R_cw, t_cw = T_cw1[:3, :3], T_cw1[:3, 3]
    pose_true = pose7_from_Rt(R_cw, t_cw)
    R_d, t_d = disturb_pose(R_cw, t_cw, r_offset, t_offset)
    pose_init = pose7_from_Rt(R_d, t_d)
    return Pair(rows=R_, cols=C_, cell=G_, fx=fx, fy=fy, cx=cx, cy=cy,
                im0=np.ascontiguousarray(tex), im1=np.ascontiguousarray(im1),
                depth_u16=np.ascontiguousarray(depth_u16), T_wc0=T_wc0,
                pose_true=pose_true, pose_init=pose_init)


def perturb_pose7(p, omega, upsilon):
    """Left-multiply by exp((omega, upsilon)) -- numpy twin of VertexSE3Expmap::oplusImpl
    (types_six_dof_expmap.h:74-77), used to build pose sequences for benches/tests."""
    omega = np.asarray(omega, dtype=np.float64)
    upsilon = np.asarray(upsilon, dtype=np.float64)
    th = float(np.linalg.norm(omega))
    Om = np.array([[0, -omega[2], omega[1]], [omega[2], 0, -omega[0]], [-omega[1], omega[0], 0]])
    Om2 = Om @ Om
    if th < 1e-5:
        R = np.eye(3) + Om + Om2
        V = R
    else:
        R = np.eye(3) + math.sin(th) / th * Om + (1 - math.cos(th)) / th ** 2 * Om2
        V = np.eye(3) + (1 - math.cos(th)) / th ** 2 * Om + (th - math.sin(th)) / th ** 3 * Om2
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ upsilon
    M = T @ pose7_to_matrix(p)
    return pose7_from_Rt(M[:3, :3], M[:3, 3])
