"""Constructed parity cases (round 5): frame pairs and poses that PLACE samples on the reference's
decision points instead of waiting for a random generator to hit them.

The camera looks along the world's z axis (T_wc0 = identity) at a fronto-parallel or gently varying depth map, and the
evaluation poses are the identity plus translations -- so a reference pixel (r, c) projects to (c + s_x(z), r + s_y(z)) with
s = f * t / z: integer coordinates for t = 0 (every border test a tie decided by rounding), a chosen sub-pixel offset for a
chosen t, and ulp-sized offsets around either (t in multiples of 2^-48).  The target image then puts what the test is after
under those sample positions:

  knots    integer grey levels ON the knots k * 255 / S of the bin axis (S | 255: 4, 6, 8 bins) and next to them, so that
           interpolated samples cross knots within ulps                  (types_six_dof_expmap.cpp:577-580, B-spline spans)
  ends     0 / 1 / 254 / 255 patches and ramps: samples at and around the clamp 255 -> 254.999 and at 0, the END knots
           whose weights are linear in the distance                                         (:572-575)
  borders  the identity and ulp-shifted poses: u == 0, v == 0, u + 3 == cols (cost) / cols - 1 (Jacobian), v + 3 == rows
                                                                                               (:565, :433)
  counts   depth holes that leave a cell exactly 300 / 299 in-frame pixels at the reference pose (the activity threshold,
           :609-612), and poses that leave a cell 1..8 samples
  edges    vertical step edges of 200+ grey levels with the sub-pixel offset that puts the interpolated sample on a knot
           or within the last 1/8 before 255 -- the steepest transfer from an ulp of u to the bin position (sweep seed 511576)

`adversarial_case(synth, seed)` -> (pair, bins, href_pose, poses, kind, textured); seeds are independent."""
import dataclasses

import numpy as np

KINDS = ["knots", "ends", "borders", "counts", "edges"]


def _pose(t):
    return np.array([0.0, 0.0, 0.0, 1.0, float(t[0]), float(t[1]), float(t[2])])


def adversarial_case(synth, seed):
    rng = np.random.default_rng(900000 + seed)
    kind = KINDS[seed % len(KINDS)]
    G = int(rng.integers(1, 4))
    rb, cb = int(rng.integers(14, 33)), int(rng.integers(22, 41))
    if kind == "counts":
        rb, cb = max(rb, 20), max(cb, 26)
    rows, cols = G * rb + int(rng.integers(0, 2)), G * cb + int(rng.integers(0, 2))
    f = float(rng.choice([96.0, 120.3, 128.0, 200.0]))
    base = synth.make_pair("S", rows=rows, cols=cols, cell=G)
    nb = int(rng.choice([4, 6, 8])) if kind in ("knots", "edges") else int(rng.choice([4, 5, 6, 8, 10, 12]))
    S = nb - 3
    # depth: a plane (every pixel the same shift) or two planes / a gentle ramp, exact in u16 counts
    z0 = int(rng.choice([5000, 8192, 10000, 12500]))
    depth = np.full((rows, cols), z0, dtype=np.uint16)
    if rng.random() < 0.5:
        depth[:, cols // 2:] = z0 * 2 if z0 * 2 < 65535 else z0 // 2
    if rng.random() < 0.3:
        depth = (depth.astype(np.int64) + (np.arange(cols)[None, :] % 7) * 16).astype(np.uint16)
    zm = z0 / 5000.0
    im0 = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    if rng.random() < 0.5:   # few reference levels: reference samples on knots too
        im0 = np.array([0, 51, 85, 102, 153, 170, 204, 255], dtype=np.uint8)[rng.integers(0, 8, (rows, cols))]
    cc, rr = np.meshgrid(np.arange(cols), np.arange(rows))
    textured = True
    ulp = 2.0 ** -48   # translations in multiples of this move u by a few ulps (f * t / z ~ 1e-12 .. 1e-13 px)
    tiny = [np.array([a, b, 0.0]) * ulp * zm for a, b in ((0, 0), (1, 0), (-1, 0), (0, 1), (3, -2), (-4, 4), (16, 16), (-64, 32))]
    px = zm / f            # one pixel of shift in x

    if kind == "knots":
        knots = [k * 255 // S for k in range(S + 1)]
        lv = knots[int(rng.integers(0, S + 1))]
        # around one knot: the knot itself, +-1, and a ramp through it
        pat = np.array([0, 1, -1, 0, 2, -2, 0, 0], dtype=np.int64)
        im1 = np.clip(lv + pat[(cc + 3 * rr) % 8], 0, 255)
        im1[rows // 2:, :] = np.clip(np.array(knots)[(cc[rows // 2:, :] // 3) % (S + 1)] + ((rr[rows // 2:, :] % 3) - 1), 0, 255)
        shifts = [0.0, 0.5, 0.25, 1.0 / 3.0]
    elif kind == "ends":
        im1 = rng.integers(0, 256, (rows, cols))
        im1[: rows // 3, :] = 255 - (rng.random((rows // 3, cols)) < 0.2)            # 255 with specks of 254
        im1[rows // 3: 2 * rows // 3, :] = (rng.random((2 * rows // 3 - rows // 3, cols)) < 0.2)   # 0 with specks of 1
        im1[:, : cols // 4] = np.where(cc[:, : cols // 4] % 2 == 0, 255, 250)        # stripes just below the clamp
        if rng.random() < 0.5:
            im0 = im0.copy()
            im0[: rows // 2, : cols // 2] = 255                                       # saturated reference: tiny reference weights
        shifts = [0.0, 0.5, 2.0 ** -10, 1.0 - 2.0 ** -10]
    elif kind == "borders":
        im1 = rng.integers(0, 256, (rows, cols))
        if rng.random() < 0.3:
            im1[:, -6:] = 255
            im1[:4, :] = 0
        shifts = [0.0, 2.0 ** -30, -(2.0 ** -30), 2.0 ** -21, -(2.0 ** -21), 1.0, -1.0]
    elif kind == "counts":
        im1 = rng.integers(0, 256, (rows, cols))
        # cell 0: exactly 300 (or 299) pixels with a depth, all of them in frame at the reference pose; the cell at the right
        # border keeps 1..8 samples under a shift of `lose` pixels
        target = 300 - int(rng.integers(0, 2))
        if rb * cb >= 320:
            inframe = (cc[:rb, :cb] + 3 <= cols) & (rr[:rb, :cb] + 3 <= rows)   # (a single cell reaches the frame's border)
            blk = np.zeros(rb * cb, dtype=bool)
            blk[rng.permutation(np.flatnonzero(inframe))[:target]] = True
            blk = blk.reshape(rb, cb)
            depth = depth.copy()
            sub = depth[:rb, :cb]
            sub[~blk] = 0
            if G > 1:   # right-most cell of the first row of cells
                c0 = (G - 1) * cb
                keep = int(rng.integers(1, 9))
                lose = 6
                m = np.zeros((rb, cb), dtype=bool)
                band = np.zeros((rb, cb), dtype=bool)
                band[:, cb - 3 - lose: cb - 3] = True        # in frame at the identity (c + 3 <= cols), out under +lose px
                idx = np.flatnonzero(band)
                m.flat[rng.permutation(idx)[: min(len(idx), 300 - keep)]] = True
                left = np.zeros((rb, cb), dtype=bool)
                left[:, : cb // 3] = True
                idl = np.flatnonzero(left)
                m.flat[rng.permutation(idl)[:keep + max(0, 300 - keep - min(len(idx), 300 - keep))]] = True
                sub2 = depth[:rb, c0: c0 + cb]
                sub2[~m] = 0
        shifts = [0.0, 6.0, 5.0, 6.0 + 2.0 ** -20]
    else:  # edges
        lo, hi = [(55, 255), (0, 200), (30, 250), (5, 255)][int(rng.integers(0, 4))]
        period = int(rng.integers(3, 7))
        im1 = np.where((cc // period) % 2 == 0, lo, hi)
        if rng.random() < 0.5:
            im1 = np.where(((cc + rr) // period) % 2 == 0, lo, hi)
        knots = [k * 255.0 / S for k in range(1, S + 1)]
        targets = [k for k in knots if lo < k <= hi] + [hi - 0.01, hi - 0.1]
        shifts = [0.0] + [float((t_ - lo) / (hi - lo)) for t_ in targets[:5]]
    im1 = np.ascontiguousarray(np.clip(im1, 0, 255).astype(np.uint8))
    T_wc0 = np.eye(4)
    ident = _pose([0, 0, 0])
    pair = dataclasses.replace(base, fx=f, fy=-f, cx=(cols - 1) / 2.0, cy=(rows - 1) / 2.0, im0=np.ascontiguousarray(im0), im1=im1,
                               depth_u16=np.ascontiguousarray(depth), T_wc0=T_wc0, pose_true=ident, pose_init=ident)
    poses = []
    for s in shifts:
        for t in (tiny[int(rng.integers(0, len(tiny)))], tiny[0]):
            poses.append(_pose(np.array([s * px, 0.0, 0.0]) + t))
    if kind in ("borders", "knots"):
        poses.append(_pose(np.array([0.0, -0.5 * px, 0.0]) + tiny[2]))   # a vertical half pixel (fy < 0)
    # a mild rotation on top of one of them: off the lattice, but still close to the decision points
    poses.append(synth.perturb_pose7(poses[1], rng.normal(0, 2e-4, 3), rng.normal(0, 2e-4, 3)))
    return pair, nb, ident, poses[:12], kind, textured
