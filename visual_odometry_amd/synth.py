"""Deterministic synthetic inputs for the VO front end.

"Blob scenes": 3-D points seen from two camera poses, each rendered as an isotropic Gaussian spot on a
flat background plus low-amplitude band-limited noise.  The reference ships no data (its rosbags are
not in the repository, launch/*.launch:8), so every test and benchmark input comes from here.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# intrinsics of the reference's two YAML files (config/camera_intrinsic.yaml:2-5,
# config/rapallo_intrinsic.yaml:2-6 scaled to its 640x505 desired_size)
K_PARETE = np.array([[407.0646129842357, 0.0, 384.5], [0.0, 407.0646129842357, 246.5], [0.0, 0.0, 1.0]])
K_1080P = np.array([[1400.0, 0.0, 959.5], [0.0, 1400.0, 539.5], [0.0, 0.0, 1.0]])
K_4K = np.array([[2800.0, 0.0, 1919.5], [0.0, 2800.0, 1079.5], [0.0, 0.0, 1.0]])


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(rvec))
    if th < 1e-15:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


@dataclasses.dataclass
class Scene:
    img1: np.ndarray  # uint8 [H, W]
    img2: np.ndarray
    K: np.ndarray  # 3x3 float64
    R: np.ndarray  # X2 = R @ X1 + t
    t: np.ndarray  # unit 3-vector
    pts3d: np.ndarray  # [P, 3] in frame 1


def _value_noise(rng: np.random.Generator, h: int, w: int, cell: int, amp: float) -> np.ndarray:
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-amp, amp, size=(gh, gw))
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def _render(h: int, w: int, uv: np.ndarray, sigma: np.ndarray, amp: np.ndarray, noise: np.ndarray) -> np.ndarray:
    img = np.full((h, w), 128.0) + noise
    for (u, v), s, a in zip(uv, sigma, amp):
        r = int(np.ceil(3.5 * s)) + 1
        x0, x1 = int(np.floor(u)) - r, int(np.floor(u)) + r + 1
        y0, y1 = int(np.floor(v)) - r, int(np.floor(v)) + r + 1
        if x1 <= 0 or y1 <= 0 or x0 >= w or y0 >= h:
            continue
        x0c, x1c, y0c, y1c = max(x0, 0), min(x1, w), max(y0, 0), min(y1, h)
        xs = np.arange(x0c, x1c) - u
        ys = np.arange(y0c, y1c) - v
        gx = np.exp(-0.5 * (xs / s) ** 2)
        gy = np.exp(-0.5 * (ys / s) ** 2)
        img[y0c:y1c, x0c:x1c] += a * gy[:, None] * gx[None, :]
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def blob_scene(
    h: int,
    w: int,
    K: np.ndarray,
    n_blobs: int,
    seed: int,
    rot_deg: float = 3.0,
    baseline: float = 0.3,
    depth=(3.0, 12.0),
    sigma_px=(1.6, 7.0),
    noise_amp: float = 2.0,
    noise_cell: int = 6,
) -> Scene:
    """Two views of n_blobs random 3-D points.  View 2 = (R, t) with rotvec ~ U(-rot_deg, rot_deg)^3 and
    |t| = baseline (returned t is the unit direction, as recoverPose reports it)."""
    rng = np.random.default_rng(seed)
    rvec = np.deg2rad(rng.uniform(-rot_deg, rot_deg, size=3))
    R = rodrigues(rvec)
    tdir = rng.normal(size=3)
    tdir /= np.linalg.norm(tdir)
    t = tdir * baseline
    # features: uniform pixels in an enlarged view-1 frame, depth uniform.  Each feature is a main spot
    # plus two satellites at fixed 3-D offsets (same depth), so it has a dominant orientation and an
    # asymmetric neighbourhood: isotropic spots alone give SURF no repeatable orientation.
    m = 0.08
    u = rng.uniform(-m * w, (1 + m) * w, size=n_blobs)
    v = rng.uniform(-m * h, (1 + m) * h, size=n_blobs)
    z = rng.uniform(depth[0], depth[1], size=n_blobs)
    s1 = np.exp(rng.uniform(np.log(sigma_px[0]), np.log(sigma_px[1]), size=n_blobs))
    amp = rng.uniform(30.0, 90.0, size=n_blobs) * rng.choice([-1.0, 1.0], size=n_blobs)
    # three satellites per feature with random polar offset, size and signed amplitude: every feature gets
    # its own asymmetric neighbourhood (distinct descriptors), rigidly attached to the 3-D point
    ns = 3
    phi = rng.uniform(0, 2 * np.pi, size=(ns, n_blobs))
    rad = rng.uniform(1.2, 2.8, size=(ns, n_blobs)) * s1
    sfac = rng.uniform(0.5, 0.9, size=(ns, n_blobs))
    afac = rng.uniform(0.35, 0.9, size=(ns, n_blobs)) * rng.choice([-1.0, 1.0], size=(ns, n_blobs))
    du = np.concatenate([np.zeros(n_blobs)] + [rad[k] * np.cos(phi[k]) for k in range(ns)])
    dv = np.concatenate([np.zeros(n_blobs)] + [rad[k] * np.sin(phi[k]) for k in range(ns)])
    ua, va, za = np.tile(u, ns + 1) + du, np.tile(v, ns + 1) + dv, np.tile(z, ns + 1)
    sa = np.concatenate([s1] + [sfac[k] * s1 for k in range(ns)])
    aa = np.concatenate([amp] + [afac[k] * amp for k in range(ns)])
    Kinv = np.linalg.inv(K)
    Xa = (Kinv @ np.stack([ua, va, np.ones_like(ua)])) * za  # [3, 3P]
    Xb = R @ Xa + t[:, None]
    uvb = (K @ Xb)[:2] / Xb[2]
    sb = sa * za / Xb[2]
    n1 = _value_noise(rng, h, w, noise_cell, noise_amp)
    n2 = _value_noise(rng, h, w, noise_cell, noise_amp)
    img1 = _render(h, w, np.stack([ua, va], 1), sa, aa, n1)
    img2 = _render(h, w, uvb.T, sb, aa, n2)
    X1 = Xa[:, :n_blobs]
    return Scene(img1, img2, K.copy(), R, tdir, X1.T.copy())


def correspondences(
    n: int, K: np.ndarray, seed: int, inlier_frac: float = 0.5, noise_px: float = 0.5, h: int = 1080, w: int = 1920,
    rot_deg: float = 5.0, depth=(4.0, 20.0)
):
    """C5-style synthetic matches: inliers from a known pose (+ pixel noise), outliers uniform in the image.
    Returns p1 [n,2] f32, p2 [n,2] f32, R, t(unit), inlier flags."""
    rng = np.random.default_rng(seed)
    R = rodrigues(np.deg2rad(rng.uniform(-rot_deg, rot_deg, size=3)))
    t = rng.normal(size=3)
    t /= np.linalg.norm(t)
    u = rng.uniform(0, w, size=n)
    v = rng.uniform(0, h, size=n)
    z = rng.uniform(depth[0], depth[1], size=n)
    X1 = (np.linalg.inv(K) @ np.stack([u, v, np.ones(n)])) * z
    X2 = R @ X1 + t[:, None]
    uv2 = ((K @ X2)[:2] / X2[2]).T
    p1 = np.stack([u, v], 1) + rng.normal(0, noise_px, size=(n, 2))
    p2 = uv2 + rng.normal(0, noise_px, size=(n, 2))
    is_in = rng.uniform(size=n) < inlier_frac
    out2 = np.stack([rng.uniform(0, w, size=n), rng.uniform(0, h, size=n)], 1)
    p2 = np.where(is_in[:, None], p2, out2)
    return p1.astype(np.float32), p2.astype(np.float32), R, t, is_in


def surf_like_descriptors(n: int, seed: int) -> np.ndarray:
    """[n,64] f32 unit vectors with SURF-like statistics (signed sums and absolute sums per cell)."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 16, 4))
    d[:, :, 2:] = np.abs(d[:, :, 2:]) + np.abs(d[:, :, :2])
    d = d.reshape(n, 64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d.astype(np.float32)


# ---- named workloads of BASELINE.json / SURVEY.md 8d -------------------------------------------------------
def scene_c1(seed: int = 1001) -> Scene:
    """C1: Parete-like 640x480 pair, ~2k keypoints per frame (BASELINE configs[0]) and 500-900 matches,
    camera_intrinsic.yaml intrinsics."""
    return blob_scene(480, 640, K_PARETE, 1400, seed, depth=(6.0, 9.0))


def scene_c2(seed: int = 2002) -> Scene:
    """C2 (headline): synthetic 1080p textured pair, 8192 +- 5 % keypoints per frame at minHessian = 50."""
    return blob_scene(1080, 1920, K_1080P, 4300, seed, depth=(6.5, 8.5), baseline=0.3, sigma_px=(1.6, 5.0))


def roi_for(h: int, w: int, border: float = 0.04):
    """ROI of the reference scaled to the frame: the image minus a 4 % border."""
    return (round(border * w), round(border * h), round((1 - border) * w), round((1 - border) * h))


@dataclasses.dataclass
class SequenceScene:
    frames: list  # uint8 [H, W] each
    K: np.ndarray
    Rw: list  # X_k = Rw[k] @ X_0 + tw[k]
    tw: list

    def relative(self, j: int, k: int):
        """(R, t_unit, |t|) with X_k = R X_j + t, the convention of recoverPose."""
        R = self.Rw[k] @ self.Rw[j].T
        t = self.tw[k] - R @ self.tw[j]
        n = float(np.linalg.norm(t))
        return R, (t / n if n > 0 else t), n


def sequence_scene(h: int, w: int, K: np.ndarray, n_blobs: int, seed: int, poses, depth=(5.0, 9.0), sigma_px=(1.6, 6.0),
                   noise_amp: float = 2.0, blank=()) -> SequenceScene:
    """C3-style run: one static blob world seen from `poses` = [(rotvec_deg[3], position[3]) ...] (camera k sits at
    position c_k in the frame of camera 0 with orientation rodrigues(rotvec)); frames listed in `blank` are rendered
    without any structure (FAIL_DETECTION).  Repeating a pose gives a stationary frame (NOT_MOVING) with fresh noise."""
    rng = np.random.default_rng(seed)
    m = 0.35  # the world extends beyond view 0 so that later views stay covered
    u = rng.uniform(-m * w, (1 + m) * w, size=n_blobs)
    v = rng.uniform(-m * h, (1 + m) * h, size=n_blobs)
    z = rng.uniform(depth[0], depth[1], size=n_blobs)
    s1 = np.exp(rng.uniform(np.log(sigma_px[0]), np.log(sigma_px[1]), size=n_blobs))
    amp = rng.uniform(30.0, 90.0, size=n_blobs) * rng.choice([-1.0, 1.0], size=n_blobs)
    ns = 3
    phi = rng.uniform(0, 2 * np.pi, size=(ns, n_blobs))
    rad = rng.uniform(1.2, 2.8, size=(ns, n_blobs)) * s1
    sfac = rng.uniform(0.5, 0.9, size=(ns, n_blobs))
    afac = rng.uniform(0.35, 0.9, size=(ns, n_blobs)) * rng.choice([-1.0, 1.0], size=(ns, n_blobs))
    du = np.concatenate([np.zeros(n_blobs)] + [rad[k] * np.cos(phi[k]) for k in range(ns)])
    dv = np.concatenate([np.zeros(n_blobs)] + [rad[k] * np.sin(phi[k]) for k in range(ns)])
    ua, va, za = np.tile(u, ns + 1) + du, np.tile(v, ns + 1) + dv, np.tile(z, ns + 1)
    sa = np.concatenate([s1] + [sfac[k] * s1 for k in range(ns)])
    aa = np.concatenate([amp] + [afac[k] * amp for k in range(ns)])
    X0 = (np.linalg.inv(K) @ np.stack([ua, va, np.ones_like(ua)])) * za
    frames, Rw, tw = [], [], []
    for k, (rv, c) in enumerate(poses):
        Rk = rodrigues(np.deg2rad(np.asarray(rv, np.float64))).T  # world (camera 0) -> camera k
        tk = -Rk @ np.asarray(c, np.float64)
        Xk = Rk @ X0 + tk[:, None]
        noise = _value_noise(rng, h, w, 6, noise_amp)
        if k in blank:
            img = np.clip(np.rint(128.0 + noise), 0, 255).astype(np.uint8)
        else:
            ok = Xk[2] > 0.5
            uv = ((K @ Xk)[:2] / Xk[2]).T
            img = _render(h, w, uv[ok], (sa * za / Xk[2])[ok], aa[ok], noise)
        frames.append(img)
        Rw.append(Rk)
        tw.append(tk)
    return SequenceScene(frames, K.copy(), Rw, tw)


# rapallo camera of the reference (config/rapallo_camera_params.yaml): 2560 x 2056 frames, desired_size 640 x 505
K_RAPALLO_FULL = np.array([[2305.66, 0.0, 1281.94], [0.0, 2303.95, 1028.35], [0.0, 0.0, 1.0]])
DIST_RAPALLO = np.array([0.0626, 0.3415, 0.00312, -0.00555])


def bgr_frame(rows: int, cols: int, seed: int, contrast: float = 0.5) -> np.ndarray:
    """Seeded colour frame [rows, cols, 3] uint8 (BGR order as cv_bridge hands it to the node): low-contrast, blue-green
    tinted value noise at three scales plus pixel noise - the kind of underwater frame CLAHE is there for."""
    rng = np.random.default_rng(seed)
    base = _value_noise(rng, rows, cols, max(rows // 6, 4), 60.0) + _value_noise(rng, rows, cols, max(rows // 40, 2), 35.0)
    fine = _value_noise(rng, rows, cols, 3, 25.0)
    out = np.empty((rows, cols, 3), np.uint8)
    for c, (gain, offs) in enumerate(((1.0, 120.0), (0.85, 105.0), (0.45, 60.0))):
        ch = offs + contrast * gain * (base + fine + _value_noise(rng, rows, cols, 9, 12.0)) + rng.normal(0.0, 2.0, (rows, cols))
        out[..., c] = np.clip(np.rint(ch), 0, 255).astype(np.uint8)
    return out


# ---- C3: Rapallo-like survey run -------------------------------------------------------------------------
def k_rapallo_640() -> np.ndarray:
    """K of the node on the rapallo camera: config/rapallo_intrinsic.yaml through updateCameraMatrix for the 2560 x 2056
    frames of that camera and its desired_size 640 x 505 (camera_utility.hpp:138-145)."""
    import os

    from .config import load_camera_config

    cfg = load_camera_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "rapallo_intrinsic.yaml"))
    return cfg.K_for(2560, 2056)


_C3 = {}


def _c3_init(world):
    _C3.update(world)


def _c3_frame(k):
    """frame k of the run held in _C3 (set by _c3_init in this process): blobs in view rendered over fresh seeded noise"""
    g = _C3
    h, w, K = g["h"], g["w"], g["K"]
    Rk, tk = g["Rw"][k], g["tw"][k]
    Xk = Rk @ g["X0"] + tk[:, None]
    uv = (K @ Xk)[:2] / Xk[2]
    sig = g["sa"] * g["za"] / Xk[2]
    m = 4.0 * sig + 2.0
    vis = (Xk[2] > 0.5) & (uv[0] > -m) & (uv[0] < w + m) & (uv[1] > -m) & (uv[1] < h + m)
    noise = _value_noise(np.random.default_rng([g["seed"], k]), h, w, 6, g["noise_amp"])
    return _render(h, w, uv[:, vis].T, sig[vis], g["aa"][vis], noise)


def sequence_c3(n_frames: int = 512, seed: int = 3003, h: int = 505, w: int = 640, K: np.ndarray | None = None, step_m: float = 0.36,
                stationary=((60, 10), (250, 6), (400, 9)), depth: float = 7.0, relief: float = 1.2, density: float = 11.0,
                workers: int = 1) -> SequenceScene:
    """BASELINE configs[2] stand-in (the reference's rapallo bags are not in its repository): a down-looking camera over a
    blob seabed (plane at `depth` m with +-`relief` m of relief), moving `step_m` per frame along a cubic spline through
    seeded waypoints with slowly varying roll / pitch / yaw - about 30 px of image motion per frame at the rapallo K -
    and standing still for the `stationary` stretches ((first frame, length), ...: their pairs are NOT_MOVING, >= 16 frames
    by default).  Frames are independent given the world, so `workers` > 1 renders them in a process pool (spawned: call
    before anything initialises the GPU)."""
    from scipy.interpolate import CubicSpline

    K = k_rapallo_640() if K is None else np.asarray(K, np.float64)
    rng = np.random.default_rng(seed)
    still = np.zeros(n_frames, bool)
    for f0, ln in stationary:
        still[f0 : min(f0 + ln, n_frames)] = True
    still[0] = False
    n_move = int((~still).sum())
    length = step_m * n_move
    # waypoints: a lawnmower-like walk inside a box, so that the world stays small while the path is long
    box = np.array([max(30.0, length / 6.0), max(16.0, length / 12.0)])
    way, total = [np.zeros(2)], 0.0
    while total < 1.15 * length + 20.0 or len(way) < 6:
        for _ in range(50):
            a = rng.uniform(0, 2 * np.pi)
            cand = way[-1] + rng.uniform(9.0, 16.0) * np.array([np.cos(a), np.sin(a)])
            if np.all(np.abs(cand) < box / 2):
                break
        cand = np.clip(cand, -box / 2, box / 2)
        total += float(np.linalg.norm(cand - way[-1]))
        way.append(cand)
    way = np.array(way)
    chord = np.r_[0.0, np.cumsum(np.linalg.norm(np.diff(way, axis=0), axis=1))]
    spl = CubicSpline(chord, way)
    dense = spl(np.linspace(0, chord[-1], 20000))
    arc = np.r_[0.0, np.cumsum(np.linalg.norm(np.diff(dense, axis=0), axis=1))]
    if arc[-1] < length:
        raise ValueError("path shorter than the run: more waypoints needed")
    s_of = np.interp(step_m * np.arange(n_move), arc, np.arange(len(arc)))
    xy_move = np.stack([np.interp(s_of, np.arange(len(arc)), dense[:, 0]), np.interp(s_of, np.arange(len(arc)), dense[:, 1])], 1)
    # world: features over the band the views sweep
    half = np.array([0.5 * w / K[0, 0], 0.5 * h / K[1, 1]]) * (depth + relief) * 1.5
    lo, hi = dense.min(0) - half, dense.max(0) + half
    n_feat = int(density * np.prod(hi - lo))
    fx_, fy_ = rng.uniform(lo[0], hi[0], n_feat), rng.uniform(lo[1], hi[1], n_feat)
    fz = depth + _value_noise(rng, 64, 64, 8, relief)[rng.integers(0, 64, n_feat), rng.integers(0, 64, n_feat)] + rng.uniform(-0.2, 0.2, n_feat)
    s_m = np.exp(rng.uniform(np.log(0.02), np.log(0.075), n_feat))  # spot radius in metres
    amp = rng.uniform(30.0, 90.0, n_feat) * rng.choice([-1.0, 1.0], n_feat)
    ns = 3
    phi = rng.uniform(0, 2 * np.pi, (ns, n_feat))
    rad = rng.uniform(1.2, 2.8, (ns, n_feat)) * s_m
    sfac = rng.uniform(0.5, 0.9, (ns, n_feat))
    afac = rng.uniform(0.35, 0.9, (ns, n_feat)) * rng.choice([-1.0, 1.0], (ns, n_feat))
    X0 = np.stack([np.concatenate([fx_] + [fx_ + rad[i] * np.cos(phi[i]) for i in range(ns)]),
                   np.concatenate([fy_] + [fy_ + rad[i] * np.sin(phi[i]) for i in range(ns)]), np.tile(fz, ns + 1)])
    # spot size in pixels at frame k = f * radius / Z: stored as sigma at unit focal length, za = f so that sa * za / Z works
    sa = np.concatenate([s_m] + [sfac[i] * s_m for i in range(ns)])
    aa = np.concatenate([amp] + [afac[i] * amp for i in range(ns)])
    za = np.full(sa.shape, 0.5 * (K[0, 0] + K[1, 1]))
    # poses: camera k at (x, y, 0) of the world frame with a slowly varying attitude; a stationary frame repeats the pose
    Rw, tw, mv = [], [], 0
    for k in range(n_frames):
        if not still[k]:
            i = mv
            mv += 1
        c = np.array([xy_move[i, 0], xy_move[i, 1], 0.0])
        rv = np.array([1.5 * np.sin(0.021 * i + 0.3), 1.5 * np.cos(0.017 * i), 25.0 * np.sin(0.006 * i + 1.0)])
        Rk = rodrigues(np.deg2rad(rv)).T
        Rw.append(Rk)
        tw.append(-Rk @ c)
    world = {"h": h, "w": w, "K": K, "Rw": Rw, "tw": tw, "X0": X0, "sa": sa, "aa": aa, "za": za, "seed": seed, "noise_amp": 2.0}
    if workers > 1:
        import multiprocessing as mp

        pool = mp.get_context("spawn").Pool(workers, initializer=_c3_init, initargs=(world,))
        try:
            frames = pool.map(_c3_frame, range(n_frames), chunksize=max(1, n_frames // (4 * workers)))
        finally:
            pool.close()  # workers leave by themselves (a terminate() under a profiler's signal handler can hang)
            pool.join()
    else:
        _c3_init(world)
        frames = [_c3_frame(k) for k in range(n_frames)]
    # ground truth relative to camera 0 (SequenceScene convention: X_k = Rw[k] X_0 + tw[k] with frame 0 the reference)
    R0, t0 = Rw[0], tw[0]
    Rrel = [Rk @ R0.T for Rk in Rw]
    trel = [tk - Rr @ t0 for tk, Rr in zip(tw, Rrel)]
    return SequenceScene(frames, K.copy(), Rrel, trel)
