"""Offline replay of a frame sequence with the semantics of the node's main loop (src/visual_Odometry.cpp:268-470):
gates, prev_img bookkeeping (FAIL_DETECTION advances it, NOT_MOVING keeps it), relative pose carry-over on
FAIL_RECOVER, scale factor from the altitude, absolute pose chain and the vo_results / fail_check fields.

All consecutive pairs of a chunk run on the GPU in one sequence batch (every frame detected once); the
data-dependent part - which pair follows a NOT_MOVING gap - is a cheap sequential host pass that fetches the rare
re-paired results on demand.  No ROS, tf or OpenCV is involved: records come back as dictionaries."""
from __future__ import annotations

import numpy as np

from . import chain
from .frontend import Batch, Context, Sequence


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class _PairResults:
    """Results of pair (j, k): consecutive pairs come from the chunk's sequence batch, others are computed on demand."""

    def __init__(self, ctx: Context, frames, K, chunk: int, prep=None):
        self.ctx, self.frames, self.K, self.chunk = ctx, frames, np.ascontiguousarray(K, np.float64), max(2, chunk)
        self.cache: dict[tuple[int, int], dict] = {}
        self.prep = prep  # visual_odometry_amd.prep.Prep: frames are raw BGR and go through get_image on the device
        self.h, self.w = (prep.params.dst_rows, prep.params.dst_cols) if prep is not None else frames[0].shape

    def _put(self, batch, slot, frame):
        if self.prep is not None:
            self.prep.to_batch(frame, batch, slot)
        elif isinstance(batch, Sequence):
            batch.set_frame(slot, frame)

    @staticmethod
    def _collect(batch, p):
        out = batch.get(p)
        p1, p2, m, mask, Xk = batch.get_points(p)
        keep = mask.astype(bool)
        return {"status": out.status, "p1": p1, "p2": p2, "success": bool(out.success), "R": np.array(out.R).reshape(3, 3),
                "t": np.array(out.t), "n_inl": out.n_inl, "in1": p1[keep], "in2": p2[keep], "Xk": Xk, "n_kp": (out.n_kp1, out.n_kp2)}

    def _run_chunk(self, f0: int):
        f1 = min(f0 + self.chunk, len(self.frames))
        if f1 - f0 < 2:
            return
        seq = Sequence(self.ctx, f1 - f0, self.h, self.w)
        for f in range(f0, f1):
            self._put(seq, f - f0, self.frames[f])
        for p in range(f1 - f0 - 1):
            seq.set_pair(p, p, p + 1, self.K)
        seq.run()
        seq.sync()
        for p in range(f1 - f0 - 1):
            self.cache[(f0 + p, f0 + p + 1)] = self._collect(seq, p)
        seq.close()

    def get(self, j: int, k: int) -> dict:
        if (j, k) not in self.cache:
            if k == j + 1:
                self._run_chunk(j)
            else:  # pair across a NOT_MOVING gap
                b = Batch(self.ctx, 1, self.h, self.w)
                if self.prep is not None:
                    self._put(b, 0, self.frames[j])
                    self._put(b, 1, self.frames[k])
                    b.set_k(0, self.K)
                else:
                    b.set_pair(0, self.frames[j], self.frames[k], self.K)
                b.run()
                b.sync()
                self.cache[(j, k)] = self._collect(b, 0)
                b.close()
        return self.cache[(j, k)]


def replay(ctx: Context, frames, stamps, altitudes, K, orientation0, location0, Rbc=None, motion2D=False, gt_z=None,
           chunk: int = 64, prep=None):
    """frames: list of H x W uint8 preprocessed frames, or - with prep, a visual_odometry_amd.prep.Prep - the raw BGR frames
    of the camera topic, which then go through the node's get_image on the device; K is the camera matrix of the
    preprocessed frames (after updateCameraMatrix); stamps [s] and altitudes [m] per frame; orientation0 ({camera} -> {W}) and location0 as the node initialises them from the ground truth.
    Returns one record per loop iteration (frames 1..N-1)."""
    Rbc = np.eye(3) if Rbc is None else np.asarray(Rbc, np.float64)
    res = _PairResults(ctx, frames, K, chunk, prep)
    orientation, location = np.array(orientation0, np.float64), np.array(location0, np.float64)
    R, t, SF = np.eye(3), np.zeros(3), 1.0
    prev, prev_time = 0, stamps[0]
    records = []
    for k in range(1, len(frames)):
        pr = res.get(prev, k)
        rec = {"frame": k, "prev": prev, "n_matches": len(pr["p1"])}
        if chain.checkMinFeat(len(pr["p1"])):  # :303
            rec["fail"] = chain.FAIL_DETECTION
            records.append(rec)
            prev = k
            continue
        if not chain.checkIfMoving(pr["p1"], pr["p2"]):  # :316
            rec["fail"] = chain.NOT_MOVING
            records.append(rec)
            continue
        success = pr["success"]
        if success:  # optRelativePose leaves R, t untouched on failure (:669-682)
            R, t = pr["R"].copy(), pr["t"].copy()
        rec["fail"] = chain.SUCCESS if success else chain.FAIL_RECOVER
        world_points = None
        if motion2D:  # :364-371, before the triangulation
            t = t.copy()
            t[2] = 0.0
            R = _rotz(chain.mat2Euler(R)[2])
            if success:
                world_points = ctx.triangPoints(pr["in1"], pr["in2"], R, t, K)
        elif success:
            world_points = pr["Xk"]
        if success and world_points is not None and world_points.shape[1] > 0:
            SF = chain.scaleFactor(float(altitudes[k]), world_points)  # :387
        if success:
            location, orientation, wpw = chain.absolutePose(orientation, location, R, t, SF, world_points)
            rec["world_pointsW"] = wpw
        else:
            location, orientation = chain.absolutePose(orientation, location, R, t, SF)
        if motion2D and gt_z is not None:
            location = location.copy()
            location[2] = gt_z[k]
        dt = float(stamps[k] - prev_time)
        rpy = chain.mat2Euler(orientation @ Rbc.T)  # publish_VOResults :553
        lin, ang = chain.estimateTwist(dt, R, t, SF, rpy, Rbc, motion2D) if dt != 0 else (np.zeros(3), np.zeros(3))
        rec.update({"R": R.copy(), "t": t.copy(), "SF": SF, "estimate_pos": location.copy(), "estimate_rpy": rpy,
                    "twist_linear": lin, "twist_angular": ang, "orientation": orientation.copy(), "n_inliers": pr["n_inl"]})
        records.append(rec)
        prev, prev_time = k, stamps[k]
    return records
