"""Offline replay of a frame sequence with the semantics of the node's main loop (src/visual_Odometry.cpp:268-470):
gates, prev_img bookkeeping (FAIL_DETECTION advances it, NOT_MOVING keeps it), relative pose carry-over on
FAIL_RECOVER, scale factor from the altitude, absolute pose chain and the vo_results / fail_check fields.

All consecutive pairs of a chunk run on the GPU in one sequence batch (every frame detected once); the
data-dependent part - which pair follows a NOT_MOVING gap - is a cheap sequential host pass.  `replay` fetches the rare
re-paired results on demand on one GPU; `replay_sharded` (BASELINE configs[2]) gives every rank a contiguous block of
frames with one overlapping frame, gathers fixed-size pair summaries over the process group, runs the same host pass on
every rank and computes the pairs it found missing (the ones across NOT_MOVING gaps) in further, small waves.
No ROS, tf or OpenCV is involved: records come back as dictionaries."""
from __future__ import annotations

import numpy as np

from . import chain, shard
from .frontend import Batch, Context, Sequence


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


# ---- fixed-size summary of one pair: what the gates and the pose chain of the node need from it ----------------------
# [n_matches, moving, median_px, success, n_inliers, n_world, sf, status, R (9), t (3)]; sf = scaleFactor(altitude_k, Xk)
# (camera_utility.hpp:384) when the pair succeeded and kept world points, NaN otherwise
SUMMARY_WIDTH = 20


def summarize(pr: dict, altitude_k: float) -> np.ndarray:
    s = np.zeros(SUMMARY_WIDTH)
    n = len(pr["p1"])
    moving, med = (False, 0.0)
    if n > 0:
        moving, med = chain.checkIfMoving(pr["p1"], pr["p2"], return_median=True)
    ok = bool(pr["success"])
    Xk = pr.get("Xk")
    nw = 0 if Xk is None else Xk.shape[1]
    s[0], s[1], s[2], s[3], s[4], s[5] = n, float(moving), med, float(ok), pr["n_inl"], nw
    s[6] = chain.scaleFactor(float(altitude_k), Xk) if ok and nw > 0 else np.nan
    s[7] = pr["status"]
    s[8:17] = np.asarray(pr["R"], np.float64).ravel()
    s[17:20] = np.asarray(pr["t"], np.float64).ravel()
    return s


class _PairResults:
    """Results of pair (j, k) on one GPU: consecutive pairs come from sequence batches, others from pair batches.  The
    batches belong to the context (Context.cached_batch) and are reused from call to call; two sequence batches alternate so
    that chunk c + 1 is staged, uploaded and running while the results of chunk c are read (one download per chunk)."""

    def __init__(self, ctx: Context, frames, K, chunk: int, prep=None):
        self.ctx, self.frames, self.K, self.chunk = ctx, frames, np.ascontiguousarray(K, np.float64), max(2, chunk)
        self.cache: dict[tuple[int, int], dict] = {}
        self.prep = prep  # visual_odometry_amd.prep.Prep: frames are raw BGR and go through get_image on the device
        self.h, self.w = (prep.params.dst_rows, prep.params.dst_cols) if prep is not None else frames[0].shape
        self._flip = 0
        self.altitudes = None  # set by replay_sharded: pair summaries are then made as the results come in, under the next chunk

    def close(self):  # the batches stay with the context
        pass

    def _seq(self, n_frames: int) -> Sequence:
        self._flip ^= 1
        return self.ctx.cached_batch(("seq", n_frames, self.h, self.w, self._flip), lambda: Sequence(self.ctx, n_frames, self.h, self.w))

    def _batch(self, n_pairs: int) -> Batch:
        return self.ctx.cached_batch(("pairs", n_pairs, self.h, self.w), lambda: Batch(self.ctx, n_pairs, self.h, self.w))

    def _put(self, batch, slot, frame):
        if self.prep is not None:
            self.prep.to_batch(frame, batch, slot)
        elif isinstance(batch, Sequence):
            batch.set_frame(slot, frame)

    def _collect_all(self, batch, keys):
        """results of every pair of the batch's latest run (waits for that run only)"""
        outs, pts = batch.get_all()
        for key, out, (p1, p2, m, mask, Xk) in zip(keys, outs, pts):
            keep = mask.astype(bool)
            pr = {"status": out.status, "p1": p1, "p2": p2, "success": bool(out.success), "R": np.array(out.R).reshape(3, 3),
                  "t": np.array(out.t), "n_inl": out.n_inl, "in1": p1[keep], "in2": p2[keep], "Xk": Xk, "n_kp": (out.n_kp1, out.n_kp2)}
            if self.altitudes is not None:
                pr["summary"] = summarize(pr, self.altitudes[key[1]])
            self.cache[key] = pr

    def _start_chunk(self, f0: int, f1: int):
        """enqueue the consecutive pairs of frames f0 .. f1 - 1; returns (batch, keys) for _collect_all"""
        seq = self._seq(f1 - f0)
        for f in range(f0, f1):
            self._put(seq, f - f0, self.frames[f])
        for p in range(f1 - f0 - 1):
            seq.set_pair(p, p, p + 1, self.K)
        seq.run()
        return seq, [(f0 + p, f0 + p + 1) for p in range(f1 - f0 - 1)]

    def _run_chunk(self, f0: int, f1: int | None = None):
        """consecutive pairs of frames f0 .. f1 - 1 (default: one chunk from f0)"""
        f1 = min(f0 + self.chunk, len(self.frames)) if f1 is None else f1
        if f1 - f0 >= 2:
            self._collect_all(*self._start_chunk(f0, f1))

    def run_block(self, f0: int, f1: int):
        """every consecutive pair of the frames f0 .. f1 (inclusive), chunk by chunk (chunks overlap by one frame); the next
        chunk is enqueued before the previous one is read"""
        f, pending = f0, None
        while f < f1:
            e = min(f + self.chunk - 1, f1)
            started = self._start_chunk(f, e + 1) if e + 1 - f >= 2 else None
            if pending is not None:
                self._collect_all(*pending)
            pending = started
            f = e
        if pending is not None:
            self._collect_all(*pending)

    def run_pairs(self, pairs):
        """arbitrary pairs (those across NOT_MOVING gaps) in pair batches"""
        pairs = [p for p in pairs if p not in self.cache]
        for c0 in range(0, len(pairs), self.chunk):
            part = pairs[c0 : c0 + self.chunk]
            b = self._batch(len(part))
            for i, (j, k) in enumerate(part):
                if self.prep is not None:
                    self._put(b, 2 * i, self.frames[j])
                    self._put(b, 2 * i + 1, self.frames[k])
                    b.set_k(i, self.K)
                else:
                    b.set_pair(i, self.frames[j], self.frames[k], self.K)
            b.run()
            self._collect_all(b, part)

    def get(self, j: int, k: int) -> dict:
        if (j, k) not in self.cache:
            if k == j + 1:
                self._run_chunk(j)
            else:  # pair across a NOT_MOVING gap
                self.run_pairs([(j, k)])
        return self.cache[(j, k)]


def _walk(n_frames, stamps, altitudes, summary_of, full_of, ctx, K, orientation0, location0, Rbc, motion2D, gt_z, gates_only=False):
    """The node's loop over frames 1 .. n - 1 (src/visual_Odometry.cpp:268-470) on pair summaries.
    summary_of(j, k) -> summary vector or None (not computed yet); full_of(j, k) -> the full result (matched points, world
    points) if this process holds it, else None.  Returns (records, missing): when `missing` is not empty the walk went on
    speculatively past the missing pairs - assuming that a pair which starts in a stationary stretch is NOT_MOVING too - so
    that every pair a second pass is likely to need is listed at once; its records are then void.
    gates_only: only the gate logic that decides which pair follows which (checkMinFeat, checkIfMoving, prev_img
    bookkeeping) - what the sharded replay needs to find the pairs still to be computed; no pose chain, no records."""
    orientation, location = np.array(orientation0, np.float64), np.array(location0, np.float64)
    R, t, SF = np.eye(3), np.zeros(3), 1.0
    prev, prev_time = 0, stamps[0]
    records, missing = [], []
    for k in range(1, n_frames):
        s = summary_of(prev, k)
        if s is None:
            missing.append((prev, k))
            near = summary_of(k - 1, k) if prev != k - 1 else None
            still = near is not None and near[0] >= chain.MIN_NUM_FEATURES and not near[1]
            if not still:
                prev, prev_time = k, stamps[k]
            continue
        n_matches = int(s[0])
        rec = {"frame": k, "prev": prev, "n_matches": n_matches}
        if chain.checkMinFeat(n_matches):  # :303
            rec["fail"] = chain.FAIL_DETECTION
            records.append(rec)
            prev = k
            continue
        if not s[1]:  # checkIfMoving :316
            rec["fail"] = chain.NOT_MOVING
            records.append(rec)
            continue
        if gates_only:
            prev, prev_time = k, stamps[k]
            continue
        success = bool(s[3])
        if success:  # optRelativePose leaves R, t untouched on failure (:669-682)
            R, t = s[8:17].reshape(3, 3).copy(), s[17:20].copy()
        rec["fail"] = chain.SUCCESS if success else chain.FAIL_RECOVER
        pr = full_of(prev, k)
        world_points = None
        if motion2D:  # :364-371, before the triangulation
            t = t.copy()
            t[2] = 0.0
            R = _rotz(chain.mat2Euler(R)[2])
            if success:
                if pr is None:
                    raise NotImplementedError("motion2D needs the inlier points of every pair: use replay(), not replay_sharded()")
                world_points = ctx.triangPoints(pr["in1"], pr["in2"], R, t, K)
                if world_points.shape[1] > 0:
                    SF = chain.scaleFactor(float(altitudes[k]), world_points)  # :387
        elif success:
            world_points = pr["Xk"] if pr is not None else None
            if s[5] > 0:
                SF = float(s[6])  # :387 scaleFactor(altitude, world_points), evaluated where the points are
        dt = float(stamps[k] - prev_time)
        if not motion2D:  # one library call: absolutePose, then rpy (publish_VOResults :553) and the twist
            orientation, location = np.ascontiguousarray(orientation), np.ascontiguousarray(location)
            rpy, lin, ang, wpw = chain.loopPoseUpdate(orientation, location, R, t, SF, dt, Rbc, False,
                                                      world_points if success and world_points is not None else None)
            if wpw is not None:
                rec["world_pointsW"] = wpw
        else:
            if success and world_points is not None:
                location, orientation, wpw = chain.absolutePose(orientation, location, R, t, SF, world_points)
                rec["world_pointsW"] = wpw
            else:  # the pose part of absolutePose does not depend on the points
                location, orientation = chain.absolutePose(orientation, location, R, t, SF)
            if gt_z is not None:
                location = location.copy()
                location[2] = gt_z[k]
            rpy = chain.mat2Euler(orientation @ Rbc.T)  # publish_VOResults :553
            lin, ang = chain.estimateTwist(dt, R, t, SF, rpy, Rbc, motion2D) if dt != 0 else (np.zeros(3), np.zeros(3))
        rec.update({"R": R.copy(), "t": t.copy(), "SF": SF, "estimate_pos": location.copy(), "estimate_rpy": rpy,
                    "twist_linear": lin, "twist_angular": ang, "orientation": orientation.copy(), "n_inliers": int(s[4])})
        records.append(rec)
        prev, prev_time = k, stamps[k]
    return records, missing


def replay(ctx: Context, frames, stamps, altitudes, K, orientation0, location0, Rbc=None, motion2D=False, gt_z=None,
           chunk: int = 64, prep=None):
    """frames: list of H x W uint8 preprocessed frames, or - with prep, a visual_odometry_amd.prep.Prep - the raw BGR frames
    of the camera topic, which then go through the node's get_image on the device; K is the camera matrix of the
    preprocessed frames (after updateCameraMatrix); stamps [s] and altitudes [m] per frame; orientation0 ({camera} -> {W}) and location0 as the node initialises them from the ground truth.
    Returns one record per loop iteration (frames 1..N-1)."""
    Rbc = np.eye(3) if Rbc is None else np.asarray(Rbc, np.float64)
    res = _PairResults(ctx, frames, K, chunk, prep)

    def summary_of(j, k):
        return summarize(res.get(j, k), altitudes[k])

    try:
        records, missing = _walk(len(frames), stamps, altitudes, summary_of, res.get, ctx, K, orientation0, location0, Rbc, motion2D, gt_z)
    finally:
        res.close()
    assert not missing
    return records


def replay_sharded(ctx, frames, stamps, altitudes, K, orientation0, location0, rank: int, world: int, Rbc=None, chunk: int = 128,
                   prep=None, engine=None, device="cpu", stats: dict | None = None):
    """BASELINE configs[2]: the replay of one sequence over `world` ranks (one process and one GPU each; every rank holds the
    frame list).  Rank r detects the frames of block r (contiguous, one frame shared with the next block) once each and
    computes the block's consecutive pairs; the pair summaries (SUMMARY_WIDTH doubles per pair) are all-gathered - the only
    collective, RCCL when `device` is "cuda"; every rank then runs the node's sequential loop on the table.  Pairs across
    NOT_MOVING gaps (the node keeps prev_img, :316-326) are not in the table: the loop lists them, they are computed round
    robin over the ranks, gathered, and the loop runs again, until it completes (normally one extra wave).
    Returns the same records as replay() on every rank; `world_pointsW` only for the pairs the rank computed itself.
    engine: object with run_block(f0, f1), run_pairs(pairs), cache - the GPU engine by default; tests pass a CPU stand-in."""
    Rbc = np.eye(3) if Rbc is None else np.asarray(Rbc, np.float64)
    n = len(frames)
    eng = engine if engine is not None else _PairResults(ctx, frames, K, chunk, prep)
    if engine is None:
        eng.altitudes = altitudes
    table: dict[tuple[int, int], np.ndarray] = {}

    def exchange(local_pairs, counts):
        rows = np.zeros((len(local_pairs), 2 + SUMMARY_WIDTH))
        for i, (j, k) in enumerate(local_pairs):
            rows[i, 0], rows[i, 1] = j, k
            pr = eng.cache[(j, k)]
            rows[i, 2:] = pr["summary"] if "summary" in pr else summarize(pr, altitudes[k])
        for row in shard.gather_rows(rows, counts, rank, world, device=device):
            table[(int(row[0]), int(row[1]))] = row[2:].copy()

    # wave 1: consecutive pairs, contiguous frame blocks
    blocks = shard.frame_blocks(n, world)
    f0, f1 = blocks[rank]
    if f1 > f0:
        eng.run_block(f0, f1)
    exchange([(f, f + 1) for f in range(f0, f1)], [b1 - b0 for b0, b1 in blocks])
    waves, extra = 1, 0
    while True:
        # the gate logic alone finds the pairs that are still missing; the pose chain runs once, at the end
        _, missing = _walk(n, stamps, altitudes, lambda j, k: table.get((j, k)), lambda j, k: None, ctx, K, orientation0, location0, Rbc,
                           False, None, gates_only=True)
        if not missing:
            break
        # further waves: the pairs the loop could not find, round robin over the ranks
        mine = [missing[i] for i in shard.pairs_of_rank(len(missing), rank, world)]
        eng.run_pairs(mine)
        exchange(mine, [len(shard.pairs_of_rank(len(missing), r, world)) for r in range(world)])
        waves += 1
        extra += len(missing)
    records, missing = _walk(n, stamps, altitudes, lambda j, k: table.get((j, k)), lambda j, k: eng.cache.get((j, k)), ctx, K,
                             orientation0, location0, Rbc, False, None)
    assert not missing
    if engine is None:
        eng.close()
    if stats is not None:
        stats.update({"waves": waves, "gap_pairs": extra, "block": (f0, f1)})
    return records
