#!/usr/bin/env python3
"""Offline stand-in for `roslaunch visual_odometry rapallo.launch` (rosbag play + the node): reads the camera and altitude
topics of a bag (or an image directory), runs get_image + the front end + the node's gates and pose chain on the GPU and
prints one JSON line per processed frame with the fields of vo_results.msg / fail_check.msg.

    python tools/replay_bag.py Rapallo.bag --camera config/rapallo_intrinsic.yaml [--max-frames N] [--motion2d]
    python tools/replay_bag.py frames_dir/ --camera config/camera_intrinsic.yaml --altitude 7.0
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--camera", required=True, help="the node's intrinsic YAML (config/*.yaml)")
    ap.add_argument("--image-topic", default="/pylon_camera/image_raw/compressed")
    ap.add_argument("--altitude-topic", default="/drivers/altitude")
    ap.add_argument("--altitude", type=float, default=None, help="constant altitude [m] (image directories have none)")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--motion2d", action="store_true")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--decode-workers", type=int, default=min(16, len(os.sched_getaffinity(0))), help="host threads decoding JPEG / PNG payloads")
    a = ap.parse_args()

    from visual_odometry_amd import bagio, chain, replay
    from visual_odometry_amd import prep as P
    from visual_odometry_amd.config import load_camera_config
    from visual_odometry_amd.frontend import Context

    cam = load_camera_config(a.camera)
    if os.path.isdir(a.source):
        frames, stamps, _ = bagio.read_image_dir(a.source)
        frames, stamps = frames[:a.max_frames], stamps[:a.max_frames]
        alts = np.full(len(frames), a.altitude if a.altitude is not None else 1.0)
    else:
        frames, stamps, alts = bagio.read_sequence(a.source, a.image_topic, a.altitude_topic, max_frames=a.max_frames,
                                                     relative_stamps=True, decode_workers=a.decode_workers)  # exact dt from the integer (sec, nsec) stamps
        if a.altitude is not None:
            alts = np.full(len(frames), a.altitude)
    if len(frames) < 2:
        sys.exit("fewer than two frames")
    rows, cols = frames[0].shape[:2]
    dw, dh = cam.desired_width or cols, cam.desired_height or rows
    ctx = Context()
    K = P.updateCameraMatrix((rows, cols), dw, dh, cam.K(), ctx.lib)
    pr = P.Prep(ctx, (rows, cols), dw, dh, K, cam.dist())
    recs = replay.replay(ctx, frames, stamps, alts, K, np.eye(3), np.zeros(3), motion2D=a.motion2d, chunk=a.chunk, prep=pr)
    names = {chain.FAIL_DETECTION: "fail_detect", chain.NOT_MOVING: "not_moving", chain.FAIL_RECOVER: "fail_pose", chain.SUCCESS: "success"}
    for r in recs:
        line = {"frame": r["frame"], "stamp": float(stamps[r["frame"]]), "fail_check": names[r["fail"]], "n_matches": r["n_matches"]}
        if "estimate_pos" in r:
            line.update({"estimate_pos": r["estimate_pos"].tolist(), "estimate_rpy": np.asarray(r["estimate_rpy"]).tolist(),
                         "estimate_twist": {"linear": np.asarray(r["twist_linear"]).tolist(), "angular": np.asarray(r["twist_angular"]).tolist()},
                         "SF": float(r["SF"]), "n_inliers": int(r["n_inliers"])})
        print(json.dumps(line))
    pr.close()
    ctx.close()


if __name__ == "__main__":
    main()
