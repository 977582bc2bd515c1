#!/usr/bin/env python3
"""Latency of the three reference calls on ONE frame pair through the synchronous C-ABI entry points (host buffers in and out),
as the ROS node would call them: opt_DetectFeatures -> optRelativePose -> triangPoints.  python tools/latency_single.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import synth
from visual_odometry_amd._abi import default_params
from visual_odometry_amd.frontend import Context

def run(name, sc, prm, reps=20):
    ctx = Context(prm)
    t = {"detect_match": [], "relative_pose": [], "triang": [], "total": []}
    for i in range(reps + 3):
        t0 = time.perf_counter()
        g = ctx.opt_DetectFeatures(sc.img1, sc.img2)
        t1 = time.perf_counter()
        st, R, tv, inl, ok = ctx.optRelativePose(g, sc.K)
        t2 = time.perf_counter()
        pts = ctx.triangPoints(inl.Kpoints1, inl.Kpoints2, R, tv, sc.K) if ok else None
        t3 = time.perf_counter()
        if i >= 3:
            t["detect_match"].append(t1 - t0); t["relative_pose"].append(t2 - t1); t["triang"].append(t3 - t2); t["total"].append(t3 - t0)
    print(name, "matches", len(g.match), " ".join("%s %.2f ms" % (k, 1e3 * float(np.median(v))) for k, v in t.items()), flush=True)
    ctx.close()

prm = default_params()
run("640x505 ", synth.blob_scene(505, 640, synth.K_PARETE, 700, 77), prm)
p2 = default_params()
p2.roi_x_lo, p2.roi_y_lo, p2.roi_x_hi, p2.roi_y_hi = synth.roi_for(1080, 1920)
run("1920x1080", synth.scene_c2(2002), p2)
