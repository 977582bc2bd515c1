#!/usr/bin/env python3
"""Per-stage device time of ONE synchronous opt_DetectFeatures call (library stage timers) next to its wall time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import synth
from visual_odometry_amd._abi import default_params
from visual_odometry_amd.frontend import Context
prm = default_params()
if len(sys.argv) > 1:  # any argument: the 1080p benchmark pair instead of the reference's 640x505
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(1080, 1920)
    sc = synth.scene_c2(2002)
else:
    sc = synth.blob_scene(505, 640, synth.K_PARETE, 700, 77)
ctx = Context(prm)
for _ in range(3): ctx.opt_DetectFeatures(sc.img1, sc.img2)
ctx.profile_enable(True)
n = 10
t0 = time.perf_counter()
for _ in range(n): g = ctx.opt_DetectFeatures(sc.img1, sc.img2)
wall = (time.perf_counter() - t0) / n
prof = ctx.profile()
ctx.profile_enable(False)
tot = 0
for k, v in prof.items():
    if v[0] > 0: print("%-22s %8.3f ms" % (k, v[0] / n)); tot += v[0] / n
print("sum of stages %.3f ms, wall per call %.3f ms (profiling on)" % (tot, wall * 1e3))
t0 = time.perf_counter()
for _ in range(n): g = ctx.opt_DetectFeatures(sc.img1, sc.img2)
print("wall per call %.3f ms (profiling off), %d matches" % ((time.perf_counter() - t0) / n * 1e3, len(g.match)))
