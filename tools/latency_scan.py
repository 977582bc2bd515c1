#!/usr/bin/env python3
"""Stage times of one detect+describe call against the number of keypoints in the frame (fixed cost of the launches)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context
ctx = Context()
for nb in (0, 3, 30, 300, 1500):
    sc = synth.blob_scene(505, 640, synth.K_PARETE, max(nb, 1), 77)
    img = sc.img1 if nb else np.full((505, 640), 128, np.uint8)
    for _ in range(2): k, d = ctx.detect_describe(img)
    ctx.profile_enable(True)
    for _ in range(5): k, d = ctx.detect_describe(img)
    prof = ctx.profile(); ctx.profile_enable(False)
    win = (21 * (k["size"].astype(np.float32) * np.float32(1.2) / np.float32(9.0))).astype(np.int64) if len(k) else np.zeros(0, np.int64)
    cls = [int(((win > lo) & (win <= hi)).sum()) for lo, hi in ((0, 64), (64, 128), (128, 320), (320, 768))]
    print("blobs %5d keypoints %5d classes %s  " % (nb, len(k), cls) + " ".join("%s %.3f" % (s.replace("describe_", ""), v[0] / 5) for s, v in prof.items() if s.startswith("describe") or s in ("orientation", "hessian_oct0_23")), flush=True)
