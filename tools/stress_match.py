#!/usr/bin/env python3
"""Stress: GPU opt_DetectFeatures vs the CPU oracle on many random scenes (matcher certificate + exact-scan fallback),
plus raw 2-NN on random descriptor sets of awkward sizes.  python tools/stress_match.py [n_scenes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import oracle_lib as O
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = Context()
rng = np.random.default_rng(12345)
bad = 0
redo_total = 0
t0 = time.time()
for i in range(n):
    nb = int(rng.integers(100, 1500))
    # every third scene at a random frame size (odd widths take the byte-wise footprint / column-sum kernels)
    hh, ww = (480, 640) if i % 3 else (int(rng.integers(120, 520)), int(rng.integers(160, 720)))
    sc = synth.blob_scene(hh, ww, synth.K_PARETE, nb, 5000 + i)
    g = ctx.opt_DetectFeatures(sc.img1, sc.img2)
    r = C.c_int32(0); ctx.lib.vo_debug_fetch(ctx._h, 4, C.byref(r), C.c_size_t(4)); redo_total += r.value
    p1, p2, m = O.detect_match(sc.img1, sc.img2)
    ok = len(m) == len(g.match) and np.array_equal(g.Kpoints1, p1) and np.array_equal(g.Kpoints2, p2) and \
        np.array_equal(g.match["distance"].view(np.uint32), m["distance"].view(np.uint32)) and np.array_equal(g.match["trainIdx"], m["trainIdx"])
    bad += not ok
    if not ok: print("MISMATCH scene", i, nb, len(m), len(g.match))
for i in range(n):
    na, nb = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    a = synth.surf_like_descriptors(na, 100 + i); b = synth.surf_like_descriptors(nb, 900 + i)
    k = min(na, nb) // 4
    if k: b[rng.choice(nb, k, replace=False)] = a[rng.choice(na, k, replace=False)] + rng.normal(0, 10.0 ** rng.uniform(-7, -2), (k, 64)).astype(np.float32)
    idx, dist = ctx.match_knn2(a, b)
    r = C.c_int32(0); ctx.lib.vo_debug_fetch(ctx._h, 4, C.byref(r), C.c_size_t(4)); redo_total += r.value
    ridx, rdist = O.match_knn2(a, b)
    ok = np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    bad += not ok
    if not ok: print("MISMATCH knn", i, na, nb, int((idx != ridx).sum()))
# the batch path (>= 8 pairs: unsplit matcher, one pair per XCD) at the benchmark size
if len(sys.argv) > 2:
    from visual_odometry_amd.frontend import Batch
    from visual_odometry_amd._abi import default_params
    H, W = 1080, 1920
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    c2 = Context(prm)
    scenes = [synth.scene_c2(7000 + i) for i in range(8)]
    b = Batch(c2, 8, H, W)
    for i, sc in enumerate(scenes): b.set_pair(i, sc.img1, sc.img2, sc.K)
    b.run(); b.sync()
    r = C.c_int32(0); c2.lib.vo_debug_fetch(c2._h, 4, C.byref(r), C.c_size_t(4)); redo_total += r.value
    for i, sc in enumerate(scenes):
        p1, p2, m, mask, X = b.get_points(i)
        rp1, rp2, rm = O.detect_match(sc.img1, sc.img2, prm)
        ok = len(rm) == len(m) and np.array_equal(p1, rp1) and np.array_equal(p2, rp2) and np.array_equal(m["distance"].view(np.uint32), rm["distance"].view(np.uint32))
        bad += not ok
        print("1080p pair", i, len(rm), "matches", "ok" if ok else "MISMATCH", flush=True)
print(f"stress: {2 * n} cases, {bad} mismatches, {redo_total} exact scans, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
