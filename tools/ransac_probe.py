"""Phase timing of k_ransac_hyp (needs a library built with -DVO_RANSAC_TIMING): C5 stress and a headline-like hard case."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context

ctx = Context(lib_path=os.environ.get("VO_LIB"))
lib = ctx.lib
names = ["nullspace", "polys", "gauss-jordan", "detB", "durand-kerner", "backsub", "scoring"]

def report(tag, fn, reps=3):
    fn()
    buf = (C.c_ulonglong * 16)()
    timing = hasattr(lib, "vo_debug_ransac_timing")
    if timing:
        lib.vo_debug_ransac_timing(None, 1)
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    wall = (time.perf_counter() - t0) / reps
    ms = ctx.profile()["ransac_solve_score"][0] / reps
    ctx.profile_enable(False)
    if timing:
        lib.vo_debug_ransac_timing(buf, 0)
    v = list(buf)
    nw = max(v[15], 1)
    print(f"== {tag}: solve+score {ms:.3f} ms device, call {1e3*wall:.3f} ms, wavefronts {nw//reps}, DK iterations/wave {v[14]/nw:.1f}, iters {out[4]}")
    tot = sum(v[:7])
    for n, c in zip(names, v[:7] if timing else []):
        print(f"   {n:14s} {c/nw/100.0:9.2f} us/wave  {100.0*c/max(tot,1):5.1f} %")   # wall_clock64 ticks at 100 MHz

K = synth.K_1080P
p1, p2, R, t, is_in = synth.correspondences(8192, K, 5005, inlier_frac=0.5, noise_px=0.5)
report("C5 8192 matches x 4096 hyp", lambda: ctx.essential_ransac(p1, p2, K, prob=1.0, max_iters=4096))
q1, q2, R, t, is_in = synth.correspondences(1200, K, 77, inlier_frac=0.95, noise_px=0.5)
report("1200 matches, prob 1.0, <= 1000 hyp", lambda: ctx.essential_ransac(q1, q2, K, prob=1.0, max_iters=1000))
