"""Stage timers of a 64-pair step at 640x480 (BASELINE configs[0]) - where a small-frame step spends its device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context, Batch

if __name__ == "__main__":
    H, W, P = 480, 640, 64
    K = synth.K_PARETE
    scenes = [synth.blob_scene(H, W, K, 700, 1001 + s) for s in range(8)]
    ctx = Context()
    b = Batch(ctx, P, H, W)
    for i in range(P):
        b.set_pair(i, scenes[i % 8].img1, scenes[i % 8].img2, K)
    for _ in range(3):
        b.run()
    b.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        b.run()
    b.sync()
    print("resident step: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(5):
        b.run()
    b.sync()
    pf = ctx.profile()
    ctx.profile_enable(False)
    tot = 0.0
    for k, (ms, n) in pf.items():
        if n:
            print("%-20s %.4f ms" % (k, ms / 5))
            tot += ms / 5
    print("sum %.3f ms" % tot)
    o = b.get(0)
    print("kp", o.n_kp1, o.n_kp2, "matches", o.n_match)
