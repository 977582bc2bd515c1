"""Stage timers of one 128-frame sequence chunk of the C3 run (640x505): device time per stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context, Sequence
from visual_odometry_amd._abi import default_params

if __name__ == "__main__":
    n = 128
    c3 = synth.sequence_c3(n, seed=3003, workers=8)
    ctx = Context(default_params())
    seq = Sequence(ctx, n, 505, 640)
    for f in range(n):
        seq.set_frame(f, c3.frames[f])
    for p in range(n - 1):
        seq.set_pair(p, p, p + 1, c3.K)
    for _ in range(3):
        seq.run()
    seq.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        seq.run()
    seq.sync()
    print("resident chunk of %d frames: %.3f ms" % (n, (time.perf_counter() - t0) / 10 * 1e3))
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(5):
        seq.run()
    seq.sync()
    pf = ctx.profile()
    ctx.profile_enable(False)
    tot = 0.0
    for k, (ms, cnt) in pf.items():
        if cnt:
            print("%-20s %.4f ms" % (k, ms / 5))
            tot += ms / 5
    print("sum %.3f ms" % tot)
    outs, _ = seq.get_all()
    print("kp", np.mean([o.n_kp1 for o in outs]), "matches", np.mean([o.n_match for o in outs]), "iters", np.mean([o.iters for o in outs]))
