"""cProfile of the single-rank C3 replay (BASELINE configs[2]): where the host time goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import synth, replay
from visual_odometry_amd.frontend import Context
from visual_odometry_amd._abi import default_params

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    c3 = synth.sequence_c3(n, seed=3003, step_m=0.5, depth=6.0, relief=1.8, workers=8)
    prm = default_params()
    ctx = Context(prm)
    stamps = np.arange(n) * 0.1
    alt = np.full(n, 6.0)
    for chunk in (64, 128, 256):
        replay.replay_sharded(ctx, c3.frames[: 2 * chunk], stamps[: 2 * chunk], alt[: 2 * chunk], c3.K, np.eye(3), np.zeros(3), 0, 1, chunk=chunk)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            replay.replay_sharded(ctx, c3.frames, stamps, alt, c3.K, np.eye(3), np.zeros(3), 0, 1, chunk=chunk)
            ts.append(time.perf_counter() - t0)
        print("chunk %d: replay %.1f ms (best of 3)" % (chunk, 1e3 * min(ts)))
    pr = cProfile.Profile()
    pr.enable()
    replay.replay_sharded(ctx, c3.frames, stamps, alt, c3.K, np.eye(3), np.zeros(3), 0, 1, chunk=128)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
