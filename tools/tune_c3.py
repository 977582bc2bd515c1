"""Scene tuning for the C3 stand-in: outcome counts of the replayed loop for a few (step, depth, relief) choices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if __name__ == "__main__":
    from visual_odometry_amd import synth
    combos = [(0.36, 7.0, 1.2), (0.5, 7.0, 1.2), (0.5, 6.0, 1.8), (0.6, 6.0, 2.0), (0.45, 5.0, 1.5)]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scenes = [synth.sequence_c3(n, seed=3003, step_m=s, depth=d, relief=r, workers=16) for s, d, r in combos]
    from visual_odometry_amd import replay
    from visual_odometry_amd.frontend import Context
    ctx = Context()
    for (s, d, r), sc in zip(combos, scenes):
        stamps, alt = 0.1 * np.arange(n), np.full(n, d)
        recs = replay.replay_sharded(ctx, sc.frames, stamps, alt, sc.K, np.eye(3), np.zeros(3), 0, 1)
        fails = [x["fail"] for x in recs]
        last = [x for x in recs if "estimate_pos" in x][-1]
        c_true = -sc.Rw[last["frame"]].T @ sc.tw[last["frame"]]
        err = float(np.linalg.norm(last["estimate_pos"] - c_true))
        path = float(sum(sc.relative(k - 1, k)[2] for k in range(1, n)))
        print(f"step {s} depth {d} relief {r}: success {fails.count(3)} not_moving {fails.count(1)} fail_recover {fails.count(2)} fail_detect {fails.count(0)} "
              f"end error {err:.2f} m over {path:.1f} m, mean matches {np.mean([x['n_matches'] for x in recs]):.0f}", flush=True)
