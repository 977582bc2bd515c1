"""Determinism soak at the headline size: the same 16 1080p pairs (8 k keypoints per frame) through the pipelined batch N times,
with restaging every step (uploads beside kernels, integral pass of step k + 1 under the descriptors of step k); every step's
records must equal the first step's.  Then the same for a 1080p sequence batch.  Usage: python tools/soak_1080p.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context, Batch, Sequence


def rec(o):
    return (o.n_kp1, o.n_kp2, o.n_match, o.status, o.n_inl, o.n_good, o.n_world, o.success, o.iters, tuple(o.E), tuple(o.R), tuple(o.t))


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    scs = [synth.scene_c2(2002 + i) for i in range(4)]
    P = 16
    ctx = Context()
    b = Batch(ctx, P, 1080, 1920)

    def fill():
        for i in range(P):
            b.set_pair(i, scs[i % 4].img1, scs[i % 4].img2, scs[i % 4].K)

    fill()
    b.run()
    want = [rec(o) for o in b.get_records()]
    bad = 0
    fill()
    for k in range(steps):
        b.run()
        fill()
        got = [rec(o) for o in b.get_records(age=1)] if k else want
        bad += got != want
        if k % 50 == 0:
            print("pair batch step", k, "bad", bad, flush=True)
    bad += [rec(o) for o in b.get_records()] != want
    b.close()
    seqs = [Sequence(ctx, 9, 1080, 1920) for _ in range(2)]
    frames = [scs[i % 4].img1 if i % 2 == 0 else scs[i % 4].img2 for i in range(9)]
    wants = None
    for k in range(steps // 3):
        s = seqs[k & 1]
        for f in range(9):
            s.set_frame(f, frames[f])
        for p in range(8):
            s.set_pair(p, p, p + 1, scs[0].K)
        s.run()
        if k:
            got = [rec(o) for o in seqs[(k - 1) & 1].get_records()]
            if wants is None:
                wants = got
            bad += got != wants
    print("soak 1080p:", steps, "steps,", bad, "mismatching steps; matches", want[0][2], "inliers", want[0][4])
    sys.exit(1 if bad else 0)
