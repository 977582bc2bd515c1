"""How well-conditioned is the five-point problem?  The oracle against itself on inputs moved by ONE ulp: the yardstick the
GPU parity tests measure the device solver with (tests/test_gpu_geometry.py).  CPU only."""
import os

import numpy as np

import oracle_lib as O
from visual_odometry_amd import synth

K = synth.K_1080P
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _norm(p):
    p = p.astype(np.float64)
    return np.stack([(p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1]], 1)


def _set_dist(Ea, na, Eb, nb):
    if na != nb:
        return np.inf
    d = 0.0
    for i in range(na):
        d = max(d, min(min(np.abs(Ea[i] - Eb[j]).max(), np.abs(Ea[i] + Eb[j]).max()) for j in range(nb)))
    return d


def test_five_point_candidates_move_under_one_ulp():
    """A tenth-degree polynomial system in fp64: 6-7 % of random hypotheses move by more than 1e-9 when the inputs move by one
    ulp, 1-2 % by more than 1e-7.  Any two correct fp64 solvers - OpenCV's, this oracle, the device kernel - differ by that
    much on those hypotheses; set equality within 1e-9 can only hold on the well-conditioned ones."""
    p1, p2, R, t, _ = synth.correspondences(5 * 2000, K, 11, inlier_frac=1.0, noise_px=0.3)
    x1, x2 = _norm(p1).reshape(-1, 5, 2), _norm(p2).reshape(-1, 5, 2)
    E0, n0 = O.five_point(x1, x2)
    rng = np.random.default_rng(1)
    pert = lambda x: (x.view(np.int64) + rng.integers(-1, 2, x.shape)).view(np.float64)
    E1, n1 = O.five_point(pert(x1.copy()), pert(x2.copy()))
    d = np.array([_set_dist(E0[h], n0[h], E1[h], n1[h]) for h in range(len(n0))])
    assert np.all(np.isfinite(d))  # the number of real roots did not change anywhere
    assert 0.90 < np.mean(d <= 1e-9) < 0.97
    assert 0.975 < np.mean(d <= 1e-7) < 0.995
    assert 0.5 < np.mean(d <= 1e-12) < 0.75  # and more than half of the hypotheses are well-conditioned


def test_golden_ransac_result_is_stable_under_one_ulp():
    """The committed RANSAC fixture does not move when K moves by one ulp: the GPU test may demand it exactly."""
    g = np.load(os.path.join(GOLD, "c1_small.npz"))
    rng = np.random.default_rng(7)
    for _ in range(6):
        Kp = g["K"].copy()
        for r, c in ((0, 0), (1, 1), (0, 2), (1, 2)):
            Kp[r, c] = (Kp[r : r + 1, c].view(np.int64) + rng.integers(-1, 2)).view(np.float64)[0]
        st, E, mask, ninl, iters = O.essential_ransac(g["p1"], g["p2"], Kp)
        assert st == 0 and iters == int(g["iters"]) and np.array_equal(mask, g["mask"])
        assert np.abs(E - g["E"]).max() < 1e-9
