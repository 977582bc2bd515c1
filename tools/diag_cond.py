#!/usr/bin/env python3
"""GPU-vs-oracle five-point candidate distances against the oracle's own sensitivity to 1-ulp input perturbations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as O
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Context

K = synth.K_1080P


def dist(Ea, na, Eb, nb):
    if na != nb:
        return np.inf
    d = 0.0
    for i in range(na):
        d = max(d, min(min(np.abs(Ea[i] - Eb[j]).max(), np.abs(Ea[i] + Eb[j]).max()) for j in range(nb)))
    return d


def main():
    ctx = Context()
    p1, p2, R, t, _ = synth.correspondences(5 * 2000, K, 11, inlier_frac=1.0, noise_px=0.3)
    nrm = lambda p: np.stack([(p[:, 0].astype(np.float64) - K[0, 2]) / K[0, 0], (p[:, 1].astype(np.float64) - K[1, 2]) / K[1, 1]], 1)
    x1, x2 = nrm(p1).reshape(-1, 5, 2), nrm(p2).reshape(-1, 5, 2)
    E0, n0 = O.five_point(x1, x2)
    G, gn = ctx.five_point(x1, x2)
    rng = np.random.default_rng(1)
    H = len(n0)
    cond = np.zeros(H)
    single = None
    for r in range(3):
        pert = lambda x: (x.view(np.int64) + rng.integers(-1, 2, x.shape)).view(np.float64)
        E1, n1 = O.five_point(pert(x1.copy()), pert(x2.copy()))
        d = np.array([dist(E0[h], n0[h], E1[h], n1[h]) for h in range(H)])
        if single is None:
            single = d
        cond = np.maximum(cond, d)
    g = np.array([dist(G[h], gn[h], E0[h], n0[h]) for h in range(H)])
    for tol in (1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
        print(f"tol {tol:g}: gpu-vs-oracle {np.mean(g <= tol):.4f}   oracle-vs-1ulp {np.mean(single <= tol):.4f}   (max of 3: {np.mean(cond <= tol):.4f})")
    print("count mismatches gpu", int(np.sum(np.isinf(g))), "oracle-pert", int(np.sum(np.isinf(cond))))
    for c in (1e-13, 1e-12, 1e-11):
        well = cond <= c
        print(f"well-conditioned (cond <= {c:g}): {well.mean():.3f} of hypotheses; worst gpu distance among them {g[well].max():.3g}; ratio g/cond max {np.max(g[well] / np.maximum(cond[well], 1e-16)):.3g}")
    ratio = g / np.maximum(cond, 1e-16)
    print("quantiles of gpu/cond ratio", np.quantile(ratio[np.isfinite(ratio)], [0.5, 0.9, 0.99, 0.999, 1.0]))
    ctx.close()


if __name__ == "__main__":
    main()
