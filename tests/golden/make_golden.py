#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/libvo_oracle.so) on small synthetic inputs.

These are regression fixtures of THIS repository's oracle: the reference ships no fixtures and OpenCV is not
available, so they do not pin the oracle against the reference (see oracle/vo_oracle.h, "parity unpinned").
They do pin the GPU path and later oracle edits against a committed set of inputs and outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402


def main():
    sc = synth.blob_scene(240, 320, synth.K_PARETE, 160, 424242)
    img1, img2 = sc.img1, sc.img2
    k, d = O.detect_describe(img1)
    p1, p2, m = O.detect_match(img1, img2)
    st, E, mask, ninl, iters = O.essential_ransac(p1, p2, sc.K)
    assert st == 0
    R, t, good = O.recover_pose(E, p1[mask > 0], p2[mask > 0], sc.K)
    Xk = O.triang_points(p1[mask > 0], p2[mask > 0], R, t, sc.K)
    np.savez_compressed(
        os.path.join(HERE, "c1_small.npz"), img1=img1, img2=img2, K=sc.K, kp_x=k["x"], kp_y=k["y"], kp_size=k["size"],
        kp_angle=k["angle"], kp_response=k["response"], kp_octave=k["octave"], kp_class=k["class_id"], desc=d,
        p1=p1, p2=p2, m_query=m["queryIdx"], m_train=m["trainIdx"], m_dist=m["distance"], E=E, mask=mask, iters=iters,
        R=R, t=t, good=good, Xk=Xk, tuples_1000=O.ransac_subsets(2**64 - 1, 1000, 16))
    print("c1_small:", len(k), "kp", len(m), "matches", ninl, "inliers", Xk.shape[1], "world points")


if __name__ == "__main__":
    main()
