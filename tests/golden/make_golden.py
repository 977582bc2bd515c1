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

    # N2: get_image on a small colour frame (fractional resize 130 x 173 -> 61 x 97, rapallo-like distortion, CLAHE 16 / 8 x 8)
    from visual_odometry_amd._abi import VoPrepParams

    bgr = synth.bgr_frame(130, 173, 515151)
    p = VoPrepParams()
    p.src_rows, p.src_cols, p.dst_rows, p.dst_cols = 130, 173, 61, 97
    Kp = np.array([[150.0, 0.0, 48.3], [0.0, 149.0, 30.6], [0.0, 0.0, 1.0]])
    p.K[:] = [float(v) for v in Kp.ravel()]
    p.dist[:] = [float(v) for v in synth.DIST_RAPALLO]
    p.clip_limit, p.tiles_x, p.tiles_y = 16.0, 8, 8
    o = O.prep(p, bgr)
    np.savez_compressed(os.path.join(HERE, "prep_small.npz"), bgr=bgr, K=Kp, dist=synth.DIST_RAPALLO, dst_shape=np.array([61, 97]),
                        resized=o["resized"], undist=o["undist"], gray0=o["gray0"], lut=o["lut"], map1=o["map1"], map2=o["map2"],
                        gray=o["gray"])
    print("prep_small:", o["gray"].shape, "mean", float(o["gray"].mean()))


if __name__ == "__main__":
    main()
