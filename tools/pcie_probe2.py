#!/usr/bin/env python3
"""Variants of the pipelined step, to find what separates the PCIe-inclusive rate from the resident one."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from visual_odometry_amd import synth
from visual_odometry_amd._abi import default_params
from visual_odometry_amd.frontend import Batch, Context


def main():
    H, W, P = 1080, 1920, 64
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    ctx = Context(prm, device=0)
    scenes = [synth.scene_c2(2002 + i) for i in range(8)]
    batch = Batch(ctx, P, H, W)
    out = {}

    def fill(k, pairs=P):
        for i in range(pairs):
            sc = scenes[(i + k) % len(scenes)]
            batch.set_pair(i, sc.img1, sc.img2, sc.K)

    def loop(n, mode):
        batch.sync()
        t0 = time.perf_counter()
        for k in range(n):
            if mode == "rotate":
                fill(k)
            elif mode == "same":
                fill(0)
            elif mode == "one":  # a single pair restaged: slots flip, 126 frames carried over device to device
                fill(k, 1)
            batch.run()
            if mode != "resident_nogets":
                _ = [batch.get(i, age=1) for i in range(P)]
        batch.sync()
        return 1e3 * (time.perf_counter() - t0) / n

    fill(0)
    batch.run()
    batch.run()
    batch.sync()
    for rnd in range(3):
        for mode in ("resident", "same", "rotate", "one"):
            out.setdefault(mode, []).append(round(loop(10, mode), 3))
    # host timeline of the rotating loop: where does the host wait?
    tl = []
    batch.sync()
    for k in range(8):
        t0 = time.perf_counter()
        fill(k)
        t1 = time.perf_counter()
        batch.run()
        t2 = time.perf_counter()
        _ = [batch.get(i, age=1) for i in range(P)]
        t3 = time.perf_counter()
        tl.append([round(1e3 * (t1 - t0), 2), round(1e3 * (t2 - t1), 2), round(1e3 * (t3 - t2), 2)])
    out["timeline_fill_run_get_ms"] = tl
    # host cost of the pieces
    t0 = time.perf_counter()
    fill(3)
    out["fill_host_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
    t0 = time.perf_counter()
    batch.run()
    out["run_enqueue_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
    batch.sync()
    print(json.dumps(out))
    batch.close()
    ctx.close()


if __name__ == "__main__":
    main()
