#!/usr/bin/env python3
"""N2 measurement: get_image (vo_prep_*) at the rapallo size, 2560 x 2056 BGR -> 640 x 505 gray.

Prints one JSON line: device time per stage (HIP events on the context's stream), frames/s with the source frame
uploaded from pinned staging on every call (the only way the entry point is used), the algorithmic HBM bytes per frame
and the CPU restatement's rate on one core beside it.  Run on the GPU box: python tools/prep_bench.py [--frames N]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--cpu-frames", type=int, default=5)
    a = ap.parse_args()
    from visual_odometry_amd import prep as P
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import Context, Sequence

    ctx = Context(device=0)
    src, dst = (2056, 2560), (505, 640)
    K = P.updateCameraMatrix(src, dst[1], dst[0], synth.K_RAPALLO_FULL, ctx.lib)
    pr = P.Prep(ctx, src, dst[1], dst[0], K, synth.DIST_RAPALLO)
    frames = [synth.bgr_frame(src[0], src[1], 100 + i) for i in range(4)]
    seq = Sequence(ctx, 8, dst[0], dst[1])
    for i in range(8):  # warm-up
        pr.to_batch(frames[i % 4], seq, i)
    ctx.lib.vo_batch_sync(seq._h)
    t0 = time.perf_counter()
    for i in range(a.frames):
        pr.to_batch(frames[i % 4], seq, i % 8)
    ctx.lib.vo_batch_sync(seq._h)
    dt = time.perf_counter() - t0
    ctx.profile_enable(True)
    # device time per stage with the frame resident and the GPU kept busy (an idle GPU between uploads drops its clocks)
    ctx._chk(ctx.lib.vo_prep_rerun(pr._h, 50))
    ctx.lib.vo_batch_sync(seq._h)
    ctx.profile_reset()
    ctx._chk(ctx.lib.vo_prep_rerun(pr._h, 500))
    ctx.lib.vo_batch_sync(seq._h)
    prof = {k: v for k, v in ctx.profile().items() if k.startswith("prep_")}
    stage_us = {k: round(1e3 * ms / max(n, 1), 2) for k, (ms, n) in prof.items()}
    src_bytes, px = src[0] * src[1] * 3, dst[0] * dst[1]
    # algorithmic traffic per frame: resize reads the source and writes 3 B/px; remap reads 3 B/px + 6 B/px of maps, writes 3 + 1;
    # CLAHE reads the gray frame twice (histograms, blending) and writes it once
    alg = {"prep_resize": src_bytes + 3 * px, "prep_remap_gray": 3 * px + 6 * px + 4 * px, "prep_clahe": 3 * px}
    out = {
        "metric": "preprocessed_frames_per_sec", "workload": "rapallo 2560x2056 BGR -> 640x505 gray (get_image)",
        "value": round(a.frames / dt, 1), "frames": a.frames, "includes": "host memcpy to pinned staging + PCIe upload of every frame",
        "stage_us": stage_us,
        "stage_GBps": {k: round(alg[k] / (stage_us[k] * 1e-6) / 1e9, 1) for k in alg if stage_us.get(k)},
        "device_us_per_frame": round(sum(stage_us.values()), 2),
    }
    import oracle_lib as O

    O.set_threads(1)
    t0 = time.perf_counter()
    for i in range(a.cpu_frames):
        O.prep(pr.params, frames[i % 4])
    out["cpu_port_frames_per_sec_1_thread"] = round(a.cpu_frames / (time.perf_counter() - t0), 2)
    print(json.dumps(out))
    pr.close()
    seq.close()
    ctx.close()


if __name__ == "__main__":
    main()
