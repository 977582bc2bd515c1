#!/usr/bin/env python3
"""Where does the PCIe-inclusive rate lose against the resident one?  Times, on one GPU:
  1. the host-side staging copy of one step's frames (pageable -> pinned), 1 and 4 threads;
  2. the H2D transfer of one step's frames alone (pinned, a stream of its own);
  3. a resident step alone, and the same step with that H2D transfer running beside it on another stream.
Prints one JSON line."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from visual_odometry_amd import synth
from visual_odometry_amd._abi import default_params
from visual_odometry_amd.frontend import Batch, Context


def main():
    H, W, P = 1080, 1920, 64
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    ctx = Context(prm, device=0)
    scenes = [synth.scene_c2(2002 + i) for i in range(4)]
    batch = Batch(ctx, P, H, W)
    for i in range(P):
        sc = scenes[i % 4]
        batch.set_pair(i, sc.img1, sc.img2, sc.K)
    batch.sync()
    for _ in range(3):
        batch.run()
    batch.sync()
    out = {}

    # 1. staging copy: 128 distinct pageable frames -> one pinned buffer
    rng = np.random.default_rng(0)
    src = [rng.integers(0, 255, (H, W), dtype=np.uint8) for _ in range(2 * P)]
    pinned = torch.empty((2 * P, H, W), dtype=torch.uint8).pin_memory()
    pin_np = pinned.numpy()
    for rep in range(2):
        t0 = time.perf_counter()
        for f in range(2 * P):
            np.copyto(pin_np[f], src[f])
        out["stage_1thr_ms"] = round(1e3 * (time.perf_counter() - t0), 2)

    def work(lo, hi):
        for f in range(lo, hi):
            np.copyto(pin_np[f], src[f])

    for nt in (4, 8):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k * 2 * P // nt, (k + 1) * 2 * P // nt)) for k in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
        out[f"stage_{nt}thr_ms"] = round(1e3 * (time.perf_counter() - t0), 2)

    # 2. H2D alone
    dev = torch.empty((2 * P, H, W), dtype=torch.uint8, device="cuda")
    cs = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        with torch.cuda.stream(cs):
            e0.record()
            dev.copy_(pinned, non_blocking=True)
            e1.record()
        cs.synchronize()
        out["h2d_alone_ms"] = round(e0.elapsed_time(e1), 3)
    out["h2d_GBs"] = round(pinned.numel() / out["h2d_alone_ms"] / 1e6, 2)
    # per-frame copies (what vo_batch_set_pair issues): 128 x 2 MB
    with torch.cuda.stream(cs):
        e0.record()
        for f in range(2 * P):
            dev[f].copy_(pinned[f], non_blocking=True)
        e1.record()
    cs.synchronize()
    out["h2d_128x2MB_ms"] = round(e0.elapsed_time(e1), 3)

    # 3. resident step alone vs with the H2D beside it
    def steps(n, with_copy):
        batch.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if with_copy:
                with torch.cuda.stream(cs):
                    dev.copy_(pinned, non_blocking=True)
            batch.run()
        batch.sync()
        cs.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out["step_alone_ms"] = round(steps(8, False), 3)
    out["step_with_h2d_beside_ms"] = round(steps(8, True), 3)
    out["step_alone_again_ms"] = round(steps(8, False), 3)

    # 4. the library's own upload path: set_pair for the whole batch, then sync (no run)
    t0 = time.perf_counter()
    for i in range(P):
        batch.set_pair(i, src[2 * i], src[2 * i + 1], scenes[0].K)
    t1 = time.perf_counter()
    batch.sync()
    t2 = time.perf_counter()
    out["set_pair_x64_host_ms"] = round(1e3 * (t1 - t0), 2)
    out["set_pair_x64_drain_ms"] = round(1e3 * (t2 - t1), 2)
    # get() of all records
    batch.run()
    batch.sync()
    t0 = time.perf_counter()
    _ = [batch.get(i) for i in range(P)]
    out["get_x64_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    print(json.dumps(out))
    batch.close()
    ctx.close()


if __name__ == "__main__":
    main()
