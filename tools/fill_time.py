"""Host cost of staging one 64-pair 1080p step (vo_batch_set_pair x 64: memcpy into pinned memory) against the device step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd.frontend import Context, Batch

if __name__ == "__main__":
    H, W, P = 1080, 1920, 64
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 255, (H, W), dtype=np.uint8) for _ in range(128)]
    K = np.array([[1400.0, 0, 959.5], [0, 1400.0, 539.5], [0, 0, 1]])
    ctx = Context()
    b = Batch(ctx, P, H, W)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        for p in range(P):
            b.set_pair(p, frames[2 * p], frames[2 * p + 1], K)
        ts.append(time.perf_counter() - t0)
        b.run()
        b.sync()
    print("fill of 64 pairs (265 MB): %.2f ms median, %.1f GB/s" % (1e3 * np.median(ts[1:]), 0.2654 / np.median(ts[1:])))
