"""Rate of the SURF constructor variants (vo_params.n_octaves / n_octave_layers / upright / extended) through the batch path:
16 synthetic 1080p pairs (synth.scene_c2) per run, frames resident, median of 5 runs after a warm-up.  The reference's values
(4, 3, false, false) take the tuned kernels; every other value takes a plain form of the same arithmetic."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Batch, Context, default_params


def main():
    n = 16
    scenes = [synth.scene_c2(2002 + i) for i in range(4)]
    out = []
    for name, kw in [("reference (4, 3, false, false)", {}), ("n_octaves=3", {"n_octaves": 3}), ("n_octave_layers=2", {"n_octave_layers": 2}),
                     ("n_octave_layers=4", {"n_octave_layers": 4}), ("n_octave_layers=6", {"n_octave_layers": 6}), ("n_octaves=5", {"n_octaves": 5}), ("upright", {"upright": 1}),
                     ("extended", {"extended": 1})]:
        p = default_params()
        for k, v in kw.items():
            setattr(p, k, v)
        h, w = scenes[0].img1.shape
        p.roi_x_lo, p.roi_y_lo, p.roi_x_hi, p.roi_y_hi = synth.roi_for(h, w)
        p.kp_capacity = 16384
        c = Context(p)
        b = Batch(c, n, h, w)
        for i in range(n):
            sc = scenes[i % len(scenes)]
            b.set_pair(i, sc.img1, sc.img2, sc.K)
        ts = []
        for r in range(6):
            t = time.perf_counter()
            b.run()
            b.sync()
            ts.append(time.perf_counter() - t)
        o = b.get(0)
        ms = float(np.median(ts[1:])) * 1e3
        rec = {"variant": name, "ms_per_16_pairs": round(ms, 2), "pairs_per_s": round(n / ms * 1e3, 1), "kp1": int(o.n_kp1), "matches": int(o.n_match),
               "status": int(o.status)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        b.close()
        c.close()
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
