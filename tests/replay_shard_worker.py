"""Worker of tests/test_cpu_shard_replay.py: one rank of replay_sharded on gloo with a CPU stand-in for the GPU pair engine
(the partition, the gather of pair summaries, the host loop and the gap-pair waves are what is under test).
usage: replay_shard_worker.py <repo root> <out.json>   (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, sys.argv[1])

from visual_odometry_amd import replay  # noqa: E402


class FakeEngine:
    """Deterministic pair results from hidden camera positions: pair (j, k) moves iff the positions differ enough; blank
    frames give too few matches; far-apart pairs fail recoverPose.  Counts what it was asked to compute."""

    def __init__(self, pos, blank=()):
        self.pos, self.blank, self.cache = np.asarray(pos, np.float64), set(blank), {}
        self.blocks, self.pair_calls = [], []

    def _make(self, j, k):
        rng = np.random.default_rng(1000 * j + k)
        n = 5 if (j in self.blank or k in self.blank) else 120
        p1 = rng.uniform(50, 500, (n, 2)).astype(np.float32)
        d = self.pos[k] - self.pos[j]
        p2 = (p1 + np.float32(60.0 * d) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
        ok = abs(d) < 2.0 and n >= 20
        a = 0.01 * d + 1e-3 * ((j * 7 + k) % 5)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        t = np.array([np.sign(d) if d else 1.0, 0.1 * ((j + k) % 3), 0.2])
        t = t / np.linalg.norm(t)
        m = 40 + (j + k) % 17 if ok else 0
        Xk = np.stack([rng.normal(0, 1, m), rng.normal(0, 1, m), rng.uniform(4, 9, m)]) if m else np.zeros((3, 0))
        return {"status": 0, "p1": p1, "p2": p2, "success": ok, "R": R, "t": t, "n_inl": int(0.8 * n), "in1": p1, "in2": p2, "Xk": Xk}

    def run_block(self, f0, f1):
        self.blocks.append((f0, f1))
        for f in range(f0, f1):
            self.cache[(f, f + 1)] = self._make(f, f + 1)

    def run_pairs(self, pairs):
        self.pair_calls.append(list(pairs))
        for j, k in pairs:
            self.cache[(j, k)] = self._make(j, k)

    def get(self, j, k):
        if (j, k) not in self.cache:
            self.cache[(j, k)] = self._make(j, k)
        return self.cache[(j, k)]


def scenario(n=97, seed=5):
    """positions with stationary stretches (one straddling the 2-rank block boundary at frame 48, one at the very end), a slow
    drift (pairs that only move against an older prev), blank frames and one jump (FAIL_RECOVER)"""
    rng = np.random.default_rng(seed)
    step = rng.uniform(0.4, 0.7, n)
    step[0] = 0
    for a, b in ((10, 16), (44, 53), (90, 97)):
        step[a:b] = 0.0
    step[60:66] = 0.12  # drift: consecutive pairs below the 20 px gate, but (59, 62) moves
    step[30] = 3.0  # too far: FAIL_RECOVER
    pos = np.cumsum(step)
    stamps = 100.0 + 0.1 * np.arange(n)
    alt = 6.0 + 0.01 * np.arange(n)
    return pos, (20, 21, 75), stamps, alt


def digest(records):
    h = hashlib.sha256()
    for r in records:
        h.update(json.dumps([r["frame"], r["prev"], r["n_matches"], r["fail"]]).encode())
        for key in ("R", "t", "SF", "estimate_pos", "estimate_rpy", "twist_linear", "twist_angular"):
            if key in r:
                h.update(np.ascontiguousarray(r[key], np.float64).tobytes())
    return h.hexdigest()


def reference_records(pos, blank, stamps, alt):
    eng = FakeEngine(pos, blank)
    recs, missing = replay._walk(len(pos), stamps, alt, lambda j, k: replay.summarize(eng.get(j, k), alt[k]), eng.get, None, np.eye(3),
                                 np.eye(3), np.zeros(3), np.eye(3), False, None)
    assert not missing
    return recs


def main():
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, blank, stamps, alt = scenario()
    eng = FakeEngine(pos, blank)
    stats = {}
    recs = replay.replay_sharded(None, [None] * len(pos), stamps, alt, np.eye(3), np.eye(3), np.zeros(3), rank, world, engine=eng,
                                 stats=stats)
    out = {"rank": rank, "digest": digest(recs), "fails": [r["fail"] for r in recs], "prevs": [r["prev"] for r in recs],
           "blocks": eng.blocks, "pair_calls": eng.pair_calls, "stats": stats,
           "with_points": sum("world_pointsW" in r for r in recs)}
    json.dump(out, open(sys.argv[2] + f".{rank}", "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
