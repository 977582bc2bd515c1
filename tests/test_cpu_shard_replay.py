"""BASELINE configs[2] / SURVEY 8e on the CPU: the sharded replay (contiguous frame blocks per rank, all-gather of pair
summaries, the node's sequential loop on every rank, extra waves for the pairs across NOT_MOVING gaps) with a stand-in for
the GPU pair engine, on gloo with 1, 2 and 3 ranks, against the unsharded loop."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "replay_shard_worker.py")


def test_frame_blocks_cover_every_pair_once():
    from visual_odometry_amd.shard import frame_blocks

    for n, w in [(512, 8), (97, 2), (97, 3), (5, 8), (2, 1), (1, 4)]:
        b = frame_blocks(n, w)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == n - 1
        pairs = [(f, f + 1) for f0, f1 in b for f in range(f0, f1)]
        assert pairs == [(f, f + 1) for f in range(n - 1)]
        assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))  # neighbouring blocks share exactly one frame
        sizes = [f1 - f0 for f0, f1 in b]
        assert max(sizes) - min(sizes) <= 1


def _run(world, tmp_path, port):
    out = str(tmp_path / f"w{world}")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, WORKER, ROOT, out], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    return [json.load(open(out + f".{r}")) for r in range(world)]


@pytest.mark.parametrize("world,port", [(1, 29551), (2, 29552), (3, 29553)])
def test_sharded_replay_equals_the_sequential_loop(world, port, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import replay_shard_worker as W
    from visual_odometry_amd import chain
    from visual_odometry_amd.shard import frame_blocks

    pos, blank, stamps, alt = W.scenario()
    ref = W.reference_records(pos, blank, stamps, alt)
    fails = [r["fail"] for r in ref]
    # the scenario does exercise every branch of the loop
    assert {chain.FAIL_DETECTION, chain.NOT_MOVING, chain.FAIL_RECOVER, chain.SUCCESS} <= set(fails)
    assert sum(f == chain.NOT_MOVING for f in fails) >= 16
    assert any(r["prev"] < r["frame"] - 1 and r["fail"] == chain.SUCCESS for r in ref)  # a pair across a gap succeeded
    res = _run(world, tmp_path, port)
    blocks = frame_blocks(len(pos), world)
    for r, out in enumerate(res):
        assert out["digest"] == W.digest(ref), (world, r)
        assert out["fails"] == fails and out["prevs"] == [q["prev"] for q in ref]
        assert [tuple(b) for b in out["blocks"]] == [blocks[r]]  # each rank detects its own block only
        assert out["stats"]["waves"] >= 2 and out["stats"]["gap_pairs"] > 0
    # the gap pairs were spread over the ranks, none computed twice
    all_gap = [tuple(p) for out in res for call in out["pair_calls"] for p in call]
    assert len(all_gap) == len(set(all_gap)) == res[0]["stats"]["gap_pairs"]
    if world > 1:
        assert all(any(len(c) for c in out["pair_calls"]) for out in res)
    # world points in {W} exist exactly for the successful pairs a rank computed itself: together they cover all of them
    assert sum(out["with_points"] for out in res) == sum(f == chain.SUCCESS for f in fails)
