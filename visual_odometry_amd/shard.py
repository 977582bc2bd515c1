"""Multi-GPU sharding of frame pairs.

A frame pair needs only its two frames and K, so pairs are independent units: pair p runs on rank p mod world,
no data-path collective.  The only exchange is the gather of the fixed-size per-pair result records at the end
(RCCL when the backend is "nccl"; the same code runs on gloo for the CPU tests).
"""
from __future__ import annotations

import numpy as np

RECORD_FIELDS = ("status", "n_kp1", "n_kp2", "n_match", "n_inl", "n_good", "n_world", "success", "iters")
RECORD_DOUBLES = 21  # E[9], R[9], t[3]


def pairs_of_rank(n_pairs: int, rank: int, world: int) -> list[int]:
    """Global pair indices handled by `rank` (round robin)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_pairs, world))


def pack_records(outs) -> np.ndarray:
    """vo_pair_out structs -> [n, 9 + 21] float64 records (integers are exact in float64)."""
    rec = np.zeros((len(outs), len(RECORD_FIELDS) + RECORD_DOUBLES), np.float64)
    for i, o in enumerate(outs):
        rec[i, : len(RECORD_FIELDS)] = [getattr(o, f) for f in RECORD_FIELDS]
        rec[i, len(RECORD_FIELDS) :] = list(o.E) + list(o.R) + list(o.t)
    return rec


def gather_records(local: np.ndarray, n_pairs: int, rank: int, world: int, device="cpu") -> np.ndarray | None:
    """All ranks call this with their local records (rows in the order of pairs_of_rank).  Returns the [n_pairs, 30]
    table in global pair order on every rank (all_gather of padded blocks)."""
    import torch
    import torch.distributed as dist

    width = len(RECORD_FIELDS) + RECORD_DOUBLES
    per = (n_pairs + world - 1) // world
    buf = torch.zeros((per, width), dtype=torch.float64, device=device)
    if len(local):
        buf[: len(local)] = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    out = np.zeros((n_pairs, width), np.float64)
    for r in range(world):
        idx = pairs_of_rank(n_pairs, r, world)
        out[idx] = parts[r][: len(idx)].cpu().numpy()
    return out
