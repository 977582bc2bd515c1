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


def frame_blocks(n_frames: int, world: int) -> list[tuple[int, int]]:
    """Contiguous frame blocks for the replay of ONE sequence: block r = frames f0 .. f1 inclusive and owns the
    consecutive pairs (f, f + 1), f0 <= f < f1; neighbouring blocks share one frame, so every frame is detected once per
    block it belongs to and no features are exchanged.  Pairs are spread as evenly as possible; a rank may get none."""
    if world <= 0 or n_frames < 1:
        raise ValueError("bad n_frames/world")
    n_pairs = n_frames - 1
    out, f = [], 0
    for r in range(world):
        cnt = n_pairs // world + (1 if r < n_pairs % world else 0)
        out.append((f, f + cnt))
        f += cnt
    return out


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
    if world > 1 or (dist.is_available() and dist.is_initialized()):  # a one-rank group still goes through the collective
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    out = np.zeros((n_pairs, width), np.float64)
    for r in range(world):
        idx = pairs_of_rank(n_pairs, r, world)
        out[idx] = parts[r][: len(idx)].cpu().numpy()
    return out


def gather_rows(local: np.ndarray, counts: list[int], rank: int, world: int, device="cpu") -> np.ndarray:
    """All-gather of float64 rows when every rank knows how many rows each rank contributes (`counts`; the partitions
    above are deterministic, so no size exchange is needed).  Returns the rows of rank 0, 1, ... concatenated."""
    local = np.ascontiguousarray(local, np.float64)
    if local.ndim != 2 or local.shape[0] != counts[rank]:
        raise ValueError("local rows do not match counts[rank]")
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    width, per = local.shape[1], max(max(counts), 1)
    buf = torch.zeros((per, width), dtype=torch.float64, device=device)
    if len(local):
        buf[: len(local)] = torch.from_numpy(local).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return np.concatenate([parts[r][: counts[r]].cpu().numpy() for r in range(world)], axis=0)
