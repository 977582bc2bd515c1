#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of the VO front end (2x SURF detect+describe, brute-force 2-NN + ratio,
E-RANSAC + recoverPose, triangulation) on synthetic 1080p pairs with ~8k keypoints per frame (BASELINE config 2).

A "step" is one pass of the device-resident batch pipeline over BATCH frame pairs whose frames are already in
HBM.  With --gpus N the pairs are sharded over N ranks (one process per GPU, no data-path collective); the
only collective is the gather of the per-pair result records at the end (RCCL via torch.distributed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense FP32 MFMA peak (= vector rate)


def layer_samples(h, w):
    return sum(5 * (h >> o) * (w >> o) for o in range(4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="frame pairs per step per GPU")
    ap.add_argument("--scenes", type=int, default=4, help="distinct synthetic pairs per GPU (replicated to fill the batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=6)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev)

    from visual_odometry_amd import synth
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Batch, Context

    H, W = 1080, 1920
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    ctx = Context(prm, device=dev)
    scenes = [synth.scene_c2(2002 + rank * 100 + i) for i in range(args.scenes)]
    batch = Batch(ctx, args.batch, H, W)
    for i in range(args.batch):
        sc = scenes[i % len(scenes)]
        batch.set_pair(i, sc.img1, sc.img2, sc.K)
    batch.sync()  # frames are resident in HBM from here on

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    batch.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
    batch.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # result records of this rank's pairs; the gather of these records is the only collective of the job
    outs = [batch.get(i) for i in range(args.batch)]
    rec = torch.tensor([[o.status, o.n_kp1, o.n_kp2, o.n_match, o.n_inl, o.n_good, o.n_world, o.success] for o in outs],
                       dtype=torch.int32, device="cuda")
    if world > 1:
        allrec = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
        rec_all = torch.cat(allrec).cpu().numpy()
    else:
        rec_all = rec.cpu().numpy()

    # per-stage device time of one more pass (HIP events on the library's stream)
    ctx.profile_enable(True)
    ctx.profile_reset()
    nprof = 3
    for _ in range(nprof):
        batch.run()
    batch.sync()
    prof = ctx.profile()
    ctx.profile_enable(False)

    if rank == 0:
        kp = float(rec_all[:, 1:3].mean())
        stage_ms = {k: v[0] / nprof for k, v in prof.items()}
        launches = {k: v[1] // nprof for k, v in prof.items()}
        frames = 2 * args.batch
        S = layer_samples(H, W)
        I = 4 * (H + 1) * (W + 1)
        # ALGORITHMIC bytes / flops per launch (SURVEY 8d per-frame figures x frames per launch)
        alg = {
            "integral": ("hbm", frames * (H * W + I)),
            "hessian_nms": ("hbm", frames * ((I + 8 * S) + (4 * S + 28 * kp))),
            "describe": ("hbm", frames * (H * W + I + 256 * kp)),
            "match_mfma": ("mfma", args.batch * 2 * 64 * float(rec_all[:, 1].mean()) * float(rec_all[:, 2].mean())),
        }
        dom = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        bound, work = alg[dom]
        dur_s = stage_ms[dom] / 1e3 / max(launches[dom] if dom != "describe" else 1, 1)
        if bound == "hbm":
            achieved, peak, unit = work / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = work / dur_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        roofline = {"kernel": dom, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 5), "traffic": None,
                    "launch_ms": round(dur_s * 1e3, 4), "algorithmic_per_launch": work}

        cpu = None
        if not args.no_cpu_baseline:
            import oracle_lib as O

            cores = min(len(os.sched_getaffinity(0)), 16)
            O.set_threads(cores)
            O.pair(scenes[0].img1, scenes[0].img2, scenes[0].K, prm)  # warm-up
            tc = time.perf_counter()
            n_cpu = 0
            while n_cpu < args.cpu_pairs and (time.perf_counter() - tc) < 30.0:
                sc = scenes[n_cpu % len(scenes)]
                O.pair(sc.img1, sc.img2, sc.K, prm)
                n_cpu += 1
            cdt = time.perf_counter() - tc
            cpu = {"value": round(n_cpu / cdt, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                   "sample": f"{n_cpu} of the same synthetic 1080p pairs through oracle/libvo_oracle.so (OpenMP, {cores} threads)"}

        pairs = args.batch * args.steps * world
        line = {
            "metric": "frame-pairs/sec (detect+match+E-RANSAC) 1080p/8k kpts",
            "value": round(pairs / dt, 3),
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (SURF, matcher) / f64 (RANSAC, pose, DLT)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1080p textured frame pairs, SURF ~8k kpts/frame, full "
                                   "detect x2 / BF match / E-RANSAC / recoverPose / triangulate",
                       "pairs_per_step_per_gpu": args.batch, "distinct_pairs_per_gpu": len(scenes),
                       "mean_keypoints_per_frame": round(kp, 1), "mean_matches": round(float(rec_all[:, 3].mean()), 1),
                       "mean_inliers": round(float(rec_all[:, 4].mean()), 1),
                       "pairs_with_pose": int(rec_all[:, 7].sum()), "pairs_total": int(len(rec_all)),
                       "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        print(json.dumps(line))
    batch.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
