#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of the VO front end (2x SURF detect+describe, brute-force 2-NN + ratio,
E-RANSAC + recoverPose, triangulation) on synthetic 1080p pairs with ~8k keypoints per frame (BASELINE configs[1]).

A "step" is one pass of the batch pipeline over BATCH frame pairs.  `value` is the SURVEY 8d metric: steady state with both
frames of every pair uploaded from host memory in every step (pinned staging + the batch's copy stream: the upload of step
k + 1 runs beside the kernels of step k) and the result records read back on the host.  The rate with the frames resident in
HBM is reported beside it (`resident`).  With --gpus N the pairs are sharded over N ranks (one process per GPU, no data-path
collective); the only collective is the gather of the per-pair result records (RCCL via torch.distributed).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (fresh child processes, before this
process touches torch or HIP); under torchrun it is one rank.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense FP32 MFMA peak (= vector rate)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6  # vector FP64 (MI355X_MICROARCH.md)
N_SIMD, VALU_CLOCK_HZ = 1024, 2.4e9
SAMPLE_ISSUE_CYCLES = 86  # VALU issue cycles of one bilinear window sample per wavefront
MFMA_EXEC_FACTOR = 25.0 / 8.0  # executed bf16 MFMA flop per algorithmic FP32 flop of the matcher (split-bf16 products + norm step)
C3_STEP_M, C3_DEPTH_M, C3_RELIEF_M = 0.5, 6.0, 1.8  # the configs[2] stand-in scene (synth.sequence_c3)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")  # PMC passes of `bench.py --profile-region`


def layer_samples(h, w):
    return sum(5 * (h >> o) * (w >> o) for o in range(4))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per step per GPU (~75 MB of HBM per pair)")
    ap.add_argument("--scenes", type=int, default=64, help="distinct synthetic pairs per GPU (cycled to fill the batch)")
    ap.add_argument("--matcher", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="matrix-core formulation of the matcher's contraction (fp32 = v_mfma_f32_32x32x2_f32, for A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the committed profile instead of running the two rocprofv3 --pmc passes first")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only: skip the legs for the other BASELINE configs")
    ap.add_argument("--profile-region", action="store_true",
                    help="for rocprofv3: ONLY the per-kernel timing pass of the headline workload (every kernel alone on the main "
                         "stream, the condition roofline.launch_ms is measured under), so that the profiler's average duration of a "
                         "kernel is comparable with it")
    ap.add_argument("--c3-frames", type=int, default=512)
    return ap.parse_args()


def launch_ranks(args) -> int:
    """--gpus N outside torchrun: start the N ranks as children of this (GPU-free) process, as the driver would."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _scene_job(job):
    """worker of the generation pool (fresh process, numpy only)"""
    from visual_odometry_amd import synth

    kind, seed = job
    sc = synth.scene_c2(seed) if kind == "c2" else synth.scene_c1(seed)
    return sc.img1, sc.img2, sc.K


def make_inputs(args, rank, world, extras):
    """All synthetic inputs, rendered on the host BEFORE anything initialises the GPU (the pool forks/spawns processes)."""
    import multiprocessing as mp

    from visual_odometry_amd import synth

    ncpu = len(os.sched_getaffinity(0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    workers = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "16")), ncpu // max(local_world, 1)))
    jobs = [("c2", 2002 + rank * 1000 + i) for i in range(args.scenes)]
    if extras:
        jobs += [("c1", 1001 + i) for i in range(16)]
    t0 = time.perf_counter()
    if workers > 1:
        pool = mp.get_context("spawn").Pool(workers)
        try:
            res = pool.map(_scene_job, jobs, chunksize=1)
        finally:
            pool.close()  # let the workers exit by themselves: under rocprofv3 a terminate() (SIGTERM) can hang them in the tool's handler
            pool.join()
    else:
        res = [_scene_job(j) for j in jobs]
    c2 = res[: args.scenes]
    c1 = res[args.scenes :]
    c3 = None
    if args.c3_frames >= 8 and not args.profile_region:
        # 0.5 m per frame over a seabed at 6 +- 1.8 m: ~48 px of image motion and enough relief for the cheirality gate
        # (the gentler default scene of the tests - 0.36 m, 7 +- 1.2 m - sends 16 % of the pairs down the FAIL_RECOVER branch)
        c3 = synth.sequence_c3(args.c3_frames, seed=3003, step_m=C3_STEP_M, depth=C3_DEPTH_M, relief=C3_RELIEF_M, workers=workers)
    return c2, c1, c3, workers, time.perf_counter() - t0


def live_traffic(args):
    """HBM-side bytes per kernel launch of THIS workload on THIS machine: FETCH_SIZE and WRITE_SIZE in two separate
    `rocprofv3 --pmc` passes (MI355X_MICROARCH.md, HBM / rocprofv3 section) over `bench.py --profile-region`, run as child
    processes BEFORE this process touches the GPU.  Returns ({kernel name: bytes per launch}, note) or (None, why not)."""
    import csv
    import glob
    import re
    import shutil
    import statistics
    import subprocess
    import tempfile

    # already under a profiler (rocprofv3 -- python3 bench.py ...): no profiler inside a profiler
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="vo_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    acc = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--profile-region", "--steps", "2", "--batch", str(args.batch), "--scenes", str(args.scenes), "--matcher", args.matcher]
            # own process group: on a timeout the profiler AND the program it started are killed, nothing is left on the GPU
            # beside the timed loop
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=90)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                pr.wait()
                return None, f"rocprofv3 --pmc {ctr} pass timed out"
            if rc != 0:
                return None, f"rocprofv3 --pmc {ctr} pass exited with {rc}"
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr:
                        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                        vals.setdefault(name, []).append(float(row["Counter_Value"]))
            if not vals:
                return None, f"rocprofv3 --pmc {ctr} pass produced no counters"
            acc[ctr] = {k: statistics.median(v) for k, v in vals.items()}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    # KB units; FETCH_SIZE doubled: the guide's gfx950 correction for wide reads (uncalibrated for the gathers of these kernels)
    return ({k: (2.0 * f + acc["WRITE_SIZE"].get(k, 0.0)) * 1024.0 for k, f in acc["FETCH_SIZE"].items()},
            "two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of `bench.py --profile-region --steps 2` run by this bench before its timed "
            "loop: median over launches, (2*FETCH_SIZE + WRITE_SIZE) KB")


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    live, live_note = None, "not requested"
    if ("RANK" not in os.environ and not args.profile_region and not args.no_live_traffic and not args.no_extras
            and args.batch == 64):
        live, live_note = live_traffic(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    extras = world == 1 and not args.no_extras and not args.profile_region
    c2, c1, c3, gen_workers, gen_s = make_inputs(args, rank, world, extras)

    import numpy as np
    import torch

    ndev = torch.cuda.device_count()
    backend = os.environ.get("VO_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun even a single rank goes through the process group (RCCL)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the backend of the real runs; gloo rehearses the N > 1 path with several ranks on one GPU (RCCL
        # refuses two ranks on the same device) and is chosen by itself when there are fewer GPUs than ranks
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if use_dist and backend == "nccl" else "cpu"
    dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)

    from visual_odometry_amd import shard, synth
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Batch, Context, Sequence

    H, W, B = 1080, 1920, args.batch
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    ctx = Context(prm, device=dev)
    if args.matcher == "fp32":
        ctx.debug_set(2, 1)
    batch = Batch(ctx, B, H, W)

    def fill(k):
        # step k stages pair (i + k) mod scenes at position i: every step uploads different frames into every slot
        for i in range(B):
            img1, img2, K = c2[(i + k) % len(c2)]
            batch.set_pair(i, img1, img2, K)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return dt

    def stage_pass(nprof):
        """per-kernel device time: stage timers on = every kernel alone on the main stream, HIP events around each"""
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(nprof):
            batch.run()
            batch.sync()
        prof = ctx.profile()
        ctx.profile_enable(False)
        return {k: v[0] / nprof for k, v in prof.items()}

    if args.profile_region:
        fill(0)
        stage_ms = stage_pass(max(args.steps, 1))  # every launch of this process with the stage timers on: each kernel alone
        if rank == 0:
            print(json.dumps({"profile_region": True, "steps": max(args.steps, 1), "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()}}))
        batch.close()
        ctx.close()
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- headline: PCIe-inclusive steady state ---------------------------------------------------------------------
    # A step = stage + upload both frames of every pair, run the pipeline on them, have the result records on the host.  The
    # batch is a two-deep pipeline: while run k executes, the host stages the frames of step k + 1 into the other frame slot.
    # The warm-up leaves the pipeline primed (frames of the first timed step staged), so the timed region holds exactly
    # `steps` runs, `steps` stagings and `steps` record read-backs; the upload of the first timed step runs with the GPU
    # idle (it is inside the region, after the barrier).
    fill(0)
    for k in range(args.warmup):
        batch.run()
        fill(k + 1)
        if k > 0:
            _ = batch.get_records(age=1)
    batch.sync()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        batch.run()  # upload (copy stream) + kernels of step k
        fill(args.warmup + k + 1)  # pinned staging of step k + 1, beside step k's kernels
        if k > 0:
            _ = batch.get_records(age=1)  # records of step k - 1: on the host since that step finished
    batch.sync()
    outs = batch.get_records()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    # result records of this rank's pairs of the last step; their gather is the only collective of the job
    rec_local = shard.pack_records(outs)
    rec_all = shard.gather_records(rec_local, B * world, rank, world, device=coll_dev)

    # frames resident in HBM (reported beside `value`): the same batch re-run without new uploads
    barrier()
    tr = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
    batch.sync()
    barrier()
    dt_res = max_over_ranks(time.perf_counter() - tr)

    stage_ms = stage_pass(3)
    import ctypes as C

    redo = C.c_int32(-1)  # queries of the last step whose matcher shortlist was not certified (exact scan instead)
    ctx._chk(ctx.lib.vo_debug_fetch(ctx._h, 4, C.byref(redo), C.c_size_t(4)))

    # ---- BASELINE configs[2]: the C3 run, frames in blocks over the ranks (replay_sharded) ----------------------------
    c3_out = None
    if c3 is not None:
        from visual_odometry_amd import replay
        from visual_odometry_amd._abi import default_params as dp

        ctx3 = Context(dp(), device=dev)  # the reference's own ROI (28/20/614/482) is the one for 640 x 505 frames
        n3 = len(c3.frames)
        stamps, alt = 0.1 * np.arange(n3), np.full(n3, C3_DEPTH_M)
        stats = {}
        # warm-up: the same replay once (the context keeps its sequence / pair batches, so the timed run allocates nothing)
        replay.replay_sharded(ctx3, c3.frames, stamps, alt, c3.K, np.eye(3), np.zeros(3), rank, world, device=coll_dev)
        d3s = []
        for _ in range(3):  # three timed replays, the fastest is reported (all three are in the line: `seconds_all`)
            barrier()
            t3 = time.perf_counter()
            stats = {}
            recs = replay.replay_sharded(ctx3, c3.frames, stamps, alt, c3.K, np.eye(3), np.zeros(3), rank, world, device=coll_dev, stats=stats)
            barrier()
            d3s.append(max_over_ranks(time.perf_counter() - t3))
        d3 = min(d3s)
        fails = [r["fail"] for r in recs]
        last = [r for r in recs if "estimate_pos" in r][-1]  # the last iteration that produced a pose (a stationary tail publishes none)
        c_true = -c3.Rw[last["frame"]].T @ c3.tw[last["frame"]]
        c3_out = {"value": round((n3 - 1) / d3, 2), "unit": "frames/s (one loop iteration of the node per frame)", "frames": n3, "size": "640x505",
                  "K": "config/rapallo_intrinsic.yaml via updateCameraMatrix", "ranks": world, "seconds_all": [round(v, 4) for v in d3s],
                  "scene": {"step_m": C3_STEP_M, "depth_m": C3_DEPTH_M, "relief_m": C3_RELIEF_M, "stationary_frames": 25}, "seconds": round(d3, 4),
                  "success": fails.count(3), "not_moving": fails.count(1), "fail_recover": fails.count(2), "fail_detection": fails.count(0),
                  "waves": stats.get("waves"), "gap_pairs": stats.get("gap_pairs"),
                  "end_position_error_m": round(float(np.linalg.norm(last["estimate_pos"] - c_true)), 3),
                  "path_length_m": round(float(sum(c3.relative(k - 1, k)[2] for k in range(1, n3))), 2),
                  "note": "includes the per-chunk upload of the frames, result read-back, the gather of pair summaries and the host loop; second replay of the sequence on a context that keeps its batches (no allocation inside the timed run)"}
        ctx3.close()

    extras_out = {}
    if extras and rank == 0:
        extras_out = run_extras(np, ctx, prm, c2, c1, dev, args)

    if rank == 0:
        assert rec_all.shape[0] == B * world
        kp = float(rec_all[:, 1:3].mean())
        frames = 2 * B
        S = layer_samples(H, W)
        I = 4 * (H + 1) * (W + 1)
        # share of the Hessian samples per octave, and of the keypoints / window samples per descriptor class
        S_oct = [5 * (H >> o) * (W >> o) for o in range(4)]
        kp1, _ = ctx.detect_describe(c2[0][0])
        win = (21 * (kp1["size"].astype(np.float32) * np.float32(1.2) / np.float32(9.0))).astype(np.int64)
        cls_share, cls_samples = {}, {}
        for name, lo, hi in (("describe_w64", 0, 64), ("describe_w128", 64, 128), ("describe_w320", 128, 320), ("describe_w768", 320, 768)):
            m = (win > lo) & (win <= hi)
            cls_share[name] = float(m.mean())
            # bilinear window samples of the class per launch: win^2 per keypoint (frame 1 of the first scene, scaled to kp)
            cls_samples[name] = frames * kp * float((win[m].astype(np.float64) ** 2).sum()) / max(len(win), 1)
        # ALGORITHMIC bytes / flops per launch (SURVEY 8d per-frame figures x frames per launch; per-kernel shares of them)
        hess_bytes = lambda So: frames * ((I + 8 * So) + (4 * So + 28 * kp * So / S))
        alg = {
            "integral": ("hbm", frames * (H * W + I)),
            "hessian_oct0_23": ("hbm", hess_bytes(S_oct[0] + S_oct[2] + S_oct[3])),
            "hessian_oct1": ("hbm", hess_bytes(S_oct[1])),
            # 2*64*N1*N2 algorithmic flop per pair; the kernel EXECUTES 3.125x that on the bf16 matrix pipe (three bf16 partial
            # products per FP32 product plus the norm step: 25 v_mfma_f32_32x32x16_bf16 per two 32x32 tiles instead of 8)
            "match_mfma": ("mfma", B * 2 * 64 * float(rec_all[:, 1].mean()) * float(rec_all[:, 2].mean())),
        }
        for name, sh in cls_share.items():
            alg[name] = ("hbm", frames * (sh * (H * W + I) + 256 * kp * sh))

        # HBM-side traffic per launch: PMC counters cannot be read from inside this process; the figures come from the
        # committed rocprofv3 --pmc passes of `bench.py --profile-region` (profiles/): default workload only
        traffic, traffic_source = {}, None
        prefix = {"describe_w64": "k_describe_tiled<64,", "describe_w128": "k_describe_tiled<128,",
                  "describe_w320": "k_describe_tiled<320,", "describe_w768": "k_describe_tiled<768,",
                  "hessian_oct0_23": "k_hessian_nms_fused<", "hessian_oct1": "k_hessian_nms_lds<", "integral": "k_integral",
                  "match_mfma": "k_match_bf3"}
        if live:  # measured by this very invocation (live_traffic)
            traffic_source = live_note
            for st, pf in prefix.items():
                for kname, v in live.items():
                    if kname.startswith(pf):
                        traffic[st] = v
        elif B == 64 and os.path.exists(TRAFFIC_JSON):
            traffic_source = f"committed profile {os.path.relpath(TRAFFIC_JSON, ROOT)} (live passes: {live_note})"
            tj = json.load(open(TRAFFIC_JSON))
            for st, pf in prefix.items():
                for kname, v in tj.items():
                    if kname.startswith(pf):
                        traffic[st] = v["hbm_bytes_per_launch"]

        def roof(name):
            bound, work = alg[name]
            dur_s = stage_ms[name] / 1e3
            if bound == "hbm":
                achieved, peak, unit = work / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                if args.matcher == "fp32":  # 33 FP32 MFMAs per 32 algorithmic ones (norm step)
                    achieved, peak, unit = (33.0 / 32.0) * work / dur_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
                else:
                    achieved, peak, unit = MFMA_EXEC_FACTOR * work / dur_s / 1e12, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s"
            out = {"kernel": name, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                   "frac": round(achieved / peak, 5), "traffic": traffic.get(name), "traffic_source": traffic_source if name in traffic else None,
                   "launch_ms": round(dur_s * 1e3, 4),
                   "algorithmic_per_launch": work}
            if name in cls_samples:
                # What bounds these kernels is VALU issue: a window sample is 24 instructions = 86 issue cycles per wavefront
                # (10 fp64 / integer coordinate and address operations, 4 byte->float conversions, 4 packed multiplies and the
                # u8 pack at 4 cycles, 5 FP32 adds at 2; costs measured by tools/ubench/valu_rates.hip).  1024 SIMDs x 2.4 GHz.
                wsamples = cls_samples[name] / 64.0
                cyc = dur_s * VALU_CLOCK_HZ * N_SIMD / max(wsamples, 1.0)
                out["valu_issue"] = {"window_samples_per_launch": round(cls_samples[name]), "cycles_per_wavefront_sample": round(cyc, 1),
                                     "floor_cycles": SAMPLE_ISSUE_CYCLES, "frac_of_floor": round(SAMPLE_ISSUE_CYCLES / cyc, 4)}
                # ... and by the gathers themselves (round 3 knock-outs, DESIGN.md 5.1: the kernels lose 0.6 ms of 19.9 without the
                # bilinear arithmetic and 4.4 ms without the loads): one 64-lane gather of 4-byte footprint words per wavefront-sample,
                # each touching ~8 cache lines of the strip-tiled footprint image (L2 -> L1 fills at 64 B per clock and CU)
                out["gathers"] = {"wavefront_gathers_per_launch": round(wsamples), "ns_per_gather_per_cu": round(dur_s * 1e9 * 256 / max(wsamples, 1.0), 2),
                                  "clocks_per_gather_per_cu": round(dur_s * VALU_CLOCK_HZ * 256 / max(wsamples, 1.0), 1)}
            if bound == "mfma" and args.matcher != "fp32":
                out["note"] = ("executed bf16 MFMA flop (3.125 x the algorithmic 2*64*N1*N2) against the dense bf16 peak; "
                               f"algorithmic rate {round(work / dur_s / 1e12, 1)} TFLOP/s = "
                               f"{round(work / dur_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 2)} x the FP32 MFMA peak it used to run on")
            return out

        dom = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        roofline = roof(dom)
        # The descriptor and Hessian kernels are bound by VALU issue and the texture addresser's gather rate, not by HBM
        # (PMC evidence under profiles/): their HBM fraction is small by construction.
        roofline["note"] = ("dominant kernel by device time, timed alone on the main stream (the condition `bench.py --profile-region` "
                            "reproduces under rocprofv3); descriptor/Hessian kernels are VALU-issue / gather bound, see roofline_all and DESIGN.md")
        roofline_all = [roof(k) for k in sorted(alg, key=lambda k: -stage_ms.get(k, 0.0))]

        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU port is timed at N = 1 only (rank 0 is the only rank there)
            cpu = cpu_baseline(np, c2, prm)

        pairs = B * args.steps * world
        line = {
            "metric": "frame-pairs/sec (detect+match+E-RANSAC) 1080p/8k kpts",
            "value": round(pairs / dt, 3),
            "unit": "frame-pairs/s",
            "n_gpus": min(world, max(ndev, 1)) if backend != "nccl" else world,
            "ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (SURF, matcher distances; shortlist contraction on " + ("FP32 MFMA" if args.matcher == "fp32" else "split-bf16 MFMA")
                     + ") / f64 (RANSAC, pose, DLT)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1080p textured frame pairs, SURF ~8k kpts/frame, full "
                                   "detect x2 / BF match / E-RANSAC / recoverPose / triangulate; both frames of every pair uploaded from "
                                   "host memory every step, result records read back (SURVEY 8d)",
                       "pairs_per_step_per_gpu": B, "distinct_pairs_per_gpu": len(c2),
                       "mean_keypoints_per_frame": round(kp, 1), "mean_matches": round(float(rec_all[:, 3].mean()), 1),
                       "mean_inliers": round(float(rec_all[:, 4].mean()), 1), "mean_ransac_iterations": round(float(rec_all[:, 8].mean()), 1),
                       "pairs_with_pose": int(rec_all[:, 7].sum()), "pairs_total": int(len(rec_all)),
                       "parallelism": f"pairs sharded over {world} rank(s) on {min(world, max(ndev, 1)) if backend != 'nccl' else world} GPU(s), no data-path collective; result records all-gathered ({backend})",
                       "input_generation_s": round(gen_s, 1), "input_generation_workers": gen_workers},
            "resident": {"value": round(pairs / dt_res, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt_res / args.steps * 1e3, 4),
                         "note": "same batch re-run with the frames already in HBM (no upload): what `value` would be without PCIe"},
            "pcie_gap": round(1.0 - (pairs / dt) / (pairs / dt_res), 4),
            "c3_replay": c3_out,
            "roofline": roofline,
            "roofline_all": roofline_all,
            "cpu_baseline": cpu,
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "match_exact_scans_per_step": redo.value,
        }
        line.update(extras_out)
        print(json.dumps(line))
    batch.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(np, c2, prm):
    """The CPU restatement (oracle/, OpenMP over layers / keypoints / query rows like OpenCV's parallel_for_) on the same
    pairs, on this host: median of 5 repetitions of a bounded sample, all the threads this job may use."""
    import oracle_lib as O

    avail = len(os.sched_getaffinity(0))
    threads = max(1, min(avail, int(os.environ.get("OMP_NUM_THREADS", str(avail)))))  # the GPU box caps its jobs' thread pools with OMP_NUM_THREADS
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    O.set_threads(threads)
    sample = c2[: min(4, len(c2))]
    O.pair(sample[0][0], sample[0][1], sample[0][2], prm)  # warm-up
    rates, t_all = [], time.perf_counter()
    for _ in range(5):
        tc = time.perf_counter()
        for img1, img2, K in sample:
            O.pair(img1, img2, K, prm)
        rates.append(len(sample) / (time.perf_counter() - tc))
        if time.perf_counter() - t_all > 25.0 and len(rates) >= 3:
            break
    O.set_threads(1)
    t1 = time.perf_counter()
    O.pair(sample[0][0], sample[0][1], sample[0][2], prm)
    one_thread = 1.0 / (time.perf_counter() - t1)
    O.set_threads(threads)
    return {"value": round(float(np.median(rates)), 4), "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "host_cpus_visible": avail, "cpu_model": model, "value_1_thread": round(one_thread, 4),
            "repetitions": [round(r, 4) for r in rates],
            "sample": f"median of {len(rates)} repetitions of {len(sample)} of the same synthetic 1080p pairs through oracle/libvo_oracle.so "
                      f"(OpenMP, {threads} threads: the thread budget of this job on the host, {avail} CPUs visible)"}


def run_extras(np, ctx, prm, c2, c1, dev, args):
    """Short legs for the other BASELINE configs (N = 1 only), each with its own roofline entry."""
    import ctypes as C

    from visual_odometry_amd import synth
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Batch, Context, Sequence

    out = {}
    H, W, B = 1080, 1920, args.batch

    def pipelined_rate(ctx_, batch_, warm, steps):
        """the headline loop (stage + upload every step, run, records of the previous step) on another context / setting"""
        def fill_(k):
            for i in range(B):
                img1, img2, K = c2[(i + k) % len(c2)]
                batch_.set_pair(i, img1, img2, K)

        fill_(0)
        for k in range(warm):
            batch_.run()
            fill_(k + 1)
        batch_.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            batch_.run()
            fill_(warm + k + 1)
            if k > 0:
                _ = batch_.get_records(age=1)
        batch_.sync()
        d = time.perf_counter() - t0
        return B * steps / d, 1e3 * d / steps, batch_.get_records()

    # ---- the headline loop with the contraction on the FP32 matrix instruction (north star's formulation; same results)
    if args.matcher != "fp32":
        bx = Batch(ctx, B, H, W)
        ctx.debug_set(2, 1)
        rate, ms, recs = pipelined_rate(ctx, bx, 2, 8)
        bx.sync()
        ctx.debug_set(2, 0)
        out["headline_fp32_mfma"] = {"value": round(rate, 2), "unit": "frame-pairs/s", "ms_per_step": round(ms, 4), "steps": 8,
                                     "mean_matches": round(float(np.mean([o.n_match for o in recs])), 1),
                                     "note": "the same timed loop (PCIe in, records out) with the matcher's shortlist contraction on "
                                             "v_mfma_f32_32x32x2_f32 instead of three split-bf16 products; results are bit-identical"}
        bx.close()

    # ---- the headline loop with RANSAC made to work: confidence 1.0 -> RANSACUpdateNumIters never drops below a few hundred
    # iterations at these inlier ratios (the reference's 0.99 stops after ~4 on these scenes)
    prm_h = default_params()
    prm_h.roi_x_lo, prm_h.roi_y_lo, prm_h.roi_x_hi, prm_h.roi_y_hi = prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi
    prm_h.ransac_prob = 1.0
    ctx_h = Context(prm_h, device=dev)
    bh = Batch(ctx_h, B, H, W)
    rate, ms, recs = pipelined_rate(ctx_h, bh, 2, 8)
    out["headline_hard_ransac"] = {"value": round(rate, 2), "unit": "frame-pairs/s", "ms_per_step": round(ms, 4), "steps": 8,
                                   "ransac_prob": 1.0, "ransac_max_iters": int(prm_h.ransac_max_iters),
                                   "mean_ransac_iterations": round(float(np.mean([o.iters for o in recs])), 1),
                                   "mean_matches": round(float(np.mean([o.n_match for o in recs])), 1),
                                   "mean_inliers": round(float(np.mean([o.n_inl for o in recs])), 1),
                                   "pairs_with_pose": int(sum(o.success for o in recs)),
                                   "note": "same frames and loop as `value`; every pair runs hundreds of five-point hypotheses x all "
                                           "its matches on the geometry stream beside the next step's detection"}
    bh.close()
    ctx_h.close()

    # consecutive-frame mode (reported beside `value`, never as `value`): the reference re-detects prev_img in every loop
    # iteration, so the headline counts two detections per pair; an offline replay detects each frame once (vo_seq_*)
    seq = Sequence(ctx, B + 1, H, W)
    for f in range(B + 1):
        sc = c2[(f // 2) % len(c2)]
        seq.set_frame(f, sc[0] if f % 2 == 0 else sc[1])  # alternate the two views: every consecutive pair has parallax
    for p in range(B):
        seq.set_pair(p, p, p + 1, c2[0][2])
    seq.run()
    seq.sync()
    ts = time.perf_counter()
    for _ in range(4):
        seq.run()
    seq.sync()
    out["consecutive_frames"] = {"value": round(B * 4 / (time.perf_counter() - ts), 2), "unit": "frame-pairs/s",
                                 "note": "sequence batch, frames resident: batch + 1 frames detected once each, pair p = (frame p, frame p + 1); not the headline workload"}
    seq.close()

    # single-pair latency of the reference's three calls through the synchronous entry points, pageable host buffers
    lat = {"opt_DetectFeatures": [], "optRelativePose": [], "triangPoints": []}
    img1, img2, K = c2[0]
    for it in range(13):
        t0 = time.perf_counter()
        g = ctx.opt_DetectFeatures(img1, img2)
        t1 = time.perf_counter()
        _st, R1, tv1, inl1, ok1 = ctx.optRelativePose(g, K)
        t2 = time.perf_counter()
        if ok1:
            ctx.triangPoints(inl1.Kpoints1, inl1.Kpoints2, R1, tv1, K)
        t3 = time.perf_counter()
        if it >= 3:
            lat["opt_DetectFeatures"].append(t1 - t0), lat["optRelativePose"].append(t2 - t1), lat["triangPoints"].append(t3 - t2)
    single = {k + "_ms": round(1e3 * float(np.median(v)), 3) for k, v in lat.items()}
    single["total_ms"] = round(sum(single.values()), 3)
    single["note"] = "one 1080p pair, synchronous C-ABI calls with pageable host buffers in and out (median of 10)"
    out["single_pair_latency"] = single

    # ---- configs[0] (C1): 640 x 480 Parete-like pairs, reference ROI, PCIe-inclusive like the headline
    ctx1 = Context(default_params(), device=dev)
    b1 = Batch(ctx1, 64, 480, 640)

    def fill1(k):
        for i in range(64):
            a, b, K1 = c1[(i + k) % len(c1)]
            b1.set_pair(i, a, b, K1)

    fill1(0)
    for k in range(2):
        b1.run()
        fill1(k + 1)
    b1.sync()
    t0 = time.perf_counter()
    n1 = 10
    for k in range(n1):
        b1.run()
        fill1(3 + k)
        if k > 0:
            _ = b1.get_records(age=1)
    b1.sync()
    d1 = time.perf_counter() - t0
    o1 = b1.get_records()
    S1 = layer_samples(480, 640)
    kp1 = float(np.mean([o.n_kp1 for o in o1]))
    alg1 = 2 * (2 * 480 * 640 + 3 * 4 * 481 * 641 + 12 * S1 + 284 * kp1)  # SURVEY 8d "detect total per frame" x 2 frames
    out["c1_640x480"] = {"value": round(64 * n1 / d1, 1), "unit": "frame-pairs/s", "ms_per_step": round(1e3 * d1 / n1, 3), "pairs_per_step": 64,
                         "distinct_pairs": len(c1), "mean_keypoints_per_frame": round(kp1, 1), "mean_matches": round(float(np.mean([o.n_match for o in o1])), 1),
                         "pairs_with_pose": int(sum(o.success for o in o1)),
                         "roofline": {"bound": "hbm", "achieved": round(alg1 * 64 * n1 / d1 / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(alg1 * 64 * n1 / d1 / 1e9 / HBM_PEAK_GBS, 5),
                                      "note": "whole pipeline: SURVEY 8d algorithmic detect bytes per pair x pairs/s"},
                         "note": "BASELINE configs[0]: Parete-like 640x480 pairs, camera_intrinsic.yaml K, reference ROI; frames uploaded every step"}
    b1.close()
    ctx1.close()

    # ---- configs[3] (C4): 32k x 32k descriptor matcher, both matrix formulations (kernel time from the stage timer)
    prm4 = default_params()
    c4 = {}
    da, db = synth.surf_like_descriptors(32768, 4004), synth.surf_like_descriptors(32768, 4005)
    for name, key in (("bf16x3", 0), ("fp32", 1)):
        ctx4 = Context(prm4, device=dev)
        ctx4.debug_set(2, key)
        ctx4.match_knn2(da, db)  # warm-up (allocations)
        ctx4.profile_enable(True)
        ctx4.profile_reset()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            idx, dist_ = ctx4.match_knn2(da, db)
        wall = (time.perf_counter() - t0) / reps
        pf = ctx4.profile()
        ctx4.profile_enable(False)
        ms = pf["match_mfma"][0] / reps
        ms_all = (pf["match_prep"][0] + pf["match_mfma"][0] + pf["match_select"][0]) / reps
        flop = 2.0 * 64 * 32768 * 32768
        if name == "fp32":
            ach, peak = (33.0 / 32.0) * flop / (ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS
        else:
            ach, peak = MFMA_EXEC_FACTOR * flop / (ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS
        c4[name] = {"mfma_kernel_ms": round(ms, 4), "matcher_device_ms": round(ms_all, 4), "call_wall_ms": round(1e3 * wall, 3),
                    "algorithmic_tflops": round(flop / (ms * 1e-3) / 1e12, 2),
                    "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                 "note": "executed matrix flop of the shortlist kernel against the dense peak of its instruction"}}
        if name == "fp32":
            c4["identical_results"] = bool(np.array_equal(idx, c4.pop("_idx")) and np.array_equal(dist_, c4.pop("_dist")))
        else:
            c4["_idx"], c4["_dist"] = idx, dist_
        ctx4.close()
    c4["note"] = "BASELINE configs[3]: 32768 x 32768 SURF-like descriptors (137.4 GFLOP algorithmic), exact 2-NN; call_wall includes the 16 MB upload and the result download"
    out["c4_matcher_32k"] = c4

    # ---- configs[4] (C5): RANSAC stress (8192 matches, 50 % outliers, 4096 hypotheses, adaptive stop off) + 100k-point DLT
    K5 = synth.K_1080P
    p1, p2, R5, t5, is_in = synth.correspondences(8192, K5, 5005, inlier_frac=0.5, noise_px=0.5)
    ctx.essential_ransac(p1, p2, K5, prob=1.0, max_iters=4096)  # warm-up; prob = 1 keeps niters at max_iters (RANSACUpdateNumIters)
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        st, E5, mask5, ninl5, iters5 = ctx.essential_ransac(p1, p2, K5, prob=1.0, max_iters=4096)
    wall5 = (time.perf_counter() - t0) / reps
    pf = ctx.profile()
    ms5 = (pf["ransac_sample"][0] + pf["ransac_solve_score"][0] + pf["ransac_select"][0]) / reps
    ms5_solve = pf["ransac_solve_score"][0] / reps
    nm = np.zeros(4096, np.int32)
    ctx._chk(ctx.lib.vo_debug_fetch(ctx._h, 1, nm.ctypes.data_as(C.c_void_p), C.c_size_t(nm.nbytes)))
    models = int(nm.sum())
    flop5 = 35.0 * 8192 * models + 25e3 * 4096  # Sampson scoring of every model on every match + the solver
    # triangulation: 100 000 correspondences under the true pose
    q1, q2, R6, t6, _ = synth.correspondences(100000, K5, 5006, inlier_frac=1.0, noise_px=0.3)
    P0 = K5 @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = K5 @ np.hstack([R6, t6[:, None]])
    ctx.triangulate(P0, P1, q1, q2)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        X4 = ctx.triangulate(P0, P1, q1, q2)
    wall6 = (time.perf_counter() - t0) / reps
    ms6 = ctx.profile()["triangulate"][0] / reps
    ctx.profile_enable(False)
    X3 = X4[:3] / X4[3]
    Xt = (np.linalg.inv(K5) @ np.vstack([q1.T.astype(np.float64), np.ones(len(q1))]))
    out["c5_ransac_triang"] = {
        "ransac": {"hypotheses": 4096, "matches": 8192, "iterations_run": int(iters5), "models_scored": models, "inliers": int(ninl5),
                   "true_inlier_fraction": round(float(is_in.mean()), 3), "device_ms": round(ms5, 3), "solve_score_ms": round(ms5_solve, 3),
                   "call_wall_ms": round(1e3 * wall5, 3), "hypotheses_per_s": round(4096 / (ms5 * 1e-3), 0),
                   "roofline": {"bound": "fp64 valu", "achieved": round(flop5 / (ms5_solve * 1e-3) / 1e9, 1), "peak": FP64_VALU_PEAK_TFLOPS * 1e3,
                                "unit": "GFLOP/s", "frac": round(flop5 / (ms5_solve * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 5),
                                "note": "35 flop per (model, match) Sampson evaluation + 25 kflop per five-point solve, over the solve+score kernels"}},
        "triangulation": {"points": 100000, "kernel_ms": round(ms6, 4), "call_wall_ms": round(1e3 * wall6, 3),
                          "roofline": {"bound": "hbm", "achieved": round(32.0 * 100000 / (ms6 * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(32.0 * 100000 / (ms6 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                       "note": "32 algorithmic bytes per point (2 x float2 in, float4 out); the kernel is ~1.5 kflop of fp64 Jacobi per point and one launch: latency bound"},
                          "median_depth_m": round(float(np.median(X3[2])), 3)},
        "note": "BASELINE configs[4]: 50 % outliers, 4096 hypotheses per pair with the adaptive stop off (prob = 1), 100k triangulated points"}
    del Xt
    return out


if __name__ == "__main__":
    main()
