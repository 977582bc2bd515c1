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
MFMA_F32_PEAK_TFLOPS = 157.3  # dense FP32 MFMA peak (= vector rate); the matcher no longer runs on it
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
N_SIMD, VALU_CLOCK_HZ = 1024, 2.4e9
SAMPLE_ISSUE_CYCLES = 86  # VALU issue cycles of one bilinear window sample per wavefront
MFMA_EXEC_FACTOR = 25.0 / 8.0  # executed bf16 MFMA flop per algorithmic FP32 flop of the matcher (split-bf16 products + norm step)


def layer_samples(h, w):
    return sum(5 * (h >> o) * (w >> o) for o in range(4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per step per GPU (resident in HBM; ~75 MB per pair)")
    ap.add_argument("--scenes", type=int, default=4, help="distinct synthetic pairs per GPU (replicated to fill the batch)")
    ap.add_argument("--matcher", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="matrix-core formulation of the matcher's contraction (fp32 = v_mfma_f32_32x32x2_f32, for A/B runs)")
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
        # RCCL ("nccl") is the backend of the real runs; VO_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on
        # one GPU (RCCL refuses two ranks on the same device)
        backend = os.environ.get("VO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if world > 1 and os.environ.get("VO_BENCH_BACKEND", "nccl") == "nccl" else "cpu"
    dev = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev)

    from visual_odometry_amd import synth
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Batch, Context

    H, W = 1080, 1920
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(H, W)
    ctx = Context(prm, device=dev)
    if args.matcher == "fp32":
        ctx.debug_set(2, 1)
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
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # result records of this rank's pairs; the gather of these records is the only collective of the job
    from visual_odometry_amd import shard

    outs = [batch.get(i) for i in range(args.batch)]
    rec_local = shard.pack_records(outs)
    rec_all = shard.gather_records(rec_local, args.batch * world, rank, world, device=coll_dev)

    # PCIe-inclusive rate (reported beside `value`, never as `value`): every step uploads both frames of every pair from
    # host memory and reads the result records back.  Two batches on two contexts (= two HIP streams) alternate, so the
    # host-side staging and the H2D copies of one step overlap the kernels of the previous one.
    h2d_steps = 6
    ctx_b = Context(prm, device=dev)
    batch_b = Batch(ctx_b, args.batch, H, W)
    duo = [batch, batch_b]

    def fill(bt):
        for i in range(args.batch):
            sc = scenes[i % len(scenes)]
            bt.set_pair(i, sc.img1, sc.img2, sc.K)

    fill(batch_b)
    batch_b.run()
    batch_b.sync()  # warm-up of the second context
    barrier()
    th = time.perf_counter()
    for k in range(h2d_steps):
        cur = duo[k % 2]
        fill(cur)
        cur.run()
        if k > 0:
            prev = duo[(k - 1) % 2]
            prev.sync()
            _ = [prev.get(i) for i in range(args.batch)]
    last = duo[(h2d_steps - 1) % 2]
    last.sync()
    _ = [last.get(i) for i in range(args.batch)]
    barrier()
    pcie_rate = args.batch * h2d_steps * world / (time.perf_counter() - th)
    # the same two batches, frames resident: the latency-bound kernels of one (RANSAC, sort, pose: ~2 ms of single wavefronts)
    # run under the VALU-bound kernels of the other.  Reported beside `value` (whose kernels run alone, so that the per-kernel
    # durations behind `roofline` mean something).
    for b2 in duo:
        b2.run()
    for b2 in duo:
        b2.sync()
    barrier()
    t2 = time.perf_counter()
    duo_steps = 4
    for _ in range(duo_steps):
        for b2 in duo:
            b2.run()
    for b2 in duo:
        b2.sync()
    barrier()
    duo_rate = 2 * args.batch * duo_steps * world / (time.perf_counter() - t2)
    batch_b.close()
    ctx_b.close()

    # Consecutive-frame mode (reported beside `value`, never as `value`): the reference re-detects prev_img in every loop
    # iteration, so the headline counts two detections per pair; an offline replay of consecutive frames detects each
    # frame once (vo_seq_*: batch + 1 frames, pair p = frames p, p + 1), the same matcher / RANSAC work per pair.
    from visual_odometry_amd.frontend import Sequence

    seq = Sequence(ctx, args.batch + 1, H, W)
    for f in range(args.batch + 1):
        sc = scenes[(f // 2) % len(scenes)]
        seq.set_frame(f, sc.img1 if f % 2 == 0 else sc.img2)  # alternate the two views: every consecutive pair has parallax
    for p in range(args.batch):
        seq.set_pair(p, p, p + 1, scenes[0].K)
    seq.run()
    seq.sync()
    barrier()
    ts = time.perf_counter()
    seq_steps = 4
    for _ in range(seq_steps):
        seq.run()
    seq.sync()
    barrier()
    seq_rate = args.batch * seq_steps * world / (time.perf_counter() - ts)
    seq.close()

    # per-stage device time of one more pass (HIP events on the library's stream)
    ctx.profile_enable(True)
    ctx.profile_reset()
    nprof = 3
    for _ in range(nprof):
        batch.run()
        batch.sync()  # one run at a time: no geometry-stream overlap inside the per-kernel durations
    prof = ctx.profile()
    ctx.profile_enable(False)
    import ctypes as C

    redo = C.c_int32(-1)  # queries of the last step whose matcher shortlist was not certified (exact scan instead)
    ctx._chk(ctx.lib.vo_debug_fetch(ctx._h, 4, C.byref(redo), C.c_size_t(4)))

    # Latency of the reference's three calls on ONE pair through the synchronous entry points, host buffers in and out - how
    # the ROS node calls them (reported beside `value`; rank 0 only)
    single = None
    if rank == 0:
        lat = {"opt_DetectFeatures": [], "optRelativePose": [], "triangPoints": []}
        sc0 = scenes[0]
        for it in range(13):
            t0 = time.perf_counter()
            g = ctx.opt_DetectFeatures(sc0.img1, sc0.img2)
            t1 = time.perf_counter()
            _st, R1, tv1, inl1, ok1 = ctx.optRelativePose(g, sc0.K)
            t2 = time.perf_counter()
            if ok1:
                ctx.triangPoints(inl1.Kpoints1, inl1.Kpoints2, R1, tv1, sc0.K)
            t3 = time.perf_counter()
            if it >= 3:
                lat["opt_DetectFeatures"].append(t1 - t0), lat["optRelativePose"].append(t2 - t1), lat["triangPoints"].append(t3 - t2)
        single = {k + "_ms": round(1e3 * float(np.median(v)), 3) for k, v in lat.items()}
        single["total_ms"] = round(sum(single.values()), 3)
        single["note"] = "one 1080p pair, synchronous C-ABI calls with pageable host buffers in and out (median of 10)"

    if rank == 0:
        assert rec_all.shape[0] == args.batch * world
        kp = float(rec_all[:, 1:3].mean())
        stage_ms = {k: v[0] / nprof for k, v in prof.items()}
        frames = 2 * args.batch
        S = layer_samples(H, W)
        I = 4 * (H + 1) * (W + 1)
        # share of the Hessian samples per octave, and of the keypoints / window samples per descriptor class
        S_oct = [5 * (H >> o) * (W >> o) for o in range(4)]
        kp1, _ = ctx.detect_describe(scenes[0].img1)
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
            "match_mfma": ("mfma", args.batch * 2 * 64 * float(rec_all[:, 1].mean()) * float(rec_all[:, 2].mean())),
        }
        for name, sh in cls_share.items():
            alg[name] = ("hbm", frames * (sh * (H * W + I) + 256 * kp * sh))

        # HBM-side traffic per launch: PMC counters cannot be read from inside this process; the figures come from the
        # committed rocprofv3 --pmc passes of this same command (profiles/) and apply to the default
        # workload only
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "r01_r_hbm_traffic.json")  # PMC passes of this command
        if args.batch == 64 and os.path.exists(tpath):
            prefix = {"describe_w64": "k_describe_tiled<64,", "describe_w128": "k_describe_tiled<128,",
                      "describe_w320": "k_describe_tiled<320,", "describe_w768": "k_describe_tiled<768,",
                      "hessian_oct0_23": "k_hessian_nms_fused<", "hessian_oct1": "k_hessian_nms_lds<", "integral": "k_integral<",
                      "match_mfma": "k_match_bf3"}
            tj = json.load(open(tpath))
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
                   "frac": round(achieved / peak, 5), "traffic": traffic.get(name), "launch_ms": round(dur_s * 1e3, 4),
                   "algorithmic_per_launch": work}
            if name in cls_samples:
                # What bounds these kernels is VALU issue: a window sample is 24 instructions = 86 issue cycles per wavefront
                # (10 fp64 / integer coordinate and address operations, 4 byte->float conversions, 4 packed multiplies and the
                # u8 pack at 4 cycles, 5 FP32 adds at 2; costs measured by tools/ubench/valu_rates.hip).  1024 SIMDs x 2.4 GHz.
                wsamples = cls_samples[name] / 64.0
                cyc = dur_s * VALU_CLOCK_HZ * N_SIMD / max(wsamples, 1.0)
                out["valu_issue"] = {"window_samples_per_launch": round(cls_samples[name]), "cycles_per_wavefront_sample": round(cyc, 1),
                                     "floor_cycles": SAMPLE_ISSUE_CYCLES, "frac_of_floor": round(SAMPLE_ISSUE_CYCLES / cyc, 4)}
            if bound == "mfma" and args.matcher != "fp32":
                out["note"] = ("executed bf16 MFMA flop (3.125 x the algorithmic 2*64*N1*N2) against the dense bf16 peak; "
                               f"algorithmic rate {round(work / dur_s / 1e12, 1)} TFLOP/s = "
                               f"{round(work / dur_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 2)} x the FP32 MFMA peak it used to run on")
            return out

        dom = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        roofline = roof(dom)
        # The descriptor and Hessian kernels are bound by VALU issue and the texture addresser's gather rate, not by HBM
        # (PMC evidence under profiles/): their HBM fraction is small by construction.
        roofline["note"] = "dominant kernel by device time; descriptor/Hessian kernels are VALU-issue / gather bound, see roofline_all and DESIGN.md"
        roofline_all = [roof(k) for k in sorted(alg, key=lambda k: -stage_ms.get(k, 0.0))]

        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU port is timed at N = 1 only (rank 0 is the only rank there)
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
            O.set_threads(1)
            t1 = time.perf_counter()
            O.pair(scenes[0].img1, scenes[0].img2, scenes[0].K, prm)
            one_thread = 1.0 / (time.perf_counter() - t1)
            O.set_threads(cores)
            cpu = {"value": round(n_cpu / cdt, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                   "value_1_thread": round(one_thread, 4),
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
            "dtype": "f32 (SURF, matcher distances; shortlist contraction on " + ("FP32 MFMA" if args.matcher == "fp32" else "split-bf16 MFMA")
                     + ") / f64 (RANSAC, pose, DLT)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1080p textured frame pairs, SURF ~8k kpts/frame, full "
                                   "detect x2 / BF match / E-RANSAC / recoverPose / triangulate",
                       "pairs_per_step_per_gpu": args.batch, "distinct_pairs_per_gpu": len(scenes),
                       "mean_keypoints_per_frame": round(kp, 1), "mean_matches": round(float(rec_all[:, 3].mean()), 1),
                       "mean_inliers": round(float(rec_all[:, 4].mean()), 1),
                       "pairs_with_pose": int(rec_all[:, 7].sum()), "pairs_total": int(len(rec_all)),
                       "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective"},
            "pcie_inclusive": {"value": round(pcie_rate, 2), "unit": "frame-pairs/s",
                               "note": "both frames of every pair uploaded from host memory every step (pinned staging, two alternating batches), result records read back"},
            "two_batches_in_flight": {"value": round(duo_rate, 2), "unit": "frame-pairs/s",
                                      "note": "two resident batches on two contexts (two HIP streams) alternating; same work per pair as `value`"},
            "consecutive_frames": {"value": round(seq_rate, 2), "unit": "frame-pairs/s",
                                   "note": "sequence batch: batch + 1 frames detected once each, pair p = (frame p, frame p + 1); not the headline workload"},
            "single_pair_latency": single,
            "roofline": roofline,
            "roofline_all": roofline_all,
            "cpu_baseline": cpu,
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "match_exact_scans_per_step": redo.value,
        }
        print(json.dumps(line))
    batch.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
