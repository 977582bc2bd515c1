"""GPU parity of the device-resident batch pipeline (vo_batch_*: detect x2 -> match -> E-RANSAC -> recoverPose ->
triangPoints with no host round trip) against the per-stage entry points and the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Batch, KpAsPoint2f_Match

pytestmark = pytest.mark.gpu


def _angle(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_batch_matches_stagewise_and_oracle(ctx):
    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 1001), (500, 3), (900, 4)]]
    b = Batch(ctx, len(scenes), 480, 640)
    for i, sc in enumerate(scenes):
        b.set_pair(i, sc.img1, sc.img2, sc.K)
    for rep in range(2):  # second run reuses every device buffer: nothing may leak between runs
        b.run()
        b.sync()
        for i, sc in enumerate(scenes):
            out = b.get(i)
            p1, p2, m, mask, X = b.get_points(i)
            rp1, rp2, rm = O.detect_match(sc.img1, sc.img2)
            assert out.status == 0 and out.n_match == len(rm)
            assert np.array_equal(p1, rp1) and np.array_equal(p2, rp2)
            assert np.array_equal(m["queryIdx"], rm["queryIdx"]) and np.array_equal(m["trainIdx"], rm["trainIdx"])
            # geometry of the batch == the single-call entry points on the same matches (bit for bit)
            st, E, smask, ninl, iters = ctx.essential_ransac(p1, p2, sc.K)
            assert st == 0 and np.array_equal(mask, smask) and out.n_inl == ninl and out.iters == iters
            assert np.array_equal(np.array(out.E).reshape(3, 3), E)
            st2, R, t, inl, ok = ctx.optRelativePose(KpAsPoint2f_Match(p1, p2, m), sc.K)
            assert ok == bool(out.success)
            if ok:
                assert np.array_equal(np.array(out.R).reshape(3, 3), R) and np.array_equal(np.array(out.t), t)
                Xs = ctx.triangPoints(inl.Kpoints1, inl.Kpoints2, R, t, sc.K)
                assert X.shape == Xs.shape and np.array_equal(X, Xs)
                Xr = O.triang_points(inl.Kpoints1, inl.Kpoints2, R, t, sc.K)
                assert X.shape == Xr.shape and np.allclose(X, Xr, rtol=1e-5, atol=1e-6)
            else:
                assert out.n_world == 0
    b.close()


def test_batch_1080p_pose_close_to_ground_truth_and_oracle():
    """BASELINE config 2 workload: 1080p, ~8k keypoints per frame, ROI scaled to the frame."""
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Context

    sc = synth.scene_c2(2002)
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(1080, 1920)
    c = Context(prm)
    b = Batch(c, 2, 1080, 1920)
    b.set_pair(0, sc.img1, sc.img2, sc.K)
    b.set_pair(1, sc.img2, sc.img1, sc.K)
    b.run()
    b.sync()
    out = b.get(0)
    ref = O.pair(sc.img1, sc.img2, sc.K, prm)
    assert out.status == 0
    assert 0.95 * 8192 <= out.n_kp1 <= 1.05 * 8192
    assert (out.n_kp1, out.n_kp2, out.n_match) == (ref.n_kp1, ref.n_kp2, ref.n_match)
    assert out.success == ref.success == 1
    R = np.array(out.R).reshape(3, 3)
    assert _angle(R, np.array(ref.R).reshape(3, 3)) < 0.1  # north star: pose vs the reference path
    assert _angle(R, sc.R) < 1.0
    assert abs(out.n_inl - ref.n_inl) <= 0.05 * ref.n_inl
    out1 = b.get(1)
    assert out1.status == 0 and out1.n_kp1 == out.n_kp2 and out1.n_kp2 == out.n_kp1
    b.close()
    c.close()


def test_sequence_batch_shares_frame_features(ctx):
    """vo_seq_*: three frames detected once each; pairs (0,1) and (1,2) give exactly what two independent pair-batch
    entries give, and re-pairing (the NOT_MOVING case) works."""
    from visual_odometry_amd.frontend import Batch, Sequence

    sc = synth.blob_scene(480, 640, synth.K_PARETE, 700, 1001)
    frames = [sc.img1, sc.img2, sc.img1]
    seq = Sequence(ctx, 3, 480, 640)
    for f, im in enumerate(frames):
        seq.set_frame(f, im)
    seq.set_pair(0, 0, 1, sc.K)
    seq.set_pair(1, 1, 2, sc.K)
    seq.run()
    seq.sync()
    ref = Batch(ctx, 2, 480, 640)
    ref.set_pair(0, sc.img1, sc.img2, sc.K)
    ref.set_pair(1, sc.img2, sc.img1, sc.K)
    ref.run()
    ref.sync()
    for p in range(2):
        a, b = seq.get(p), ref.get(p)
        assert (a.status, a.n_kp1, a.n_kp2, a.n_match, a.n_inl, a.n_good, a.n_world, a.success, a.iters) == (
            b.status, b.n_kp1, b.n_kp2, b.n_match, b.n_inl, b.n_good, b.n_world, b.success, b.iters)
        assert list(a.E) == list(b.E) and list(a.R) == list(b.R) and list(a.t) == list(b.t)
        pa, pb = seq.get_points(p), ref.get_points(p)
        for x, y in zip(pa, pb):
            assert np.array_equal(x, y)
    assert seq.get(0).n_match > 50
    # re-pair: frame 0 against frame 2 (identical images) -> every kept match has distance 0 and zero displacement
    seq.set_pair(1, 0, 2, sc.K)
    seq.run()
    seq.sync()
    p1, p2, m, _, _ = seq.get_points(1)
    assert len(m) > 100 and np.array_equal(p1, p2) and np.all(m["distance"] == 0)
    seq.close()
    ref.close()


def test_replay_follows_the_node_loop(ctx):
    """C3-style sequence through visual_odometry_amd.replay: gate sequence, prev_img bookkeeping across NOT_MOVING and
    FAIL_DETECTION frames, pose carry-over, scale factor and the pose chain (src/visual_Odometry.cpp:268-470)."""
    from visual_odometry_amd import chain, replay

    step = 0.45
    poses = [((0, 0, 0), (0, 0, 0)), ((0.5, -0.3, 1.0), (step, 0.02, 0.0)), ((1.0, 0.2, 2.0), (2 * step, 0.05, 0.02))]
    poses += [poses[2], poses[2]]  # frames 3, 4: the camera stands still
    poses += [((0.8, 0.5, 3.0), (3 * step, 0.0, 0.03)), ((0, 0, 0), (4 * step, 0, 0)), ((0.3, 0.2, 3.5), (5 * step, 0.03, 0.0)),
              ((0.1, 0.6, 4.0), (6 * step, 0.06, 0.01))]
    sc = synth.sequence_scene(480, 640, synth.K_PARETE, 900, 3003, poses, blank=(6,))
    n = len(sc.frames)
    stamps = np.arange(n) * 0.5
    alt = np.full(n, 7.0)
    recs = replay.replay(ctx, sc.frames, stamps, alt, sc.K, np.eye(3), np.zeros(3), chunk=4)
    assert [r["frame"] for r in recs] == list(range(1, n))
    fails = [r["fail"] for r in recs]
    S, NM, FD = chain.SUCCESS, chain.NOT_MOVING, chain.FAIL_DETECTION
    assert fails == [S, S, NM, NM, S, FD, FD, S], fails
    assert [r["prev"] for r in recs] == [0, 1, 2, 2, 2, 5, 6, 7]
    for r in recs:
        if r["fail"] != S:
            assert "R" not in r
            continue
        Rgt, tgt, base = sc.relative(r["prev"], r["frame"])
        assert _angle(r["R"], Rgt) < 0.5, (r["frame"], _angle(r["R"], Rgt))
        assert np.degrees(np.arccos(np.clip(r["t"] @ tgt, -1, 1))) < 10.0  # ~150 matches, 2 px RANSAC threshold
        assert r["world_pointsW"].shape[0] == 3 and r["world_pointsW"].shape[1] > 20
        # scaleFactor = altitude / mean depth in |t| = 1 units; the scene is 5-9 m deep and the altitude 7 m, so SF must
        # come out near the true baseline
        assert 0.5 * base < r["SF"] < 2.0 * base, (r["SF"], base)
    # the chain: position advances along +x of the start frame by roughly SF * |t| per successful step, and the
    # absolute orientation is the product of the relative ones
    last = recs[-1]
    Rtot = np.eye(3)
    for r in recs:
        if r["fail"] == S:
            Rtot = Rtot @ r["R"].T
    np.testing.assert_allclose(last["orientation"], Rtot, atol=1e-12)
    assert last["estimate_pos"][0] > 1.0


def test_batch_of_eight_takes_the_unsplit_matcher_path(ctx):
    """Eight pairs or more: one train split per workgroup and one pair per XCD (k_match_bf3 with ksplit = 1); fewer pairs
    split the train range four ways.  Both must give the oracle's matches, with ragged keypoint counts."""
    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in
              [(700, 11), (150, 12), (900, 13), (40, 14), (500, 15), (650, 16), (300, 17), (800, 18)]]
    b = Batch(ctx, len(scenes), 480, 640)
    for i, sc in enumerate(scenes):
        b.set_pair(i, sc.img1, sc.img2, sc.K)
    b.run()
    b.sync()
    for i, sc in enumerate(scenes):
        out = b.get(i)
        p1, p2, m, mask, X = b.get_points(i)
        single = ctx.opt_DetectFeatures(sc.img1, sc.img2)  # one pair: the four-way split
        assert np.array_equal(p1, single.Kpoints1) and np.array_equal(p2, single.Kpoints2)
        assert np.array_equal(m["trainIdx"], single.match["trainIdx"]) and np.array_equal(m["distance"], single.match["distance"])
        if i in (0, 3):
            rp1, rp2, rm = O.detect_match(sc.img1, sc.img2)
            assert out.n_match == len(rm) and np.array_equal(p1, rp1) and np.array_equal(p2, rp2)
            assert np.array_equal(m["distance"].view(np.uint32), rm["distance"].view(np.uint32))
    b.close()


def test_bag_to_poses_end_to_end(ctx, tmp_path):
    """N4 + N2 + N1 together: a rosbag with CompressedImage (PNG) frames and altitudes -> bagio -> get_image on the device
    -> sequence batches -> the node loop.  Must equal the replay of the same frames preprocessed by the CPU restatement."""
    import io

    from PIL import Image

    from visual_odometry_amd import bagio, chain, replay
    from visual_odometry_amd import prep as P

    step = 0.45
    poses = [((0, 0, 0), (0, 0, 0)), ((0.5, -0.3, 1.0), (step, 0.02, 0.0)), ((1.0, 0.2, 2.0), (2 * step, 0.05, 0.02)),
             ((1.0, 0.2, 2.0), (2 * step, 0.05, 0.02)), ((0.8, 0.5, 3.0), (3 * step, 0.0, 0.03))]
    sc = synth.sequence_scene(480, 640, synth.K_PARETE, 900, 4004, poses)
    # the "camera": 960 x 1280 colour frames (the gray scene on all three channels, pixel-doubled)
    big = [np.repeat(np.repeat(np.stack([f, f, f], -1), 2, axis=0), 2, axis=1) for f in sc.frames]
    alt_def = f"Header header\nfloat32 altitude\nbool validity\n{'=' * 80}\nMSG: std_msgs/Header\n{bagio.HEADER_DEF}"
    path = str(tmp_path / "run.bag")
    w = bagio.BagWriter(path, "bz2", chunk_bytes=1 << 19)
    w.add_connection("/pylon_camera/image_raw/compressed", "sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    w.add_connection("/drivers/altitude", "marta_msgs/Altitude", alt_def)
    ci = bagio.MsgCodec("sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    al = bagio.MsgCodec("marta_msgs/Altitude", alt_def)
    for i, f in enumerate(big):
        b = io.BytesIO()
        Image.fromarray(f[..., ::-1]).save(b, "PNG")
        w.write("/drivers/altitude", 50 + 0.5 * i - 0.02, al.encode({"header": {"seq": i, "stamp": 50 + 0.5 * i - 0.02, "frame_id": ""}, "altitude": 7.0, "validity": True}))
        w.write("/pylon_camera/image_raw/compressed", 50 + 0.5 * i, ci.encode({"header": {"seq": i, "stamp": 50 + 0.5 * i, "frame_id": "cam"}, "format": "png", "data": b.getvalue()}))
    w.close()

    frames, stamps, alts = bagio.read_sequence(path)
    assert len(frames) == 5 and frames[0].shape == (960, 1280, 3) and np.allclose(alts, 7.0)
    K_full = np.diag([2.0, 2.0, 1.0]) @ sc.K
    K = P.updateCameraMatrix((960, 1280), 640, 480, K_full, ctx.lib)
    pr = P.Prep(ctx, (960, 1280), 640, 480, K, [0.01, -0.005, 0.0002, -0.0002])
    try:
        recs = replay.replay(ctx, frames, stamps, alts, K, np.eye(3), np.zeros(3), chunk=3, prep=pr)
        gray = [O.prep(pr.params, f)["gray"] for f in frames]
        ref = replay.replay(ctx, gray, stamps, alts, K, np.eye(3), np.zeros(3), chunk=3)
    finally:
        pr.close()
    assert [r["fail"] for r in recs] == [chain.SUCCESS, chain.SUCCESS, chain.NOT_MOVING, chain.SUCCESS]
    assert [r["prev"] for r in recs] == [0, 1, 2, 2]
    for a, b in zip(recs, ref):
        assert a["fail"] == b["fail"] and a["prev"] == b["prev"] and a["n_matches"] == b["n_matches"]
        if a["fail"] == chain.SUCCESS:
            for k in ("R", "t", "estimate_pos", "estimate_rpy", "twist_linear", "orientation", "world_pointsW"):
                assert np.array_equal(a[k], b[k]), k
            assert a["SF"] == b["SF"]
    Rgt, tgt, base = sc.relative(0, 1)
    assert _angle(recs[0]["R"], Rgt) < 1.0


def _adaptor_exe():
    import os

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "cv_adaptor_check")
    if not os.path.exists(exe):
        pytest.fail("examples/cv_adaptor_check missing: run __graft_entry__.build()")
    return exe


def test_cv_adaptor_runs_the_node_call_sequence():
    """N3: the switch headers + cv::Mat adaptor (mock cv types) run get_image -> opt_DetectFeatures -> optRelativePose ->
    triangPoints as src/visual_Odometry.cpp:228-384 calls them.  The binary is built by __graft_entry__.build()."""
    import subprocess

    out = subprocess.run([_adaptor_exe(), "run"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "success 1" in out.stdout, out.stdout
    # the reference's parameter globals (camera_utility.hpp:33-63) drive the library: edited ROI / minHessian take effect
    assert "globals: roi ok" in out.stdout and "minHessian ok" in out.stdout and "restored ok" in out.stdout, out.stdout


def test_cv_adaptor_outputs_equal_the_oracle(tmp_path):
    """N3 against the oracle, not just `success 1`: two colour frames through the node's call sequence on the adaptor (the
    reference's four signatures on cv:: types) must give the oracle's preprocessed frames, matches, inliers, pose and world
    points - get_image = oracle_prep, opt_DetectFeatures = O.detect_match (bit for bit), optRelativePose = O.relative_pose,
    triangPoints = O.triang_points."""
    import subprocess

    from visual_odometry_amd._abi import VoPrepParams

    sc = synth.blob_scene(480, 640, synth.K_PARETE, 900, 1001, noise_amp=0.4)  # little noise: CLAHE (clip 16) amplifies it
    rng = np.random.default_rng(5)
    frames = []
    for g in (sc.img1, sc.img2):  # colour frames whose channels differ, so that the BGR weighting of get_image matters
        bgr = np.stack([g, np.clip(g.astype(np.int16) + rng.integers(-1, 2, g.shape), 0, 255).astype(np.uint8),
                        np.clip(g.astype(np.int16) // 2 + 40, 0, 255).astype(np.uint8)], axis=2)
        frames.append(np.ascontiguousarray(bgr))
    dist = np.array([0.02, -0.01, 0.001, -0.0005])
    K = synth.K_PARETE
    files = []
    for i, f in enumerate(frames):
        path = tmp_path / f"f{i}.bgr"
        f.tofile(path)
        files.append(str(path))
    out_txt = str(tmp_path / "out.txt")
    args = [_adaptor_exe(), "files", files[0], files[1], "480", "640", "640", "480", repr(float(K[0, 0])), repr(float(K[1, 1])), repr(float(K[0, 2])), repr(float(K[1, 2]))]
    args += [repr(float(d)) for d in dist] + [out_txt]
    run = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = open(out_txt).read().split("\n")
    hx = float.fromhex
    # the oracle on the same frames
    pp = VoPrepParams()
    pp.src_rows, pp.src_cols, pp.dst_rows, pp.dst_cols = 480, 640, 480, 640
    pp.K[:] = list(K.ravel())
    pp.dist[:] = list(dist)
    pp.clip_limit, pp.tiles_x, pp.tiles_y = 16.0, 8, 8
    g0, g1 = O.prep(pp, frames[0])["gray"], O.prep(pp, frames[1])["gray"]

    def fnv(a):
        h = 1469598103934665603
        for v in a.ravel().tolist():
            h = ((h ^ v) * 1099511628211) & (2**64 - 1)
        return h

    assert lines[0] == "gray 480 640"
    assert lines[1] == f"hash {fnv(g0)} {fnv(g1)}"  # get_image
    n = int(lines[2].split()[1])
    rows = [ln.split() for ln in lines[3 : 3 + n]]
    p1 = np.array([[hx(r[0]), hx(r[1])] for r in rows], np.float32)
    p2 = np.array([[hx(r[2]), hx(r[3])] for r in rows], np.float32)
    rp1, rp2, rm = O.detect_match(g0, g1)
    assert n == len(rm) and n > 30
    assert np.array_equal(p1, rp1) and np.array_equal(p2, rp2)
    assert [int(r[4]) for r in rows] == rm["queryIdx"].tolist() and [int(r[5]) for r in rows] == rm["trainIdx"].tolist()
    assert np.array_equal(np.array([hx(r[6]) for r in rows], np.float32), rm["distance"])
    head = lines[3 + n].split()
    ok, ni = int(head[1]), int(head[3])
    st, R, t, i1, i2, rok = O.relative_pose(rp1, rp2, K)
    assert ok == int(rok) == 1 and ni == len(i1)
    irows = [ln.split() for ln in lines[4 + n : 4 + n + ni]]
    assert np.array_equal(np.array([[hx(r[0]), hx(r[1])] for r in irows], np.float32), i1)
    assert np.array_equal(np.array([[hx(r[2]), hx(r[3])] for r in irows], np.float32), i2)
    Rg = np.array([hx(v) for v in lines[4 + n + ni].split()[1:]]).reshape(3, 3)
    tg = np.array([hx(v) for v in lines[5 + n + ni].split()[1:]])
    assert np.abs(Rg - R).max() < 1e-9 and np.abs(tg - t).max() < 1e-9
    nw = int(lines[6 + n + ni].split()[1])
    Xg = np.array([[hx(v) for v in ln.split()] for ln in lines[7 + n + ni : 7 + n + ni + nw]]).T
    Xr = O.triang_points(i1, i2, R, t, K)
    assert Xg.shape == Xr.shape and nw > 5 and np.allclose(Xg, Xr, rtol=1e-5, atol=1e-6)


def test_back_to_back_runs_with_new_inputs_in_between(ctx):
    """The geometry of run i works on the geometry stream while the main stream already detects run i + 1: new frames and
    new K written between two runs WITHOUT a sync in between must not disturb either run."""
    A = [synth.blob_scene(480, 640, synth.K_PARETE, 700, 21), synth.blob_scene(480, 640, synth.K_PARETE, 500, 22)]
    Bs = [synth.blob_scene(480, 640, synth.K_PARETE * np.array([[1.1], [1.1], [1.0]]), 600, 23), synth.blob_scene(480, 640, synth.K_PARETE, 800, 24)]

    def fields(b, i):
        o = b.get(i)
        return (o.status, o.n_kp1, o.n_kp2, o.n_match, o.n_inl, o.n_good, o.n_world, o.success, o.iters, tuple(o.E), tuple(o.R), tuple(o.t))

    ref = {}
    for name, scenes in (("A", A), ("B", Bs)):
        r = Batch(ctx, 2, 480, 640)
        for i, sc in enumerate(scenes):
            r.set_pair(i, sc.img1, sc.img2, sc.K)
        r.run()
        r.sync()
        ref[name] = [(fields(r, i), [x.copy() for x in r.get_points(i)]) for i in range(2)]
        r.close()
    b = Batch(ctx, 2, 480, 640)
    for i, sc in enumerate(A):
        b.set_pair(i, sc.img1, sc.img2, sc.K)
    b.run()
    for i, sc in enumerate(Bs):  # no sync: run A's geometry may still be in flight
        b.set_pair(i, sc.img1, sc.img2, sc.K)
    b.run()
    b.run()  # and once more on the same inputs
    b.sync()
    for i in range(2):
        assert fields(b, i) == ref["B"][i][0]
        for x, y in zip(b.get_points(i), ref["B"][i][1]):
            assert np.array_equal(x, y)
    for i, sc in enumerate(A):
        b.set_pair(i, sc.img1, sc.img2, sc.K)
    b.run()
    assert fields(b, 0) == ref["A"][0][0] and fields(b, 1) == ref["A"][1][0]  # get without an explicit sync
    b.close()


def _rec(o):
    return (o.status, o.n_kp1, o.n_kp2, o.n_match, o.n_inl, o.n_good, o.n_world, o.success, o.iters, tuple(o.E), tuple(o.R), tuple(o.t))


def test_pipelined_runs_with_double_buffered_frames(ctx):
    """The frame slots: while run k executes, the frames of run k + 1 are staged and uploaded on the copy stream and run
    k + 1 is enqueued; run k's records are then read with age 1.  Every run must equal a fresh synchronous batch on the
    same inputs, bit for bit; frames that are not replaced carry over to the other slot."""
    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 1001), (500, 3), (900, 4), (650, 8)]]
    P = 3

    def fresh(pairs):
        bb = Batch(ctx, P, 480, 640)
        for i, (a, c) in enumerate(pairs):
            bb.set_pair(i, a, c, synth.K_PARETE)
        bb.run()
        bb.sync()
        out = [_rec(bb.get(i)) for i in range(P)]
        bb.close()
        return out

    steps = []
    for k in range(5):  # step k: pairs rotate through the scenes, some swapped
        steps.append([(scenes[(k + i) % 4].img1, scenes[(k + i) % 4].img2) if (k + i) % 2 == 0 else
                      (scenes[(k + i) % 4].img2, scenes[(k + i) % 4].img1) for i in range(P)])
    want = [fresh(s) for s in steps]
    b = Batch(ctx, P, 480, 640)
    got = [None] * len(steps)
    for k, s in enumerate(steps):
        for i, (a, c) in enumerate(s):
            b.set_pair(i, a, c, synth.K_PARETE)
        b.run()  # enqueued behind run k - 1, which may still be executing
        if k > 0:
            got[k - 1] = [_rec(b.get(i, age=1)) for i in range(P)]
    got[-1] = [_rec(b.get(i)) for i in range(P)]
    assert got == want
    # partial restaging: only pair 1 replaced, pairs 0 and 2 carry over from the other slot
    b.set_pair(1, scenes[3].img1, scenes[3].img2, synth.K_PARETE)
    b.run()
    b.sync()
    mixed = [steps[-1][0], (scenes[3].img1, scenes[3].img2), steps[-1][2]]
    assert [_rec(b.get(i)) for i in range(P)] == fresh(mixed)
    # nothing restaged: same slot again, same records
    b.run()
    assert [_rec(b.get(i)) for i in range(P)] == fresh(mixed)
    b.close()


def test_pair_without_camera_matrix_fails_alone(ctx):
    """A pair that never got a K reports VO_ERR_INVALID_ARG in ITS record; the run and the other pairs are unaffected
    (sequence batches are created with K = 0 for every pair, partly configured ones used to fail as a whole)."""
    from visual_odometry_amd.frontend import Sequence

    sc = synth.blob_scene(480, 640, synth.K_PARETE, 300, 5)
    ref = Batch(ctx, 1, 480, 640)
    ref.set_pair(0, sc.img1, sc.img2, sc.K)
    ref.run()
    want = _rec(ref.get(0))
    ref.close()
    b = Batch(ctx, 2, 480, 640)
    b.set_pair(0, sc.img1, sc.img2, sc.K)  # pair 1 never gets a K
    b.run()
    assert _rec(b.get(0)) == want
    o = b.get(1)
    assert o.status == -1 and o.success == 0
    b.close()
    seq = Sequence(ctx, 3, 480, 640)
    for f, im in enumerate([sc.img1, sc.img2, sc.img1]):
        seq.set_frame(f, im)
    seq.set_pair(0, 0, 1, sc.K)  # pair 1 = (1, 2) keeps the default pairing and no K
    seq.run()
    assert _rec(seq.get(0)) == want
    assert seq.get(1).status == -1 and seq.get(1).success == 0
    seq.close()


def test_partial_restaging_of_a_batch_filled_through_prep(ctx):
    """Carry-over out of a frame slot that vo_prep_to_batch filled: those frames were written by kernels on the MAIN stream,
    the carry-over copy runs on the COPY stream.  Run k (all frames through prep), then - without any host wait - one pair
    re-staged from the host and run k + 1: the pairs that carried over must equal a fresh batch (the copy stream
    used to be unordered against the prep kernels)."""
    from visual_odometry_amd import prep as P

    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 31), (500, 32), (900, 33)]]
    big = lambda g: np.ascontiguousarray(np.repeat(np.repeat(np.stack([g, g, g], -1), 2, axis=0), 2, axis=1))
    K = P.updateCameraMatrix((960, 1280), 640, 480, np.diag([2.0, 2.0, 1.0]) @ synth.K_PARETE, ctx.lib)
    pr = P.Prep(ctx, (960, 1280), 640, 480, K, [0.01, -0.005, 0.0002, -0.0002])
    try:
        gray = [(pr.get_image(big(sc.img1)), pr.get_image(big(sc.img2))) for sc in scenes]
        n = 3
        b = Batch(ctx, n, 480, 640)
        for rep in range(3):
            for i, sc in enumerate(scenes):
                pr.to_batch(big(sc.img1), b, 2 * i)
                pr.to_batch(big(sc.img2), b, 2 * i + 1)
                b.set_k(i, K)
            b.run()
            # no sync: pair 1 comes from the host, pairs 0 and 2 carry over from the slot the prep kernels are still writing
            b.set_pair(1, gray[0][1], gray[0][0], K)
            b.run()
            got = [_rec(b.get(i)) for i in range(n)]
            fresh = Batch(ctx, n, 480, 640)
            for i, (a, c) in enumerate([gray[0], (gray[0][1], gray[0][0]), gray[2]]):
                fresh.set_pair(i, a, c, K)
            fresh.run()
            want = [_rec(fresh.get(i)) for i in range(n)]
            fresh.close()
            assert got == want, rep
        b.close()
    finally:
        pr.close()


def test_pipelined_1080p_steps_equal_a_fresh_batch():
    """The overlap the headline number rests on, at the size where it really happens: 16 pairs of 1080p / 8 k keypoints,
    24 pipelined steps with restaging every step (upload on the copy stream beside the kernels, the integral pass of step
    k + 1 on the auxiliary stream under step k's descriptor kernels - vo_detect.hip, enqueue_detect), read with age 1.
    Every step's records must equal those of a fresh synchronous batch on the same pairs; then the same for a pair of
    alternating 1080p sequence batches."""
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Context, Sequence

    scs = [synth.scene_c2(2002 + i) for i in range(2)]
    prm = default_params()
    prm.roi_x_lo, prm.roi_y_lo, prm.roi_x_hi, prm.roi_y_hi = synth.roi_for(1080, 1920)
    c = Context(prm)
    P = 16
    variants = [(scs[0].img1, scs[0].img2), (scs[1].img1, scs[1].img2), (scs[0].img2, scs[0].img1), (scs[1].img2, scs[1].img1)]
    # two different step contents, alternating, so that a step which read stale frames cannot pass
    layouts = [[variants[(i + s) % 4] for i in range(P)] for s in range(2)]
    want = []
    for lay in layouts:
        f = Batch(c, P, 1080, 1920)
        for i, (a, d) in enumerate(lay):
            f.set_pair(i, a, d, scs[0].K)
        f.run()
        f.sync()
        want.append([_rec(o) for o in f.get_records()])
        f.close()
    assert want[0] != want[1] and want[0][0][3] > 500 and want[0][0][7] == 1
    b = Batch(c, P, 1080, 1920)

    def fill(k):
        for i, (a, d) in enumerate(layouts[k & 1]):
            b.set_pair(i, a, d, scs[0].K)

    steps = 24
    fill(0)
    bad = []
    for k in range(steps):
        b.run()
        fill(k + 1)  # staged while run k executes
        if k:
            if [_rec(o) for o in b.get_records(age=1)] != want[(k - 1) & 1]:
                bad.append(k - 1)
    if [_rec(o) for o in b.get_records()] != want[(steps - 1) & 1]:
        bad.append(steps - 1)
    b.close()
    assert not bad, bad
    # sequence batches (every frame detected once), two batches alternating as the replay uses them
    frames = [scs[0].img1, scs[0].img2, scs[1].img1, scs[1].img2, scs[0].img2, scs[0].img1, scs[1].img2, scs[1].img1, scs[0].img1]
    seqs = [Sequence(c, 9, 1080, 1920) for _ in range(2)]
    first, bad = None, []
    for k in range(8):
        s = seqs[k & 1]
        for f in range(9):
            s.set_frame(f, frames[f])
        for p in range(8):
            s.set_pair(p, p, p + 1, scs[0].K)
        s.run()
        if k:
            got = [_rec(o) for o in seqs[(k - 1) & 1].get_records()]
            first = first or got
            if got != first:
                bad.append(k - 1)
    for s in seqs:
        s.close()
    c.close()
    assert not bad and first[0] == want[0][0], bad


def test_failed_pair_reports_identity_pose_not_the_previous_runs(ctx):
    """Every run starts every pair from R = I, t = 0: a pair that fails after a successful run must not report that run's pose."""
    sc = synth.blob_scene(480, 640, synth.K_PARETE, 700, 1001)
    flat = np.full((480, 640), 128, np.uint8)
    b = Batch(ctx, 1, 480, 640)
    b.set_pair(0, sc.img1, sc.img2, sc.K)
    b.run()
    o = b.get(0)
    assert o.success == 1 and not np.allclose(np.array(o.R), np.eye(3).ravel())
    b.set_pair(0, flat, flat, sc.K)
    b.run()
    o = b.get(0)
    assert o.success == 0 and o.n_match == 0
    assert np.array_equal(np.array(o.R), np.eye(3).ravel()) and np.array_equal(np.array(o.t), np.zeros(3))
    b.close()


def test_two_seeds_on_one_device_do_not_share_a_sample_stream():
    """The cv::RNG tables belong to the context and are keyed by seed: a batch whose geometry is still in flight keeps its
    stream when a single call (same context) or a second context asks for another seed."""
    from visual_odometry_amd._abi import default_params
    from visual_odometry_amd.frontend import Context

    sc = synth.blob_scene(480, 640, synth.K_PARETE, 700, 1001)
    p1, p2, _, _, _ = synth.correspondences(1500, synth.K_PARETE, 9, inlier_frac=0.5, noise_px=0.5, h=480, w=640)
    prm_a, prm_b = default_params(), default_params()
    prm_b.ransac_seed = 777
    ca, cb = Context(prm_a), Context(prm_b)
    ref_a = ca.essential_ransac(p1, p2, synth.K_PARETE, seed=2**64 - 1)
    ref_b = ca.essential_ransac(p1, p2, synth.K_PARETE, seed=777)
    assert ref_a[4] != ref_b[4] or not np.array_equal(ref_a[2], ref_b[2])  # the two seeds do give different runs
    ba, bb = Batch(ca, 4, 480, 640), Batch(cb, 4, 480, 640)
    for i in range(4):
        ba.set_pair(i, sc.img1, sc.img2, sc.K)
        bb.set_pair(i, sc.img1, sc.img2, sc.K)
    ba.run()
    ba.sync()
    bb.run()
    bb.sync()
    want_a, want_b = [_rec(ba.get(i)) for i in range(4)], [_rec(bb.get(i)) for i in range(4)]
    assert want_a != want_b
    for rep in range(3):
        ba.run()
        bb.run()  # second context, other seed, same device, no sync in between
        got_single = ca.essential_ransac(p1, p2, synth.K_PARETE, seed=777)  # other seed on ca while its batch is in flight
        assert got_single[4] == ref_b[4] and np.array_equal(got_single[2], ref_b[2]) and np.array_equal(got_single[1], ref_b[1])
        assert [_rec(ba.get(i)) for i in range(4)] == want_a
        assert [_rec(bb.get(i)) for i in range(4)] == want_b
    # more seeds than a context keeps tables for: the oldest is recycled after a drain
    for s in range(1, 12):
        ca.essential_ransac(p1, p2, synth.K_PARETE, seed=s)
    again = ca.essential_ransac(p1, p2, synth.K_PARETE, seed=2**64 - 1)
    assert again[4] == ref_a[4] and np.array_equal(again[2], ref_a[2])
    ba.close(), bb.close(), ca.close(), cb.close()


def test_c3_sequence_through_the_sharded_replay(ctx):
    """BASELINE configs[2] on the GPU: a rapallo-like 640 x 505 run (K from config/rapallo_intrinsic.yaml through
    updateCameraMatrix, 18 stationary frames) through replay_sharded - frame block, pair summaries, host loop, gap-pair
    wave - against the on-demand replay (identical records), the CPU oracle on sampled pairs, and the ground truth."""
    from visual_odometry_amd import chain, replay

    n = 72
    sc = synth.sequence_c3(n, seed=3003, stationary=((20, 10), (50, 8)))
    assert sc.frames[0].shape == (505, 640) and abs(sc.K[0, 0] - 2305.660253962050 * 640 / 2560) < 1e-9
    stamps = 10.0 + 0.1 * np.arange(n)
    alt = np.full(n, 7.0)
    stats = {}
    recs = replay.replay_sharded(ctx, sc.frames, stamps, alt, sc.K, np.eye(3), np.zeros(3), 0, 1, chunk=32, stats=stats)
    base = replay.replay(ctx, sc.frames, stamps, alt, sc.K, np.eye(3), np.zeros(3), chunk=24)
    assert len(recs) == len(base) == n - 1
    for a, b in zip(recs, base):
        assert (a["frame"], a["prev"], a["n_matches"], a["fail"]) == (b["frame"], b["prev"], b["n_matches"], b["fail"])
        for key in ("R", "t", "SF", "estimate_pos", "estimate_rpy", "twist_linear", "twist_angular", "world_pointsW"):
            assert (key in a) == (key in b)
            if key in a:
                assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), (a["frame"], key)
    fails = [r["fail"] for r in recs]
    still = {r["frame"] for r in recs if r["fail"] == chain.NOT_MOVING}
    # the 18 stationary frames, plus the frames of a tight turn of the path (37 -> 38, 39: under 20 px against frame 37)
    assert set(range(20, 30)) | set(range(50, 58)) <= still and len(still) <= 22, sorted(still)
    assert stats["waves"] >= 2 and stats["gap_pairs"] >= 2
    # ~30 px of sideways motion over a seabed with +-17 % of relief is a weak geometry: some pairs fail the cheirality gate
    # (FAIL_RECOVER, the node then keeps the previous R, t) - on the GPU exactly where the loop's branch was taken
    assert all(f in (chain.SUCCESS, chain.NOT_MOVING, chain.FAIL_RECOVER) for f in fails), fails
    moving = [r for r in recs if r["fail"] != chain.NOT_MOVING]
    assert sum(r["fail"] == chain.SUCCESS for r in moving) >= 0.6 * len(moving)
    # after a stationary stretch the pair is (last moving frame, first frame that moved again)
    assert [r["prev"] for r in recs if r["frame"] in (30, 58)] == [19, 49]
    # sampled pairs against the CPU oracle: detect / match exactly; same verdict and pose within the north star's 0.1 degrees
    # wherever both sides found the same consensus set
    sample = [recs[i] for i in (0, 1, 5, 12, 29, 31, 40, 48, 57, 60, 66, 70)]
    close = 0
    for r in sample:
        j, k = r["prev"], r["frame"]
        ref = O.pair(sc.frames[j], sc.frames[k], sc.K)
        assert r["n_matches"] == ref.n_match, (j, k)
        if r["fail"] == chain.NOT_MOVING:
            continue
        assert (r["fail"] == chain.SUCCESS) == bool(ref.success), (j, k, r["fail"], ref.n_inl, ref.n_good)
        if not ref.success:
            continue
        if r["n_inliers"] == ref.n_inl:
            close += 1
            assert _angle(r["R"], np.array(ref.R).reshape(3, 3)) < 0.1, (j, k)
            assert np.degrees(np.arccos(np.clip(r["t"] @ np.array(ref.t), -1, 1))) < 0.5, (j, k)
        Rgt, tgt, bl = sc.relative(j, k)
        assert _angle(r["R"], Rgt) < 3.0 and np.degrees(np.arccos(np.clip(r["t"] @ tgt, -1, 1))) < 30.0, (j, k)  # weak geometry
        assert 0.5 * bl < r["SF"] < 2.0 * bl
    assert close >= 4, close
    # the chain ends near the true camera position (scale from the altitude / mean depth: a few per cent per step)
    c_true = -sc.Rw[-1].T @ sc.tw[-1]
    assert np.linalg.norm(recs[-1]["estimate_pos"] - c_true) < 0.5 * np.linalg.norm(c_true) + 1.0


@pytest.mark.gpu
def test_bulk_download_equals_the_per_pair_reads(ctx):
    """vo_batch_get_run_all / vo_batch_get_points_all against vo_batch_get / vo_batch_get_points pair by pair (pairs with
    different match counts, one of them empty: a black frame)."""
    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 1001), (300, 5), (900, 6)]]
    b = Batch(ctx, 4, 480, 640)
    for i, sc in enumerate(scenes):
        b.set_pair(i, sc.img1, sc.img2, synth.K_PARETE)
    b.set_pair(3, np.zeros((480, 640), np.uint8), scenes[0].img2, synth.K_PARETE)
    b.run()
    outs, pts = b.get_all()
    assert len({o.n_match for o in outs}) >= 3 and outs[3].n_match == 0
    for i in range(4):
        assert _rec(outs[i]) == _rec(b.get(i))
        one = b.get_points(i, cap=max(outs[i].n_match, 1))
        for got, want in zip(pts[i], one):
            assert got.shape == want.shape and np.array_equal(got, want)
    # rows too short for the fullest pair: refused as a whole, nothing half-written is reported as success
    import ctypes as C

    small = np.zeros((4, 1, 2), np.float32)
    st = ctx.lib.vo_batch_get_points_all(b._h, small.ctypes.data_as(C.POINTER(C.c_float)), None, None, None, 1, None, 0)
    assert st == -4  # VO_ERR_CAPACITY
    b.close()


def test_batch_hypothesis_chunks_with_hard_pairs(ctx):
    """Batches of eight pairs or more open RANSAC with a chunk of 16 hypotheses (then 64, 256, ...), a single pair with 64: the
    same hypotheses in the same order either way, so a batch must reproduce the single-call result bit for bit also when
    pairs need many iterations - here four pairs whose second frame belongs to ANOTHER scene (few accidental matches, no
    consensus: the bound never drops) beside four ordinary ones."""
    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 31), (600, 32), (800, 33), (500, 34)]]
    pairs = [(sc.img1, sc.img2) for sc in scenes] + [(scenes[i].img1, scenes[(i + 1) % 4].img2) for i in range(4)]
    b = Batch(ctx, 8, 480, 640)
    for i, (a, c) in enumerate(pairs):
        b.set_pair(i, a, c, synth.K_PARETE)
    b.run()
    outs, pts = b.get_all()
    iters = [o.iters for o in outs]
    assert min(iters[:4]) < 64 and max(iters[4:]) > 64, iters  # both sides of the chunk bounds are exercised
    for i, o in enumerate(outs):
        p1, p2, m, mask, X = pts[i]
        if len(p1) < 5:
            continue
        st, E, smask, ninl, it = ctx.essential_ransac(p1, p2, synth.K_PARETE)
        assert st == 0 and it == o.iters and ninl == o.n_inl, (i, it, o.iters)
        assert np.array_equal(mask, smask) and np.array_equal(np.array(o.E).reshape(3, 3), E)
    b.close()


def test_cxx_host_program_runs_the_pipelined_loop():
    """examples/batch_pipeline.cpp (built by __graft_entry__.build()): the pipelined batch loop of INTEGRATION.md from a C++
    host on the C ABI; the program compares every pipelined step with a synchronous run and exits non-zero on a difference."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "batch_pipeline")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "records differing from the synchronous runs: 0" in out.stdout and "4 of 4 pairs with a pose" in out.stdout, out.stdout


def test_cxx_multi_context_driver_equals_one_context():
    """examples/multi_gpu_batch.cpp: vo_pair_batch with one host thread per context (two contexts on device 0 when the box
    has a single GPU; one per GPU otherwise), pair p on context p mod n, host gather - every record must equal the
    single-context run of the same 150 pairs bit for bit, over three pipelined steps per context, and an argument error
    must come back as a status."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "multi_gpu_batch")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe, "150"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "records differing: 0" in out.stdout and "mixed sizes -> invalid argument" in out.stdout, out.stdout


def test_pair_batch_from_python_matches_the_batch_api(ctx):
    """vo_pair_batch through ctypes with THREE contexts on one device and an odd number of pairs (uneven shares, a last
    step that is not full) against Batch on the same pairs."""
    import ctypes as C

    from visual_odometry_amd._abi import VoPairIn, VoPairOut
    from visual_odometry_amd.frontend import Context

    scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 41), (500, 42), (900, 43)]]
    frames = []
    for k in range(7):
        sc = scenes[k % 3]
        frames.append((sc.img1, sc.img2) if k < 4 else (sc.img2, sc.img1))
    ref = Batch(ctx, 7, 480, 640)
    for i, (a, b) in enumerate(frames):
        ref.set_pair(i, a, b, synth.K_PARETE)
    ref.run()
    want = [_rec(o) for o in ref.get_records()]
    ref.close()
    extra = [Context(), Context()]
    ctxs = (C.c_void_p * 3)(ctx._h, extra[0]._h, extra[1]._h)
    ins = (VoPairIn * 7)()
    for i, (a, b) in enumerate(frames):
        ins[i].img1, ins[i].step1 = a.ctypes.data_as(C.POINTER(C.c_uint8)), a.strides[0]
        ins[i].img2, ins[i].step2 = b.ctypes.data_as(C.POINTER(C.c_uint8)), b.strides[0]
        ins[i].rows, ins[i].cols = 480, 640
        ins[i].K[:] = list(synth.K_PARETE.ravel())
    outs = (VoPairOut * 7)()
    for rep in range(2):  # the second call reuses the contexts' cached batches
        st = ctx.lib.vo_pair_batch(ctxs, 3, ins, 7, outs)
        assert st == 0
        assert [_rec(o) for o in outs] == want
    assert ctx.lib.vo_pair_batch(ctxs, 3, ins, 0, outs) == 0
    dup = (C.c_void_p * 2)(ctx._h, ctx._h)
    assert ctx.lib.vo_pair_batch(dup, 2, ins, 7, outs) == -1  # one context cannot serve two threads
    for e in extra:
        e.close()


@pytest.mark.parametrize("n_oct,n_layers,upright,extended", [(3, 3, 1, 0), (4, 4, 0, 0), (4, 2, 0, 0), (4, 3, 0, 1)])
def test_batch_with_surf_variants_equals_the_single_calls_and_the_oracle(n_oct, n_layers, upright, extended):
    """vo_params.n_octaves = 3 and upright through the batch path (the octave-1 launch stays, octave 3's tiles are gone, every
    keypoint is described by the sequential kernel), and layer counts other than 3 (the plain det-plane / NMS kernels in place of
    the two fused launches, all frames of the batch in one group), and 128-float descriptors (sequential describer, exact-scan
    matcher): records and matches equal the synchronous calls and the oracle."""
    from visual_odometry_amd.frontend import Context, default_params

    p = default_params()
    p.n_octaves, p.n_octave_layers, p.upright, p.extended = n_oct, n_layers, upright, extended
    c = Context(p)
    try:
        scenes = [synth.blob_scene(480, 640, synth.K_PARETE, nb, seed) for nb, seed in [(700, 21), (400, 22), (900, 23)]]
        b = Batch(c, len(scenes), 480, 640)
        for i, sc in enumerate(scenes):
            b.set_pair(i, sc.img1, sc.img2, sc.K)
        b.run()
        b.sync()
        for i, sc in enumerate(scenes):
            out = b.get(i)
            p1, p2, m, mask, X = b.get_points(i)
            single = c.opt_DetectFeatures(sc.img1, sc.img2)
            assert np.array_equal(p1, single.Kpoints1) and np.array_equal(p2, single.Kpoints2)
            rp1, rp2, rm = O.detect_match(sc.img1, sc.img2, params=p)
            assert out.n_match == len(rm) and len(rm) > 20
            assert np.array_equal(p1, rp1) and np.array_equal(p2, rp2)
            assert np.array_equal(m["distance"].view(np.uint32), rm["distance"].view(np.uint32))
        b.close()
    finally:
        c.close()
