"""N2: get_image on the device (vo_prep_*, csrc/vo_prep.hip) against the CPU restatement, stage by stage, bit for bit.
Every stage is integer / fixed-point or a float computation in OpenCV's operation order: the bar is exact equality.
(The oracle itself is parity-unpinned for this row: see tests/test_cpu_prep.py.)"""
import numpy as np
import pytest

import oracle_lib as O
from visual_odometry_amd import prep as P
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import Batch, Sequence

pytestmark = pytest.mark.gpu

K_SMALL = np.array([[300.0, 0, 80.0], [0, 300.0, 60.0], [0, 0, 1.0]])


def _check(ctx, src_shape, dst_shape, K, dist, seed, clip=16.0, tiles=(8, 8), frames=1):
    pr = P.Prep(ctx, src_shape, dst_shape[1], dst_shape[0], K, dist, clip, tiles)
    try:
        for f in range(frames):
            img = synth.bgr_frame(src_shape[0], src_shape[1], seed + f)
            ref = O.prep(pr.params, img)
            out = pr.get_image(img)
            assert np.array_equal(pr.fetch(4), ref["map1"]) and np.array_equal(pr.fetch(5), ref["map2"]), "undistortion maps"
            assert np.array_equal(pr.fetch(0), ref["resized"]), "resize"
            assert np.array_equal(pr.fetch(1), ref["undist"]), "remap"
            assert np.array_equal(pr.fetch(2), ref["gray0"]), "gray"
            assert np.array_equal(pr.fetch(3), ref["lut"]), "CLAHE look-up tables"
            assert np.array_equal(out, ref["gray"]), "CLAHE output"
    finally:
        pr.close()


def test_rapallo_frames_2560x2056_to_640x505(ctx):
    K = P.updateCameraMatrix((2056, 2560), 640, 505, synth.K_RAPALLO_FULL, ctx.lib)
    assert np.array_equal(K, np.diag([640 / 2560, 505 / 2056, 1.0]) @ synth.K_RAPALLO_FULL)
    _check(ctx, (2056, 2560), (505, 640), K, synth.DIST_RAPALLO, 11, frames=3)  # three frames: staging slots wrap around


def test_integer_scales_and_no_resize(ctx):
    _check(ctx, (960, 1280), (480, 640), K_SMALL * [[4], [4], [1]], [0.05, -0.1, 0.001, 0.002], 21)      # 2 x 2 fast path
    _check(ctx, (1440, 1920), (480, 640), K_SMALL * [[4], [4], [1]], [-0.2, 0.07, -0.001, 0.0005], 22)   # 3 x 3 fast path
    _check(ctx, (505, 640), (505, 640), K_SMALL * [[4], [4], [1]], synth.DIST_RAPALLO, 23)               # parete / soller case: no resize


def test_ragged_sizes_and_tile_grids(ctx):
    _check(ctx, (131, 173), (61, 97), K_SMALL * [[0.6], [0.5], [1]], [0.3, 0.2, 0.01, -0.01], 31)           # both axes fractional
    _check(ctx, (131, 173), (131, 97), K_SMALL * [[0.6], [1.0], [1]], [0.0, 0.0, 0.0, 0.0], 32, tiles=(4, 2))  # one axis only
    _check(ctx, (64, 64), (32, 32), K_SMALL * [[0.2], [0.25], [1]], [0.0, 0.0, 0.0, 0.0], 33, clip=0.0, tiles=(2, 2))  # clip off
    _check(ctx, (200, 300), (100, 150), K_SMALL, [2.5, 1.0, 0.2, 0.2], 34, clip=1.0, tiles=(16, 16))        # sources far outside the frame
    _check(ctx, (333, 512), (61, 93), K_SMALL * [[0.3], [0.5], [1]], [0.1, 0.0, 0.0, 0.0], 38)                 # staged resize, 8 x 8 taps
    _check(ctx, (480, 640), (200, 270), K_SMALL * [[0.9], [1.6], [1]], [0.0, 0.0, 0.01, 0.0], 39)              # staged resize, 4 x 4 taps
    _check(ctx, (330, 520), (41, 64), K_SMALL * [[0.4], [0.3], [1]], [0.1, 0.0, 0.0, 0.0], 35, tiles=(4, 4))   # scale ~8.1: any-scale kernel
    _check(ctx, (400, 350), (40, 70), K_SMALL * [[0.4], [0.3], [1]], [0.1, 0.0, 0.0, 0.0], 36, tiles=(2, 2))   # 5 x 10 integer boxes
    _check(ctx, (300, 450), (60, 150), K_SMALL * [[0.8], [0.5], [1]], [0.0, 0.1, 0.0, 0.0], 37, tiles=(3, 5))  # 3 x 5 boxes, odd tile grid


def test_extreme_images(ctx):
    K = K_SMALL * [[0.5], [0.5], [1]]
    pr = P.Prep(ctx, (160, 200), 100, 80, K, [0.1, 0.0, 0.0, 0.0])
    try:
        for img in (np.zeros((160, 200, 3), np.uint8), np.full((160, 200, 3), 255, np.uint8)):
            assert np.array_equal(pr.get_image(img), O.prep(pr.params, img)["gray"])
        rng = np.random.default_rng(5)
        img = rng.integers(0, 256, (160, 200, 3), dtype=np.uint8)  # white noise: every histogram bin near the clip level
        assert np.array_equal(pr.get_image(img), O.prep(pr.params, img)["gray"])
        padded = np.zeros((160, 220, 3), np.uint8)  # a frame with a row stride
        padded[:, :200] = img
        assert np.array_equal(pr.get_image(padded[:, :200]), O.prep(pr.params, img)["gray"])
    finally:
        pr.close()


def test_prep_feeds_a_batch_on_the_device(ctx):
    """get_image straight into the frame slots of a batch == get_image to the host followed by set_pair."""
    h, w = 480, 640
    sc = synth.blob_scene(h, w, synth.K_PARETE, 500, 77)
    K_full = np.diag([2.0, 2.0, 1.0]) @ sc.K
    big = [np.repeat(np.repeat(np.stack([im, im, im], axis=-1), 2, axis=0), 2, axis=1) for im in (sc.img1, sc.img2)]
    K = P.updateCameraMatrix((2 * h, 2 * w), w, h, K_full, ctx.lib)
    pr = P.Prep(ctx, (2 * h, 2 * w), w, h, K, [0.02, -0.01, 0.0005, -0.0005])
    try:
        g = [O.prep(pr.params, f)["gray"] for f in big]
        a = Batch(ctx, 1, h, w)
        a.set_pair(0, g[0], g[1], K)
        a.run(); a.sync()
        b = Batch(ctx, 1, h, w)
        pr.to_batch(big[0], b, 0)
        pr.to_batch(big[1], b, 1)
        b.set_k(0, K)
        b.run(); b.sync()
        oa, ob = a.get(0), b.get(0)
        assert oa.n_kp1 > 200 and oa.n_match > 30
        for f in ("status", "n_kp1", "n_kp2", "n_match", "n_inl", "n_good", "n_world", "success", "iters"):
            assert getattr(oa, f) == getattr(ob, f), f
        assert np.array_equal(np.array(oa.E), np.array(ob.E)) and np.array_equal(np.array(oa.R), np.array(ob.R))
        for x, y in zip(a.get_points(0), b.get_points(0)):
            assert np.array_equal(x, y)
        # and a sequence batch
        s = Sequence(ctx, 2, h, w)
        pr.to_batch(big[0], s, 0)
        pr.to_batch(big[1], s, 1)
        s.set_pair(0, 0, 1, K)
        s.run(); s.sync()
        os_ = s.get(0)
        assert (os_.n_kp1, os_.n_kp2, os_.n_match, os_.n_inl) == (oa.n_kp1, oa.n_kp2, oa.n_match, oa.n_inl)
        for bb in (a, b, s):
            bb.close()
    finally:
        pr.close()


def test_rejects_what_the_node_never_asks_for(ctx):
    from visual_odometry_amd.frontend import VoError

    with pytest.raises(VoError):
        P.Prep(ctx, (240, 320), 640, 480, K_SMALL, [0, 0, 0, 0])  # enlargement: INTER_AREA would turn into INTER_LINEAR
    pr = P.Prep(ctx, (240, 320), 160, 120, K_SMALL, [0, 0, 0, 0])
    with pytest.raises(ValueError):
        pr.get_image(np.zeros((240, 320), np.uint8))  # not a colour frame
    b = Batch(ctx, 1, 480, 640)
    with pytest.raises(VoError):
        pr.to_batch(np.zeros((240, 320, 3), np.uint8), b, 0)  # batch of another frame size
    b.close()
    pr.close()
