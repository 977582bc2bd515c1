"""vo_triang_points_noalloc (the entry point that allocates nothing: static device scratch, pinned buffers of the context)
against vo_triang_points, through the C ABI: bit-identical output on two calls in a row, the second with fewer points, so
that anything the first call left in the shared scratch array would show."""
import ctypes as C

import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu
K = synth.K_1080P


def _inliers(n, seed):
    p1, p2, R, t, _ = synth.correspondences(n, K, seed, inlier_frac=1.0, noise_px=0.15)
    return p1, p2, R, t


def test_noalloc_equals_triang_points_twice_second_smaller(ctx):
    for n, seed in ((3000, 71), (500, 72)):
        p1, p2, R, t = _inliers(n, seed)
        ref = ctx.triangPoints(p1, p2, R, t, K)
        got = ctx.triangPoints_noalloc(p1, p2, R, t, K)
        assert ref.shape[1] > n // 2, ref.shape
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


def test_noalloc_limits(ctx):
    from visual_odometry_amd import _abi

    n = 16384 + 1
    p = np.zeros((n, 2), np.float32)
    X = np.zeros((3, n), np.float64)
    m = C.c_int(-1)
    f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
    R, t = np.eye(3), np.array([1.0, 0.0, 0.0])
    st = ctx.lib.vo_triang_points_noalloc(ctx._h, p.ctypes.data_as(f32p), p.ctypes.data_as(f32p), n, R.ctypes.data_as(f64p),
                                          t.ctypes.data_as(f64p), np.ascontiguousarray(K).ctypes.data_as(f64p), X.ctypes.data_as(f64p), n,
                                          C.byref(m))
    assert st == _abi.VO_ERR_CAPACITY and m.value == 0
    ctx.lib.vo_last_hip_error.restype = C.c_char_p
    assert b"VO_NOALLOC_MAX_POINTS" in ctx.lib.vo_last_hip_error()
    # the maximum itself is accepted and equals the allocating entry point
    p1, p2, R, t = _inliers(16384, 73)
    assert np.array_equal(ctx.triangPoints_noalloc(p1, p2, R, t, K), ctx.triangPoints(p1, p2, R, t, K))
    # an empty input is fine
    assert ctx.triangPoints_noalloc(p[:0], p[:0], R, t, K).shape == (3, 0)
