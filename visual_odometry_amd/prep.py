"""N2: the node's per-frame preprocessing on the device (csrc/vo_prep.hip).

Mirrors the reference helpers of include/visual_odometry/camera_utility.hpp:
    desiredResize(img, w, h)                      :119
    updateCameraMatrix(image, w, h, K)            :138
    get_image(img, w, h, K, dist)                 :147   resize -> undistort -> RGB2GRAY (on BGR) -> CLAHE(16)
Everything runs in libvo_mi355x.so; there is no Python or CPU implementation behind it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._abi import VoPrepParams, load_library
from .frontend import Batch, Context

u8p = C.POINTER(C.c_uint8)
f64p = C.POINTER(C.c_double)

CLIP_LIMIT = 16.0  # camera_utility.hpp:10


def updateCameraMatrix(src_shape, desired_width, desired_height, K, lib=None):
    """updateCameraMatrix, camera_utility.hpp:138: K of the resized frames.  src_shape = (rows, cols)."""
    lib = lib or load_library()
    K = np.ascontiguousarray(K, np.float64)
    out = np.zeros((3, 3), np.float64)
    st = lib.vo_update_camera_matrix(int(src_shape[0]), int(src_shape[1]), int(desired_height), int(desired_width),
                                     K.ctypes.data_as(f64p), out.ctypes.data_as(f64p))
    if st != 0:
        raise ValueError(f"vo_update_camera_matrix status {st}")
    return out


def make_params(src_shape, desired_width, desired_height, K_resized, dist, clip_limit=CLIP_LIMIT, tiles=(8, 8)) -> VoPrepParams:
    p = VoPrepParams()
    p.src_rows, p.src_cols = int(src_shape[0]), int(src_shape[1])
    p.dst_rows, p.dst_cols = int(desired_height), int(desired_width)
    p.K[:] = [float(v) for v in np.asarray(K_resized, np.float64).ravel()]
    d = [float(v) for v in np.asarray(dist, np.float64).ravel()[:4]]
    p.dist[:] = d + [0.0] * (4 - len(d))
    p.clip_limit = float(clip_limit)
    p.tiles_x, p.tiles_y = int(tiles[0]), int(tiles[1])
    return p


def _bgr(img, rows, cols):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape != (rows, cols, 3):
        raise ValueError(f"expected a CV_8UC3 frame of {rows} x {cols}")
    if img.strides[2] != 1 or img.strides[1] != 3:
        img = np.ascontiguousarray(img)
    return img


class Prep:
    """One vo_prep: the maps and tables of one camera.  get_image(frame) is the reference's get_image."""

    def __init__(self, ctx: Context, src_shape, desired_width, desired_height, K_resized, dist, clip_limit=CLIP_LIMIT, tiles=(8, 8)):
        self.ctx = ctx
        self.params = make_params(src_shape, desired_width, desired_height, K_resized, dist, clip_limit, tiles)
        self._h = C.c_void_p()
        ctx._chk(ctx.lib.vo_prep_create(ctx._h, C.byref(self.params), C.byref(self._h)))

    def close(self):
        if self._h:
            self.ctx.lib.vo_prep_destroy.argtypes = [C.c_void_p]
            self.ctx.lib.vo_prep_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_image(self, frame):
        p = self.params
        frame = _bgr(frame, p.src_rows, p.src_cols)
        out = np.empty((p.dst_rows, p.dst_cols), np.uint8)
        self.ctx._chk(self.ctx.lib.vo_prep_run(self._h, frame.ctypes.data_as(u8p), C.c_size_t(frame.strides[0]), out.ctypes.data_as(u8p),
                                               C.c_size_t(out.strides[0])))
        return out

    def to_batch(self, frame, batch: Batch, slot: int):
        """get_image with the result left on the device as frame `slot` of a Batch (2 * pair + {0, 1}) or Sequence."""
        p = self.params
        frame = _bgr(frame, p.src_rows, p.src_cols)
        self.ctx._chk(self.ctx.lib.vo_prep_to_batch(self._h, frame.ctypes.data_as(u8p), C.c_size_t(frame.strides[0]), batch._h, int(slot)))

    _SHAPES = {0: ("u1", 3), 1: ("u1", 3), 2: ("u1", 1), 4: ("i2", 2), 5: ("u2", 1)}

    def fetch(self, what):
        """Intermediates of the last frame: 0 resized BGR, 1 undistorted BGR, 2 gray, 3 CLAHE LUTs, 4 map1, 5 map2."""
        p = self.params
        if what == 3:
            out = np.empty((p.tiles_y, p.tiles_x, 256), np.uint8)
        else:
            dt, ch = self._SHAPES[what]
            out = np.empty((p.dst_rows, p.dst_cols, ch) if ch > 1 else (p.dst_rows, p.dst_cols), dt)
        self.ctx._chk(self.ctx.lib.vo_prep_fetch(self._h, int(what), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out
