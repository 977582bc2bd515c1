"""ctypes mirror of include/vo_mi355x.h (structs + library loader).

The library is the product: there is no Python or CPU fallback.  If libvo_mi355x.so is missing
the import of the front end fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvo_mi355x.so")

VO_OK = 0
VO_ERR_INVALID_ARG = -1
VO_ERR_NO_DEVICE = -2
VO_ERR_HIP = -3
VO_ERR_CAPACITY = -4
VO_ERR_UNSUPPORTED = -5
VO_ERR_TOO_FEW_POINTS = -6
VO_ERR_NO_MODEL = -7


class VoKeypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


class VoDMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32), ("distance", C.c_float)]


class VoParams(C.Structure):
    _fields_ = [
        ("hessian_threshold", C.c_float), ("n_octaves", C.c_int32), ("n_octave_layers", C.c_int32),
        ("extended", C.c_int32), ("upright", C.c_int32), ("ratio_thresh", C.c_float),
        ("roi_x_lo", C.c_float), ("roi_y_lo", C.c_float), ("roi_x_hi", C.c_float), ("roi_y_hi", C.c_float),
        ("ransac_prob", C.c_double), ("ransac_threshold", C.c_double), ("ransac_max_iters", C.c_int32),
        ("ransac_seed", C.c_uint64), ("inlier_threshold", C.c_float), ("vpf_threshold", C.c_float),
        ("distance_threshold", C.c_double), ("reprojection_tolerance", C.c_float),
        ("kp_capacity", C.c_int32), ("match_capacity", C.c_int32),
    ]


class VoPairOut(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("n_kp1", C.c_int32), ("n_kp2", C.c_int32), ("n_match", C.c_int32),
        ("n_inl", C.c_int32), ("n_good", C.c_int32), ("n_world", C.c_int32), ("success", C.c_int32),
        ("iters", C.c_int32), ("pad_", C.c_int32),
        ("E", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3),
    ]


class VoPairIn(C.Structure):
    """vo_pair_in: one frame pair of vo_pair_batch (cv::Mat-style buffers + K)."""
    _fields_ = [
        ("img1", C.POINTER(C.c_uint8)), ("step1", C.c_size_t), ("img2", C.POINTER(C.c_uint8)), ("step2", C.c_size_t),
        ("rows", C.c_int32), ("cols", C.c_int32), ("K", C.c_double * 9),
    ]


class VoPrepParams(C.Structure):
    _fields_ = [
        ("src_rows", C.c_int32), ("src_cols", C.c_int32), ("dst_rows", C.c_int32), ("dst_cols", C.c_int32),
        ("K", C.c_double * 9), ("dist", C.c_double * 4), ("clip_limit", C.c_double),
        ("tiles_x", C.c_int32), ("tiles_y", C.c_int32),
    ]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KP_DTYPE.itemsize == C.sizeof(VoKeypoint) == 28
assert DM_DTYPE.itemsize == C.sizeof(VoDMatch) == 16


def default_params() -> VoParams:
    """camera_utility.hpp:33-63 defaults (the same values vo_params_default() fills)."""
    p = VoParams()
    p.hessian_threshold = 50.0
    p.n_octaves, p.n_octave_layers, p.extended, p.upright = 4, 3, 0, 0
    p.ratio_thresh = 0.7
    p.roi_x_lo, p.roi_y_lo, p.roi_x_hi, p.roi_y_hi = 28, 20, 614, 482
    p.ransac_prob, p.ransac_threshold, p.ransac_max_iters = 0.99, 2.0, 1000
    p.ransac_seed = 2**64 - 1
    p.inlier_threshold, p.vpf_threshold = 0.3, 0.85
    p.distance_threshold = 50.0
    p.reprojection_tolerance = 0.5
    p.kp_capacity = 16384
    p.match_capacity = 16384
    return p


def load_library(path: str | None = None) -> C.CDLL:
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    return C.CDLL(path)
