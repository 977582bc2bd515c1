"""N1: the gates and the pose chain the reference node runs around the three hot-path calls
(src/visual_Odometry.cpp:296-430, 536-614).  Thin ctypes wrappers over the host-only entry points of libvo_mi355x.so
(csrc/vo_chain.hip); names follow the reference (camera_utility.hpp / math_utility.hpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi

DISTANCE = 20.0  # camera_utility.hpp:11
MIN_NUM_FEATURES = 20.0  # camera_utility.hpp:12
# enum fail_V0 (visual_Odometry.cpp:58)
FAIL_DETECTION, NOT_MOVING, FAIL_RECOVER, SUCCESS = 0, 1, 2, 3

_lib = None
_f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)


def _L():
    global _lib
    if _lib is None:
        _lib = _abi.load_library()
        _lib.vo_wrap2pi.restype = C.c_double
        _lib.vo_wrap2pi.argtypes = [C.c_double]
    return _lib


def _d(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _chk(st):
    if st != 0:
        raise ValueError(f"libvo_mi355x status {st}")


def checkMinFeat(n_matches: int) -> bool:
    return bool(_L().vo_check_min_feat(int(n_matches), C.c_double(MIN_NUM_FEATURES)))


def checkIfMoving(p1, p2, distance: float = DISTANCE, return_median: bool = False):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    moving, med = C.c_int(0), C.c_double(0)
    _chk(_L().vo_check_if_moving(p1.ctypes.data_as(_f32p), p2.ctypes.data_as(_f32p), len(p1), C.c_double(distance), C.byref(moving),
                                 C.byref(med)))
    return (bool(moving.value), med.value) if return_median else bool(moving.value)


def scaleFactor(distance: float, world_points) -> float:
    X = _d(world_points)
    assert X.ndim == 2 and X.shape[0] == 3
    sf = C.c_double(0)
    _chk(_L().vo_scale_factor(C.c_float(distance), X.ctypes.data_as(_f64p), X.shape[1], C.byref(sf)))
    return sf.value


def absolutePose(rotm, tran, orient, loc, SF: float, world_points=None):
    """Returns [scaled_locW (3,), orient_wk (3,3)] and, with world_points (3 x n), world_pointsW (3 x n)."""
    rotm, orient, tran, loc = _d(rotm, (3, 3)), _d(orient, (3, 3)), _d(tran, (3,)), _d(loc, (3,))
    t_wk, R_wk = np.zeros(3), np.zeros((3, 3))
    X = Xw = None
    n = 0
    if world_points is not None:
        X = _d(world_points)
        n = X.shape[1]
        Xw = np.zeros_like(X)
    _chk(_L().vo_absolute_pose(rotm.ctypes.data_as(_f64p), tran.ctypes.data_as(_f64p), orient.ctypes.data_as(_f64p),
                               loc.ctypes.data_as(_f64p), C.c_double(SF), X.ctypes.data_as(_f64p) if X is not None else None, n,
                               t_wk.ctypes.data_as(_f64p), R_wk.ctypes.data_as(_f64p),
                               Xw.ctypes.data_as(_f64p) if Xw is not None else None))
    return [t_wk, R_wk] + ([Xw] if Xw is not None else [])


def mat2Euler(R) -> np.ndarray:
    R = _d(R, (3, 3))
    rpy = np.zeros(3)
    _chk(_L().vo_mat2euler(R.ctypes.data_as(_f64p), rpy.ctypes.data_as(_f64p)))
    return rpy


def wrap2pi(a: float) -> float:
    return _L().vo_wrap2pi(float(a))


def absDiff_Vec3(v1, v2, isAngle: bool = False) -> np.ndarray:
    a, b, out = _d(v1, (3,)), _d(v2, (3,)), np.zeros(3)
    _chk(_L().vo_abs_diff3(a.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p), int(isAngle), out.ctypes.data_as(_f64p)))
    return out


def estimateTwist(deltaT: float, R, t, SF: float, estimate_rpy, Rbc, motion2D: bool):
    R, Rbc, t, rpy = _d(R, (3, 3)), _d(Rbc, (3, 3)), _d(t, (3,)), _d(estimate_rpy, (3,))
    lin, ang = np.zeros(3), np.zeros(3)
    _chk(_L().vo_estimate_twist(C.c_double(deltaT), R.ctypes.data_as(_f64p), t.ctypes.data_as(_f64p), C.c_double(SF),
                                rpy.ctypes.data_as(_f64p), Rbc.ctypes.data_as(_f64p), int(motion2D), lin.ctypes.data_as(_f64p),
                                ang.ctypes.data_as(_f64p)))
    return lin, ang


def loopPoseUpdate(orientation, location, R, t, SF: float, dt: float, Rbc, motion2D: bool, world_points=None):
    """absolutePose + mat2Euler(orientation Rbc^T) + estimateTwist in one library call (vo_loop_pose_update).
    orientation (3,3) and location (3,) must be C-contiguous float64 arrays; they are updated IN PLACE.
    Returns (rpy, twist_linear, twist_angular, world_pointsW or None)."""
    R, t, Rbc = _d(R, (3, 3)), _d(t, (3,)), _d(Rbc, (3, 3))
    rpy, lin, ang = np.zeros(3), np.zeros(3), np.zeros(3)
    X = Xw = None
    n = 0
    if world_points is not None:
        X = _d(world_points)
        n = X.shape[1]
        Xw = np.zeros_like(X)
    _chk(_L().vo_loop_pose_update(orientation.ctypes.data_as(_f64p), location.ctypes.data_as(_f64p), R.ctypes.data_as(_f64p),
                                  t.ctypes.data_as(_f64p), C.c_double(SF), X.ctypes.data_as(_f64p) if X is not None else None, n,
                                  Xw.ctypes.data_as(_f64p) if Xw is not None else None, C.c_double(dt), Rbc.ctypes.data_as(_f64p),
                                  int(motion2D), rpy.ctypes.data_as(_f64p), lin.ctypes.data_as(_f64p), ang.ctypes.data_as(_f64p)))
    return rpy, lin, ang, Xw


def gate(p1, p2):
    """The gate sequence of the node's main loop (visual_Odometry.cpp:303-326): returns None when the pair goes on to
    optRelativePose, else the fail_V0 code the node would publish."""
    if checkMinFeat(len(np.asarray(p1).reshape(-1, 2))):
        return FAIL_DETECTION
    if not checkIfMoving(p1, p2):
        return NOT_MOVING
    return None
