"""Reader for the node's intrinsic YAML files (reference: config/camera_intrinsic.yaml, config/rapallo_intrinsic.yaml;
the node reads them through the ROS parameter server in getCameraParam, src/visual_Odometry.cpp:482-500).

Keys: camera_intrinsic/{fx, fy, ccxLeft, ccyLeft, skew}, distortion_coefficient/{radial/{k1,k2}, tangential/{p1,p2}},
desired_size/{width, height}.  camera_intrinsic.yaml has neither `skew` nor `desired_size` (SURVEY.md section 0,
discrepancy 3): skew defaults to 0 and desired_size to the incoming image size.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import yaml


@dataclasses.dataclass
class CameraConfig:
    fx: float
    fy: float
    cx: float
    cy: float
    skew: float = 0.0
    k1: float = 0.0
    k2: float = 0.0
    p1: float = 0.0
    p2: float = 0.0
    desired_width: int | None = None
    desired_height: int | None = None

    def K(self) -> np.ndarray:
        """K = [fx skew cx; 0 fy cy; 0 0 1] (src/visual_Odometry.cpp:220)."""
        return np.array([[self.fx, self.skew, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])

    def dist(self) -> np.ndarray:
        """[k1 k2 p1 p2] (src/visual_Odometry.cpp:221)."""
        return np.array([self.k1, self.k2, self.p1, self.p2])

    def K_for(self, image_width: int, image_height: int) -> np.ndarray:
        """updateCameraMatrix (camera_utility.hpp:138-145): K <- diag(RW, RH, 1) K for the desired size."""
        dw = self.desired_width or image_width
        dh = self.desired_height or image_height
        S = np.diag([dw / image_width, dh / image_height, 1.0])
        return S @ self.K()


def load_camera_config(path: str) -> CameraConfig:
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    ci = y.get("camera_intrinsic") or {}
    for k in ("fx", "fy", "ccxLeft", "ccyLeft"):
        if k not in ci:
            raise KeyError(f"{path}: camera_intrinsic/{k} missing")
    dc = y.get("distortion_coefficient") or {}
    rad, tan = dc.get("radial") or {}, dc.get("tangential") or {}
    ds = y.get("desired_size") or {}
    return CameraConfig(
        fx=float(ci["fx"]), fy=float(ci["fy"]), cx=float(ci["ccxLeft"]), cy=float(ci["ccyLeft"]),
        skew=float(ci.get("skew", 0.0) or 0.0),
        k1=float(rad.get("k1", 0.0)), k2=float(rad.get("k2", 0.0)), p1=float(tan.get("p1", 0.0)), p2=float(tan.get("p2", 0.0)),
        desired_width=int(ds["width"]) if "width" in ds else None,
        desired_height=int(ds["height"]) if "height" in ds else None,
    )
