"""Calibrated fisheye camera description (host side).

Only the numbers the hot path needs are kept: the world->camera polynomial and the image
centre, as read by the reference's `FishEyeCameraCalibrated.__init__`
(`utils/fisheye/FishEyeCalibrated.py:7-15`).  The projection itself runs in the HIP energy
kernel (`csrc/energy.hip`); `project_numpy` is a float64 host twin used for synthesising inputs
(it follows `FishEyeCalibrated.py:57-87`).
"""
import json
import os
from dataclasses import dataclass

import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_CALIBRATION = os.path.join(_DATA_DIR, "fisheye.calibration.json")
ALT_CALIBRATION = os.path.join(_DATA_DIR, "fisheye.calibration_new.json")

MAX_POLY = 16  # capacity of the coefficient array in the C ABI (gem_config.poly)


@dataclass(frozen=True)
class FisheyeCamera:
    poly_w2c: tuple      # rho(theta) = sum_i poly_w2c[i] * theta**i   (pixels)
    poly_c2w: tuple
    cx: float
    cy: float
    width: int
    height: int

    @staticmethod
    def from_json(path):
        with open(path) as f:
            d = json.load(f)
        intr = np.asarray(d["intrinsic"], dtype=np.float64)
        w2c = tuple(float(c) for c in d["polynomialW2C"])
        if len(w2c) > MAX_POLY:
            raise ValueError("polynomialW2C has %d coefficients, max %d" % (len(w2c), MAX_POLY))
        return FisheyeCamera(poly_w2c=w2c, poly_c2w=tuple(float(c) for c in d["polynomialC2W"]),
                             cx=float(intr[0, 2]), cy=float(intr[1, 2]),
                             width=int(d["size"][0]), height=int(d["size"][1]))

    def project_numpy(self, points):
        """float64 [n,3] camera-frame points -> [n,2] pixels (FishEyeCalibrated.py:57-87)."""
        p = np.asarray(points, dtype=np.float64)
        n = np.hypot(p[:, 0], p[:, 1])
        if not (n != 0).all():
            raise Exception("norm is zero!")
        theta = np.arctan(-p[:, 2] / n)
        rho = np.zeros_like(theta)
        for c in reversed(self.poly_w2c):
            rho = rho * theta + c
        return np.stack([p[:, 0] / n * rho + self.cx, p[:, 1] / n * rho + self.cy], axis=1)
