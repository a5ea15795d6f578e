"""Shared builders for the tests (oracle side)."""
import numpy as np

from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from oracle import np_oracle as O

TINY = vae_schema.VAEShape(latent_dim=32, hidden=(16, 16, 32, 32, 64))
FULL = vae_schema.VAEShape()


def oracle_camera(path=DEFAULT_CALIBRATION):
    c = FisheyeCamera.from_json(path)
    return O.Camera(poly=np.asarray(c.poly_w2c), cx=c.cx, cy=c.cy)


def heat_from_centres(centres):
    return synth.gaussian_heatmaps(centres[..., 0], centres[..., 1])


def sd_from_npz(npz, prefix):
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}
