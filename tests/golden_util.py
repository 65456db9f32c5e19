import glob
import os

import numpy as np

import sdf_amd as S
from sdf_amd import _ffi

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    buf = S.SdfTreeBuffer(z["dt"].view(_ffi.DT_DTYPE), z["ut"].view(_ffi.UT_DTYPE))
    c = z["cam"]
    cam = _ffi.Camera()
    for k, name in enumerate(("eye", "origin", "du", "dv")):
        for a in range(3):
            getattr(cam, name)[a] = float(c[3 * k + a])
    m = z["march"]
    mp = _ffi.MarchParams(int(m[0]), float(m[1]), float(m[2]), float(m[3]), float(m[4]), 0)
    return z, buf, cam, mp
