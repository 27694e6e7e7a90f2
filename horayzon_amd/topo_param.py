"""horayzon.topo_param.sky_view_factor on MI355X (the only topo_param routine
on the horizon hot path; reference: horayzon/topo_param.pyx:377-460)."""
import numpy as np

from . import _lib
from ._lib import ptr


def sky_view_factor(azim, hori, vec_tilt, *, device=0):
    """Sky view factor (SVF) computation.

    Same arguments, checks and result as the reference
    (topo_param.pyx:377-409): azim float32 (azim), hori float32 (y, x, azim)
    [radian], vec_tilt float32 (y, x, 3); returns svf float32 (y, x)."""
    # Check arguments (topo_param.pyx:398-404)
    if (len(azim) != hori.shape[2]) or (hori.shape[:2] != vec_tilt.shape[:2])\
            or (vec_tilt.shape[2] != 3):
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    if ((azim.dtype != "float32") or (hori.dtype != "float32")
            or (vec_tilt.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    azim = np.ascontiguousarray(azim)
    hori = np.ascontiguousarray(hori)
    vec_tilt = np.ascontiguousarray(vec_tilt)
    svf = np.empty(hori.shape[:2], dtype=np.float32)
    _lib.check(_lib.lib().hz_sky_view_factor(ptr(azim), ptr(hori), ptr(vec_tilt),
                                             hori.shape[0], hori.shape[1], hori.shape[2],
                                             ptr(svf), device))
    return svf
