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
    if len(azim) < 2:   # azim[1] - azim[0] is read (topo_param.pyx:433): out of bounds in the reference
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    azim = np.ascontiguousarray(azim)
    hori = np.ascontiguousarray(hori)
    vec_tilt = np.ascontiguousarray(vec_tilt)
    svf = np.empty(hori.shape[:2], dtype=np.float32)
    _lib.check(_lib.lib().hz_sky_view_factor(ptr(azim), ptr(hori), ptr(vec_tilt),
                                             hori.shape[0], hori.shape[1], hori.shape[2],
                                             ptr(svf), device))
    return svf


def visible_sky_fraction(azim, hori, vec_tilt, *, device=0):
    """Visible sky fraction (solid angle of the visible sky); arguments, checks and result as the
    reference (topo_param.pyx:465-496)."""
    if (len(azim) != hori.shape[2]) or (hori.shape[:2] != vec_tilt.shape[:2])\
            or (vec_tilt.shape[2] != 3):
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    if ((azim.dtype != "float32") or (hori.dtype != "float32")
            or (vec_tilt.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if len(azim) < 2:   # azim[1] - azim[0] is read (topo_param.pyx:520): out of bounds in the reference
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    azim = np.ascontiguousarray(azim)
    hori = np.ascontiguousarray(hori)
    vec_tilt = np.ascontiguousarray(vec_tilt)
    vsf = np.empty(hori.shape[:2], dtype=np.float32)
    _lib.check(_lib.lib().hz_visible_sky_fraction(ptr(azim), ptr(hori), ptr(vec_tilt), hori.shape[0],
                                                  hori.shape[1], hori.shape[2], ptr(vsf), device))
    return vsf


def topographic_openness(azim, hori, *, device=0):
    """Positive topographic openness (Yokoyama et al. 2002) [radian]; arguments, checks and result as
    the reference (topo_param.pyx:548-574)."""
    if len(azim) != hori.shape[2]:
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    if (azim.dtype != "float32") or (hori.dtype != "float32"):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    azim = np.ascontiguousarray(azim)
    hori = np.ascontiguousarray(hori)
    top = np.empty(hori.shape[:2], dtype=np.float32)
    _lib.check(_lib.lib().hz_topographic_openness(ptr(azim), ptr(hori), hori.shape[0], hori.shape[1],
                                                  hori.shape[2], ptr(top), device))
    return top


def _slope(which, x, y, z, rot_mat, output_rot, device):
    # Check arguments (topo_param.pyx:59-72 / :262-277)
    if (x.shape != y.shape) or (y.shape != z.shape):
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if ((x.dtype != "float32") or (y.dtype != "float32")
            or (z.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if which == 1 and output_rot and (rot_mat is None):
        raise ValueError("'rot_mat' must be provided for 'output_rot = True'")
    if rot_mat is not None:
        if ((x.shape[0] != rot_mat.shape[0])
                or (x.shape[1] != rot_mat.shape[1])):
            raise ValueError("Inconsistent shapes / number of dimensions of "
                             + "input arrays")
        if rot_mat.dtype != "float32":
            raise ValueError("'rot mat' has incorrect data type")
        rot_mat = np.ascontiguousarray(rot_mat)
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y)
    z = np.ascontiguousarray(z)
    vec_tilt = np.empty(x.shape + (3,), dtype=np.float32)
    L = _lib.lib()
    fn = L.hz_slope_plane_meth if which == 0 else L.hz_slope_vector_meth
    _lib.check(fn(ptr(x), ptr(y), ptr(z), x.shape[0], x.shape[1], ptr(rot_mat), int(bool(output_rot)),
                  ptr(vec_tilt), device))
    return vec_tilt


def slope_plane_meth(x, y, z, rot_mat=None, output_rot=False, *, device=0):
    """Plane-based slope computation (surface normal of the least-squares plane through the
    centre and its 8 neighbours).  Arguments and result as the reference
    (topo_param.pyx:16-82): x, y, z float32 (y, x); optional rot_mat float32 (y, x, 3, 3);
    returns vec_tilt float32 (y, x, 3) with NaN on the outermost ring."""
    return _slope(0, x, y, z, rot_mat, output_rot, device)


def slope_vector_meth(x, y, z, rot_mat=None, output_rot=False, *, device=0):
    """Vector-based slope computation (average normal of the 4 adjacent triangles,
    Corripio 2003).  Arguments and result as the reference (topo_param.pyx:230-281)."""
    return _slope(1, x, y, z, rot_mat, output_rot, device)
