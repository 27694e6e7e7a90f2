"""horayzon.direction -- surface-normal and north vectors (ECEF) for curved-DEM input of the
horizon / shadow path, on MI355X (reference: horayzon/direction.pyx)."""
import numpy as np

from . import _lib
from ._lib import ptr
from .transform import _ELLPS


def surf_norm(lon, lat, *, device=0):
    """Unit vectors perpendicular to the ellipsoid at (lon, lat) [degree, float64]; returns float32
    (..., 3) in ECEF (direction.pyx:15-45)."""
    if lon.shape != lat.shape:
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if (lon.dtype != "float64") or (lat.dtype != "float64"):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    shp = lon.shape
    lo = np.ascontiguousarray(lon).ravel()
    la = np.ascontiguousarray(lat).ravel()
    out = np.empty((lo.size, 3), np.float32)
    _lib.check(_lib.lib().hz_surf_norm(ptr(lo), ptr(la), lo.size, ptr(out), device))
    return out.reshape(shp + (3,))


def north_dir(x_ecef, y_ecef, z_ecef, vec_norm_ecef, ellps, *, device=0):
    """Unit vectors pointing towards North in the local tangent plane (direction.pyx:75-122)."""
    if (x_ecef.shape != y_ecef.shape) or (y_ecef.shape != z_ecef.shape):
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if ((x_ecef.dtype != "float64") or (y_ecef.dtype != "float64")
            or (z_ecef.dtype != "float64")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if (x_ecef.shape + (3,)) != vec_norm_ecef.shape:
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if vec_norm_ecef.dtype != "float32":
        raise ValueError("Input array has incorrect data type")
    if ellps not in ("sphere", "GRS80", "WGS84"):
        raise ValueError("Unknown value for 'ellps'")
    shp = x_ecef.shape
    a = [np.ascontiguousarray(v).ravel() for v in (x_ecef, y_ecef, z_ecef)]
    vn = np.ascontiguousarray(vec_norm_ecef).reshape(-1, 3)
    out = np.empty(vn.shape, np.float32)
    _lib.check(_lib.lib().hz_north_dir(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(vn), a[0].size, _ELLPS[ellps],
                                       ptr(out), device))
    return out.reshape(shp + (3,))
