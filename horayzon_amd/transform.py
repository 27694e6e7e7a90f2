"""horayzon.transform -- the coordinate transforms that prepare curved-DEM input for the
horizon / shadow path, on MI355X (reference: horayzon/transform.pyx).
lonlat2ecef, ecef2enu, ecef2enu_vector, TransformerEcef2enu, rotation_matrix_glob2loc and the Swiss projection
pair wgs2swiss / swiss2wgs (the swissALTI3D input path of the reference's examples)."""
import numpy as np

from . import _lib
from ._lib import ptr

_ELLPS = {"sphere": 0, "GRS80": 1, "WGS84": 2}


class TransformerEcef2enu:
    """Attributes to transform from ECEF to ENU coordinates; the ENU origin lies on the surface of
    the sphere / ellipsoid at (lon_or, lat_or) [degree] (reference transform.pyx:438-487)."""

    def __init__(self, lon_or, lat_or, ellps):
        if (lon_or < -180.0) or (lon_or > 180.0):
            raise ValueError("Value for 'lon_or' is outside of valid range")
        if (lat_or < -90.0) or (lat_or > 90.0):
            raise ValueError("Value for 'lat_or' is outside of valid range")
        if ellps not in _ELLPS:
            raise ValueError("Unknown value for 'ellps'")
        self.lon_or = lon_or
        self.lat_or = lat_or
        self.ellps = ellps
        lo, la = np.deg2rad(lon_or), np.deg2rad(lat_or)
        if ellps == "sphere":
            n = zfac = 6370997.0
        else:
            a = 6378137.0
            f = (1.0 / 298.257222101) if ellps == "GRS80" else (1.0 / 298.257223563)
            b = a * (1.0 - f)
            e_2 = 1.0 - (b ** 2 / a ** 2)
            n = a / np.sqrt(1.0 - e_2 * np.sin(la) ** 2)
            zfac = b ** 2 / a ** 2 * n
        self.x_ecef_or = n * np.cos(la) * np.cos(lo)
        self.y_ecef_or = n * np.cos(la) * np.sin(lo)
        self.z_ecef_or = zfac * np.sin(la)


def lonlat2ecef(lon, lat, h, ellps, *, device=0):
    """Geodetic longitude/latitude [degree] (float64) and ellipsoidal height (float32) to ECEF
    coordinates (float64); arguments and checks as the reference (transform.pyx:15-57)."""
    if (lon.shape != lat.shape) or (lat.shape != h.shape):
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if ((lon.dtype != "float64") or (lat.dtype != "float64")
            or (h.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if ellps not in ("sphere", "GRS80", "WGS84"):
        raise ValueError("Unknown value for 'ellps'")
    shp = lon.shape
    lon = np.ascontiguousarray(lon).ravel()
    lat = np.ascontiguousarray(lat).ravel()
    h = np.ascontiguousarray(h).ravel()
    out = [np.empty(lon.size, np.float64) for _ in range(3)]
    _lib.check(_lib.lib().hz_lonlat2ecef(ptr(lon), ptr(lat), ptr(h), lon.size, _ELLPS[ellps],
                                         ptr(out[0]), ptr(out[1]), ptr(out[2]), device))
    return out[0].reshape(shp), out[1].reshape(shp), out[2].reshape(shp)


def ecef2enu(x_ecef, y_ecef, z_ecef, trans_ecef2enu, *, device=0):
    """ECEF (float64) to ENU (float32) coordinates (transform.pyx:108-149)."""
    if (x_ecef.shape != y_ecef.shape) or (y_ecef.shape != z_ecef.shape):
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if ((x_ecef.dtype != "float64") or (y_ecef.dtype != "float64")
            or (z_ecef.dtype != "float64")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if not isinstance(trans_ecef2enu, TransformerEcef2enu):
        raise ValueError("Last input argument must be instance of class "
                         + "'TransformerEcef2enu'")
    shp = x_ecef.shape
    a = [np.ascontiguousarray(v).ravel() for v in (x_ecef, y_ecef, z_ecef)]
    out = [np.empty(a[0].size, np.float32) for _ in range(3)]
    t = trans_ecef2enu
    _lib.check(_lib.lib().hz_ecef2enu(ptr(a[0]), ptr(a[1]), ptr(a[2]), a[0].size, float(t.lon_or), float(t.lat_or),
                                      _ELLPS[getattr(t, "ellps", "WGS84")], ptr(out[0]), ptr(out[1]), ptr(out[2]),
                                      device))
    return out[0].reshape(shp), out[1].reshape(shp), out[2].reshape(shp)


def ecef2enu_vector(vec_ecef, trans_ecef2enu, *, device=0):
    """Rotate vectors (float32, components in the last dimension) from ECEF to ENU
    (transform.pyx:194-228)."""
    if (vec_ecef.ndim < 2) or (vec_ecef.shape[vec_ecef.ndim - 1] != 3):
        raise ValueError("Incorrect shape / number of dimensions of input "
                         + "array")
    if vec_ecef.dtype != "float32":
        raise ValueError("Input array has incorrect data type")
    if not isinstance(trans_ecef2enu, TransformerEcef2enu):
        raise ValueError("Last input argument must be instance of class "
                         + "'TransformerEcef2enu'")
    shp = vec_ecef.shape
    v = np.ascontiguousarray(vec_ecef).reshape(-1, 3)
    out = np.empty(v.shape, np.float32)
    t = trans_ecef2enu
    _lib.check(_lib.lib().hz_ecef2enu_vector(ptr(v), v.shape[0], float(t.lon_or), float(t.lat_or),
                                             _ELLPS[getattr(t, "ellps", "WGS84")], ptr(out), device))
    return out.reshape(shp)


def _triple(a, b, c, sym, *, device):
    """shared body of wgs2swiss / swiss2wgs: (f64, f64, f32) arrays of one shape -> (f64, f64, f32)"""
    if (a.shape != b.shape) or (b.shape != c.shape):
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    if ((a.dtype != "float64") or (b.dtype != "float64")
            or (c.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    shp = a.shape
    a, b, c = (np.ascontiguousarray(v).ravel() for v in (a, b, c))
    o0, o1, o2 = np.empty(a.size, np.float64), np.empty(a.size, np.float64), np.empty(a.size, np.float32)
    _lib.check(getattr(_lib.lib(), sym)(ptr(a), ptr(b), ptr(c), a.size, ptr(o0), ptr(o1), ptr(o2), device))
    return o0.reshape(shp), o1.reshape(shp), o2.reshape(shp)


# wgs2swiss / swiss2wgs: outside the scope table (SURVEY.md section 8(f)4 cites transform.pyx:60-103, 152-189, 231-261;
# coordinate transforms are otherwise out of scope, section 2).  Kept from round 2 as a convenience for the swissALTI3D
# input path of the reference's examples; no parity or coverage claim rests on them.
def wgs2swiss(lon, lat, h_wgs, *, device=0):
    """Ellipsoidal WGS84 longitude / latitude [degree] (float64) and height above the ellipsoid (float32) to Swiss
    projection coordinates LV95: ``e``, ``n`` [metre] (float64) and ``h_ch`` (float32).  Arguments, checks and
    formulas as the reference (transform.pyx:266-345)."""
    return _triple(lon, lat, h_wgs, "hz_wgs2swiss", device=device)


def swiss2wgs(e, n, h_ch, *, device=0):
    """Swiss projection coordinates LV95 [metre] (float64, height float32) to WGS84 longitude / latitude [degree]
    (float64) and height above the ellipsoid (float32); reference transform.pyx:349-432."""
    return _triple(e, n, h_ch, "hz_swiss2wgs", device=device)


def rotation_matrix_glob2loc(vec_north_enu, vec_norm_enu):
    """Matrices (y + 2, x + 2, 3, 3; NaN ring) that rotate vectors from global to local ENU
    coordinates: rows = east (north x norm), north, norm (transform.pyx:490-530)."""
    if vec_north_enu.shape != vec_norm_enu.shape:
        raise ValueError("Inconsistent shapes / number of dimensions of "
                         + "input arrays")
    rot = np.full((vec_north_enu.shape[0] + 2, vec_north_enu.shape[1] + 2, 3, 3), np.nan, dtype=np.float32)
    rot[1:-1, 1:-1, 0, :] = np.cross(vec_north_enu, vec_norm_enu, axisa=2, axisb=2)
    rot[1:-1, 1:-1, 1, :] = vec_north_enu
    rot[1:-1, 1:-1, 2, :] = vec_norm_enu
    return rot
