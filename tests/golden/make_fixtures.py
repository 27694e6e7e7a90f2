"""Generate golden vectors from the parts of the reference that CAN be built here.

The reference's ray-casting core needs Intel Embree (absent), but its Embree-free
Cython modules (horayzon/topo_param.pyx, transform.pyx, direction.pyx) compile with
Cython + gcc.  This script is run ONCE in the development container:

    mkdir /tmp/refbuild && cd /tmp/refbuild
    cp /root/reference/horayzon/{topo_param,transform,direction}.pyx .
    # setup with extra_compile_args ["-O3", "-ffast-math"] as reference setup.py:24
    python setup_ref.py build_ext --inplace
    python /root/repo/tests/golden/make_fixtures.py /tmp/refbuild

and writes small .npz files (inputs + expected outputs = data only) next to itself.
Nothing of the reference travels with the repo; the GPU box never needs it.
"""
import os
import sys

import numpy as np

refdir = sys.argv[1] if len(sys.argv) > 1 else "/tmp/refbuild"
sys.path.insert(0, refdir)
import topo_param   # noqa: E402  (reference module, built from /root/reference sources)
import transform    # noqa: E402
import direction    # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))


def rand_tilt(rng, shape, max_slope_deg):
    slope = np.deg2rad(rng.uniform(0.0, max_slope_deg, shape))
    aspect = rng.uniform(0.0, 2 * np.pi, shape)
    t = np.stack([np.sin(slope) * np.sin(aspect), np.sin(slope) * np.cos(aspect), np.cos(slope)], axis=-1)
    return np.ascontiguousarray(t, np.float32)


# ---- sky_view_factor: random horizons / tilts --------------------------------------------
rng = np.random.default_rng(20220621)
out = {}
for name, (ny, nx, na, hmax, smax) in {"a": (16, 16, 36, 40.0, 35.0),
                                        "b": (8, 8, 360, 60.0, 50.0),
                                        "c": (12, 20, 90, 10.0, 5.0)}.items():
    azim = np.empty(na, np.float32)
    for i in range(na):
        azim[i] = ((2 * np.pi) / na * i)
    hori = np.deg2rad(rng.uniform(-5.0, hmax, (ny, nx, na))).astype(np.float32)
    tilt = rand_tilt(rng, (ny, nx), smax)
    svf = topo_param.sky_view_factor(azim, hori, tilt)
    out["azim_" + name] = azim
    out["hori_" + name] = hori
    out["tilt_" + name] = tilt
    out["svf_" + name] = np.asarray(svf, np.float32)
    out["vsf_" + name] = np.asarray(topo_param.visible_sky_fraction(azim, hori, tilt), np.float32)
    out["top_" + name] = np.asarray(topo_param.topographic_openness(azim, hori), np.float32)
# closed forms: flat horizon & flat tilt -> 1; uniform 30 deg horizon -> cos^2(30 deg)
azim = out["azim_a"]
flat = np.zeros((2, 2, 36), np.float32)
up = np.zeros((2, 2, 3), np.float32); up[..., 2] = 1.0
out["svf_flat"] = np.asarray(topo_param.sky_view_factor(azim, flat, up), np.float32)
out["svf_30deg"] = np.asarray(topo_param.sky_view_factor(azim, flat + np.float32(np.deg2rad(30.0)), up), np.float32)
np.savez_compressed(os.path.join(here, "svf_reference.npz"), **out)
print("svf fixtures:", {k: v.shape for k, v in out.items() if k.startswith("svf")},
      out["svf_flat"].ravel()[:2], out["svf_30deg"].ravel()[:2])

# ---- curved-DEM input preparation: lon/lat -> ECEF -> ENU, surface normal, north vector --
lon = np.linspace(7.95, 8.05, 48)
lat = np.linspace(46.55, 46.45, 40)
lon2, lat2 = np.meshgrid(lon, lat)
rng = np.random.default_rng(7)
from numpy.fft import irfft2   # noqa: E402
elev = (1500.0 + 900.0 * np.sin(lon2 * 180.0) * np.cos(lat2 * 140.0)
        + 60.0 * rng.standard_normal(lon2.shape)).astype(np.float32)
x_ecef, y_ecef, z_ecef = transform.lonlat2ecef(lon2, lat2, elev, ellps="WGS84")
trans = transform.TransformerEcef2enu(lon_or=lon.mean(), lat_or=lat.mean(), ellps="WGS84")
x_enu, y_enu, z_enu = transform.ecef2enu(x_ecef, y_ecef, z_ecef, trans)
off = 6
sl = (slice(off, lat.size - off), slice(off, lon.size - off))
vec_norm_ecef = direction.surf_norm(lon2[sl], lat2[sl])
vec_north_ecef = direction.north_dir(x_ecef[sl], y_ecef[sl], z_ecef[sl], vec_norm_ecef, ellps="WGS84")
vec_norm_enu = transform.ecef2enu_vector(vec_norm_ecef, trans).astype(np.float32)
vec_north_enu = transform.ecef2enu_vector(vec_north_ecef, trans).astype(np.float32)
np.savez_compressed(os.path.join(here, "curved_dem_reference.npz"),
                    lon=lon, lat=lat, elevation=elev,
                    x_enu=x_enu.astype(np.float32), y_enu=y_enu.astype(np.float32), z_enu=z_enu.astype(np.float32),
                    vec_norm=vec_norm_enu, vec_north=vec_north_enu, offset=np.int32(off))
print("curved fixture:", x_enu.shape, vec_norm_enu.shape, float(np.abs(z_enu).max()))

# ---- slope (plane / vector method) and the input-preparation chain, all three ellipsoids -----
out = {}
rng = np.random.default_rng(11)
# planar DEM, no rotation matrices
xp = (np.arange(26) * 40.0).astype(np.float32); yp = ((22 - np.arange(23)) * 35.0).astype(np.float32)
xx, yy = np.meshgrid(xp, yp)
zz = (800.0 + 300.0 * np.sin(xx / 400.0) * np.cos(yy / 300.0) + 15.0 * rng.standard_normal(xx.shape)).astype(np.float32)
out["pl_x"], out["pl_y"], out["pl_z"] = xx, yy, zz
import io, contextlib   # the reference prints when no rot_mat is given
with contextlib.redirect_stdout(io.StringIO()):
    out["pl_tilt_plane"] = topo_param.slope_plane_meth(xx, yy, zz)
out["pl_tilt_vector"] = topo_param.slope_vector_meth(xx, yy, zz)
for ellps in ("sphere", "GRS80", "WGS84"):
    lon = np.linspace(10.0, 10.4, 30); lat = np.linspace(-33.1, -33.4, 26)
    lon2, lat2 = np.meshgrid(lon, lat)
    h = (500.0 + 2500.0 * rng.random(lon2.shape)).astype(np.float32)
    X, Y, Z = transform.lonlat2ecef(lon2, lat2, h, ellps=ellps)
    tr = transform.TransformerEcef2enu(lon_or=lon.mean(), lat_or=lat.mean(), ellps=ellps)
    xe, ye, ze = transform.ecef2enu(X, Y, Z, tr)
    vn = direction.surf_norm(lon2, lat2)
    vno = direction.north_dir(X, Y, Z, vn, ellps=ellps)
    vn_enu = transform.ecef2enu_vector(vn, tr)
    vno_enu = transform.ecef2enu_vector(vno, tr)
    rot = transform.rotation_matrix_glob2loc(vno_enu[1:-1, 1:-1], vn_enu[1:-1, 1:-1])
    with contextlib.redirect_stdout(io.StringIO()):
        t_plane = topo_param.slope_plane_meth(xe, ye, ze, rot_mat=rot, output_rot=False)
        t_plane_rot = topo_param.slope_plane_meth(xe, ye, ze, rot_mat=rot, output_rot=True)
        t_vec_rot = topo_param.slope_vector_meth(xe, ye, ze, rot_mat=rot, output_rot=True)
    t_vec = topo_param.slope_vector_meth(xe, ye, ze)
    k = ellps + "_"
    out.update({k + "lon": lon2, k + "lat": lat2, k + "h": h, k + "X": X, k + "Y": Y, k + "Z": Z,
                k + "origin": np.array([tr.lon_or, tr.lat_or, tr.x_ecef_or, tr.y_ecef_or, tr.z_ecef_or]),
                k + "x_enu": xe, k + "y_enu": ye, k + "z_enu": ze, k + "norm_ecef": vn, k + "north_ecef": vno,
                k + "norm_enu": vn_enu, k + "north_enu": vno_enu, k + "rot": rot, k + "tilt_plane": t_plane,
                k + "tilt_plane_rot": t_plane_rot, k + "tilt_vector": t_vec, k + "tilt_vector_rot": t_vec_rot})
np.savez_compressed(os.path.join(here, "prep_reference.npz"), **{k: np.asarray(v) for k, v in out.items()})
print("prep fixtures:", len(out), "arrays")

# ---- Swiss projection (LV95) <-> WGS84 (transform.pyx:266-432) and auxiliary.pad_buffer ---------------------
rng = np.random.default_rng(20220622)
sw = {}
lon = rng.uniform(5.8, 10.6, (9, 13)); lat = rng.uniform(45.7, 47.9, (9, 13))
h = rng.uniform(190.0, 4600.0, (9, 13)).astype(np.float32)
e, n, h_ch = transform.wgs2swiss(lon, lat, h)
sw.update(lon=lon, lat=lat, h_wgs=h, e=np.asarray(e), n=np.asarray(n), h_ch=np.asarray(h_ch))
e2 = rng.uniform(2.48e6, 2.84e6, 57); n2 = rng.uniform(1.07e6, 1.30e6, 57)
h2 = rng.uniform(190.0, 4600.0, 57).astype(np.float32)
lo2, la2, hw2 = transform.swiss2wgs(e2, n2, h2)
sw.update(e2=e2, n2=n2, h_ch2=h2, lon2=np.asarray(lo2), lat2=np.asarray(la2), h_wgs2=np.asarray(hw2))
np.savez_compressed(os.path.join(here, "swiss_reference.npz"), **sw)
print("swiss_reference.npz:", {k: (v.dtype.name, v.shape) for k, v in sw.items()})
