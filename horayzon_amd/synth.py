"""Synthetic inputs for the horizon / shadow path (tests and bench.py).

There is no DEM data and no network in the build environment, so the
workloads of BASELINE.json are generated here:

* ``gaussian_hill``  -- configs 1/2: 200 x 200 planar DEM, dx = dy = 50 m,
  one Gaussian hill of 1500 m.
* ``fractal_tile``   -- config 3/4/5: SRTM-like 1-arc-second tile (spectral
  synthesis, integer metres), planar approximation at 46 deg N.
* ``sun_positions``  -- config 4: analytic diurnal sun path.

``pack_vertices`` produces the ``vert_grid`` layout the reference boundary
expects (interleaved float32 xyz + >= 16 trailing zeros; the layout defined
by reference horayzon/auxiliary.py:49-95, own implementation).
"""
import numpy as np


def pack_vertices(x, y, z):
    """(y, x)-shaped float32 coordinate arrays -> 1-D interleaved xyz buffer."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    z = np.asarray(z, np.float32)
    if not (x.shape == y.shape == z.shape) or x.ndim != 2:
        raise ValueError("x, y, z must be two-dimensional and equally shaped")
    n = x.size
    buf = np.zeros(3 * n + 16 + (-(3 * n) % 4), np.float32)
    buf[0:3 * n:3] = x.ravel()
    buf[1:3 * n:3] = y.ravel()
    buf[2:3 * n:3] = z.ravel()
    return buf


def planar_frames(dim_in_0, dim_in_1):
    """vec_norm = (0, 0, 1), vec_north = (0, 1, 0) as in the planar examples
    (reference examples/horizon/gridded_planar_DEM.py:71-76)."""
    vec_norm = np.zeros((dim_in_0, dim_in_1, 3), np.float32)
    vec_norm[:, :, 2] = 1.0
    vec_north = np.zeros((dim_in_0, dim_in_1, 3), np.float32)
    vec_north[:, :, 1] = 1.0
    return vec_norm, vec_north


def gaussian_hill(n=200, dx=50.0, height=1500.0, sigma=1500.0, offset=10):
    """Config 1/2 input. Returns a dict of horizon_gridded keyword inputs."""
    x = (np.arange(n, dtype=np.float32) * np.float32(dx)).astype(np.float32)
    y = (np.arange(n, dtype=np.float32) * np.float32(dx)).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    xm, ym = np.float32(x.mean()), np.float32(y.mean())
    z = (np.float32(height) * np.exp(-((xx - xm) ** 2 + (yy - ym) ** 2)
                                     / np.float32(2.0 * sigma ** 2))).astype(np.float32)
    vec_norm, vec_north = planar_frames(n - 2 * offset, n - 2 * offset)
    return dict(vert_grid=pack_vertices(xx, yy, z), dem_dim_0=n, dem_dim_1=n,
                vec_norm=vec_norm, vec_north=vec_north, offset_0=offset,
                offset_1=offset, x=x, y=y, z=z)


def fractal_elevation(n0, n1, hurst=0.8, seed=20220621, z_min=200.0, z_max=4000.0):
    """Spectral-synthesis fractal surface, power spectrum ~ k^-(2H+1),
    scaled to [z_min, z_max] and rounded to integer metres (SRTM is int16)."""
    rng = np.random.default_rng(seed)
    m0, m1 = 1 << int(np.ceil(np.log2(n0))), 1 << int(np.ceil(np.log2(n1)))
    k0 = np.fft.fftfreq(m0)[:, None]
    k1 = np.fft.rfftfreq(m1)[None, :]
    k = np.sqrt(k0 * k0 + k1 * k1)
    k[0, 0] = 1.0
    amp = k ** (-(2.0 * hurst + 1.0) / 2.0 - 0.5)   # 2-D field: amplitude ~ k^-(H+1)
    amp[0, 0] = 0.0
    phase = rng.uniform(0.0, 2.0 * np.pi, size=amp.shape)
    spec = (amp * np.exp(1j * phase)).astype(np.complex64)
    f = np.fft.irfft2(spec, s=(m0, m1))[:n0, :n1]
    f = (f - f.min()) / (f.max() - f.min())
    return np.rint(z_min + (z_max - z_min) * f).astype(np.float32)


def fractal_tile(n=3601, offset=16, seed=20220621, dy=30.87, dx=21.44, plain_fraction=0.0):
    """Config 3 input (planar variant): n x n 1-arc-second tile at 46 deg N,
    rows run north -> south (y decreasing), like a DEM raster.  ``plain_fraction`` > 0 scales the relief of that share
    of the rows (the northern ones) down to 3 % -- rolling lowland next to high relief, with a smooth ramp over a tenth of
    the rows between them (a vertical step instead would put a 3 km wall next to the plain: the cells at its foot then
    need hundreds of rays per azimuth and their one workgroup, not the slab split, decides the run time of a small job,
    profiles/r04/half_plain_dem_slab_costs.jsonl) -- a deliberately inhomogeneous DEM for the load-balance experiments
    of bench.py."""
    z = fractal_elevation(n, n, seed=seed)
    if plain_fraction > 0.0:
        r = np.arange(n, dtype=np.float64) / max(n - 1, 1)
        t = np.clip((r - (min(plain_fraction, 1.0) - 0.05)) / 0.1, 0.0, 1.0)
        w = 0.03 + 0.97 * (3.0 * t * t - 2.0 * t * t * t)              # smoothstep from the lowland to the mountains
        zmin = float(z.min())
        z = np.rint(zmin + (z - zmin) * w[:, None]).astype(np.float32)
    x = (np.arange(n, dtype=np.float64) * dx - 0.5 * (n - 1) * dx).astype(np.float32)
    y = (0.5 * (n - 1) * dy - np.arange(n, dtype=np.float64) * dy).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    vec_norm, vec_north = planar_frames(n - 2 * offset, n - 2 * offset)
    return dict(vert_grid=pack_vertices(xx, yy, z), dem_dim_0=n, dem_dim_1=n,
                vec_norm=vec_norm, vec_north=vec_north, offset_0=offset,
                offset_1=offset, x=x, y=y, z=z)


def tilt_from_planar_dem(x, y, z, offset):
    """Surface normal ("vec_tilt") and surface enlargement factor from centred
    differences on a planar DEM -- input preparation for Terrain / SVF in the
    synthetic workloads (the reference uses topo_param.slope_plane_meth, a
    3x3 least-squares plane; this is NOT that algorithm, only a stand-in that
    yields valid unit normals)."""
    z = np.asarray(z, np.float64)
    dzdx = (z[1:-1, 2:] - z[1:-1, :-2]) / (np.asarray(x, np.float64)[2:] - np.asarray(x, np.float64)[:-2])[None, :]
    dzdy = (z[2:, 1:-1] - z[:-2, 1:-1]) / (np.asarray(y, np.float64)[2:] - np.asarray(y, np.float64)[:-2])[:, None]
    nrm = np.sqrt(dzdx ** 2 + dzdy ** 2 + 1.0)
    t = np.stack([-dzdx / nrm, -dzdy / nrm, 1.0 / nrm], axis=2)
    o = offset - 1
    sl = (slice(o, t.shape[0] - o), slice(o, t.shape[1] - o))
    vec_tilt = np.ascontiguousarray(t[sl], np.float32)
    vec_tilt /= np.sqrt((vec_tilt.astype(np.float64) ** 2).sum(axis=2, keepdims=True)).astype(np.float32)
    surf_enl_fac = np.ascontiguousarray(nrm[sl], np.float32)
    return vec_tilt, surf_enl_fac


def sun_positions(num=144, lat_deg=46.0, day_of_year=172, dist=1.496e11):
    """Config 4: one day at equal steps; declination = -23.44 deg * cos(2 pi (d + 10) / 365),
    hour angle 15 deg/h.  Returns float32 (num, 3) ENU positions (east, north, up)."""
    decl = np.deg2rad(-23.44) * np.cos(2.0 * np.pi * (day_of_year + 10) / 365.0)
    lat = np.deg2rad(lat_deg)
    hours = (np.arange(num) + 0.5) * 24.0 / num
    ha = np.deg2rad(15.0 * (hours - 12.0))
    sin_alt = np.sin(lat) * np.sin(decl) + np.cos(lat) * np.cos(decl) * np.cos(ha)
    alt = np.arcsin(sin_alt)
    # azimuth clockwise from north
    az = np.arctan2(-np.cos(decl) * np.sin(ha),
                    np.sin(decl) * np.cos(lat) - np.cos(decl) * np.sin(lat) * np.cos(ha))
    sun = dist * np.stack([np.cos(alt) * np.sin(az), np.cos(alt) * np.cos(az), np.sin(alt)], axis=1)
    return sun.astype(np.float32), alt, az
