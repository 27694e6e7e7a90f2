"""ctypes front end of the CPU oracle (oracle/hz_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of hz_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package ``horayzon_amd``.

The Python signatures mirror the reference boundary (horizon.pyx:29-49,
shadow.pyx:27-38,149-200, topo_param.pyx:377-409) so that the parity tests
read like calls to the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhz_oracle.so")
_lib = None


def build(force=False):
    """Compile libhz_oracle.so with gcc (oracle/Makefile tracks hz_oracle.c and the shared hz_crmath.h)."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []) + ["libhz_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        i32p = C.POINTER(C.c_int32)
        u64p = C.POINTER(C.c_uint64)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, i32p, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_occluded_batch.argtypes = [C.c_void_p, C.c_int64, fp, fp, fp, C.c_int, u8p]
        L.orc_tables.restype = C.c_int
        L.orc_tables.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, fp, fp,
                                 C.c_int, fp, fp, fp, fp]
        L.orc_horizon_gridded.restype = C.c_int
        L.orc_horizon_gridded.argtypes = [
            fp, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int,
            C.c_int, C.c_float, C.c_float, C.c_char_p, C.c_char_p, fp, C.c_int,
            i32p, C.c_int, C.c_float, u8p, C.c_float, C.c_float,
            C.c_int, C.c_int, C.c_int, C.c_int, u64p]
        L.orc_closest_batch.argtypes = [C.c_void_p, C.c_int64, fp, fp, fp, C.c_int, u8p, fp]
        L.orc_horizon_locations.restype = C.c_int
        L.orc_horizon_locations.argtypes = [
            fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float,
            C.c_char_p, C.c_char_p, C.c_float, fp, C.c_int, C.c_int, u64p]
        L.orc_terrain_create.restype = C.c_void_p
        L.orc_terrain_create.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp,
                                         C.c_int, C.c_int, fp, fp, u8p, C.c_float,
                                         C.c_float, C.c_int]
        L.orc_terrain_destroy.argtypes = [C.c_void_p]
        L.orc_terrain_shadow.argtypes = [C.c_void_p, fp, u8p, C.c_int, u64p]
        L.orc_terrain_sw_dir_cor.argtypes = [C.c_void_p, fp, fp, C.c_int, u64p]
        L.orc_sky_view_factor.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_libm.argtypes = [C.c_int]
        L.orc_set_quad_order.argtypes = [C.c_int]
        L.orc_set_box_start.argtypes = [C.c_float]
        L.orc_set_den_noise.argtypes = [C.c_float]
        L.orc_crmath_sweep.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, u64p]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


MODE_BVH, MODE_BRUTE, MODE_BRUTE_F64 = 0, 1, 2


def num_threads():
    return lib().orc_num_threads()


def set_libm(platform):
    """False / 0: hz_crmath.h (shared with the HIP kernels); True / 1: the platform's float routines (what the reference
    calls); 2: the oracle's own evaluation of the correctly rounded float through the long double libm (shares no code with
    the kernels)."""
    lib().orc_set_libm(int(platform))


TRI_MODES = {"plain": 0, "embree_fma_rcp": 1, "moeller_trumbore": 2, "plain_fma": 3}   # "plain_fma" = a library built with -DHZ_TRI_FMA


def set_tri_mode(mode="plain"):
    """Triangle test every later query uses (sensitivity diagnostics; "plain" is the contract)."""
    L = lib()
    L.orc_set_tri_mode.argtypes = [C.c_int]
    L.orc_set_tri_mode(TRI_MODES[mode])


def set_tri_compare(mode=None):
    """Keep the active test's decisions but also evaluate every ray with ``mode`` and count the rays whose
    decision differs (``tri_compare_counts``); ``None`` switches the comparison off.  Resets the counts."""
    L = lib()
    L.orc_set_tri_compare.argtypes = [C.c_int]
    L.orc_set_tri_compare(-1 if mode is None else TRI_MODES[mode])


def tri_compare_counts():
    """(rays compared, rays whose hit decision differed)."""
    L = lib()
    out = (C.c_uint64 * 2)()
    L.orc_tri_compare_counts.argtypes = [C.POINTER(C.c_uint64)]
    L.orc_tri_compare_counts(out)
    return int(out[0]), int(out[1])


def set_quad_order(embree_quad):
    """False (default): second triangle of a DEM quad as (b, d, c), the explicit "triangle" topology
    (horizon_comp.cpp:142-148).  True: as (d, c, b), the order Embree's quad / grid intersector forms from the quad
    (v0, v1, v2, v3) of horizon_comp.cpp:165-172 -- same triangle, rotated vertices, different rounding."""
    lib().orc_set_quad_order(int(bool(embree_quad)))


DEN_NOISE = 2.0 ** -20     # hz_oracle.c: ORC_DEN_NOISE = hz_common.h: HZ_DEN_NOISE


def set_den_noise(k=DEN_NOISE):
    """Threshold of the triangle test's parallel check |den| > k (|nx dx| + |ny dy| + |nz dz|); 0 = the exact den != 0 of rounds 1-4."""
    lib().orc_set_den_noise(float(k))


BOX_START_PADS = 16.0      # hz_oracle.c: ORC_BOX_START_PADS = hz_common.h: HZ_BOX_START_PADS


def set_box_start(tau_pads=BOX_START_PADS):
    """The tree's box tests run over [-tau, tfar + tau], tau = tau_pads * pad.  The default
    (16 pads) is the contract; 0 restores the round-4 tree, which culls the grazing-at-the-origin counter-example."""
    lib().orc_set_box_start(float(tau_pads))


def crmath_sweep(which, lo, hi, y=0.0):
    """Exhaustive sweep of hz_crmath.h over the floats in [lo, hi] (which: "acos", "tan", "cos", "sin", "pow").
    Returns (values, differing from the rounded float64 libm result, differing from the platform float routine)."""
    out = np.zeros(3, np.uint64)
    lib().orc_crmath_sweep(("acos", "tan", "cos", "sin", "pow").index(which), lo, hi, y,
                           out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return int(out[0]), int(out[1]), int(out[2])


def div_const_sweep(c, lo_bits, hi_bits):
    """hz_crmath.h's four-instruction division by the constant c > 0 against the IEEE division, for every float whose bit pattern
    lies in [lo_bits, hi_bits] (promoted to double).  Returns (values, differing results)."""
    out = np.zeros(2, np.uint64)
    f = lib().orc_div_const_sweep
    f.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    f.restype = None
    f(float(c), int(lo_bits), int(hi_bits), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return int(out[0]), int(out[1])


def tables(azim_num, hori_acc, elev_ang_low_lim, dist_search):
    """Trig tables exactly as horizon_comp.cpp:711-731 builds them."""
    L = lib()
    n = L.orc_tables(azim_num, hori_acc, elev_ang_low_lim, dist_search,
                     None, None, 0, None, None, None, None)
    out = {k: np.empty(azim_num, np.float32) for k in ("azim_sin", "azim_cos")}
    out.update({k: np.empty(n, np.float32) for k in ("elev_ang", "elev_sin", "elev_cos")})
    sc = np.empty(4, np.float32)
    L.orc_tables(azim_num, hori_acc, elev_ang_low_lim, dist_search,
                 _f(out["azim_sin"]), _f(out["azim_cos"]), n, _f(out["elev_ang"]),
                 _f(out["elev_sin"]), _f(out["elev_cos"]), _f(sc))
    out.update(elev_num=n, hori_acc=sc[0], low=sc[1], up=sc[2], dist=sc[3])
    return out


class Scene:
    """Bare occlusion queries against the grid mesh (+ optional TIN)."""

    def __init__(self, vert_grid, dem_dim_0, dem_dim_1, vert_simp=None, tri_ind_simp=None):
        vg = np.ascontiguousarray(vert_grid, np.float32)
        assert vg.size >= 3 * dem_dim_0 * dem_dim_1
        if vert_simp is None:
            self._h = lib().orc_scene_create(_f(vg), dem_dim_0, dem_dim_1, None, 0, None, 0)
        else:
            vs = np.ascontiguousarray(vert_simp, np.float32)
            ts = np.ascontiguousarray(tri_ind_simp, np.int32)
            self._h = lib().orc_scene_create(_f(vg), dem_dim_0, dem_dim_1, _f(vs),
                                             vs.size // 3, _i32(ts), ts.size // 3)

    def occluded(self, org, dirs, tfar, mode=MODE_BVH):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = org.shape[0]
        tf = np.ascontiguousarray(np.broadcast_to(np.asarray(tfar, np.float32), (n,)))
        out = np.empty(n, np.uint8)
        lib().orc_occluded_batch(self._h, n, _f(org), _f(dirs), _f(tf), mode, _u8(out))
        return out.astype(bool)

    def closest(self, org, dirs, tfar, mode=MODE_BVH):
        """(hit, distance) of the closest accepted triangle (rtcIntersect1 semantics)."""
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = org.shape[0]
        tf = np.ascontiguousarray(np.broadcast_to(np.asarray(tfar, np.float32), (n,)))
        hit = np.empty(n, np.uint8)
        dist = np.empty(n, np.float32)
        lib().orc_closest_batch(self._h, n, _f(org), _f(dirs), _f(tf), mode, _u8(hit), _f(dist))
        return hit.astype(bool), dist

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None


def horizon_locations(vert_grid, dem_dim_0, dem_dim_1, coords, vec_norm, vec_north,
                      dist_search, azim_num=360, hori_acc=0.25, ray_algorithm="binary_search",
                      geom_type="grid", elev_ang_low_lim=-89.98,
                      ray_org_elev=np.array([0.01], dtype=np.float32), hori_dist_out=False,
                      *, mode=MODE_BVH, return_stats=False):
    """CPU restatement of horayzon.horizon.horizon_locations (horizon.pyx:218-370)."""
    vert_grid = np.ascontiguousarray(vert_grid, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    vec_norm = np.ascontiguousarray(vec_norm, np.float32)
    vec_north = np.ascontiguousarray(vec_north, np.float32)
    n = coords.shape[0]
    roe = np.ascontiguousarray(ray_org_elev, np.float32)
    if len(roe) != n:
        roe = np.repeat(roe, n)
    hori = np.full((n, azim_num), np.nan, np.float32)
    dist = np.full((n if hori_dist_out else 1, azim_num), np.nan, np.float32)
    stats = np.zeros(4, np.uint64)
    rc = lib().orc_horizon_locations(
        _f(vert_grid), dem_dim_0, dem_dim_1, _f(coords), _f(vec_norm), _f(vec_north), _f(hori), _f(dist),
        n, azim_num, dist_search, hori_acc, ray_algorithm.encode(), geom_type.encode(), elev_ang_low_lim,
        _f(roe), int(bool(hori_dist_out)), mode, stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc == 1:
        raise ValueError("invalid input argument for ray_algorithm")
    if rc == 2:
        raise TypeError("horizon detection algorithm 'guess_constant' not implemented for horizon "
                        "distance computation")
    azim = np.empty(azim_num, np.float32)
    for i in range(azim_num):
        azim[i] = ((2 * np.pi) / azim_num * i)
    out = (hori, dist, azim) if hori_dist_out else (hori, azim)
    if return_stats:
        return out + (dict(rays=int(stats[0]), guards=int(stats[1]), found=int(stats[2])),)
    return out


def horizon_gridded(vert_grid, dem_dim_0, dem_dim_1, vec_norm, vec_north,
                    offset_0, offset_1, dist_search, azim_num=360, hori_acc=0.25,
                    ray_algorithm="guess_constant", geom_type="grid",
                    vert_simp=None, num_vert_simp=1, tri_ind_simp=None,
                    num_tri_simp=1, elev_ang_low_lim=-15.0, mask=None,
                    hori_fill=0.0, ray_org_elev=0.01, *, mode=MODE_BVH,
                    rows=None, slab_only=False, count_work=False, return_stats=False):
    """CPU restatement of horayzon.horizon.horizon_gridded (horizon.pyx:29-197).

    ``rows=(begin, end)`` restricts the computation to a slab of inner-domain rows;
    with ``slab_only`` only that slab is allocated and returned (large tiles)."""
    if vert_simp is None:
        vert_simp = np.zeros(4, np.float32)
    if tri_ind_simp is None:
        tri_ind_simp = np.zeros(4, np.int32)
    vert_grid = np.ascontiguousarray(vert_grid, np.float32)
    vec_norm = np.ascontiguousarray(vec_norm, np.float32)
    vec_north = np.ascontiguousarray(vec_north, np.float32)
    vert_simp = np.ascontiguousarray(vert_simp, np.float32)
    tri_ind_simp = np.ascontiguousarray(tri_ind_simp, np.int32)
    d0, d1 = vec_norm.shape[:2]
    if mask is None:
        mask = np.ones((d0, d1), np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    rb, re = (0, d0) if rows is None else rows
    stats = np.zeros(8, np.uint64)
    if slab_only:
        hori = np.full((max(re - rb, 0), d1, azim_num), np.nan, np.float32)
        # the C driver indexes by global cell: shift the slab back by rb rows
        hori_ptr = C.cast(C.c_void_p(hori.ctypes.data - 4 * rb * d1 * azim_num), C.POINTER(C.c_float))
    else:
        hori = np.full((d0, d1, azim_num), np.nan, np.float32)
        hori_ptr = _f(hori)
    rc = lib().orc_horizon_gridded(
        _f(vert_grid), dem_dim_0, dem_dim_1, _f(vec_norm), _f(vec_north),
        offset_0, offset_1, hori_ptr, d0, d1, azim_num, dist_search, hori_acc,
        ray_algorithm.encode(), geom_type.encode(), _f(vert_simp), num_vert_simp,
        _i32(tri_ind_simp), num_tri_simp, elev_ang_low_lim, _u8(mask), hori_fill,
        ray_org_elev, mode, rb, re, int(count_work),
        stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise ValueError("invalid input argument for ray_algorithm")
    azim = np.empty(azim_num, np.float32)
    for i in range(azim_num):                      # horizon.pyx:191-195
        azim[i] = ((2 * np.pi) / azim_num * i)
    if return_stats:
        return hori, azim, dict(rays=int(stats[0]), guards=int(stats[1]),
                                nodes=int(stats[2]), tris=int(stats[3]),
                                t_build_s=float(stats[4]) * 1e-9, t_rays_s=float(stats[5]) * 1e-9)
    return hori, azim


class Terrain:
    """CPU restatement of horayzon.shadow.Terrain (shadow.pyx:17-200)."""

    def __init__(self, mode=MODE_BVH):
        self._h = None
        self._mode = mode
        self.rays = 0

    def initialise(self, vert_grid, dem_dim_0, dem_dim_1, offset_0, offset_1,
                   vec_tilt, vec_norm, surf_enl_fac, elevation, mask,
                   geom_type="grid", sw_dir_cor_fill=np.nan, ang_max=89.0,
                   refrac_cor=False):
        in0, in1 = vec_tilt.shape[:2]
        a = [np.ascontiguousarray(x, np.float32) for x in
             (vert_grid, vec_tilt, vec_norm, surf_enl_fac, elevation)]
        m = np.ascontiguousarray(mask, np.uint8)
        self._shape = (in0, in1)
        self._h = lib().orc_terrain_create(_f(a[0]), dem_dim_0, dem_dim_1, offset_0,
                                           offset_1, _f(a[1]), _f(a[2]), in0, in1,
                                           _f(a[3]), _f(a[4]), _u8(m),
                                           sw_dir_cor_fill, ang_max, int(refrac_cor))

    def shadow(self, sun_position, shadow_buffer):
        sp = np.ascontiguousarray(sun_position, np.float32)
        st = np.zeros(1, np.uint64)
        lib().orc_terrain_shadow(self._h, _f(sp), _u8(shadow_buffer), self._mode,
                                 st.ctypes.data_as(C.POINTER(C.c_uint64)))
        self.rays = int(st[0])

    def sw_dir_cor(self, sun_position, sw_dir_cor_buffer):
        sp = np.ascontiguousarray(sun_position, np.float32)
        st = np.zeros(1, np.uint64)
        lib().orc_terrain_sw_dir_cor(self._h, _f(sp), _f(sw_dir_cor_buffer), self._mode,
                                     st.ctypes.data_as(C.POINTER(C.c_uint64)))
        self.rays = int(st[0])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_terrain_destroy(self._h)
            self._h = None


def sky_view_factor(azim, hori, vec_tilt):
    """CPU restatement of horayzon.topo_param.sky_view_factor (topo_param.pyx:377-460)."""
    if (len(azim) != hori.shape[2]) or (hori.shape[:2] != vec_tilt.shape[:2]) \
            or (vec_tilt.shape[2] != 3):
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    if ((azim.dtype != "float32") or (hori.dtype != "float32")
            or (vec_tilt.dtype != "float32")):
        raise ValueError("Input array(s) has/have incorrect data type(s)")
    if len(azim) < 2:    # azim[1] - azim[0] is read (topo_param.pyx:433): out of bounds in the reference
        raise ValueError("Inconsistent/incorrect shapes of input arrays")
    azim = np.ascontiguousarray(azim)
    hori = np.ascontiguousarray(hori)
    vec_tilt = np.ascontiguousarray(vec_tilt)
    svf = np.empty(hori.shape[:2], np.float32)
    lib().orc_sky_view_factor(_f(azim), _f(hori), _f(vec_tilt), hori.shape[0],
                              hori.shape[1], hori.shape[2], _f(svf))
    return svf
