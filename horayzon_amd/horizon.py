"""horayzon.horizon -- terrain horizon on MI355X.

Host-side mirror of the reference's Cython boundary ``horayzon/horizon.pyx``:
same function name, positional order, defaults, validation order and
exception classes (horizon.pyx:29-49, 109-156); the computation runs in
hand-written HIP kernels behind the C ABI (hz_horizon_gridded).
"""
import ctypes as C

import numpy as np

from . import _lib
from . import _validate as V
from ._lib import Scene, hz_opts, hz_stats, ptr

last_stats = None   # hz_stats of the most recent call as a dict (timers, ray count)
# test hook: defaults of the launch-schedule options ("left_min", "persist_grid", "left_cap_test": hz_opts) for calls that do not
# pass them -- lets the parity tests run their cases under other schedules without touching every call
schedule_overrides = {}


def _check_f32(a, ndim, name):
    # Cython's typed-buffer arguments raise ValueError on dtype/ndim mismatch
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if a.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % a.dtype.name)


def horizon_gridded(vert_grid, dem_dim_0, dem_dim_1, vec_norm, vec_north,
                    offset_0, offset_1, dist_search, azim_num=360, hori_acc=0.25,
                    ray_algorithm="guess_constant", geom_type="grid",
                    vert_simp=np.array([0.0, 0.0, 0.0, 0.0], dtype=np.float32),
                    num_vert_simp=1,
                    tri_ind_simp=np.array([0, 0, 0, 0], dtype=np.int32),
                    num_tri_simp=1, elev_ang_low_lim=-15.0, mask=None,
                    hori_fill=0.0, ray_org_elev=0.01, *, device=0, verbose=False,
                    scene=None, svf_vec_tilt=None, svf_only=False, rows=None, count_work=False, devices=None,
                    _top_nodes=-1, _regroup=-1, _hit_cache=True, _chunk_rows=0, _near_skip=True, _level_stack=False,
                    _verify_near=False, _left_min=0, _persist_grid=0):
    """Horizon computation for gridded domain.

    Parameters, units and return values are those of the reference
    (horizon.pyx:50-106): ``hori_buffer`` float32 (y, x, azim_num) [radian],
    ``azim`` float32 (azim_num) [radian].

    Additional keyword-only arguments (not in the reference): ``device`` (HIP
    ordinal), ``verbose`` (print the reference's stdout report), ``scene`` (a
    prebuilt ``Scene`` to skip the BVH build), ``svf_vec_tilt`` (tilted normals;
    when given the sky view factor is accumulated in the same kernel and
    returned as a third value; with ``svf_only`` the horizon array is never materialised --
    it lives in a bounded device buffer, chunk by chunk -- and ``None`` is returned in its place:
    the 14401^2 mosaic would need 298 GB of horizon), ``rows`` ((begin, end) slab of inner-domain rows
    to compute; the rest of ``hori_buffer`` stays NaN), ``count_work``, ``devices``
    ("all" or a sequence of HIP ordinals: the inner-domain rows are split into one slab per
    entry, balanced by ``mask``, and each slab is computed by its own host thread on its own
    GPU -- the scene is built per device, nothing is exchanged, every thread copies its rows
    straight into the returned array; the multi-process form of the same sharding is
    ``horayzon_amd.dist``).
    """
    global last_stats
    _check_f32(vert_grid, 1, "vert_grid")
    _check_f32(vec_norm, 3, "vec_norm")
    _check_f32(vec_north, 3, "vec_north")
    _check_f32(vert_simp, 1, "vert_simp")
    if not isinstance(tri_ind_simp, np.ndarray) or tri_ind_simp.ndim != 1 or tri_ind_simp.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch, expected 'int32_t' for tri_ind_simp")
    if mask is not None and (not isinstance(mask, np.ndarray) or mask.ndim != 2):
        raise ValueError("Buffer has wrong number of dimensions (expected 2) for mask")

    # Consistency and validity of the arguments: the reference's checks, classes, messages and order (horizon.pyx:109-153)
    if mask is None:
        mask = np.ones(vec_norm.shape[:2], dtype=np.uint8)
    V.run((
        (ValueError, "inconsistency between input arguments vert_grid, dem_dim_0 and dem_dim_1",
         lambda: not V.fits_grid(len(vert_grid), dem_dim_0, dem_dim_1)),
        (ValueError, "inconsistency between input arguments dem_dim_0, dem_dim_1, offset_0, offset_1 and vec_norm",
         lambda: not V.window_inside(offset_0, offset_1, vec_norm.shape, dem_dim_0, dem_dim_1)),
        (ValueError, V.MSG_NORTH, lambda: not V.same_leading_shape((vec_norm, vec_north), 3, 3)),
        (ValueError, V.MSG_ALG, lambda: ray_algorithm not in V.ALGORITHMS),
        (ValueError, V.MSG_GEOM, lambda: geom_type not in V.GEOMETRIES),
        (ValueError, "inconsistency between input arguments vert_simp and num_vert_simp", lambda: len(vert_simp) < num_vert_simp * 3),
        (ValueError, "inconsistency between input arguments tri_ind_simp and num_tri_simp", lambda: len(tri_ind_simp) < num_tri_simp * 3),
        (ValueError, "triangle indices of simplified outer domain exceed number of vertices",
         lambda: tri_ind_simp.max() > num_vert_simp - 1),
        (ValueError, V.MSG_ACC, lambda: hori_acc > 10.0),
        (ValueError, "shape of mask is inconsistent with other input", lambda: mask.shape[:2] != vec_norm.shape[:2]),
        (TypeError, V.MSG_MASK_TYPE, lambda: mask.dtype != "uint8"),
        (TypeError, V.MSG_ELEV, lambda: ray_org_elev < 0.005),
        (ValueError, V.MSG_DIM_LIMIT, lambda: max(dem_dim_0, dem_dim_1) > V.DIM_LIMIT),
        (ValueError, "vertex buffer vert_simp is larger than 16 GB", lambda: vert_simp.nbytes > 16.0e9),
    ))

    # Ensure that passed arrays are contiguous in memory (horizon.pyx:159-163)
    vert_grid = np.ascontiguousarray(vert_grid)
    vec_norm = np.ascontiguousarray(vec_norm)
    vec_north = np.ascontiguousarray(vec_north)
    vert_simp = np.ascontiguousarray(vert_simp)
    tri_ind_simp = np.ascontiguousarray(tri_ind_simp)
    mask = np.ascontiguousarray(mask)

    dim_in_0, dim_in_1 = vec_norm.shape[0], vec_norm.shape[1]
    # Allocate horizon array (horizon.pyx:170-173)
    if svf_only and svf_vec_tilt is None:
        raise ValueError("'svf_only' needs 'svf_vec_tilt'")
    hori_buffer = None if svf_only else np.empty((dim_in_0, dim_in_1, azim_num), dtype=np.float32)
    if rows is not None and hori_buffer is not None:
        hori_buffer.fill(np.nan)   # only a slab is written; every cell is written otherwise
                                   # (masked ones get hori_fill), so the 18 GB pre-fill is skipped

    opts = hz_opts()
    opts.device = device
    opts.verbose = int(bool(verbose))
    opts.top_nodes = _top_nodes
    opts.regroup = _regroup
    opts.no_hit_cache = 0 if _hit_cache else 1
    opts.chunk_rows = _chunk_rows
    # True: certificates when the scene allows them; False: never; "force": even for a mesh that is not a height field (tests)
    opts.no_near_skip = -1 if _near_skip == "force" else (0 if _near_skip else 1)
    opts.level_stack = int(_level_stack)      # True / 1: level stack from the start; -n: fast stack of n entries (tests)
    # schedule of the launches (results never depend on it; tests run the parity cases under several: schedule_overrides)
    opts.left_min = int(_left_min or schedule_overrides.get("left_min", 0))            # leftover cells: byte l = hand-over threshold of level l; 0: default; < 0: off
    opts.persist_grid = int(_persist_grid or schedule_overrides.get("persist_grid", 0))    # 0: persistent waves; < 0: one tile per workgroup; n > 0: n workgroups
    opts.left_cap_test = int(schedule_overrides.get("left_cap_test", 0))
    opts.left_tune = int(schedule_overrides.get("left_tune", 0))
    n_verify = int(_verify_near)              # False / 0: off; True / 1: every shortened ray; N: one of every N (rounded up to 2^k)
    if n_verify < 0 or n_verify != _verify_near:
        raise ValueError("_verify_near must be a non-negative integer (or a bool)")
    opts.verify_near = n_verify
    opts.skip_hori = 1 if svf_only else 0
    opts.count_work = int(bool(count_work))
    if rows is not None:
        if len(rows) != 2 or not (0 <= int(rows[0]) < int(rows[1]) <= dim_in_0):
            raise ValueError("'rows' must be (begin, end) with 0 <= begin < end <= %d" % dim_in_0)
        opts.row_begin, opts.row_end = int(rows[0]), int(rows[1])
    svf = None
    if svf_vec_tilt is not None:
        _check_f32(svf_vec_tilt, 3, "svf_vec_tilt")
        if azim_num < 2:      # the azimuth spacing azim[1] - azim[0] does not exist (topo_param.pyx:433)
            raise ValueError("sky view factor needs azim_num >= 2")
        if svf_vec_tilt.shape != vec_norm.shape:
            raise ValueError("Inconsistent/incorrect shapes of input arrays")
        svf_vec_tilt = np.ascontiguousarray(svf_vec_tilt)
        svf = np.full((dim_in_0, dim_in_1), np.nan, dtype=np.float32)
        opts.svf = ptr(svf)
        opts.vec_tilt = ptr(svf_vec_tilt)
    stats = hz_stats()
    L = _lib.lib()

    def run(o, st):
        if scene is None:
            return L.hz_horizon_gridded(
                ptr(vert_grid), dem_dim_0, dem_dim_1, ptr(vec_norm), ptr(vec_north),
                offset_0, offset_1, ptr(hori_buffer), dim_in_0, dim_in_1, azim_num,
                dist_search, hori_acc, ray_algorithm.encode("utf-8"),
                geom_type.encode("utf-8"), ptr(vert_simp), num_vert_simp,
                ptr(tri_ind_simp), num_tri_simp, elev_ang_low_lim, ptr(mask),
                hori_fill, ray_org_elev, C.byref(o), C.byref(st))
        o.device = scene.device
        return L.hz_horizon_gridded_scene(
            scene._h, ptr(vec_norm), ptr(vec_north), offset_0, offset_1,
            ptr(hori_buffer), dim_in_0, dim_in_1, azim_num, dist_search, hori_acc,
            ray_algorithm.encode("utf-8"), elev_ang_low_lim, ptr(mask), hori_fill,
            ray_org_elev, C.byref(o), C.byref(st))

    if devices is None:
        rc = run(opts, stats)
    else:
        # one host thread per GPU, one row slab each (ctypes releases the GIL during the call;
        # the library keeps its error string and build arenas per thread)
        import threading
        from .dist import row_slabs
        if scene is not None or rows is not None:
            raise ValueError("'devices' cannot be combined with 'scene' or 'rows'")
        devs = list(range(_lib.device_count())) if isinstance(devices, str) and devices == "all" \
            else [int(d) for d in devices]
        if not devs:
            raise ValueError("'devices' is empty")
        slabs = [sl for sl in row_slabs(mask, len(devs))]
        part, errs = [None] * len(devs), []

        def work(idx):
            b, e = slabs[idx]
            if e <= b:
                return
            o = hz_opts.from_buffer_copy(opts)
            o.device, o.row_begin, o.row_end = devs[idx], b, e
            st = hz_stats()
            try:
                _lib.check(run(o, st))
                part[idx] = st
            except Exception as exc:      # re-raised in the calling thread
                errs.append(exc)
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(devs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        for st in part:                    # counts add up, times are the slowest device's
            if st is None:
                continue
            for name, _ in hz_stats._fields_:
                a, b = getattr(stats, name), getattr(st, name)
                setattr(stats, name, max(a, b) if name.startswith("t_") or name in ("bvh_height", "elev_num", "scene_bytes") else a + b)
        rc = 0
    _lib.check(rc)
    last_stats = stats.as_dict()

    # Recompute azimuth (horizon.pyx:191-195)
    azim = np.empty(azim_num, dtype=np.float32)
    for i in range(azim_num):
        azim[i] = ((2 * np.pi) / azim_num * i)

    if svf is not None:
        return hori_buffer, azim, svf
    return hori_buffer, azim


def horizon_locations(vert_grid, dem_dim_0, dem_dim_1, coords, vec_norm, vec_north,
                      dist_search, azim_num=360, hori_acc=0.25,
                      ray_algorithm="binary_search", geom_type="grid",
                      elev_ang_low_lim=-89.98,
                      ray_org_elev=np.array([0.01], dtype=np.float32),
                      hori_dist_out=False, *, device=0, verbose=False, scene=None):
    """Horizon computation for arbitrary locations.

    Parameters, units and return values are those of the reference
    (horizon.pyx:233-276): returns ``hori_buffer`` float32 (number of locations,
    azim_num) [radian] and ``azim``; with ``hori_dist_out`` also the distance to the
    horizon [metre] as second value.  Keyword-only extras: ``device``, ``verbose``, ``scene``."""
    global last_stats
    _check_f32(vert_grid, 1, "vert_grid")
    _check_f32(coords, 2, "coords")
    _check_f32(vec_norm, 2, "vec_norm")
    _check_f32(vec_north, 2, "vec_north")
    _check_f32(ray_org_elev, 1, "ray_org_elev")

    # Consistency and validity of the arguments: the reference's checks, classes, messages and order (horizon.pyx:279-312)
    V.run((
        (ValueError, "inconsistency between input arguments vert_grid, dem_dim_0 and dem_dim_1",
         lambda: not V.fits_grid(len(vert_grid), dem_dim_0, dem_dim_1)),
        (ValueError, "'number of dimensions and/or dimension length(s) of 'coords' incorrect",
         lambda: coords.ndim != 2 or coords.shape != (vec_norm.shape[0], 3)),
        (ValueError, V.MSG_NORTH, lambda: not V.same_leading_shape((vec_norm, vec_north), 2, 2)),
        (ValueError, V.MSG_ALG, lambda: ray_algorithm not in V.ALGORITHMS),
        (ValueError, V.MSG_GEOM, lambda: geom_type not in V.GEOMETRIES),
        (ValueError, V.MSG_ACC, lambda: hori_acc > 10.0),
        (ValueError, "length of array 'ray_org_elev' must be either one or correspond to the number of locations",
         lambda: len(ray_org_elev) not in (1, coords.shape[0])),
        (TypeError, V.MSG_ELEV, lambda: ray_org_elev.min() < 0.005),
        (TypeError, "horizon detection algorithm 'guess_constant' not implemented for horizon distance computation",
         lambda: bool(hori_dist_out) and ray_algorithm == "guess_constant"),
        (ValueError, V.MSG_DIM_LIMIT, lambda: max(dem_dim_0, dem_dim_1) > V.DIM_LIMIT),
    ))

    # Repeat array 'ray_org_elev' if necessary (horizon.pyx:315-316)
    if len(ray_org_elev) != coords.shape[0]:
        ray_org_elev = np.repeat(ray_org_elev, coords.shape[0])

    vert_grid = np.ascontiguousarray(vert_grid)
    coords = np.ascontiguousarray(coords)
    vec_norm = np.ascontiguousarray(vec_norm)
    vec_north = np.ascontiguousarray(vec_north)
    ray_org_elev = np.ascontiguousarray(ray_org_elev)

    num_loc = vec_norm.shape[0]
    hori_buffer = np.empty((num_loc, azim_num), dtype=np.float32)
    hori_buffer.fill(np.nan)
    hori_dist_buffer = np.empty((num_loc if hori_dist_out else 1, azim_num), dtype=np.float32)
    hori_dist_buffer.fill(np.nan)

    opts = hz_opts()
    opts.device = device if scene is None else scene.device
    opts.verbose = int(bool(verbose))
    stats = hz_stats()
    L = _lib.lib()
    if num_loc == 0:    # the reference's loop over locations simply does not run
        rc = 0
    elif scene is None:
        rc = L.hz_horizon_locations(
            ptr(vert_grid), dem_dim_0, dem_dim_1, ptr(coords), ptr(vec_norm), ptr(vec_north),
            ptr(hori_buffer), ptr(hori_dist_buffer) if hori_dist_out else None, num_loc, azim_num,
            dist_search, hori_acc, ray_algorithm.encode("utf-8"), geom_type.encode("utf-8"),
            elev_ang_low_lim, ptr(ray_org_elev), int(bool(hori_dist_out)), C.byref(opts), C.byref(stats))
    else:
        rc = L.hz_horizon_locations_scene(
            scene._h, ptr(coords), ptr(vec_norm), ptr(vec_north), ptr(hori_buffer),
            ptr(hori_dist_buffer) if hori_dist_out else None, num_loc, azim_num, dist_search, hori_acc,
            ray_algorithm.encode("utf-8"), elev_ang_low_lim, ptr(ray_org_elev), int(bool(hori_dist_out)),
            C.byref(opts), C.byref(stats))
    _lib.check(rc)
    last_stats = stats.as_dict()

    azim = np.empty(azim_num, dtype=np.float32)
    for i in range(azim_num):
        azim[i] = ((2 * np.pi) / azim_num * i)

    if hori_dist_out:
        return hori_buffer, hori_dist_buffer, azim
    else:
        return hori_buffer, azim


def horizon_tables(azim_num, hori_acc, elev_ang_low_lim):
    """The trig tables the library builds on the host (for bit-for-bit checks)."""
    L = _lib.lib()
    n = C.c_int(0)
    _lib.check(L.hz_horizon_tables(azim_num, hori_acc, elev_ang_low_lim, None, None, 0,
                                   None, None, None, C.byref(n)))
    out = {k: np.empty(azim_num, np.float32) for k in ("azim_sin", "azim_cos")}
    out.update({k: np.empty(n.value, np.float32) for k in ("elev_ang", "elev_sin", "elev_cos")})
    _lib.check(L.hz_horizon_tables(azim_num, hori_acc, elev_ang_low_lim, ptr(out["azim_sin"]),
                                   ptr(out["azim_cos"]), n.value, ptr(out["elev_ang"]),
                                   ptr(out["elev_sin"]), ptr(out["elev_cos"]), C.byref(n)))
    out["elev_num"] = n.value
    return out
