"""ctypes binding of libhorayzon_hip.so (C ABI: include/horayzon_hip.h).

The library is hand-written HIP for gfx950 and is the ONLY compute path of
this package: there is no CPU fallback.  If the shared object is missing or
no GPU is usable, calls fail loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HORAYZON_HIP_LIB") or os.path.join(_HERE, "libhorayzon_hip.so")
_lib = None


class HorayzonHipError(RuntimeError):
    """The HIP library reported a failure (message from hz_last_error())."""


class hz_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("verbose", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("top_nodes", C.c_int32), ("regroup", C.c_int32),
                ("count_work", C.c_int32), ("no_hit_cache", C.c_int32),
                ("svf", C.c_void_p), ("vec_tilt", C.c_void_p),
                ("skip_hori", C.c_int32), ("chunk_rows", C.c_int32),
                ("level_stack", C.c_int32), ("hori_is_slab", C.c_int32),
                ("no_near_skip", C.c_int32), ("verify_near", C.c_int32),
                ("inputs_are_slab", C.c_int32), ("no_host_pin", C.c_int32),
                ("left_min", C.c_int32), ("persist_grid", C.c_int32), ("left_cap_test", C.c_int32), ("left_tune", C.c_int32)]


class hz_stats(C.Structure):
    _fields_ = [("num_rays", C.c_uint64), ("num_cells", C.c_uint64),
                ("guard_events", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("t_bvh_s", C.c_double),
                ("t_h2d_s", C.c_double), ("t_kernel_s", C.c_double),
                ("t_d2h_s", C.c_double), ("t_total_s", C.c_double),
                ("bvh_height", C.c_int32), ("elev_num", C.c_int32),
                ("scene_bytes", C.c_uint64), ("wave_node_iters", C.c_uint64),
                ("wave_leaf_iters", C.c_uint64), ("wave_refills", C.c_uint64),
                ("t_svf_s", C.c_double), ("stack_fallbacks", C.c_uint64),
                ("rays_shortened", C.c_uint64), ("near_violations", C.c_uint64), ("t_near_s", C.c_double),
                ("stack_redo_blocks", C.c_uint64), ("guard_cells", C.c_uint64),
                ("height_field", C.c_int32), ("near_used", C.c_int32), ("near_verified", C.c_uint64),
                ("t_left_s", C.c_double), ("left_cells", C.c_uint64),
                ("left_again", C.c_uint64), ("scratch_bytes", C.c_uint64), ("left_redo_groups", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/horayzon_hip.h declares (tests check that all are exported)
SYMBOLS = (
    "hz_last_error", "hz_abi_struct_sizes", "hz_abi_version", "hz_device_count", "hz_device_info",
    "hz_scene_create", "hz_scene_blob", "hz_scene_vertices", "hz_scene_adopt", "hz_scene_destroy",
    "hz_horizon_gridded", "hz_horizon_gridded_scene", "hz_horizon_locations",
    "hz_horizon_locations_scene", "hz_horizon_tables",
    "hz_sky_view_factor", "hz_visible_sky_fraction", "hz_topographic_openness",
    "hz_slope_plane_meth", "hz_slope_vector_meth", "hz_lonlat2ecef", "hz_ecef2enu", "hz_wgs2swiss", "hz_swiss2wgs",
    "hz_ecef2enu_vector", "hz_surf_norm", "hz_north_dir", "hz_vert_grid_len", "hz_pack_vertices",
    "hz_debug_sort_pairs", "hz_debug_exclusive_scan",
    "hz_debug_valu_peak", "hz_debug_copy_peak", "hz_debug_inst_rate", "hz_debug_set",
    "hz_terrain_create", "hz_terrain_initialise", "hz_terrain_initialise_scene",
    "hz_terrain_shadow", "hz_terrain_sw_dir_cor", "hz_terrain_shadow_batch",
    "hz_terrain_sw_dir_cor_batch", "hz_terrain_count_work", "hz_terrain_destroy",
)


def _preload_hip_runtime():
    """One process can hold only one HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so.7 (same SONAME as /opt/rocm's); whichever is mapped first serves both
    this library and torch.  If torch is installed but not imported yet, map ITS runtime
    first so that a later ``import torch`` (bench.py uses torch.distributed / RCCL for the
    multi-GPU path) still finds the GPU.  HORAYZON_HIP_RUNTIME=system skips this."""
    if os.environ.get("HORAYZON_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libhorayzon_hip.so (built by horayzon_amd/csrc/Makefile or
    __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HorayzonHipError(
            "libhorayzon_hip.so not found at %s -- build it with "
            "`make -C horayzon_amd/csrc` (hipcc, gfx950). There is no CPU "
            "fallback in this package." % LIB_PATH)
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.c_int
    L.hz_last_error.restype = C.c_char_p
    L.hz_abi_struct_sizes.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hz_device_count.argtypes = [C.POINTER(C.c_int)]
    L.hz_device_info.argtypes = [ip, C.c_char_p, ip, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.hz_scene_create.argtypes = [vp, ip, ip, C.c_char_p, vp, ip, vp, ip, ip,
                                  C.POINTER(vp), C.POINTER(hz_stats)]
    L.hz_scene_blob.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.hz_scene_vertices.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hz_scene_adopt.argtypes = [vp, C.c_size_t, ip, C.POINTER(vp)]
    L.hz_scene_destroy.argtypes = [vp]
    L.hz_horizon_gridded.argtypes = [
        vp, ip, ip, vp, vp, ip, ip, vp, ip, ip, ip, C.c_float, C.c_float,
        C.c_char_p, C.c_char_p, vp, ip, vp, ip, C.c_float, vp, C.c_float,
        C.c_float, C.POINTER(hz_opts), C.POINTER(hz_stats)]
    L.hz_horizon_gridded_scene.argtypes = [
        vp, vp, vp, ip, ip, vp, ip, ip, ip, C.c_float, C.c_float, C.c_char_p,
        C.c_float, vp, C.c_float, C.c_float, C.POINTER(hz_opts), C.POINTER(hz_stats)]
    L.hz_horizon_locations.argtypes = [
        vp, ip, ip, vp, vp, vp, vp, vp, ip, ip, C.c_float, C.c_float, C.c_char_p, C.c_char_p,
        C.c_float, vp, ip, C.POINTER(hz_opts), C.POINTER(hz_stats)]
    L.hz_horizon_locations_scene.argtypes = [
        vp, vp, vp, vp, vp, vp, ip, ip, C.c_float, C.c_float, C.c_char_p, C.c_float, vp, ip,
        C.POINTER(hz_opts), C.POINTER(hz_stats)]
    L.hz_horizon_tables.argtypes = [ip, C.c_float, C.c_float, vp, vp, ip, vp, vp, vp,
                                    C.POINTER(C.c_int)]
    L.hz_sky_view_factor.argtypes = [vp, vp, vp, ip, ip, ip, vp, ip]
    L.hz_visible_sky_fraction.argtypes = [vp, vp, vp, ip, ip, ip, vp, ip]
    L.hz_topographic_openness.argtypes = [vp, vp, ip, ip, ip, vp, ip]
    L.hz_slope_plane_meth.argtypes = [vp, vp, vp, ip, ip, vp, ip, vp, ip]
    L.hz_slope_vector_meth.argtypes = [vp, vp, vp, ip, ip, vp, ip, vp, ip]
    L.hz_lonlat2ecef.argtypes = [vp, vp, vp, C.c_size_t, ip, vp, vp, vp, ip]
    L.hz_ecef2enu.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, C.c_double, ip, vp, vp, vp, ip]
    L.hz_wgs2swiss.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp, ip]
    L.hz_swiss2wgs.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp, ip]
    L.hz_ecef2enu_vector.argtypes = [vp, C.c_size_t, C.c_double, C.c_double, ip, vp, ip]
    L.hz_surf_norm.argtypes = [vp, vp, C.c_size_t, vp, ip]
    L.hz_north_dir.argtypes = [vp, vp, vp, vp, C.c_size_t, ip, vp, ip]
    L.hz_vert_grid_len.argtypes = [C.c_size_t]
    L.hz_pack_vertices.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_size_t, ip]
    L.hz_debug_sort_pairs.argtypes = [vp, vp, C.c_size_t, ip]
    L.hz_debug_exclusive_scan.argtypes = [vp, vp, C.c_size_t, ip]
    L.hz_debug_valu_peak.argtypes = [ip, ip, ip, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.hz_debug_copy_peak.argtypes = [ip, C.c_size_t, C.POINTER(C.c_double)]
    L.hz_debug_inst_rate.argtypes = [ip, ip, C.POINTER(C.c_double)]
    L.hz_debug_set.argtypes = [C.c_char_p, ip]
    L.hz_terrain_create.argtypes = [ip, C.POINTER(vp)]
    L.hz_terrain_initialise.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, ip, vp, vp, vp,
                                        C.c_char_p, C.c_float, C.c_float, ip, C.POINTER(hz_stats)]
    L.hz_terrain_initialise_scene.argtypes = [vp, vp, ip, ip, vp, vp, ip, ip, vp, vp, vp,
                                              C.c_float, C.c_float, ip]
    L.hz_terrain_shadow.argtypes = [vp, vp, vp, C.POINTER(hz_stats)]
    L.hz_terrain_sw_dir_cor.argtypes = [vp, vp, vp, C.POINTER(hz_stats)]
    L.hz_terrain_shadow_batch.argtypes = [vp, vp, ip, vp, C.POINTER(hz_stats)]
    L.hz_terrain_sw_dir_cor_batch.argtypes = [vp, vp, ip, vp, C.POINTER(hz_stats)]
    L.hz_terrain_count_work.argtypes = [vp, ip]
    L.hz_terrain_destroy.argtypes = [vp]
    for name in SYMBOLS:
        if name not in ("hz_last_error", "hz_vert_grid_len"):
            getattr(L, name).restype = C.c_int
    L.hz_vert_grid_len.restype = C.c_size_t
    a, b = C.c_int(0), C.c_int(0)
    L.hz_abi_struct_sizes(C.byref(a), C.byref(b))
    if a.value != C.sizeof(hz_opts) or b.value != C.sizeof(hz_stats):
        raise HorayzonHipError("ABI mismatch between _lib.py and libhorayzon_hip.so "
                               "(hz_opts %d/%d, hz_stats %d/%d bytes)"
                               % (C.sizeof(hz_opts), a.value, C.sizeof(hz_stats), b.value))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().hz_last_error()
        raise HorayzonHipError("libhorayzon_hip: %s (status %d)"
                               % (msg.decode("utf-8", "replace") if msg else "unknown error", rc))


def device_count():
    n = C.c_int(0)
    lib().hz_device_count(C.byref(n))
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(128)
    cu = C.c_int(0)
    mem = C.c_uint64(0)
    check(lib().hz_device_info(device, name, 128, C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cu.value, hbm_bytes=mem.value)


def ptr(a):
    """Raw address of a NumPy array, a torch tensor (host or HBM), an int, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):          # torch.Tensor (device memory stays in HBM)
        return a.data_ptr()
    raise TypeError("unsupported buffer type %r" % type(a))


class Scene:
    """Vertices + flat LBVH resident in HBM as one blob (hz_scene_*)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        self.stats = None

    @classmethod
    def create(cls, vert_grid, dem_dim_0, dem_dim_1, geom_type="grid", vert_simp=None,
               num_vert_simp=0, tri_ind_simp=None, num_tri_simp=0, device=0):
        h = C.c_void_p()
        st = hz_stats()
        check(lib().hz_scene_create(ptr(vert_grid), dem_dim_0, dem_dim_1, geom_type.encode("utf-8"),
                                    ptr(vert_simp), num_vert_simp, ptr(tri_ind_simp), num_tri_simp,
                                    device, C.byref(h), C.byref(st)))
        sc = cls(h, device)
        sc.stats = st.as_dict()
        return sc

    @classmethod
    def adopt(cls, device_ptr, nbytes, device=0, keepalive=None):
        h = C.c_void_p()
        check(lib().hz_scene_adopt(ptr(device_ptr), nbytes, device, C.byref(h)))
        sc = cls(h, device)
        sc._keepalive = keepalive
        return sc

    def blob(self):
        p = C.c_void_p()
        n = C.c_size_t()
        check(lib().hz_scene_blob(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def vertices(self):
        """(device pointer of the f32[d0 * d1][3] vertex array inside the blob, d0, d1, height_field)."""
        p, d0, d1, hf = C.c_void_p(), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().hz_scene_vertices(self._h, C.byref(p), C.byref(d0), C.byref(d1), C.byref(hf)))
        return p.value, d0.value, d1.value, bool(hf.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().hz_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
