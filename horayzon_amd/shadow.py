"""horayzon.shadow -- shadow mask and direct-shortwave correction on MI355X.

Host-side mirror of the reference's ``cdef class Terrain``
(horayzon/shadow.pyx:17-200): same method names, argument order, defaults,
validation and exception classes; the scene (LBVH) and all per-cell inputs
live in HBM for the lifetime of the object (the reference keeps raw host
pointers, shadow_comp.cpp:332-346).
"""
import ctypes as C

import numpy as np

from . import _lib
from . import _validate as V
from ._lib import hz_stats, ptr


def _typed(a, dtype, ndim, name):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'"
                         % (np.dtype(dtype).name, a.dtype.name))


class Terrain:

    def __init__(self, *, device=0):
        self._h = C.c_void_p()
        self._shape = None
        self.device = device
        self.last_stats = None
        _lib.check(_lib.lib().hz_terrain_create(device, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().hz_terrain_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def initialise(self, vert_grid, dem_dim_0, dem_dim_1, offset_0, offset_1,
                   vec_tilt, vec_norm, surf_enl_fac, elevation, mask,
                   geom_type="grid", sw_dir_cor_fill=np.nan, ang_max=89.0,
                   refrac_cor=False, *, scene=None):
        """Initialise Terrain class with Digital Elevation Model (DEM) data.

        Arguments as in the reference (shadow.pyx:40-85).  ``scene`` (keyword
        only, not in the reference) reuses a prebuilt ``Scene``."""
        _typed(vert_grid, np.float32, 1, "vert_grid")
        _typed(vec_tilt, np.float32, 3, "vec_tilt")
        _typed(vec_norm, np.float32, 3, "vec_norm")
        _typed(surf_enl_fac, np.float32, 2, "surf_enl_fac")
        _typed(elevation, np.float32, 2, "elevation")
        _typed(mask, np.uint8, 2, "mask")

        # Consistency and validity of the arguments: the reference's checks, classes, messages and order (shadow.pyx:87-133)
        cell_arrays = (vec_tilt[..., 0], surf_enl_fac, elevation, mask) if vec_tilt.ndim == 3 else (surf_enl_fac, elevation, mask)
        V.run((
            (ValueError, "inconsistency between input arguments 'vert_grid', 'dem_dim_0' and 'dem_dim_1'",
             lambda: not V.fits_grid(len(vert_grid), dem_dim_0, dem_dim_1)),
            (ValueError, "inconsistency between input arguments 'dem_dim_0', 'dem_dim_1', 'offset_0', 'offset_1' and 'vec_norm'",
             lambda: not V.window_inside(offset_0, offset_1, vec_tilt.shape, dem_dim_0, dem_dim_1)),
            (ValueError, "Inconsistent/incorrect shape of 'vec_tilt' and/or 'vec_norm'",
             lambda: not V.same_leading_shape((vec_tilt, vec_norm), 3, 3) or vec_tilt.shape[2] != 3),
            (ValueError, "Inconsistent/incorrect shape of 'surf_enl_fac',  'elevation' and/or 'mask'",
             lambda: not V.same_leading_shape(cell_arrays, 2, 2)),
            (ValueError, "not all input arrays are C-contiguous",
             lambda: not all(a.flags["C_CONTIGUOUS"] for a in (vert_grid, vec_tilt, vec_norm, surf_enl_fac, elevation, mask))),
            (ValueError, "Vectors in 'vec_tilt' and/or 'vec_norm' are not normalised",
             lambda: not (V.unit_vectors(vec_tilt) and V.unit_vectors(vec_norm))),
            (ValueError, V.MSG_GEOM, lambda: geom_type not in V.GEOMETRIES),
            (TypeError, V.MSG_MASK_TYPE, lambda: mask.dtype != "uint8"),
            (TypeError, "'ang_max' must be in the range [85.0, 89.99]", lambda: ang_max < 85.0 or ang_max > 89.99),
            (ValueError, V.MSG_DIM_LIMIT, lambda: max(dem_dim_0, dem_dim_1) > V.DIM_LIMIT),
        ))

        L = _lib.lib()
        st = hz_stats()
        if scene is None:
            rc = L.hz_terrain_initialise(
                self._h, ptr(vert_grid), dem_dim_0, dem_dim_1, offset_0, offset_1,
                ptr(vec_tilt), ptr(vec_norm), vec_tilt.shape[0], vec_tilt.shape[1],
                ptr(surf_enl_fac), ptr(elevation), ptr(mask), geom_type.encode("utf-8"),
                sw_dir_cor_fill, ang_max, int(refrac_cor), C.byref(st))
        else:
            self._scene = scene   # keep the borrowed scene alive
            self.device = scene.device   # the terrain follows the scene's GPU (hz_terrain_initialise_scene)
            rc = L.hz_terrain_initialise_scene(
                self._h, scene._h, offset_0, offset_1, ptr(vec_tilt), ptr(vec_norm),
                vec_tilt.shape[0], vec_tilt.shape[1], ptr(surf_enl_fac), ptr(elevation),
                ptr(mask), sw_dir_cor_fill, ang_max, int(refrac_cor))
        _lib.check(rc)
        self._shape = (vec_tilt.shape[0], vec_tilt.shape[1])
        self.last_stats = st.as_dict()

    def _check_out(self, buf, name):
        if self._shape is None:
            raise _lib.HorayzonHipError("Terrain is not initialised")
        if tuple(buf.shape[-2:]) != self._shape:
            raise ValueError("array '%s' has incorrect shape" % name)

    def shadow(self, sun_position, shadow_buffer):
        """Compute shadow mask for specified sun position
        (0: illuminated, 1: self-shaded, 2: terrain-shaded, 3: masked)."""
        _typed(sun_position, np.float32, 1, "sun_position")
        _typed(shadow_buffer, np.uint8, 2, "shadow_buffer")
        # Check consistency and validity of input arguments (shadow.pyx:165-168)
        if (sun_position.ndim != 1) or (sun_position.size != 3):
            raise ValueError("array 'sun_position' has incorrect shape")
        if not shadow_buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("array 'shadow_buffer' is not C-contiguous")
        self._check_out(shadow_buffer, "shadow_buffer")
        st = hz_stats()
        _lib.check(_lib.lib().hz_terrain_shadow(self._h, ptr(np.ascontiguousarray(sun_position)),
                                                ptr(shadow_buffer), C.byref(st)))
        self.last_stats = st.as_dict()

    def sw_dir_cor(self, sun_position, sw_dir_cor_buffer):
        """Compute shortwave correction factor for specified sun position."""
        _typed(sun_position, np.float32, 1, "sun_position")
        _typed(sw_dir_cor_buffer, np.float32, 2, "sw_dir_cor_buffer")
        # Check consistency and validity of input arguments (shadow.pyx:195-198)
        if (sun_position.ndim != 1) or (sun_position.size != 3):
            raise ValueError("array 'sun_position' has incorrect shape")
        if not sw_dir_cor_buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("array 'sw_dir_cor_buffer' is not C-contiguous")
        self._check_out(sw_dir_cor_buffer, "sw_dir_cor_buffer")
        st = hz_stats()
        _lib.check(_lib.lib().hz_terrain_sw_dir_cor(self._h, ptr(np.ascontiguousarray(sun_position)),
                                                    ptr(sw_dir_cor_buffer), C.byref(st)))
        self.last_stats = st.as_dict()

    def count_work(self, on=True):
        """Additive: later calls also count BVH node visits / triangle tests (``last_stats``; slower)."""
        _lib.check(_lib.lib().hz_terrain_count_work(self._h, int(bool(on))))

    # --- additive batch API (not in the reference): many sun positions, one call ------
    @staticmethod
    def _batch_out(buf, np_dtype, name):
        """The batch forms (additive API) also take torch tensors in HBM: outputs of 144 sun positions of a
        3601^2 tile are 1.8 GB / 7.3 GB and usually consumed on the GPU."""
        if isinstance(buf, np.ndarray):
            _typed(buf, np_dtype, 3, name)
            if not buf.flags["C_CONTIGUOUS"]:
                raise ValueError("array '%s' is not C-contiguous" % name)
            return
        if not hasattr(buf, "data_ptr"):
            raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray or torch.Tensor, got %s)"
                            % (name, type(buf).__name__))
        if buf.dim() != 3:
            raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % buf.dim())
        if str(buf.dtype).split(".")[-1] != np.dtype(np_dtype).name:
            raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'" % (np.dtype(np_dtype).name, buf.dtype))
        if not buf.is_contiguous():
            raise ValueError("array '%s' is not C-contiguous" % name)

    def shadow_batch(self, sun_positions, shadow_buffers):
        """``shadow`` for sun_positions f32[num][3] -> shadow_buffers u8[num][y][x] (NumPy or torch/HBM)."""
        _typed(sun_positions, np.float32, 2, "sun_positions")
        self._batch_out(shadow_buffers, np.uint8, "shadow_buffers")
        if sun_positions.shape[1] != 3 or shadow_buffers.shape[0] != sun_positions.shape[0]:
            raise ValueError("array 'sun_positions' has incorrect shape")
        self._check_out(shadow_buffers, "shadow_buffers")
        st = hz_stats()
        _lib.check(_lib.lib().hz_terrain_shadow_batch(
            self._h, ptr(np.ascontiguousarray(sun_positions)), sun_positions.shape[0],
            ptr(shadow_buffers), C.byref(st)))
        self.last_stats = st.as_dict()

    def sw_dir_cor_batch(self, sun_positions, sw_dir_cor_buffers):
        """``sw_dir_cor`` for sun_positions f32[num][3] -> sw_dir_cor_buffers f32[num][y][x] (NumPy or torch/HBM)."""
        _typed(sun_positions, np.float32, 2, "sun_positions")
        self._batch_out(sw_dir_cor_buffers, np.float32, "sw_dir_cor_buffers")
        if sun_positions.shape[1] != 3 or sw_dir_cor_buffers.shape[0] != sun_positions.shape[0]:
            raise ValueError("array 'sun_positions' has incorrect shape")
        self._check_out(sw_dir_cor_buffers, "sw_dir_cor_buffers")
        st = hz_stats()
        _lib.check(_lib.lib().hz_terrain_sw_dir_cor_batch(
            self._h, ptr(np.ascontiguousarray(sun_positions)), sun_positions.shape[0],
            ptr(sw_dir_cor_buffers), C.byref(st)))
        self.last_stats = st.as_dict()
