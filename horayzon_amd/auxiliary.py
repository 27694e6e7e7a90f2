"""horayzon.auxiliary -- the routines of the reference's auxiliary module that sit on the hot path's
input side: ``rearrange_pad_buffer`` (reference horayzon/auxiliary.py:49-95) on MI355X, and ``pad_buffer``
(:100-135; host-side: it prepares the few-element TIN buffers ``vert_simp`` / ``tri_ind_simp``).

NumPy in -> NumPy out as the reference; torch tensors in HBM in -> a torch tensor in HBM out (the chain
lon/lat/elevation -> ENU -> vert_grid -> scene -> horizon / SVF / shadow then never touches host memory)."""
import numpy as np

from . import _lib
from ._lib import ptr


def rearrange_pad_buffer(x, y, z, *, device=0):
    """Rearrange digital elevation model data (two-dimensional float32 x, y, z) into the interleaved,
    zero-padded one-dimensional geometry buffer ``vert_grid``."""
    is_np = [isinstance(a, np.ndarray) for a in (x, y, z)]
    if all(is_np):
        if (x.dtype != np.float32) or (y.dtype != np.float32) or (z.dtype != np.float32):
            raise TypeError("Not all input arguments are 32-bit floats")
        if any(i != 2 for i in (x.ndim, y.ndim, z.ndim)) or not x.shape == y.shape == z.shape:
            raise ValueError("Dimensions of input arguments are erroneous/inconsistent")
        x, y, z = (np.ascontiguousarray(a) for a in (x, y, z))
        n = x.size
        out = np.empty(_lib.lib().hz_vert_grid_len(n), np.float32)
    elif not any(is_np) and all(hasattr(a, "data_ptr") for a in (x, y, z)):
        import torch
        if any(a.dtype != torch.float32 for a in (x, y, z)):
            raise TypeError("Not all input arguments are 32-bit floats")
        if any(a.dim() != 2 for a in (x, y, z)) or not tuple(x.shape) == tuple(y.shape) == tuple(z.shape):
            raise ValueError("Dimensions of input arguments are erroneous/inconsistent")
        x, y, z = (a.contiguous() for a in (x, y, z))
        n = x.numel()
        out = torch.empty(_lib.lib().hz_vert_grid_len(n), dtype=torch.float32, device=x.device)
        if x.device.type == "cuda":
            device = x.device.index or 0
    else:
        raise TypeError("One or more input arguments are of invalid type")
    _lib.check(_lib.lib().hz_pack_vertices(ptr(x), ptr(y), ptr(z), n, ptr(out), len(out), device))
    return out


def pad_buffer(buffer):
    """Padding of a one-dimensional geometry buffer as the reference does it (auxiliary.py:100-135): at least 16 zero
    elements are appended and the byte size becomes a multiple of 16.  This library copies its inputs to HBM and
    never reads the padding -- the function exists so that code written for the reference runs unchanged."""
    if not isinstance(buffer, np.ndarray):
        raise ValueError("argument 'buffer' has invalid type")
    if buffer.ndim != 1:
        raise ValueError("argument 'buffer' must be one-dimensional")
    add_elem = 16
    if not (buffer.nbytes % 16) == 0:
        add_elem += ((16 - (buffer.nbytes % 16)) // buffer[0].nbytes)
    return np.append(buffer, np.zeros(add_elem, dtype=buffer.dtype))
