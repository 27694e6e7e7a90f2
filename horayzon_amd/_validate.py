"""Argument validation of the drop-in boundary as data.

The reference checks its arguments in a fixed order and raises fixed exception classes with fixed messages
(horizon.pyx:108-153, :279-312; shadow.pyx:87-133); callers and tests depend on class, message and ORDER, so those are the
contract.  Here every entry point states its checks as a table of rules -- (exception class, message, predicate that is true
when the argument is BAD) -- and `run` raises the first rule that fires.  Predicates are callables so that a later rule may
rely on what an earlier one established (shapes before contents)."""
import numpy as np

ALGORITHMS = ("discrete_sampling", "binary_search", "guess_constant")
GEOMETRIES = ("triangle", "quad", "grid")
DIM_LIMIT = 32767          # Embree's 16-bit grid resolution, kept as the reference's limit (horizon.pyx:149-151)

MSG_ALG = "invalid input argument for ray_algorithm"
MSG_GEOM = "invalid input argument for geom_type"
MSG_ACC = "limit of hori_acc (10 degree) is exceeded"
MSG_ELEV = "minimal allowed value for 'ray_org_elev' is 0.005 m"
MSG_MASK_TYPE = "data type of mask must be 'uint8'"
MSG_DIM_LIMIT = "maximal allowed input length for dem_dim_0 and dem_dim_1 is 32'767"
MSG_NORTH = "dimension (lengths) of vec_norm and/or vec_north is/are erroneous"


def run(rules):
    """rules: iterable of (exception class, message, is_bad) -- raises the first whose predicate returns true."""
    for exc, message, is_bad in rules:
        if is_bad():
            raise exc(message)


def same_leading_shape(arrays, ndim, n_lead):
    """True when every array has `ndim` dimensions and they agree with the first one in their leading `n_lead` lengths."""
    first = arrays[0]
    return all(a.ndim == ndim for a in arrays) and all(a.shape[:n_lead] == first.shape[:n_lead] for a in arrays)


def fits_grid(n_elements, dem_dim_0, dem_dim_1):
    return n_elements >= dem_dim_0 * dem_dim_1 * 3


def window_inside(offset_0, offset_1, shape, dem_dim_0, dem_dim_1):
    return offset_0 + shape[0] <= dem_dim_0 and offset_1 + shape[1] <= dem_dim_1


def unit_vectors(v, tol=1.0e-5):
    return float(np.abs((v ** 2).sum(axis=2) - 1.0).max()) <= tol
