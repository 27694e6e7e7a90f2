"""Drop-in alias: ``import horayzon`` resolves the hot-path modules
(``horizon``, ``shadow``, ``topo_param`` sky view factor / slope, and the ``transform`` /
``direction`` / ``auxiliary`` routines that prepare the input) to the MI355X implementation in ``horayzon_amd``.
Everything else of the reference package (DEM and geoid loaders, ocean masking, domain helpers: file and network
I/O) is out of scope."""
import sys as _sys

from horayzon_amd import auxiliary, direction, horizon, shadow, topo_param, transform   # noqa: F401

_sys.modules[__name__ + ".horizon"] = horizon
_sys.modules[__name__ + ".shadow"] = shadow
_sys.modules[__name__ + ".topo_param"] = topo_param
_sys.modules[__name__ + ".transform"] = transform
_sys.modules[__name__ + ".direction"] = direction
_sys.modules[__name__ + ".auxiliary"] = auxiliary
