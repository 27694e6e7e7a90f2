"""horayzon_amd -- MI355X-native (gfx950) terrain-horizon and shadow ray casting.

Drop-in for ``horayzon.horizon.horizon_gridded`` / ``horayzon.shadow.Terrain`` /
``horayzon.topo_param.sky_view_factor`` of ChristianSteger/HORAYZON: NumPy in,
NumPy out, same signatures; the computation runs in hand-written HIP kernels
(libhorayzon_hip.so, C ABI in include/horayzon_hip.h).
"""
from . import horizon      # noqa: F401
from . import shadow       # noqa: F401
from . import topo_param   # noqa: F401
from . import transform    # noqa: F401
from . import direction    # noqa: F401
from . import auxiliary    # noqa: F401
from . import synth        # noqa: F401
from ._lib import HorayzonHipError, Scene, device_count, device_info   # noqa: F401

__version__ = "0.1.0"
