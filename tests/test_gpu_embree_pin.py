"""HIP kernels against outputs of the Embree reference (tests/golden/embree_*.npz), when present."""
import json
import os

import pytest

from tests import embree_pin

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(embree_pin.HORIZON), reason=embree_pin.MISSING)
def test_horizon_against_embree_reference(hip):
    print(json.dumps(embree_pin.compare_horizon(hip.horizon.horizon_gridded)))


@pytest.mark.skipif(not os.path.exists(embree_pin.SHADOW), reason=embree_pin.MISSING)
def test_shadow_against_embree_reference(hip):
    print(json.dumps(embree_pin.compare_shadow(hip.shadow.Terrain)))
