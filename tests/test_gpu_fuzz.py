"""Seeded random configurations: GPU (C ABI) against the CPU oracle, bit for bit.
HZ_FUZZ_N / HZ_FUZZ_SEED widen the sweep when hunting (default: 24 configurations)."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def test_random_configurations(hip, orc):
    n = int(os.environ.get("HZ_FUZZ_N", "24"))
    rng = np.random.default_rng(int(os.environ.get("HZ_FUZZ_SEED", "20260928")))
    for it in range(n):
        kw, par = cases.random_config(rng)
        h_gpu, a_gpu = hip.horizon.horizon_gridded(**kw, **par)
        st = hip.horizon.last_stats
        h_cpu, a_cpu, so = orc.horizon_gridded(**kw, **par, return_stats=True)
        desc = "config %d: dem %dx%d %s" % (it, kw["dem_dim_0"], kw["dem_dim_1"],
                                           {k: v for k, v in par.items() if np.isscalar(v)})
        assert np.array_equal(a_gpu, a_cpu), desc
        assert np.array_equal(h_gpu, h_cpu), desc
        assert st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"], desc
