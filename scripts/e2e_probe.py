"""Where does the drop-in call (NumPy in, 18 GB NumPy out) spend its time?  hz_horizon_gridded_scene on the config-3 tile
with host output: untouched result array, pre-touched result array, one chunk, and the pieces of hz_stats."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import horayzon_amd as hz
from horayzon_amd import _lib, synth
L = _lib.lib()
n, off, A = 3601, 16, 360
g = synth.fractal_tile(n=n, offset=off)
in0 = in1 = n - 2 * off
sc = hz.Scene.create(g["vert_grid"], n, n)
mask = np.ones((in0, in1), np.uint8)
def call(hori, chunk_rows=0, no_pin=0):
    o = _lib.hz_opts(); o.chunk_rows = chunk_rows; o.no_host_pin = no_pin
    st = _lib.hz_stats()
    t0 = time.perf_counter()
    _lib.check(L.hz_horizon_gridded_scene(sc._h, g["vec_norm"].ctypes.data, g["vec_north"].ctypes.data, off, off, hori.ctypes.data, in0, in1, A,
                                          50.0, 0.25, b"guess_constant", -15.0, mask.ctypes.data, 0.0, 0.01, C.byref(o), C.byref(st)))
    w = time.perf_counter() - t0
    return {"wall_s": w, "lib_total_s": st.t_total_s, "h2d_s": st.t_h2d_s, "kernel_s": st.t_kernel_s, "near_s": st.t_near_s, "d2h_tail_s": st.t_d2h_s}
res = {}
t0 = time.perf_counter(); h = np.empty((in0, in1, A), np.float32); res["np_empty_s"] = time.perf_counter() - t0
res["untouched"] = call(h)
res["touched_same_array"] = call(h)
h2 = np.empty((in0, in1, A), np.float32)
res["untouched_no_pin"] = call(h2, 0, 1)
res["touched_no_pin"] = call(h2, 0, 1)
del h2
ref = h.copy()
h3 = np.empty((in0, in1, A), np.float32)
res["untouched_again"] = call(h3)
res["equal"] = bool(np.array_equal(h3, ref))
print(json.dumps(res, indent=1))
