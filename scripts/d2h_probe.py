"""Host-side copy rates that decide how the drop-in call (NumPy out, 18 GB) should move its result: pageable
device-to-host copies, hipHostRegister of the caller's array, copies into a registered array, hipHostMalloc."""
import ctypes as C, json, time, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from horayzon_amd import _lib
_lib.lib()
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)     # the runtime the library is using
hip = C.CDLL(path)
GB = 1 << 30
n = 4 * GB
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
hip.hipMemset(d, 1, C.c_size_t(n)); hip.hipDeviceSynchronize()
res = {}
def t(f):
    t0 = time.perf_counter(); f(); hip.hipDeviceSynchronize(); return time.perf_counter() - t0
a = np.empty(n, np.uint8)
res["d2h_pageable_first_touch_gbs"] = n / t(lambda: hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(n), 2)) / 1e9
res["d2h_pageable_touched_gbs"] = n / t(lambda: hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(n), 2)) / 1e9
b = np.empty(n, np.uint8)
res["host_register_untouched_s_per_gb"] = t(lambda: hip.hipHostRegister(C.c_void_p(b.ctypes.data), C.c_size_t(n), 0)) / (n / GB)
res["d2h_registered_gbs"] = n / t(lambda: hip.hipMemcpy(C.c_void_p(b.ctypes.data), d, C.c_size_t(n), 2)) / 1e9
res["host_unregister_s_per_gb"] = t(lambda: hip.hipHostUnregister(C.c_void_p(b.ctypes.data))) / (n / GB)
res["host_register_touched_s_per_gb"] = t(lambda: hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(n), 0)) / (n / GB)
hip.hipHostUnregister(C.c_void_p(a.ctypes.data))
p = C.c_void_p()
res["host_malloc_s_per_gb"] = t(lambda: hip.hipHostMalloc(C.byref(p), C.c_size_t(n), 0)) / (n / GB)
res["d2h_hostmalloc_gbs"] = n / t(lambda: hip.hipMemcpy(p, d, C.c_size_t(n), 2)) / 1e9
c = np.empty(n, np.uint8)
src = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value))
res["cpu_memcpy_pinned_to_untouched_numpy_gbs_1thread"] = n / t(lambda: np.copyto(c, src)) / 1e9
res["cpu_memcpy_pinned_to_touched_numpy_gbs_1thread"] = n / t(lambda: np.copyto(c, src)) / 1e9
e = np.empty(n, np.uint8)
MADV_POPULATE_WRITE = 23
libc = C.CDLL(None, use_errno=True)
t0 = time.perf_counter(); rc = libc.madvise(C.c_void_p(e.ctypes.data & ~4095), C.c_size_t(n), MADV_POPULATE_WRITE); dt = time.perf_counter() - t0
res["madvise_populate_write_rc"] = rc; res["madvise_populate_write_s_per_gb"] = dt / (n / GB)
res["d2h_pageable_after_populate_gbs"] = n / t(lambda: hip.hipMemcpy(C.c_void_p(e.ctypes.data), d, C.c_size_t(n), 2)) / 1e9
print(json.dumps(res, indent=1))
