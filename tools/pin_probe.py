"""scratch measurement: strategies for the 336 MB signal download (D2H into a fresh NumPy array)"""
import ctypes, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
N = 336 * 1024 * 1024
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(N)) == 0
hip.hipMemset(d, 1, ctypes.c_size_t(N)); hip.hipDeviceSynchronize()
pool = ThreadPoolExecutor(8)
def t(label, f, n=3):
    for i in range(n):
        t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
        print(f"{label} #{i}: {dt*1e3:.1f} ms", flush=True)
    return r
def pageable():
    a = np.empty(N, np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(a.ctypes.data), d, ctypes.c_size_t(N), 2) == 0
    return a
t("np.empty + hipMemcpy D2H", pageable)
for T in (2, 4, 8):
    def threaded():
        a = np.empty(N, np.uint8)
        step = N // T
        def part(i):
            off = i * step
            cnt = step if i < T - 1 else N - off
            return hip.hipMemcpy(ctypes.c_void_p(a.ctypes.data + off), ctypes.c_void_p(d.value + off), ctypes.c_size_t(cnt), 2)
        assert all(r == 0 for r in pool.map(part, range(T)))
        return a
    r = t(f"np.empty + {T} threads hipMemcpy slices", threaded)
    assert (r == 1).all()
for T in (4, 8):
    def prefault():
        a = np.empty(N, np.uint8)
        step = N // T
        def part(i):
            a[i * step:(i + 1) * step:4096] = 0
        list(pool.map(part, range(T)))
        assert hip.hipMemcpy(ctypes.c_void_p(a.ctypes.data), d, ctypes.c_size_t(N), 2) == 0
        return a
    t(f"np.empty + {T}-thread prefault + hipMemcpy", prefault)
