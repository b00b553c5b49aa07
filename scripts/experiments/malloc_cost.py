"""hipMalloc / hipFree cost probe (development aid)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def t(f, n=1):
    s = time.perf_counter(); r = f(); return (time.perf_counter() - s) * 1e3, r
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
for mb in (1, 10, 40, 100, 400):
    ms_a = []; ms_f = []
    for rep in range(5):
        p = ctypes.c_void_p()
        a, _ = t(lambda: hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mb << 20)))
        hip.hipMemset(p, 0, ctypes.c_size_t(mb << 20)); hip.hipDeviceSynchronize()
        f, _ = t(lambda: hip.hipFree(p))
        ms_a.append(a); ms_f.append(f)
    print("%4d MB: hipMalloc %.3f ms  hipFree %.3f ms (min of 5: %.3f / %.3f)" % (mb, sum(ms_a) / 5, sum(ms_f) / 5, min(ms_a), min(ms_f)))
