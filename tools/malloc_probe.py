import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
n = 13 << 30
for i in range(5):
    d = C.c_void_p()
    t0 = time.perf_counter(); assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0; t1 = time.perf_counter()
    hip.hipMemset(d, 0, C.c_size_t(n)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipMemset(d, 0, C.c_size_t(n)); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    hip.hipFree(d); t4 = time.perf_counter()
    print(f"round {i}: hipMalloc {1e3*(t1-t0):7.1f} ms, first memset {1e3*(t2-t1):7.1f}, second memset {1e3*(t3-t2):7.1f}, hipFree {1e3*(t4-t3):7.1f}")
