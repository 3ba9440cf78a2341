"""Device allocation cost vs size (diagnostics for the pair-pool sizing)."""
import ctypes, time, sys
hip = ctypes.CDLL("libamdhip64.so")
def malloc(n):
    p = ctypes.c_void_p(); t = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)); dt = time.perf_counter() - t
    return rc, p, dt
hip.hipInit(0)
for gb in (1, 4, 8, 16, 24, 32, 40, 51, 64, 100):
    rc, p, dt = malloc(gb << 30)
    t = time.perf_counter(); hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); dm = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(p); df = time.perf_counter() - t
    print(f"{gb:4d} GB: hipMalloc rc={rc} {dt*1e3:8.1f} ms   memset {dm*1e3:7.1f} ms   hipFree {df*1e3:7.1f} ms", flush=True)
# many 8-GB slabs
t = time.perf_counter(); ps = [malloc(8 << 30)[1] for _ in range(8)]; print(f"8 x 8 GB: {1e3*(time.perf_counter()-t):.1f} ms")
