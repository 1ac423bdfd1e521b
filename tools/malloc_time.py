"""how long a first 17 GB hipMalloc takes in this process (the F matrix of the 64k problem), with and without the library loaded"""
import ctypes as C, sys, time, os
sys.path.insert(0, '.')
from daisyriot_amd import api
L = api.load_library()
hip = C.CDLL("libamdhip64.so.7" if os.environ.get("DR_SYSTEM_ROCM") else os.path.join(api._torch_rocm_dir() or "", "libamdhip64.so"))
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
c = api.Context(0)
for rep in range(3):
    p = C.c_void_p()
    t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), 17179869184); dt = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(p); df = time.perf_counter() - t
    print("MALLOC rc", rc, "ms", round(dt * 1e3, 1), "free ms", round(df * 1e3, 1), flush=True)
