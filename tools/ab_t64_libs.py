import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_t64.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
r = []
for i in range(6):
    ms = ctypes.c_float(); L.rife_hip_bench_t64(0, 544, 960, 0, 200, ctypes.byref(ms)); r.append(round(ms.value * 1e3, 1))
print(os.environ.get("RIFE_HIP_BENCH_LIB", "default (no SLP)"), "t64 full, 200 launches each:", r)
