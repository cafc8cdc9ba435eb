import ctypes, importlib, os, sys
sys.path.insert(0, os.getcwd())
amd = importlib.import_module("rife-ncnn-vulkan_amd")
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_h2b.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
for rep in range(2):
  for h in (544, 272):
    for name, v in (("fixed", 0), ("pingpong", 8192), ("fixed, zero data", 16384), ("pingpong, zero data", 8192 + 16384)):
        ms = ctypes.c_float(); rc = L.rife_hip_bench_h2b(0, h, 960, v, 20, ctypes.byref(ms))
        print("h=%4d %-20s rc=%d %.4f ms  -> %.4f ms per 544 rows" % (h, name, rc, ms.value, ms.value * 544 / h))
