"""Ablation timing of the fused assemble + stem-0 kernel (bench-only entry point rife_hip_bench_stemf)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("rife-ncnn-vulkan_amd")
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_stemf.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
for name, v in [("full", 0), ("full", 0), ("no mfma/epilogue", 1), ("no stores", 16), ("no MFMAs (LDS reads kept)", 32)]:
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_stemf(0, 3840, 2176, v, 10, ctypes.byref(ms))
    print("%-28s rc=%d  %.4f ms" % (name, rc, ms.value))
