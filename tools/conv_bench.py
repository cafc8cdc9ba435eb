"""Ablation timing of the trunk conv kernel (bench-only entry point rife_hip_bench_conv8)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("rife-ncnn-vulkan_amd")
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_conv8.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_float)]
h, w, c = 544, 960, 64
gf = 2.0 * c * c * 9 * h * w / 1e9
for rep in range(2):
    for name, v in [("full", 0), ("no_stores", 256), ("no_loads", 512), ("no_barriers", 1024), ("no_loads_no_stores", 768), ("mfma+lds only", 1792)]:
        ms = ctypes.c_float()
        rc = L.rife_hip_bench_conv8(0, c, h, w, v, 20, ctypes.byref(ms))
        print("%-20s rc=%d  %.4f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (name, rc, ms.value, gf / ms.value, gf / ms.value / 1.573))

print("---- split-f16 trunk kernel (conv_h2b_kernel<2,10>) ----")
L.rife_hip_bench_h2b.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
nbytes = 2.0 * h * w * c * 4
for name, v in [("full", 0), ("no_stores", 256), ("no_prefetch_loads", 512), ("no_barriers", 1024), ("no_lds_staging", 2048),
                ("no_loads_no_staging", 2560), ("no_loads_staging_stores", 2816), ("mfma+lds reads only", 3840)]:
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_h2b(0, h, w, v, 20, ctypes.byref(ms))
    print("%-26s rc=%d  %.4f ms  %.0f TFLOP/s-equiv  %.2f TB/s algorithmic" % (name, rc, ms.value, gf / ms.value, nbytes / ms.value / 1e9))
