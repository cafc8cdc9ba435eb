"""librife_hip_bench.so = the product sources + csrc/bench_hooks.h (kernel ablation benches, hardware probes, clock-stamp traces).
Built by `make -C rife-ncnn-vulkan_amd/csrc bench` (and by __graft_entry__.build()); loaded by tools/*.py and one hardware-probe
test - never by the product package."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("RIFE_HIP_BENCH_LIB") or os.path.join(ROOT, "rife-ncnn-vulkan_amd", "librife_hip_bench.so")      # the override: A/B of two bench builds
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rife-ncnn-vulkan_amd", "csrc"), "bench"])
    return PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise RuntimeError("librife_hip_bench.so is not built (make -C rife-ncnn-vulkan_amd/csrc bench)")
        try:
            import torch  # noqa: F401  (one libamdhip64 per process)
        except Exception:
            pass
        _lib = ctypes.CDLL(PATH)
        _lib.rife_hip_last_error.restype = ctypes.c_char_p
    return _lib
