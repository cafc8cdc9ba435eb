"""Bare matrix-pipe throughput for the trunk tile's instruction mix on random operands (rife_hip_bench_mfma_mix):
hi + lo as f16 (today) vs hi as f16 + lo as scaled fp8 vs hi alone.  512 workgroups x 8 waves, `tiles` tiles per wave."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("rife-ncnn-vulkan_amd")
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_mfma_mix.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
tiles = 16
out = (ctypes.c_float * 6)()
L.rife_hip_probe_fp8.argtypes = [ctypes.c_int, ctypes.c_void_p]
assert L.rife_hip_probe_fp8(0, out) == 0
print("fp8 probe: legacy mfma 16 x (1.0 * 2.0) = %g (32 = OCP e4m3fn); cvt(1, 448) = 0x%04x (0x7e38 = OCP); cvt(1000, -0.3) = 0x%04x; cvt(2^-9, 2^-10) = 0x%04x; scaled mfma = %g (32 expected); k-pairing probe = %g"
      % (out[0], int(out[1]), int(out[2]), int(out[3]), out[4], out[5]))
for rep in range(2):
    for mix, name in ((0, "80 f16 hi + 80 f16 lo"), (1, "80 f16 hi + 20 fp8x64 lo"), (3, "80 f16 hi + 80 fp8x16 lo"), (2, "80 f16 hi only")):
        ms = ctypes.c_float()
        rc = L.rife_hip_bench_mfma_mix(0, mix, tiles, 20, ctypes.byref(ms))
        assert rc == 0, L.rife_hip_last_error()
        # per wave per tile: f16 instr = 32 cycles nominal, fp8x64 = 64 cycles nominal; 4 SIMDs x 4 waves each
        n16 = {0: 160, 1: 80, 2: 80, 3: 160}[mix]; n8 = 20 if mix == 1 else 0
        waves = 512 * 8; per_simd = waves / (256 * 4) * tiles
        cyc = ms.value * 1e-3 * 2.4e9 / per_simd
        print("%-28s %.3f ms  = %.0f cycles@2.4GHz per tile per wave-slot (nominal %d)" % (name, ms.value, cyc, n16 * 32 + n8 * 64))
