"""GPU box: determinism of the fused stem kernels in isolation (rife_hip_probe_stem_det in the bench build): `reps` launches on the same random
inputs, number of output floats that differ from the first launch.  variant = S + 16 x ABL (csrc/stem_fused.h)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
reps = int(os.environ.get("REPS", "6"))
VARIANTS = (("S=4", 4), ("S=2", 2), ("S=1 (64-byte records)", 1 + 16 * 256), ("S=1 (80-byte records)", 1), ("S=4, second pixel in a second round", 4 + 16 * 2),
            ("S=4, direct-store epilogue", 4 + 16 * 64), ("S=4, both", 4 + 16 * 66))
for name, v in ((("S=4 with the per-thread dump", 4 + 16 * 1024),) if len(sys.argv) > 1 else VARIANTS):
    for (wp, hp) in ((3840, 2176), (1920, 1088)):
        mm = (ctypes.c_longlong * reps)()
        rc = L.rife_hip_probe_stem_det(0, v, wp, hp, reps, mm)
        print("%-40s %dx%d rc=%d mismatching floats per launch vs launch 0: %s" % (name, wp, hp, rc, list(mm)), flush=True)
