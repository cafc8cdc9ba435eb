"""GPU box: stem_rs_kernel (csrc/stem_rs.h) alone on a 3840 x 2176 frame: time per launch, with parts of the kernel removed (bench build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_stem_rs.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
NOTAPS, NOFM, NOMATH, NOSTORE, NOFINISH = 1, 2, 4, 8, 16
wp, hp = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2176)
for name, v in [("full", 0), ("full", 0), ("one workgroup per CU", 0x100), ("no image taps", NOTAPS), ("no F, M loads", NOFM), ("no global loads", NOTAPS | NOFM),
                ("no MFMAs", NOMATH), ("no stores", NOSTORE), ("no gather arithmetic / ring A writes", NOFINISH), ("matrix work + stores only", NOTAPS | NOFM | NOFINISH),
                ("matrix work only", NOTAPS | NOFM | NOFINISH | NOSTORE), ("gather arithmetic only", NOTAPS | NOFM | NOMATH | NOSTORE)]:
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_stem_rs(0, wp, hp, v, 20, ctypes.byref(ms), None)
    print("%-40s rc=%d  %.1f us" % (name, rc, ms.value * 1e3), flush=True)
# where a step's cycles go: per wave, mean shader cycles per step between the stamps (the clock stamps themselves cost a few per cent)
PH = ["load issue", "stem 0", "barrier", "stem 1", "early finish", "barrier", "combine/finish/store", "barrier"]
for name, v in [("full", 32), ("matrix work only", 32 | NOTAPS | NOFM | NOFINISH | NOSTORE)]:
    ms = ctypes.c_float(); st = (ctypes.c_longlong * 64)()
    rc = L.rife_hip_bench_stem_rs(0, wp, hp, v, 5, ctypes.byref(ms), st)
    print("clock stamps, %s: rc=%d %.1f us per launch; cycles per step" % (name, rc, ms.value * 1e3))
    print("  wave " + " ".join("%22s" % p for p in PH) + "   total")
    for w in range(8):
        print("  %4d " % w + " ".join("%22d" % st[w * 8 + i] for i in range(8)) + "   %d" % sum(st[w * 8 + i] for i in range(8)))
