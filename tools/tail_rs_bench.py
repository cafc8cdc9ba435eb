"""GPU box: tail_rs_kernel (csrc/tail_rs.h) alone on a 3840 x 2176 frame: time per launch, with parts of the kernel removed (bench build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_tail_rs.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
NOTAPS, NOFM, NOMATH, NOSTORE, NOPIX, NOROW = 1, 2, 4, 8, 16, 32
wp, hp = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2176)
for name, v in [("full", 0), ("full", 0), ("no image taps", NOTAPS), ("no F, M loads", NOFM), ("no trunk row loads", NOROW), ("no global loads", NOTAPS | NOFM | NOROW),
                ("no MFMAs", NOMATH), ("no stores", NOSTORE), ("no pixel arithmetic", NOPIX), ("matrix work only", NOTAPS | NOFM | NOROW | NOPIX),
                ("pixel arithmetic only", NOTAPS | NOFM | NOROW | NOMATH | NOSTORE)]:
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_tail_rs(0, wp, hp, v, 20, ctypes.byref(ms))
    print("%-28s rc=%d  %.1f us" % (name, rc, ms.value * 1e3), flush=True)
