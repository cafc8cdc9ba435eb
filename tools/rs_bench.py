"""GPU box: the row-streaming block-3 trunk kernel (csrc/conv_rs.h) against conv_t64_kernel (csrc/conv_t64.h).

    python tools/rs_bench.py [quick]
1. agreement with conv_t64 (same products, another summation order: |difference| ~1e-6) on random S16 tensors at aligned, ragged and tiny sizes, walking down and up, with 1 .. #CU workgroups;
2. per-launch time of both kernels (interleaved rounds in one process), ablations of conv_rs (no stores / no LDS-DMA / no matrix work /
   raised priority / non-temporal loads and stores), the clock probe."""
# NOTE (round 6): the timing loops of this tool feed every launch its predecessor's output (ping-pong); after a few hundred launches the tensor has converged to
# constants and the matrix pipe rewards that with a higher clock (conv_rs2: 110 us at 2.26 GHz in this mode, 144 - 154 us at 1.63 GHz on data that stays random:
# tools/rs2_bench.py, profiles/r6/rs2_bench_real_data.txt).  Read these figures as A/B ratios, not as what a launch costs inside a pass.
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_rs.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_bench_t64.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
NOSTORE, NODMA, NOMATH, PRIO, NTLOAD, NTSTORE, CLK = 0x100, 0x200, 0x400, 0x2000, 0x4000, 0x8000, 0x40000
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

def rs(h, w, variant, iters, check=False):
    ms = ctypes.c_float()
    st = (ctypes.c_longlong * 8)()
    rc = L.rife_hip_bench_rs(0, h, w, variant, iters, ctypes.byref(ms), st if check else None)
    if rc:
        print("rife_hip_bench_rs rc=%d %s" % (rc, L.rife_hip_last_error().decode()))
    return rc, ms.value * 1e3, list(st)

bad = 0
for h, w, g in ((544, 960, 0), (272, 480, 0), (135, 241, 0), (17, 33, 0), (7, 70, 0), (8, 1, 0), (9, 32, 3), (544, 960, 7), (271, 479, 255), (68, 120, 0), (34, 60, 0)):
    rc, us, st = rs(h, w, g << 24, 1, check=True)
    ok = rc == 0 and 0 <= st[2] < 20000 and 0 <= st[3] < 20000      # |difference of the stored values| < 2e-5 (summation order only)
    bad += not ok
    print("%4dx%-4d workgroups %-3s vs conv_t64: bytes differing down %d, up %d of %d; max |value difference| down %.2e, up %.2e %s%s" % (h, w, g or "CUs", st[0], st[1], st[6],
          st[2] * 1e-9, st[3] * 1e-9, "OK" if ok else "MISMATCH", "" if ok else " worst at chunk %d padded row %d col %d" % (st[4], st[5], st[7])), flush=True)
print("parity: %s" % ("all within summation-order noise" if not bad else "%d cases differ" % bad), flush=True)
sizes = ((544, 960),) if quick else ((544, 960), (272, 480))
for h, w in sizes:
    for rep in range(2 if quick else 3):
        ms = ctypes.c_float()
        rc = L.rife_hip_bench_t64(0, h, w, 0, 40, ctypes.byref(ms))
        print("%dx%d conv_t64 full                               rc=%d %.1f us" % (h, w, rc, ms.value * 1e3), flush=True)
        for name, v in (("full (down)", 0), ("full, layers alternate direction", 0x10000), ("full (up)", 0x20000), ("matrix waves at s_setprio 2", PRIO), ("nt loads", NTLOAD), ("nt stores", NTSTORE), ("nt loads + stores", NTLOAD | NTSTORE),
                        ("no stores", NOSTORE), ("no DMA", NODMA), ("no DMA, no stores (math only)", NODMA | NOSTORE),
                        ("no math", NOMATH), ("no math, no stores (loads only)", NOMATH | NOSTORE), ("no math, no DMA (stores only)", NOMATH | NODMA)):
            rc, us, _ = rs(h, w, v, 40)
            print("%dx%d conv_rs  %-36s rc=%d %.1f us" % (h, w, name, rc, us), flush=True)
if not quick:
    for name, v in (("full", 0), ("math only", NODMA | NOSTORE), ("no math", NOMATH)):
        rc, us, _ = rs(544, 960, v | CLK, 200)
        print("544x960 clock probe, %-12s rc=%d %.1f us per launch" % (name, rc, us), flush=True)
sys.exit(1 if bad else 0)
