"""GPU box: the depth-fused block-3 trunk kernel (csrc/conv_rs2.h: two 64 -> 64 residual layers per launch, layer A's rows LDS-resident) against
two launches of conv_rs_kernel (csrc/conv_rs.h).

    python tools/rs2_bench.py [quick]
1. byte equality with conv_rs(A) -> conv_rs(B) on random S16 tensors at aligned, ragged and small sizes, walking down and up, planned for 1 .. #CU
   compute units (segments per strip 1 .. 8, several segments per workgroup);
2. time per launch against 2 x conv_rs (interleaved rounds in one process), ablations (no stores / no LDS-DMA / no matrix work)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_rs.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_bench_rs2.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
NOSTORE, NODMA, NOMATH = 0x100, 0x200, 0x400
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def rs2(h, w, variant, iters, check=False):
    ms = ctypes.c_float()
    st = (ctypes.c_longlong * 8)()
    rc = L.rife_hip_bench_rs2(0, h, w, variant, iters, ctypes.byref(ms), st if check else None)
    if rc:
        print("rife_hip_bench_rs2 rc=%d %s" % (rc, L.rife_hip_last_error().decode()))
    return rc, ms.value * 1e3, list(st)


def rs(h, w, variant, iters):
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_rs(0, h, w, variant, iters, ctypes.byref(ms), None)
    return rc, ms.value * 1e3


bad = 0
cases = ((544, 960, 0), (272, 480, 0), (135, 241, 0), (17, 33, 0), (7, 70, 0), (8, 1, 0), (9, 32, 3), (544, 960, 7), (271, 479, 255), (68, 120, 0), (34, 60, 0), (544, 960, 128), (272, 480, 64),
         (100, 30, 1), (40, 31, 2), (64, 64, 5), (33, 95, 0))
for h, w, g in cases:
    rc, us, st = rs2(h, w, g << 24, 0, check=True)
    ok = rc == 0 and st[0] == 0 and st[1] == 0
    bad += not ok
    print("%4dx%-4d planned for %-3s CUs (%d segments per strip, %d workgroups) vs conv_rs x 2: bytes differing down %d, up %d of %d %s%s" % (h, w, g or "all", st[6], st[7], st[0], st[1], st[2],
          "OK" if ok else "MISMATCH", "" if ok else " first at plane %d padded row %d col %d" % (st[3], st[4], st[5])), flush=True)
print("parity: %s" % ("byte-identical everywhere" if not bad else "%d cases differ" % bad), flush=True)
# bursts of IT launches (~ 0.15 - 0.3 s per variant): 40 launches (7 ms) ran at the clock of an idle chip ramping up (177 us per conv_rs2 launch against 108 in a
# 0.8 s burst, profiles/r6/rs2_power.txt); inside a pass the chip sits at the 1,400 W cap (1.8 GHz) and a launch takes 147 us (bench.py, roofline.avg_launch_ms)
IT = 1500
sizes = ((544, 960),) if quick else ((544, 960), (272, 480))
for h, w in sizes:
    for rep in range(2 if quick else 3):
        rc, us = rs(h, w, 0x10000, IT)
        print("%dx%d conv_rs  x 2, layers alternate direction   rc=%d %.1f us" % (h, w, rc, 2 * us), flush=True)
        for name, v in (("full (down)", 0), ("full, launches alternate direction", 0x10000), ("full (up)", 0x20000), ("no stores", NOSTORE), ("no DMA", NODMA),
                        ("no DMA, no stores (math only)", NODMA | NOSTORE), ("no math", NOMATH), ("no math, no stores (loads only)", NOMATH | NOSTORE), ("no math, no DMA (stores only)", NOMATH | NODMA)):
            rc, us, _ = rs2(h, w, v, IT)
            print("%dx%d conv_rs2 %-36s rc=%d %.1f us" % (h, w, name, rc, us), flush=True)
        COLD = 0x80000
        for name, v in (("full, cold, alternate", COLD | 0x10000), ("no stores, cold", COLD | NOSTORE), ("no DMA, cold", COLD | NODMA), ("no math, cold", COLD | NOMATH),
                        ("no math, no stores, cold (loads only)", COLD | NOMATH | NOSTORE), ("no math, no DMA, cold (stores only)", COLD | NOMATH | NODMA)):
            rc, us, _ = rs2(h, w, v, IT)
            print("%dx%d conv_rs2 %-36s rc=%d %.1f us" % (h, w, name, rc, us), flush=True)
        rc, us, _ = rs2(h, w, 0x100000, IT)
        print("%dx%d conv_rs2 %-36s rc=%d %.1f us" % (h, w, "full, fixed input x -> y", rc, us), flush=True)
        for g in (128, 64):
            rc, us, _ = rs2(h, w, (g << 24) | 0x10000, IT)
            print("%dx%d conv_rs2 planned for %3d CUs                   rc=%d %.1f us" % (h, w, g, rc, us), flush=True)
for v, nm in ((0x40000, "ping-pong"), (0x40000 | 0x100000, "fixed input x -> y (267 MB like ping-pong, data never changes)"), (0x40000 | 0x80000, "cold: four rotating fixed inputs (1.07 GB)")):
    rc, us, _ = rs2(544, 960, v, 600)
    print("544x960 conv_rs2 per-workgroup clocks, %s (stderr) rc=%d %.1f us" % (nm, rc, us), flush=True)
rc, us, _ = rs2(544, 960, 0x1000, 600)
print("544x960 conv_rs2 barrier trace (stderr) rc=%d %.1f us" % (rc, us), flush=True)
rc, us, _ = rs2(544, 960, 0x1000 | 0x80000, 600)
print("544x960 conv_rs2 barrier trace, cold mode (stderr) rc=%d %.1f us" % (rc, us), flush=True)
rc, us, _ = rs2(544, 960, 0x1000 | NODMA | NOSTORE, 600)
print("544x960 conv_rs2 barrier trace, math only (stderr) rc=%d %.1f us" % (rc, us), flush=True)
sys.exit(1 if bad else 0)
