"""GPU box: the weight-stationary K-split trunk kernel (csrc/conv_ks.h) alone, through librife_hip_bench.so: ms per launch over a burst of
back-to-back launches at the tensor sizes of blocks 1 / 2 at 1080p and 4K, with parts removed (matrix work, LDS-DMA, stores, weight loads), on
half the chip, and the per-workgroup timeline of one launch from in-kernel stamps of the 100 MHz counter.
    python tools/ks_bench.py      -> gpurun_out/ks_bench.txt"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_ks.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
NOSTORE, NODMA, NOMATH, NOWEIGHTS, CLK = 0x100, 0x200, 0x400, 0x800, 0x40000
os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/ks_bench.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

def run(C, h, w, variant, iters=200, div=1, stamps=False):
    ms = ctypes.c_float(); nwg = ctypes.c_int()
    st = np.zeros(16 * 4096, np.int64)
    rc = L.rife_hip_bench_ks(0, C, h, w, variant, iters, div, ctypes.byref(ms), st.ctypes.data_as(ctypes.c_void_p) if stamps else None, ctypes.byref(nwg))
    if rc:
        say("rife_hip_bench_ks rc=%d %s" % (rc, L.rife_hip_last_error().decode())); return None, None, 0
    return ms.value * 1000.0, st[: 16 * nwg.value].reshape(nwg.value, 16), nwg.value

for (C, h, w, what) in ((128, 136, 240, "block 1 @4K"), (128, 68, 120, "block 1 @1080p"), (96, 272, 480, "block 2 @4K"), (96, 136, 240, "block 2 @1080p")):
    say("== C = %d, %d x %d pixels (%s)" % (C, h, w, what))
    for name, v in (("full", 0), ("no matrix work", NOMATH), ("no LDS-DMA", NODMA), ("no stores", NOSTORE), ("no weight loads", NOWEIGHTS),
                    ("no DMA, no stores", NODMA | NOSTORE), ("nothing but the barriers", NOMATH | NODMA | NOSTORE | NOWEIGHTS)):
        us, _, nwg = run(C, h, w, v)
        if us is not None: say("   %-26s %7.2f us per launch (%d workgroups)" % (name, us, nwg))
    for div in (2, 4):
        us, _, nwg = run(C, h, w, 0, div=div)
        if us is not None: say("   full, 1/%d of the chip      %7.2f us per launch (%d workgroups)" % (div, us, nwg))
    us, st, nwg = run(C, h, w, CLK, iters=20, stamps=True)
    if us is None: continue
    t0 = st[:, 0].min()
    def col(i): return (st[:, i] - t0) * 0.01
    names = {0: "start", 1: "weights in registers", 12: "loader: prologue rows issued", 13: "loader: rows 0-2 landed", 2: "prologue barrier passed", 3: "iteration 0: X passed",
             4: "step 0 MFMAs issued", 5: "iteration 1: X passed", 6: "step 1 MFMAs issued", 7: "iteration 2: X passed", 8: "step 2 MFMAs issued", 9: "iteration 3: X passed",
             10: "step 3 MFMAs issued", 11: "consumers done", 14: "storers done", 15: "stores retired"}
    say("   timeline of one launch (us after the first workgroup's start; median / max over %d workgroups), %.2f us per launch with stamps:" % (nwg, us))
    for i in (0, 1, 12, 13, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 14, 15):
        c = col(i)
        c = c[st[:, i] > 0]
        if len(c): say("      %-30s %7.2f / %7.2f" % (names[i], float(np.median(c)), float(c.max())))
