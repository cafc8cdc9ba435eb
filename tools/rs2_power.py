"""GPU box: steady-state time, socket power and shader clock of conv_rs2 / conv_rs and their ablations.  The trunk kernels run at the board's power cap
(1400 W; tools/power_during_bench.sh: 1,350 W at 1.9 GHz in bench.py's timed region), so what a variant costs is its ENERGY: bursts of >= 1 s per variant,
the hwmon power / clock sampled every 50 ms by a second thread, the first 300 ms (the power controller's transient) dropped.
    python tools/rs2_power.py [seconds per variant]"""
import ctypes, glob, os, sys, threading, time
sys.path.insert(0, os.getcwd())
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_rs.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_bench_rs2.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)]
NOSTORE, NODMA, NOMATH, NOFRAG, NOLO = 0x100, 0x200, 0x400, 0x10, 0x20
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2

def find(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None
PW = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
FQ = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        try:
            p = int(open(PW).read()) / 1e6 if PW else -1
            f = int(open(FQ).read()) / 1e6 if FQ else -1
        except Exception:
            p = f = -1
        samples.append((time.perf_counter(), p, f))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()

def run(fn, h, w, variant, us_guess):
    iters = max(50, int(secs * 1e6 / us_guess))
    ms = ctypes.c_float()
    t0 = time.perf_counter()
    rc = fn(0, h, w, variant, iters, ctypes.byref(ms), None)
    t1 = time.perf_counter()
    # the burst is the tail of [t0, t1] (setup: tensors, upload); keep samples of its second half
    dur = ms.value * 1e-3 * iters
    lo, hi = t1 - dur * 0.6, t1 - 0.02
    sel = [(p, f) for (t, p, f) in samples if lo <= t <= hi and p > 0]
    pw = sum(p for p, _ in sel) / max(1, len(sel)); fq = sum(f for _, f in sel) / max(1, len(sel))
    return rc, ms.value * 1e3, pw, fq, len(sel)

print("power file %s, clock file %s" % (PW, FQ), flush=True)
H, W = 544, 960
for rep in range(2):
    for name, fn, v, g in (("conv_rs  full (x 2 layers)", L.rife_hip_bench_rs, 0x10000, 90), ("conv_rs  math only", L.rife_hip_bench_rs, NODMA | NOSTORE, 50), ("conv_rs  no math", L.rife_hip_bench_rs, NOMATH, 50),
                           ("conv_rs2 full", L.rife_hip_bench_rs2, 0x10000, 170), ("conv_rs2 math only", L.rife_hip_bench_rs2, NODMA | NOSTORE, 120),
                           ("conv_rs2 math only, no fragment reads", L.rife_hip_bench_rs2, NODMA | NOSTORE | NOFRAG, 120),
                           ("conv_rs2 math only, hi products only", L.rife_hip_bench_rs2, NODMA | NOSTORE | NOLO, 70),
                           ("conv_rs2 no math", L.rife_hip_bench_rs2, NOMATH, 70), ("conv_rs2 no stores", L.rife_hip_bench_rs2, NOSTORE, 140), ("conv_rs2 no DMA", L.rife_hip_bench_rs2, NODMA, 140)):
        rc, us, pw, fq, n = run(fn, H, W, v, g)
        per_layer = us / 2 if fn is L.rife_hip_bench_rs2 else us
        print("%-40s rc=%d %7.1f us per launch (%.1f per layer)  %6.0f W  %5.0f MHz  (%d samples)  -> %.3f J per layer" % (name, rc, us, per_layer, pw, fq, n, pw * per_layer * 1e-6), flush=True)
stop[0] = True
