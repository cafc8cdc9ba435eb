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
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5

import subprocess
samples, stop = [], [False]
def sampler():      # rocm-smi's "Current Socket Graphics Package Power" follows the load within ~ 50 ms (the hwmon power1_input file is a slow average: useless for 1 s bursts)
    while not stop[0]:
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = [float(l.split(":")[-1]) for l in t.splitlines() if "Socket Graphics Package Power" in l]
            fq = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk clock level" in l]
            if pw and fq:
                samples.append((time.perf_counter(), pw[0], fq[0]))
        except Exception:
            time.sleep(0.1)
th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(1.5)
IDLE = sum(p for _, p, _ in samples) / max(1, len(samples))

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

print("idle socket power %.0f W (%d samples before the first burst)" % (IDLE, len(samples)), flush=True)
H, W = 544, 960
for rep in range(2):
    for name, fn, v, g in (("conv_rs  full (x 2 layers)", L.rife_hip_bench_rs, 0x10000, 90), ("conv_rs  math only", L.rife_hip_bench_rs, NODMA | NOSTORE, 50), ("conv_rs  no math", L.rife_hip_bench_rs, NOMATH, 50),
                           ("conv_rs2 full", L.rife_hip_bench_rs2, 0x10000, 170), ("conv_rs2 math only", L.rife_hip_bench_rs2, NODMA | NOSTORE, 120),
                           ("conv_rs2 math only, no fragment reads", L.rife_hip_bench_rs2, NODMA | NOSTORE | NOFRAG, 120),
                           ("conv_rs2 math only, hi products only", L.rife_hip_bench_rs2, NODMA | NOSTORE | NOLO, 70),
                           ("conv_rs2 no math", L.rife_hip_bench_rs2, NOMATH, 70), ("conv_rs2 no stores", L.rife_hip_bench_rs2, NOSTORE, 140), ("conv_rs2 no DMA", L.rife_hip_bench_rs2, NODMA, 140)):
        rc, us, pw, fq, n = run(fn, H, W, v, g)
        per_layer = us / 2 if fn is L.rife_hip_bench_rs2 else us
        print("%-40s rc=%d %7.1f us per launch (%.1f per layer)  %6.0f W  %5.0f MHz  (%d samples)  -> %.4f J per layer, %.4f above idle" % (name, rc, us, per_layer, pw, fq, n, pw * per_layer * 1e-6, (pw - IDLE) * per_layer * 1e-6), flush=True)
stop[0] = True
