"""GPU box: socket power, shader clock and ENERGY per launch of the rife-v4.6 4K kernels that have a bench hook, each alone in a burst of seconds (rocm-smi sampled
by a second thread): what a launch costs the 1,400 W budget every workload runs against (profiles/r6/power_workloads.txt).  The kernels run on synthetic
operands (random weights / activations / flows); conv_rs2 on a fixed random tensor (the ping-pong mode converges to constants and flatters it, tools/rs2_bench.py).
    python tools/energy_classes.py [seconds per kernel]"""
import ctypes, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
L = benchlib.lib()
ci, fp = ctypes.c_int, ctypes.POINTER(ctypes.c_float)
L.rife_hip_bench_rs2.argtypes = [ci] * 5 + [fp, ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_bench_stem_rs.argtypes = [ci] * 5 + [fp, ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_bench_tail_rs.argtypes = [ci] * 5 + [fp]
L.rife_hip_bench_stemf.argtypes = [ci] * 5 + [fp]
L.rife_hip_bench_mfma_mix.argtypes = [ci] * 4 + [ctypes.c_void_p]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = [float(l.split(":")[-1]) for l in t.splitlines() if "Socket Graphics Package Power" in l]
            fq = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk clock level" in l]
            if pw and fq:
                samples.append((time.perf_counter(), pw[0], fq[0]))
        except Exception:
            time.sleep(0.1)


threading.Thread(target=sampler, daemon=True).start()
time.sleep(1.5)
IDLE = sorted(p for _, p, _ in samples)[len(samples) // 2]
print("idle socket power %.0f W" % IDLE, flush=True)
WP, HP = 3840, 2176
KERNELS = [  # name, launches per 4K pair, call(iters, ms) -> rc
    ("conv_rs2 (block-3 trunk, two layers; fixed random input)", 4, lambda n, ms: L.rife_hip_bench_rs2(0, HP // 4, WP // 4, 0x100000, n, ms, None)),
    ("stem_rs (block-3 stems)", 1, lambda n, ms: L.rife_hip_bench_stem_rs(0, WP, HP, 0, n, ms, None)),
    ("tail_rs (block-3 head + graph tail)", 1, lambda n, ms: L.rife_hip_bench_tail_rs(0, WP, HP, 0, n, ms)),
    ("stem0_fused (tile stem, the hook's default scale)", 1, lambda n, ms: L.rife_hip_bench_stemf(0, WP, HP, 0, n, ms)),
]
tot = 0.0
for name, per_pair, call in KERNELS:
    ms = ctypes.c_float()
    rc = call(5, ctypes.byref(ms))
    if rc:
        print("%-62s rc=%d %s" % (name, rc, L.rife_hip_last_error().decode())); continue
    iters = max(10, int(secs * 1e3 / ms.value))
    rc = call(iters, ctypes.byref(ms))
    t1 = time.perf_counter()
    dur = ms.value * 1e-3 * iters
    sel = [(p, f) for (t, p, f) in samples if t1 - 0.6 * dur <= t <= t1 - 0.02]
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    pw, fq = med([p for p, _ in sel]), med([f for _, f in sel])
    j = pw * ms.value * 1e-3
    tot += per_pair * j
    print("%-62s rc=%d %8.1f us per launch  %5.0f W  %5.0f MHz (%2d samples)  %.4f J per launch (%.4f above idle)  x %d per pair = %.3f J"
          % (name, rc, ms.value * 1e3, pw, fq, len(sel), j, (pw - IDLE) * ms.value * 1e-3, per_pair, per_pair * j), flush=True)
print("sum of the kernels above per 4K pair: %.2f J (a pair costs ~ 1,370 W x 2.03 ms = 2.8 J in bench.py's timed region)" % tot)
stop[0] = True
