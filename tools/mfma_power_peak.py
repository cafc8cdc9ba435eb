"""GPU box: what the f16 matrix pipe SUSTAINS on random operands inside the 1,400 W board power cap - the peak a split-f16 trunk kernel can be priced against.
k_bench_mfma_mix (csrc/bench_hooks.h): nothing but v_mfma_f32_32x32x16_f16 on register operands (random f16 in [-1, 1), or all zeros), in bursts of seconds
while a thread samples rocm-smi (socket power, shader clock).  Variants: one wave per SIMD (256 workgroups x 4 waves: how the matrix waves of conv_rs / conv_rs2
run) with two or four accumulation chains per wave; four waves per SIMD (512 x 8) with two chains (stops at 70 % of the pipe on any data: wave arbitration) or
four.  Result (profiles/r6/mfma_power_peak.txt): zeros 2,461 - 2,477 TFLOP/s at 2.40 GHz; random operands 1,743 - 1,761 at 1.80 GHz and 1,320 W.
    python tools/mfma_power_peak.py [seconds]"""
import ctypes, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_mfma_mix.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
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
time.sleep(1.0)
TILES = 256
for rep in range(2):
    for mix, name, nmfma in ((0x300, "ONE wave per SIMD, two chains, ALL-ZERO operands", 160), (0x304, "ONE wave per SIMD, four chains, ALL-ZERO operands", 160),
                             (0x200, "ONE wave per SIMD, two chains (hi + lo like a conv_rs consumer)", 160), (0x204, "ONE wave per SIMD, four chains", 160),(0, "160 x v_mfma_f32_32x32x16_f16 per tile (the hi + lo mix)", 160), (2, "80 x v_mfma_f32_32x32x16_f16 per tile (hi only)", 80),
                             (4, "160 per tile, FOUR accumulation chains per wave (no dependent issue)", 160),
                             (0x100, "160 per tile, ALL-ZERO operands (same instruction stream, no toggling)", 160),
                             (0x104, "160 per tile, four chains, ALL-ZERO operands", 160)):
        waves = 256 * 4 if mix & 0x200 else 512 * 8
        flop = waves * TILES * nmfma * 32 * 32 * 16 * 2.0
        ms = ctypes.c_float()
        assert L.rife_hip_bench_mfma_mix(0, mix, TILES, 3, ctypes.byref(ms)) == 0, L.rife_hip_last_error()
        iters = max(5, int(secs * 1e3 / ms.value))
        t0 = time.perf_counter()
        assert L.rife_hip_bench_mfma_mix(0, mix, TILES, iters, ctypes.byref(ms)) == 0, L.rife_hip_last_error()
        t1 = time.perf_counter()
        dur = ms.value * 1e-3 * iters
        sel = [(p, f) for (t, p, f) in samples if t1 - 0.6 * dur <= t <= t1 - 0.02]
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
        pw, fq = med([p for p, _ in sel]), med([f for _, f in sel])
        tf = flop / (ms.value * 1e-3) / 1e12
        print("%-60s %.3f ms per launch, %d launches: %.0f TFLOP/s dense f16 sustained = %.3f of the 2,500 TFLOP/s peak | %.0f W, %.0f MHz (%d samples) | matrix pipe busy at that clock: %.2f"
              % (name, ms.value, iters, tf, tf / 2500.0, pw, fq, len(sel), tf / (2500.0 * fq / 2400.0)), flush=True)
stop[0] = True
