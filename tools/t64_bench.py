"""GPU box: ablation timings and the per-step phase timeline of the persistent S16 trunk kernel (csrc/conv_t64.h).

    python tools/t64_bench.py [h w]        (default 544 960 = the block-3 trunk of a 3840x2160 frame)
Variants (bench-only template instantiations; all but the first compute garbage): no stores / no LDS-DMA after the prologue /
no matrix work / no vmcnt wait, and combinations.  Then the clock probe (200 back-to-back launches per variant; the C side prints per launch the
median workgroup life, first start -> last end, the gap to the previous launch and the shader clock = cycles / 100 MHz ticks, and writes the last
launch's per-workgroup records to gpurun_out/t64_clk_<variant>.bin).  Then one stamped launch: per workgroup, wave and step the shader clock at
(3) DMA issued + epilogue done = start of the step's matrix work, (0) end of it, (1) this wave's DMA pieces landed, (2) barrier passed."""
# NOTE (round 6): the timing loops of this tool feed every launch its predecessor's output (ping-pong); after a few hundred launches the tensor has converged to
# constants and the matrix pipe rewards that with a higher clock (conv_rs2: 110 us at 2.26 GHz in this mode, 144 - 154 us at 1.63 GHz on data that stays random:
# tools/rs2_bench.py, profiles/r6/rs2_bench_real_data.txt).  Read these figures as A/B ratios, not as what a launch costs inside a pass.
import ctypes, importlib, os, struct, sys, statistics
sys.path.insert(0, os.getcwd())
amd = importlib.import_module("rife-ncnn-vulkan_amd")
from tools import benchlib
L = benchlib.lib()
L.rife_hip_bench_t64.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (544, 960)
NOSTORE, NODMA, NOMATH, NOVMWAIT, STAMPS = 0x100, 0x200, 0x400, 0x800, 0x1000
for rep in range(2):
    for name, v in (("full", 0), ("full, layers alternate direction", 0x10000), ("no stores (loads + math)", NOSTORE), ("no DMA (math + stores)", NODMA), ("no DMA, no stores (math only)", NODMA | NOSTORE), ("no math", NOMATH),
                    ("no math, no stores (loads only)", NOMATH | NOSTORE), ("no math, no DMA (stores only)", NOMATH | NODMA), ("no vmcnt wait", NOVMWAIT)):
        ms = ctypes.c_float()
        rc = L.rife_hip_bench_t64(0, h, w, v, 20, ctypes.byref(ms))
        print("%dx%d %-44s rc=%d %.1f us" % (h, w, name, rc, ms.value * 1e3), flush=True)
CLK = 0x40000
for name, v in (("full", 0), ("no stores", NOSTORE), ("no DMA", NODMA), ("math only", NODMA | NOSTORE), ("no math", NOMATH), ("loads only", NOMATH | NOSTORE), ("stores only", NOMATH | NODMA)):
    ms = ctypes.c_float()
    rc = L.rife_hip_bench_t64(0, h, w, v | CLK, 200, ctypes.byref(ms))
    print("%dx%d clock probe, %-12s rc=%d %.1f us per launch" % (h, w, name, rc, ms.value * 1e3), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
ms = ctypes.c_float()
rc = L.rife_hip_bench_t64(0, h, w, STAMPS, 1, ctypes.byref(ms))
print("stamped launch rc=%d %.1f us" % (rc, ms.value * 1e3))
raw = open("gpurun_out/t64_stamps.bin", "rb").read()
n = len(raw) // 8
st = struct.unpack("<%dq" % n, raw)
NW = 8
nwg = n // (NW * 32 * 4)
def S(b, wv, s, k): return st[((b * NW + wv) * 32 + s) * 4 + k]
print("per step, median over %d workgroups [cycles since the workgroup's previous barrier release (= max over waves of stamp 2)]" % nwg)
print("columns per wave class: start of MFMAs, end of MFMAs, DMA landed, barrier passed")
for s in range(1, 20):
    rows = {}
    for b in range(nwg):
        prev = [S(b, wv, s - 1, 2) for wv in range(NW)]
        if not all(prev) or not S(b, 0, s, 2):
            continue
        t0 = max(prev)
        for wv in range(NW):
            if S(b, wv, s, 0):
                rows.setdefault(wv, []).append([S(b, wv, s, k) - t0 for k in (3, 0, 1, 2)])
    if not rows:
        continue
    line = "step %2d " % s
    for wv in (0, 2, 4, 6, 7):
        if wv in rows:
            med = [int(statistics.median(x[k] for x in rows[wv])) for k in range(4)]
            line += "| w%-2d %5d %5d %5d %5d " % (wv, med[0], med[1], med[2], med[3])
    print(line)
