"""Where the time of the dominant trunk kernel goes, per workgroup: wave 0 of every workgroup of conv_h2b_kernel<2,10> (64 -> 64
channels, 544 x 960, 2040 workgroups) stamps the shader clock at its phase boundaries (bench-only build of the kernel, variant 32768
of rife_hip_bench_h2b); this script turns the stamps into per-phase statistics and per-CU timelines."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs("gpurun_out", exist_ok=True)
if len(sys.argv) > 1:                       # offline: python tools/h2b_phase_trace.py gpurun_out/h2b_stamps.bin <kernel us>
    st = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 16)
    kernel_us = float(sys.argv[2])
else:
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    from tools import benchlib
    L = benchlib.lib()
    L.rife_hip_bench_h2b.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
    ms = ctypes.c_float()
    assert L.rife_hip_bench_h2b(0, 544, 960, 0, 20, ctypes.byref(ms)) == 0
    kernel_us = ms.value * 1e3
    assert L.rife_hip_bench_h2b(0, 544, 960, 32768, 1, ctypes.byref(ms)) == 0, L.rife_hip_last_error()
    st = np.fromfile("gpurun_out/h2b_stamps.bin", dtype=np.int64).reshape(-1, 16)
n = st.shape[0]
hw = st[:, 15] & 0xffffffff
xcc = (st[:, 15] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
# s_memtime is consistent within a CU but not across CUs (different bases): every CU gets its own origin = its first workgroup's start,
# which is the kernel start to within the dispatch skew
st = st.copy()
for c in set(cuid.tolist()):
    m = cuid == c
    st[m, :15] -= st[m, 0].min()
t0 = 0
span = int(np.median([st[cuid == c, 14].max() for c in set(cuid.tolist())]))
tick_us = kernel_us / span              # the un-instrumented kernel time over the median per-CU span (the stamps add little)
print("workgroups %d on %d distinct CUs; median per-CU span %d ticks for a %.1f us kernel -> %.3f ns per tick (%.2f GHz)" % (n, len(set(cuid.tolist())), span, kernel_us, tick_us * 1e3, 1e-3 / tick_us))
ph = {"launch -> index math done": st[:, 8] - st[:, 0], "first loads issued": st[:, 9] - st[:, 8], "loads back, converted, chunk 0 in LDS": st[:, 10] - st[:, 9],
      "chunk-1 loads issued + barrier": st[:, 1] - st[:, 10], "PROLOGUE total": st[:, 1] - st[:, 0],
      "chunk 0": st[:, 2] - st[:, 1], "chunk 1": st[:, 3] - st[:, 2], "chunk 2": st[:, 4] - st[:, 3], "chunk 3": st[:, 5] - st[:, 4], "MATRIX total": st[:, 5] - st[:, 1],
      "epilogue barrier (slowest wave)": st[:, 6] - st[:, 5], "bias + activation + tile -> LDS": st[:, 7] - st[:, 6], "LDS -> registers -> global stores issued": st[:, 13] - st[:, 7],
      "stores drained": st[:, 14] - st[:, 13], "EPILOGUE total": st[:, 14] - st[:, 5], "whole workgroup": st[:, 14] - st[:, 0]}
for k, v in ph.items():
    v = v * tick_us
    print("%-52s mean %6.2f us   p10 %6.2f   p50 %6.2f   p90 %6.2f" % (k, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
# per CU: how many workgroups, how they overlap, the gaps between one leaving and the next arriving
gaps, conc, offs = [], [], []
for c in sorted(set(cuid.tolist())):
    idx = np.where(cuid == c)[0]
    s, e = st[idx, 0] - t0, st[idx, 14] - t0
    order = np.argsort(s); s, e = s[order], e[order]
    # time-weighted concurrency on this CU
    ev = sorted([(x, 1) for x in s] + [(x, -1) for x in e])
    cur, last, acc = 0, 0, {}
    for t, d in ev:
        acc[cur] = acc.get(cur, 0) + (t - last); last = t; cur += d
    conc.append([acc.get(k, 0) * tick_us for k in range(4)])
    # phase offset between co-resident workgroups: start of each workgroup relative to the start of the one it overlaps most
    for i in range(1, len(s)):
        j = int(np.argmax([min(e[i], e[k]) - max(s[i], s[k]) for k in range(i)]))
        if min(e[i], e[j]) > max(s[i], s[j]):
            offs.append((s[i] - s[j]) * tick_us)
    ends = np.sort(e)
    for x in s[2:]:
        prev = ends[ends <= x]
        if len(prev): gaps.append((x - prev.max()) * tick_us)
conc = np.array(conc)
print("per CU, time with 0 / 1 / 2 / 3 workgroups resident (us, mean over CUs): %s of %.1f" % (np.round(conc.mean(0), 2).tolist(), span * tick_us))
bc = np.bincount(cuid); print("workgroups per CU: min %d max %d" % (bc[bc > 0].min(), bc.max()))
print("gap between a workgroup leaving a CU and the next one starting there: mean %.2f us, p50 %.2f, p90 %.2f" % (np.mean(gaps), np.percentile(gaps, 50), np.percentile(gaps, 90)))
offs = np.array(offs); whole = (ph["whole workgroup"] * tick_us).mean()
print("start offset between co-resident workgroups: mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f (whole workgroup %.2f us: half of it would be perfect interleaving)" %
      (offs.mean(), np.percentile(offs, 10), np.percentile(offs, 50), np.percentile(offs, 90), whole))
# chip-wide: how many workgroups are in which phase over time (20 bins)
bins = np.linspace(0, span, 21)
print("chip-wide phase census (workgroups in prologue / matrix chunks / epilogue+drain) per 1/20 of the kernel:")
for b in range(20):
    t = (bins[b] + bins[b + 1]) / 2 + t0
    pro = int(((st[:, 0] <= t) & (t < st[:, 1])).sum()); mat = int(((st[:, 1] <= t) & (t < st[:, 5])).sum()); epi = int(((st[:, 5] <= t) & (t < st[:, 14])).sum())
    print("  %5.1f us: %4d / %4d / %4d" % ((t - t0) * tick_us, pro, mat, epi))
