"""Per-layer HIP-event profile of a bench.py workload in the stream layouts that matter (rocprofv3 cannot follow CU-masked streams: its queue
interception drops the mask):
    whole  - one ordinary stream, one pair in flight on the whole chip
    part   - ONE stream of rife_hip_stream_create(0, N) alone (the other parts idle)
    parts  - N streams, one per part, all busy (what bench.py's timed region runs); per-launch events on every stream
    python tools/part_profile.py [--workload v23-1080p] [--parts 4] [--pairs 12]
RIFE_HIP_PROFILE_FINE=1 (set here) makes the v2 family report one class per layer position."""
import argparse, importlib, os, sys, threading, time
os.environ.setdefault("RIFE_HIP_PROFILE_FINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import gen_frames, gen_models

WL = {"4k": ("rife-v4.6", 3840, 2160), "1080p": ("rife-v4.6", 1920, 1080), "v23-1080p": ("rife-v2.3", 1920, 1080)}
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="v23-1080p", choices=list(WL))
ap.add_argument("--parts", type=int, default=4)
ap.add_argument("--pairs", type=int, default=12)
ap.add_argument("--per-part", type=int, default=1, help="streams per part in the `parts` layout (bench.py: 4k runs 2 parts x 2)")
args = ap.parse_args()
fam, w, h = WL[args.workload]
amd = importlib.import_module("rife-ncnn-vulkan_amd")
eng = amd.RIFE(0, rife_v2=fam.startswith("rife-v2"), rife_v4=fam.startswith("rife-v4"))
eng.load(gen_models.ensure(None, fam))
fr = [torch.from_numpy(f).cuda() for f in gen_frames.tiled_real_pair(w // 640)]


def run(streams, pairs, profile):
    outs = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in streams]

    def worker(s, n):
        torch.cuda.set_device(0)
        for i in range(n):
            eng.process_device(fr[i % 2].data_ptr(), fr[(i + 1) % 2].data_ptr(), w, h, 0.5, outs[s].data_ptr(), streams[s])
    for s in range(len(streams)):
        worker(s, 2)
    torch.cuda.synchronize()
    if profile:
        eng.profile_enable(True)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(s, pairs)) for s in range(len(streams))]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = None
    if profile:
        prof = eng.profile_read()
        eng.profile_enable(False)
    return len(streams) * pairs / el, prof


layouts = {}
whole = [torch.cuda.Stream().cuda_stream]
layouts["whole"] = run(whole, args.pairs, True) + (args.pairs,)
part = [eng.stream_create(0, args.parts)]
layouts["part"] = run(part, args.pairs, True) + (args.pairs,)
parts = [eng.stream_create(i % args.parts, args.parts) for i in range(args.parts * args.per_part)]
fps_clean, _ = run(parts, args.pairs, False)
layouts["parts"] = run(parts, args.pairs, True) + (args.pairs * len(parts),)
print("# %s  frames/s: whole-chip 1 in flight %.1f | one part of %d alone %.1f | %d streams on %d parts %.1f (with per-launch events %.1f)" % (
    args.workload, layouts["whole"][0], args.parts, layouts["part"][0], len(parts), args.parts, fps_clean, layouts["parts"][0]))
names = sorted(layouts["whole"][1], key=lambda k: -layouts["parts"][1].get(k, dict(ms=0))["ms"])
print("%-18s %5s %10s %10s %12s %10s   (us per pair; `parts` = elapsed under contention on 1/%d of the chip; chip-us = parts / %d)" % (
    "class", "n", "whole", "part", "parts", "chip-us", args.parts, args.parts * args.per_part))
tot = [0.0, 0.0, 0.0]
for k in names:
    row = []
    for j, lay in enumerate(("whole", "part", "parts")):
        fps, prof, npairs = layouts[lay]
        v = prof.get(k, dict(ms=0.0, launches=0))
        row.append(v["ms"] * 1e3 / npairs)
        tot[j] += row[-1]
    n = layouts["whole"][1][k]["launches"] / layouts["whole"][2]
    print("%-18s %5.1f %10.1f %10.1f %12.1f %10.1f" % (k, n, row[0], row[1], row[2], row[2] / (args.parts * args.per_part)))
print("%-18s %5s %10.1f %10.1f %12.1f %10.1f" % ("total", "", tot[0], tot[1], tot[2], tot[2] / (args.parts * args.per_part)))
