"""Parity report per BASELINE config (SURVEY.md §8d "Parity reporting"): HIP engine vs CPU oracle on the F1 (the reference's real frame pair tiled to size), F2 (smooth synthetic, native resolution) and
F3 (noise) frames: max |diff| in LSB, share of channels with diff 0 / 1 / >= 2, PSNR.  Run on the GPU box; the output is
committed as profiles/<round>/parity_report.txt.
Round 4: every row is measured against TWO CPU references - the restated oracle (oracle/rife_oracle.cpp, pitch-correct crop) and, in the last
columns, the reference's OWN compiled CPU path (oracle/_ref/libref_rife.so = /root/reference/src/rife.cpp + warp.cpp built unmodified by
oracle/refbuild/Makefile; widths here are multiples of 32, where its flat crop and the GPU shader agree) - and all rows are filled: F3 at 4K and
F2 / F3 under -x -z too (--full; the CPU side of those rows takes ~10 minutes)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import gen_models, gen_frames
from oracle import pyoracle, pyref
amd = importlib.import_module("rife-ncnn-vulkan_amd")

CONFIGS = [  # name, family, w, h, timesteps, flags
    ("C1 rife-v2.3 640x360", "rife-v2.3", 640, 360, [0.5], {}),
    ("C2 rife-v2.3 1920x1080", "rife-v2.3", 1920, 1080, [0.5], {}),
    ("C3 rife-v4.6 1920x1080 timestep sweep", "rife-v4.6", 1920, 1080, [0.125, 0.25, 0.5, 0.7, 0.9], {}),
    ("C4 rife-v4.6 3840x2160 (-u is a no-op for v4)", "rife-v4.6", 3840, 2160, [0.5], {"uhd_mode": True}),
    ("C5 rife-v4.6 3840x2160 -x -z", "rife-v4.6", 3840, 2160, [0.5], {"tta_mode": True, "tta_temporal_mode": True}),
    ("+  rife-v4 1920x1080", "rife-v4", 1920, 1080, [0.4], {}),
    ("+  rife-v3.1 1920x1080", "rife-v3.1", 1920, 1080, [0.5], {}),
    ("+  rife (v1) 1920x1080", "rife", 1920, 1080, [0.5], {}),
    ("+  rife-HD 1920x1080 -u", "rife-HD", 1920, 1080, [0.5], {"uhd_mode": True}),
]
quick = "--quick" in sys.argv
full = "--full" in sys.argv
have_ref = pyref.available()
print("%-48s %-7s %8s %10s %10s %10s %8s | %s" % ("config", "frames", "max LSB", "diff = 0", "diff = 1", "diff >= 2", "PSNR dB",
                                                "vs the reference build: max LSB, diff = 0, oracle == reference build" if have_ref else "(oracle/_ref not built)"))
for name, fam, w, h, ts, kw in CONFIGS:
    if quick and w > 1920:
        continue
    fl = dict(kw, rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    d = gen_models.ensure(None, fam)
    g = amd.RIFE(0, **fl); g.load(d)
    nthr = min(len(os.sched_getaffinity(0)), 64)
    o = pyoracle.OracleRIFE(num_threads=nthr, **fl); o.set_gpu_crop(1); o.load(d)
    r = None
    if have_ref:
        r = pyref.RefRIFE(num_threads=nthr, **fl); r.load(d)
    for kind in ("F1 real frames tiled", "F2 smooth", "F3 noise"):
        if not full and kind == "F3 noise" and (w > 1920 or "tta_mode" in kw):
            continue                                   # keep the CPU side of the report to a few minutes
        if not full and kind == "F2 smooth" and "tta_mode" in kw:
            continue
        diffs, rdiffs, same = [], [], True
        for i, t in enumerate(ts):
            if kind.startswith("F1"):
                a, b = gen_frames.tiled_real_pair(w // 640)
                if i & 1: a, b = b, a
            elif kind == "F2 smooth":
                a, b = gen_frames.smooth_pair(w, h, 1000 + i) if w * h < 4000000 else gen_frames.smooth_pair_native(w, h, 1000 + i)
            else:
                a, b = gen_frames.noise_pair(w, h, 7 + i)
            got, want = g.process(a, b, t), o.process(a, b, t)
            diffs.append(np.abs(got.astype(np.int32) - want.astype(np.int32)).ravel())
            if r is not None and not ("tta_mode" in kw and w > 1920 and not full):      # the reference build runs the 16 passes too: only with --full at 4K
                rw = r.process(a, b, t)
                rdiffs.append(np.abs(got.astype(np.int32) - rw.astype(np.int32)).ravel())
                same = same and np.array_equal(rw, want)
        dd = np.concatenate(diffs)
        mse = float((dd.astype(np.float64) ** 2).mean())
        psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        tail = ""
        if rdiffs:
            rd = np.concatenate(rdiffs)
            tail = " | %d  %9.5f%%  %s" % (rd.max(), 100 * (rd == 0).mean(), "bit-identical" if same else "DIFFERENT")
        print("%-48s %-7s %8d %9.5f%% %9.5f%% %9.5f%% %8.2f%s" % (name, kind[:2], dd.max(), 100 * (dd == 0).mean(), 100 * (dd == 1).mean(), 100 * (dd >= 2).mean(), psnr, tail))
        sys.stdout.flush()
