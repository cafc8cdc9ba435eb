"""Parity report per BASELINE config (SURVEY.md §8d "Parity reporting"): HIP engine vs CPU oracle on the F1 (the reference's real frame pair tiled to size), F2 (smooth synthetic, native resolution) and
F3 (noise) frames: max |diff| in LSB, share of channels with diff 0 / 1 / >= 2, PSNR.  Run on the GPU box; the output is
committed as profiles/<round>/parity_report.txt."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import gen_models, gen_frames
from oracle import pyoracle
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
print("%-48s %-7s %8s %10s %10s %10s %8s" % ("config", "frames", "max LSB", "diff = 0", "diff = 1", "diff >= 2", "PSNR dB"))
for name, fam, w, h, ts, kw in CONFIGS:
    if quick and w > 1920:
        continue
    fl = dict(kw, rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    d = gen_models.ensure(None, fam)
    g = amd.RIFE(0, **fl); g.load(d)
    o = pyoracle.OracleRIFE(num_threads=min(len(os.sched_getaffinity(0)), 64), **fl); o.set_gpu_crop(1); o.load(d)
    for kind in ("F1 real frames tiled", "F2 smooth", "F3 noise"):
        if kind == "F3 noise" and (w > 1920 or "tta_mode" in kw):
            continue                                   # keep the CPU side of the report to a few minutes
        if kind == "F2 smooth" and "tta_mode" in kw:
            continue
        diffs = []
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
        dd = np.concatenate(diffs)
        mse = float((dd.astype(np.float64) ** 2).mean())
        psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        print("%-48s %-7s %8d %9.5f%% %9.5f%% %9.5f%% %8.2f" % (name, kind[:2], dd.max(), 100 * (dd == 0).mean(), 100 * (dd == 1).mean(), 100 * (dd >= 2).mean(), psnr))
        sys.stdout.flush()
