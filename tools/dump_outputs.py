"""GPU box: frames of a fixed set of cases from the library currently in place -> npz (for byte comparison of two library builds:
tools/ab_exact.sh).  Cases: rife-v4.6 plain at several sizes / timesteps incl. large flows and ragged sizes, v4.6 -x -z small, rife-v2.3 small."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tools import gen_frames, gen_models
amd = importlib.import_module("rife-ncnn-vulkan_amd")
out = {}
g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
for i, (w, h, t) in enumerate(((640, 360, 0.5), (100, 60, 0.3), (33, 47, 0.9), (1920, 1080, 0.25), (1000, 520, 0.7), (3840, 2160, 0.5))):
    a, b = gen_frames.smooth_pair(w, h, 700 + i)
    out["v46_%dx%d" % (w, h)] = g.process(a, b, t)
rng = np.random.default_rng(7)
a, b = rng.integers(0, 256, (200, 328, 3), dtype=np.uint8), rng.integers(0, 256, (200, 328, 3), dtype=np.uint8)
out["v46_noise"] = g.process(a, b, 0.5)
gt = amd.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v4=True); gt.load(gen_models.ensure(None, "rife-v4.6"))
a, b = gen_frames.smooth_pair(160, 96, 77)
out["v46_tta"] = gt.process(a, b, 0.5)
g2 = amd.RIFE(0, rife_v2=True); g2.load(gen_models.ensure(None, "rife-v2.3"))
a, b = gen_frames.smooth_pair(320, 192, 78)
out["v23"] = g2.process(a, b, 0.5)
np.savez_compressed(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: v.shape for k, v in out.items()})
