"""Extreme frame sizes through every model family: HIP engine vs CPU oracle (GPU box)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import gen_models, gen_frames
from oracle import pyoracle
amd = importlib.import_module("rife-ncnn-vulkan_amd")
bad = 0
for fam in ("rife-v4.6", "rife-v4", "rife-v2.3", "rife-v3.1", "rife", "rife-HD"):
    kw = dict(rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    d = gen_models.ensure(None, fam)
    g = amd.RIFE(0, **kw); g.load(d)
    o = pyoracle.OracleRIFE(**kw); o.set_gpu_crop(1); o.load(d)
    for (w, h) in ((1, 1), (31, 33), (8, 300), (520, 16), (33, 32)):
        a, b = gen_frames.smooth_pair(w, h, 77)
        try:
            got, want = g.process(a, b, 0.5), o.process(a, b, 0.5)
            d8 = np.abs(got.astype(int) - want.astype(int))
            ok = d8.max() <= 1
            print(fam, (w, h), "max", int(d8.max()), "exact %.3f" % (d8 == 0).mean(), "" if ok else "  <-- FAIL")
            bad += not ok
        except Exception as e:
            print(fam, (w, h), "EXC", str(e)[:100]); bad += 1
print("failures:", bad)
