import importlib, sys, os
os.environ["RIFE_HIP_GRAPH"] = "1"     # opt-in path under test
sys.path.insert(0, os.getcwd())
import numpy as np
from tools import gen_models, gen_frames
from oracle import pyoracle
amd = importlib.import_module("rife-ncnn-vulkan_amd")
for fam in ("rife-v4.6", "rife-v4"):
    d = gen_models.ensure(None, fam)
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(d)
    for (w, h) in ((160, 96), (100, 60), (160, 96)):
        for i, t in enumerate((0.5, 0.25, 0.7, 0.9, 0.125)):      # 1st warm-up, 2nd capture, 3rd.. replays with new frames / timesteps
            a, b = gen_frames.smooth_pair(w, h, 40 + i)
            got, want = g.process(a, b, t), o.process(a, b, t)
            d8 = np.abs(got.astype(int) - want.astype(int))
            print(fam, w, h, t, "max", d8.max(), "exact %.4f" % (d8 == 0).mean())
            assert d8.max() <= 1
print("graph replay parity ok")
