"""GPU box: how do the S16 / conv_t64 trunk and the per-tile trunk differ (bytes of the frame, block-3 flow), and is the new path deterministic?
Repeats a mixed-size sequence; on a mismatch prints where (block-3 trunk tile coordinates) the flows differ."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tools import gen_frames, gen_models
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # the kernel-selection switches this tool flips live in the test build (librife_hip_test.so)
d = gen_models.ensure(None, "rife-v4.6")
def eng(t64):
    os.environ["RIFE_HIP_T64"] = "1" if t64 else "0"
    g = amd.RIFE(0, rife_v4=True); g.load(d); return g
new, old = eng(True), eng(False)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for (w, h) in ((640, 360), (1, 1), (1920, 1080), (1000, 520), (100, 60), (1920, 1080)):
        a, b = gen_frames.smooth_pair(w, h, 6 + rep)
        x, y, x2 = new.process(a, b, 0.5), old.process(a, b, 0.5), new.process(a, b, 0.5)
        dd = np.abs(x.astype(int) - y.astype(int))
        exact = w * h >= 1000 * 520
        if (exact and dd.max() > 0) or dd.max() > 1 or (x != x2).any():
            bad += 1
            print(rep, w, h, "frame: differing bytes", int((dd > 0).sum()), "of", dd.size, "max", int(dd.max()), "| new vs new", int((x != x2).sum()))
            fn, fo, fn2 = new.v4_extract_flow(a, b, 0.5, 3), old.v4_extract_flow(a, b, 0.5, 3), new.v4_extract_flow(a, b, 0.5, 3)
            for name, df in (("new-old", np.abs(fn - fo).max(axis=0)), ("new-new", np.abs(fn - fn2).max(axis=0))):
                ys, xs = np.nonzero(df > 1e-4)
                print("   flow3", name, "max", float(df.max()), "pixels > 1e-4:", len(ys), "trunk tiles (ty, tx):", sorted(set(zip((ys // 32).tolist(), (xs // 128).tolist())))[:12])
print("mismatching cases:", bad)
