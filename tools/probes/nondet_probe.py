"""GPU box: is the plain rife-v4.6 pass deterministic inside one process at a large size, and if not, which stage differs first?
    python tools/probes/nondet_probe.py [w h reps]
Runs the same pair `reps` times: frames byte-compared, then the flow blobs flow0 .. flow3 (rife_hip_v4_extract_flow) compared run to run."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tools import gen_frames, gen_models
amd = importlib.import_module("rife-ncnn-vulkan_amd")
w, h, reps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160, 4)
g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
a, b = gen_frames.smooth_pair(w, h, 705)
outs = [g.process(a, b, 0.5) for _ in range(reps)]
for i in range(1, reps):
    d = outs[i] != outs[0]
    ys, xs = np.nonzero(d.any(axis=2))
    print("frame run %d vs run 0: differing bytes %d" % (i, int(d.sum())), "rows", sorted(set((ys // 8 * 8).tolist()))[:10], "cols", sorted(set((xs // 32 * 32).tolist()))[:10])
for fi in range(4):
    fl = [g.v4_extract_flow(a, b, 0.5, fi) for _ in range(reps)]
    for i in range(1, reps):
        d = np.abs(fl[i] - fl[0])
        n = int((d > 0).sum())
        print("flow%d run %d vs run 0: differing values %d of %d, max %.3g" % (fi, i, n, d.size, float(d.max())), "" if n == 0 else "channels %s" % sorted(set(np.nonzero(d)[0].tolist())))
