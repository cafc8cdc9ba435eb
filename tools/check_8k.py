import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tools import gen_models, gen_frames
from oracle import pyoracle
amd = importlib.import_module("rife-ncnn-vulkan_amd")
d = gen_models.ensure(None, "rife-v4.6")
g = amd.RIFE(0, rife_v4=True); g.load(d)
o = pyoracle.OracleRIFE(rife_v4=True, num_threads=64); o.set_gpu_crop(1); o.load(d)
w, h = 7680, 4320
base = gen_frames.smooth_pair(w // 4, h // 4, 8)
a, b = [np.ascontiguousarray(np.kron(x, np.ones((4, 4, 1), np.uint8))) for x in base]
t0 = time.time(); got = g.process(a, b, 0.5); t1 = time.time(); got = g.process(a, b, 0.5); t2 = time.time()
want = o.process(a, b, 0.5); t3 = time.time()
d8 = np.abs(got.astype(np.int16) - want.astype(np.int16))
print("8K %dx%d: max LSB %d, exact %.5f%%; HIP (host buffers) %.1f ms warm, oracle %.1f s" % (w, h, d8.max(), 100 * (d8 == 0).mean(), (t2 - t1) * 1e3, t3 - t2))
