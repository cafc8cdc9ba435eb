"""Ad-hoc GPU diagnostics (prints error magnitudes instead of asserting)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle
from tools import gen_frames, gen_models
amd = importlib.import_module("rife-ncnn-vulkan_amd")

print("devices", amd.device_count())
rng = np.random.default_rng(0)
for (cin, cout, stride, h, w) in [(16, 24, 1, 5, 7), (64, 64, 1, 24, 64), (96, 96, 1, 17, 33), (7, 96, 2, 34, 60), (12, 32, 2, 36, 70), (192, 192, 1, 8, 32)]:
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = pyoracle.conv2d(x, wt, b, stride=stride, pad=1)
    got = amd.op_conv3x3(x, wt, b, stride=stride)
    print("conv", cin, cout, stride, h, w, "maxerr", np.abs(got - want).max())
for (cin, cout, h, w) in [(64, 24, 16, 40), (32, 4, 12, 20)]:
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    print("deconv", cin, cout, h, w, "maxerr", np.abs(amd.op_deconv4x4(x, wt, b) - pyoracle.deconv2d(x, wt, b)).max())
img = rng.uniform(0, 1, (3, 45, 70)).astype(np.float32); flow = (rng.standard_normal((2, 45, 70)) * 9).astype(np.float32)
print("warp exact", np.array_equal(amd.op_warp(img, flow), pyoracle.warp(img, flow)))

d = gen_models.ensure(None, "rife-v4.6")
g = amd.RIFE(0, rife_v4=True); g.load(d)
o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(d)
a, b = gen_frames.smooth_pair(160, 96, 21)
for fi in range(4):
    got = g.v4_extract_flow(a, b, 0.5, fi); want = o.v4_extract(a, b, 0.5, "flow%d" % fi)
    print("flow", fi, "maxerr", np.abs(got - want).max(), "absmean", np.abs(want).mean())
for (w, h) in [(160, 96), (640, 360), (1920, 1080)]:
    a, b = gen_frames.smooth_pair(w, h, 1000)
    t0 = time.time(); got = g.process(a, b, 0.5); t1 = time.time(); want = o.process(a, b, 0.5); t2 = time.time()
    dd = np.abs(got.astype(int) - want.astype(int))
    print("process", w, h, "maxLSB", dd.max(), "frac0", (dd == 0).mean(), "gpu_s", t1 - t0, "oracle_s", t2 - t1)
g.profile_enable(True)
for _ in range(3): g.process(a, b, 0.5)
for k, v in sorted(g.profile_read().items(), key=lambda kv: -kv[1]["ms"]):
    print("%-14s launches %4d  ms %9.3f  TFLOP/s %7.2f" % (k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9))

# ---- rife-v2.3 ----
d2 = gen_models.ensure(None, "rife-v2.3")
g2 = amd.RIFE(0, rife_v2=True); g2.load(d2)
o2 = pyoracle.OracleRIFE(rife_v2=True); o2.set_gpu_crop(1); o2.load(d2)
for (w, h) in [(64, 64), (160, 96), (640, 360)]:
    a, b = gen_frames.smooth_pair(w, h, 300)
    got = g2.process(a, b, 0.5); want = o2.process(a, b, 0.5)
    dd = np.abs(got.astype(int) - want.astype(int))
    print("v2.3 process", w, h, "maxLSB", dd.max(), "frac0", (dd == 0).mean(), "mean", got.mean(), want.mean())
