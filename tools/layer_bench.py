"""Every matrix layer of rife-v2.3 at 1920x1088 through the product's launch_conv() (bench hook rife_hip_bench_layer): wall time per launch with ONE
stream and at saturation (S concurrent streams, own tensors each), algorithmic TFLOP/s at saturation and the fraction of the f16 matrix peak issued
(x 2: hi and lo products).  The saturated figure is what a layer costs the chip inside bench.py's timed region (several pairs in flight).
    python tools/layer_bench.py [--streams 4] [--iters 20] [--only fb2_trunk,fus8]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib

HP, WP = 1088, 1920
# name, cin, cout, input h, w divisor, kind (0 s1, 1 s2, 2 deconv), input pixel stride, launches per pair
LAYERS = []
for b, (c, s) in enumerate([(384, 8), (256, 4), (192, 2), (96, 1)]):
    LAYERS += [("fb%d_stem0" % b, 6 if b == 0 else 10, c // 2, s, 1, 8 if b == 0 else 16, 1), ("fb%d_stem1" % b, c // 2, c, 2 * s, 1, c // 2, 1),
               ("fb%d_trunk" % b, c, c, 4 * s, 0, c, 6), ("fb%d_head" % b, c, 4, 4 * s, 2, c, 1)]
CI = [3, 32, 32, 32, 32, 64, 64, 128, 128, 256]; CO = [32, 32, 32, 32, 64, 64, 128, 128, 256, 256]; DV = [1, 2, 2, 4, 4, 8, 8, 16, 16, 32]
for i in range(10):
    LAYERS.append(("ctx%d" % i, CI[i], CO[i], DV[i], 1 if i % 2 == 0 else 0, 8 if i == 0 else CI[i], 2))
FI = [10, 32, 32, 64, 128, 128, 256, 256, 512, 512]; FO = [32, 32, 64, 64, 128, 128, 256, 256, 512, 512]
for i in range(10):
    LAYERS.append(("fus%d" % i, FI[i], FO[i], DV[i], 1 if i % 2 == 0 else 0, 16 if i == 0 else FI[i], 1))
for i, (ci, co, d) in enumerate([(1024, 256, 32), (512, 128, 16), (256, 64, 8), (128, 32, 4)]):
    LAYERS.append(("fus%d" % (10 + i), ci, co, d, 2, ci, 1))

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
args = ap.parse_args()
L = benchlib.lib()
L.rife_hip_bench_layer.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_float)]
only = set(args.only.split(",")) if args.only else None
print("%-11s %5s %5s %9s %4s | %8s %8s | %7s %6s | %9s" % ("layer", "cin", "cout", "in HxW", "kind", "alone us", "sat us", "TFLOP/s", "issued", "us/pair"))
tot = 0.0
for name, ci, co, d, kind, ld, n in LAYERS:
    if only and name not in only:
        continue
    h, w = HP // d, WP // d
    ms = (ctypes.c_float * 2)()
    rc = L.rife_hip_bench_layer(0, ci, co, h, w, kind, ld, args.streams, args.iters, ms)
    if rc:
        print("%-11s failed: %s" % (name, L.rife_hip_last_error().decode()))
        continue
    mo = (h // 2) * (w // 2) if kind == 1 else h * w
    gf = 2.0 * ci * co * (16 if kind == 2 else 9) * mo / 1e9
    tf = gf / ms[0]
    tot += ms[0] * 1e3 * n
    print("%-11s %5d %5d %4dx%-4d %4d | %8.1f %8.1f | %7.1f %5.1f%% | %9.1f" % (name, ci, co, h, w, kind, ms[1] * 1e3, ms[0] * 1e3, tf, 100 * 2 * tf / 2500.0, ms[0] * 1e3 * n))
print("sum over the pair's matrix launches at saturation: %.1f us" % tot)
