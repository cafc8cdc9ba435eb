"""End-to-end throughput of the rife-hip command line (files on disk -> files on disk), directory mode, rife-v4.6.
usage: cli_e2e_bench.py [formats, e.g. png,ppm] [-j settings, e.g. 1:2:2,8:3:8]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from tools import gen_models, gen_frames
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "rife-ncnn-vulkan_amd", "rife-hip")
model = gen_models.ensure(None, "rife-v4.6")
for (w, h, nin) in ((1920, 1080, 24), (3840, 2160, 12)):
    base = gen_frames.smooth_pair(w // 4, h // 4, 3)
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        for fmt in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("png", "ppm")):
            ind, outd = os.path.join(d, "in_" + fmt), os.path.join(d, "out_" + fmt)
            os.makedirs(ind); os.makedirs(outd)
            for i in range(nin * (3 if fmt == "ppm" else 1)):      # no codec: longer run, so that start-up does not dominate
                f = np.kron(np.roll(base[i % 2], 7 * i, axis=1), np.ones((4, 4, 1), np.uint8))
                Image.fromarray(f).save(os.path.join(ind, "%04d.%s" % (i, fmt)))
            for jobs in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1:2:2", "4:2:4", "8:3:8")):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-i", ind, "-o", outd, "-m", model, "-j", jobs, "-f", "%08d." + fmt], capture_output=True, text=True,
                                   env=dict(os.environ, RIFE_HIP_CLI_TIMING="1"))
                dt = time.perf_counter() - t0
                n = len(os.listdir(outd))
                print("%dx%d %s -j %s: %d frames in %.2f s = %.1f frames/s (incl. process start + model load)%s" % (w, h, fmt, jobs, n, dt, n / dt, "" if r.returncode == 0 else "  rc=%d %s" % (r.returncode, r.stderr[:100])))
                print("    " + "".join(l for l in r.stderr.splitlines() if l.startswith("timing:")))
