"""GPU box: the weight-stationary K-split trunk kernel (csrc/conv_ks.h, round 4) against the kernels it replaces, in ONE call (boxes differ by
up to 10 %): for every RIFE_HIP_KS mask (0 = conv_row / conv_t64 as in round 3; 1 = 128 channels; 2 = 96 channels on small grids; 4 = 96 channels at
every size; 8 = 192 channels) an engine is created, checked against mask 0 (differing bytes, largest difference) and timed: per-class kernel time
per pair with one pair in flight (HIP events on the launch stream) and frames/s with two pairs in flight from resident frames.
    python tools/ks_ab.py [masks ...]      -> gpurun_out/ks_ab.txt"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # the kernel-selection switches this tool flips live in the test build (librife_hip_test.so)
import torch
from tools import gen_frames, gen_models

masks = [int(x) for x in sys.argv[1:] if not x.startswith("div=")] or [0, 1, 3, 7]
for x in sys.argv[1:]:
    if x.startswith("div="): os.environ["RIFE_HIP_KS_DIV"] = x[4:]
d = gen_models.ensure(None, "rife-v4.6")
os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/ks_ab.txt", "a")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

def engine(mask):
    os.environ["RIFE_HIP_KS"] = str(mask)
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    os.environ.pop("RIFE_HIP_KS", None)
    return g

for (w, h, npairs) in ((1920, 1080, 64), (3840, 2160, 32)):
    a, b = gen_frames.tiled_real_pair(w // 640)
    ref_out = None
    eng = {m: engine(m) for m in masks}
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for rep in range(2):
        for m in masks:
            g = eng[m]
            out = g.process(a, b, 0.5)
            if m == masks[0] and ref_out is None: ref_out = out
            df = np.abs(out.astype(np.int32) - ref_out.astype(np.int32))
            again = g.process(a, b, 0.5)
            # one pair in flight, per-class events
            g.profile_enable(True)
            for _ in range(8): g.process(a, b, 0.5)
            prof = g.profile_read(); g.profile_enable(False)
            cls = ", ".join("%s %.4f" % (k, v["ms"] / 8) for k, v in sorted(prof.items()) if k.startswith("trunk"))
            tot = sum(v["ms"] for v in prof.values()) / 8
            # two pairs in flight, resident frames
            outs = [torch.empty_like(da) for _ in range(2)]
            strs = [torch.cuda.Stream() for _ in range(2)]
            def worker(i, n):
                for _ in range(n): g.process_device(da.data_ptr(), db.data_ptr(), w, h, 0.5, outs[i].data_ptr(), strs[i].cuda_stream)
                strs[i].synchronize()
            for n in (4, npairs):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                th = [threading.Thread(target=worker, args=(i, n)) for i in range(2)]
                [t.start() for t in th]; [t.join() for t in th]
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            say("[div %s] " % os.environ.get("RIFE_HIP_KS_DIV", "1") + "%dx%d KS=%d rep %d: %.1f frames/s (2 in flight), kernel ms/pair %.3f | %s | vs KS=%d: %d of %d bytes differ, max %d; deterministic %s" %
                (w, h, m, rep, 2 * npairs / dt, tot, cls, masks[0], int((df > 0).sum()), df.size, int(df.max()), np.array_equal(out, again)))
    del eng
