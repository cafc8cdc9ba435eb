"""GPU box: resident-frame throughput of rife-v4.6, the ways a caller can keep the chip busy (one call, same frames):
  streams K   : K host threads, each its own stream, rife_hip_process_device per pair (bench.py's round 1-3 mode, K = 2)
  batch N x T : T host threads, each rife_hip_process_device_batch of N pairs per call on its own stream (lockstep groups of two)
for RIFE_HIP_KS = 0 / 1 (conv_row vs the round-4 conv_ks trunk kernel for the 128-channel block).
    python tools/batch_ab.py      -> gpurun_out/batch_ab.txt"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # the kernel-selection switches this tool flips live in the test build (librife_hip_test.so)
import torch
from tools import gen_frames, gen_models
d = gen_models.ensure(None, "rife-v4.6")
os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/batch_ab.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

def engine(mask):
    os.environ["RIFE_HIP_KS"] = str(mask)
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    os.environ.pop("RIFE_HIP_KS", None)
    return g

def run_threads(fn, T, reps):
    def worker(i):
        st = torch.cuda.Stream()
        for _ in range(reps): fn(i, st)
        st.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return time.perf_counter() - t0

for (w, h, total) in ((1920, 1080, 192), (3840, 2160, 64)):
    a, b = gen_frames.tiled_real_pair(w // 640)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    outs = [torch.empty_like(da) for _ in range(16)]
    for mask in (0, 1):
        g = engine(mask)
        for rep in range(2):
            res = []
            for K in (2, 3):
                fn = lambda i, st: g.process_device(da.data_ptr(), db.data_ptr(), w, h, 0.5, outs[i].data_ptr(), st.cuda_stream)
                run_threads(fn, K, 4)
                dt = run_threads(fn, K, total // K)
                res.append("streams %d: %.1f" % (K, (total // K) * K / dt))
            for (N, T) in ((2, 1), (4, 1), (2, 2), (4, 2), (6, 1)):
                def fn(i, st, N=N):
                    o = outs[i * N:(i + 1) * N]
                    g.process_device_batch([da.data_ptr()] * N, [db.data_ptr()] * N, w, h, [0.5] * N, [x.data_ptr() for x in o], st.cuda_stream)
                run_threads(fn, T, 2)
                reps = max(1, total // (N * T))
                dt = run_threads(fn, T, reps)
                res.append("batch %dx%d: %.1f" % (N, T, reps * N * T / dt))
            say("%dx%d KS=%d rep %d frames/s | " % (w, h, mask, rep) + " | ".join(res))
        del g
