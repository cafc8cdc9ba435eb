"""GPU box: rife_hip_process_batch from ONE caller thread, lockstep groups (batched coarse-block launches) against the per-pair
workers (RIFE_HIP_BATCH_GROUPS=0), pageable and page-locked host frames, 1920x1080 and 3840x2160; per-class kernel time of the coarse trunks.
    python tools/batch_groups_bench.py      -> gpurun_out/batch_groups.txt"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # the kernel-selection switches this tool flips live in the test build (librife_hip_test.so)
from tools import gen_frames, gen_models
g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
log = open("gpurun_out/batch_groups.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
for (w, h, n) in ((1920, 1080, 96), (3840, 2160, 48)):
    base = gen_frames.tiled_real_pair(w // 640)
    fr = [np.ascontiguousarray(np.roll(base[i & 1], (2 * (i // 2), 5 * (i // 2)), axis=(0, 1))) for i in range(4)]
    for kind in ("pageable", "page_locked"):
        hin = fr if kind == "pageable" else [amd.pinned_empty((h, w, 3)) for _ in fr]
        if kind != "pageable":
            for d, s_ in zip(hin, fr): d[...] = s_
        outs = [np.empty((h, w, 3), np.uint8) if kind == "pageable" else amd.pinned_empty((h, w, 3)) for _ in range(n)]
        a0 = [hin[i % 4] for i in range(n)]; a1 = [hin[(i + 1) % 4] for i in range(n)]; ts = [(0.5, 0.125, 0.25, 0.7, 0.9)[i % 5] for i in range(n)]
        for rep in range(2):
            for mode in ("1", "0"):
                os.environ["RIFE_HIP_BATCH_GROUPS"] = mode
                g.process_batch(a0[:6], a1[:6], ts[:6], outs[:6])
                t0 = time.perf_counter(); g.process_batch(a0, a1, ts, outs); dt = time.perf_counter() - t0
                say("%dx%d %-11s %s: %.1f frames/s (%d pairs)" % (w, h, kind, "lockstep groups (batched coarse trunks)" if mode == "1" else "per-pair workers                    ", n / dt, n))
    # kernel time of the coarse trunks per pair, events on the launch streams
    for mode in ("1", "0"):
        os.environ["RIFE_HIP_BATCH_GROUPS"] = mode
        g.profile_enable(True)
        g.process_batch(a0[:24], a1[:24], ts[:24], outs[:24])
        prof = g.profile_read(); g.profile_enable(False)
        say("%dx%d per-class ms per pair (%s): " % (w, h, "groups" if mode == "1" else "per pair") + ", ".join("%s %.4f" % (k, v["ms"] / 24) for k, v in sorted(prof.items()) if k.startswith("trunk")))
os.environ.pop("RIFE_HIP_BATCH_GROUPS", None)
