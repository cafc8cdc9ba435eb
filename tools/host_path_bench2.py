"""GPU box: PCIe-inclusive throughput of the host-frame entry points, pageable vs page-locked frames (rife_hip_host_alloc),
1..4 caller threads of process() and process_batch() from one thread; rife-v4.6."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import gen_models, gen_frames
amd = importlib.import_module("rife-ncnn-vulkan_amd")
# --test-build: the test build (reads RIFE_HIP_POOL_PARTS: 0 = whole-chip pool streams at every caller count, the round-5 behaviour); default: the product
mod = amd.test_build() if "--test-build" in sys.argv else amd
print("frames: %s; library: %s, RIFE_HIP_POOL_PARTS=%s" % ("smooth blocky pair" if "--smooth" in sys.argv else "F1 (real pair tiled)", "test build" if mod is not amd else "product", os.environ.get("RIFE_HIP_POOL_PARTS", "(unset)")), flush=True)
g = mod.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
for (w, h) in ((3840, 2160), (1920, 1080)):
    # frames: F1 like bench.py (the reference's real pair tiled to size, rolled per index) - the matrix pipe's clock depends on the data (profiles/r6/README.md), so
    # the resident-frame `value` of bench.py and these PCIe-inclusive rates are only comparable on the same frames; --smooth: the blocky smooth pair of rounds 2 - 5
    if "--smooth" in sys.argv:
        base = gen_frames.smooth_pair(w // 4, h // 4, 3)
        fr = [np.ascontiguousarray(np.kron(np.roll(base[i % 2], 5 * i, axis=1), np.ones((4, 4, 1), np.uint8))) for i in range(9)]
    else:
        base = gen_frames.tiled_real_pair(w // 640)
        fr = [np.ascontiguousarray(np.roll(base[i % 2], (2 * (i // 2), 5 * (i // 2)), axis=(0, 1))) for i in range(9)]
    n = 96
    for kind in ("pageable", "page-locked"):
        if kind == "page-locked":
            pf = [amd.pinned_empty(f.shape) for f in fr]
            for d, s in zip(pf, fr): d[...] = s
            frames, outs = pf, [amd.pinned_empty((h, w, 3)) for _ in range(n)]
        else:
            frames, outs = fr, [np.empty((h, w, 3), np.uint8) for _ in range(n)]
        pairs = [(frames[i % 8], frames[i % 8 + 1], 0.5) for i in range(n)]
        for i in range(3): g.process(*pairs[i], outimage=outs[i])
        res = []
        for nt in (1, 2, 3, 4):
            def worker(k):
                for i in range(k, n, nt): g.process(*pairs[i], outimage=outs[i])
            th = [threading.Thread(target=worker, args=(k,)) for k in range(nt)]; [x.start() for x in th]; [x.join() for x in th]      # untimed: the pool settles on this caller count
            t0 = time.perf_counter(); th = [threading.Thread(target=worker, args=(k,)) for k in range(nt)]; [x.start() for x in th]; [x.join() for x in th]
            res.append("%d thr %.1f" % (nt, n / (time.perf_counter() - t0)))
        g.process_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], outs)
        t0 = time.perf_counter()
        g.process_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], outs)
        db = time.perf_counter() - t0
        print("%dx%d %-11s process(): %s fps | process_batch() from one thread: %.1f fps" % (w, h, kind, ", ".join(res), n / db), flush=True)
