"""PCIe-inclusive throughput of the host-buffer entry points (frames in pageable host memory): process() from 1 and 2 caller
threads vs process_batch() from one thread vs stream mode (each frame of the sequence uploaded once, rife_hip_frame_*)."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import gen_models, gen_frames
amd = importlib.import_module("rife-ncnn-vulkan_amd")
g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
for (w, h) in ((3840, 2160), (1920, 1080)):
    base = gen_frames.smooth_pair(w // 4, h // 4, 3)
    fr = [np.ascontiguousarray(np.kron(np.roll(base[i % 2], 5 * i, axis=1), np.ones((4, 4, 1), np.uint8))) for i in range(9)]
    n = 24
    pairs = [(fr[i % 8], fr[i % 8 + 1], 0.5) for i in range(n)]
    for a, b, t in pairs[:3]: g.process(a, b, t)
    t0 = time.perf_counter()
    for a, b, t in pairs: g.process(a, b, t)
    d1 = time.perf_counter() - t0
    def worker(k):
        for i in range(k, n, 2): g.process(*pairs[i])
    t0 = time.perf_counter(); th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]; [x.start() for x in th]; [x.join() for x in th]
    d2 = time.perf_counter() - t0
    outs = [np.empty((h, w, 3), np.uint8) for _ in range(n)]
    g.process_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], outs)      # touches the output pages once
    t0 = time.perf_counter()
    g.process_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], outs)
    d3 = time.perf_counter() - t0
    # stream mode: a sequence of n + 1 frames, pair i = frames (i, i + 1); whoever needs a frame first uploads it
    seq = [fr[i % 9] for i in range(n + 1)]
    def stream_run(nthreads):
        res, locks = [None] * (n + 1), [threading.Lock() for _ in range(n + 1)]
        uses = [1] + [2] * (n - 1) + [1]
        def on_device(i):
            with locks[i]:
                if res[i] is None: res[i] = g.upload(seq[i])
                return res[i]
        def w2(k):
            for i in range(k, n, nthreads):
                g.process_frames(on_device(i), on_device(i + 1), 0.5, outs[i])
                for j in (i, i + 1):                       # a frame retires after the two pairs that use it
                    with locks[j]:
                        uses[j] -= 1
                        if uses[j] == 0: res[j].release()
        t0 = time.perf_counter(); th = [threading.Thread(target=w2, args=(k,)) for k in range(nthreads)]; [x.start() for x in th]; [x.join() for x in th]
        d = time.perf_counter() - t0
        return d
    stream_run(2)
    d4, d5 = stream_run(1), stream_run(2)
    t0 = time.perf_counter()
    def worker_o(k):
        for i in range(k, n, 2): g.process(pairs[i][0], pairs[i][1], 0.5, outs[i])
    th = [threading.Thread(target=worker_o, args=(k,)) for k in range(2)]; [x.start() for x in th]; [x.join() for x in th]
    d6 = time.perf_counter() - t0
    print("%dx%d stream mode (frames uploaded once): 1 thread %.1f fps, 2 threads %.1f fps; process() into reused outputs, 2 threads %.1f fps" % (w, h, n / d4, n / d5, n / d6))
    print("%dx%d host buffers: process() 1 thread %.1f fps, 2 threads %.1f fps, process_batch() 1 thread %.1f fps" % (w, h, n / d1, n / d2, n / d3))
