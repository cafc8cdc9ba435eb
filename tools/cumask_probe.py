"""GPU box: does partitioning the chip between the pairs in flight pay?  K host threads, each with its own stream restricted to 1/K of the compute
units (hipExtStreamCreateWithCUMask) and persistent grids sized for that part (RIFE_HIP_CUS), against K unrestricted streams.
    python tools/cumask_probe.py      -> gpurun_out/cumask_probe.txt"""
import ctypes, importlib, os, subprocess, sys, threading, time
import numpy as np

def child(K, layout, budget):
    if budget: os.environ["RIFE_HIP_CUS"] = str(budget)
    sys.path.insert(0, os.getcwd())
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    import torch
    from tools import gen_frames, gen_models
    hip = ctypes.CDLL("libamdhip64.so")
    g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nwords = (ncu + 31) // 32
    def make_stream(i):
        if layout == "none": return torch.cuda.Stream().cuda_stream, None
        bits = [0] * nwords
        for cu in range(ncu):
            mine = (cu * K // ncu == i) if layout == "blocks" else (cu % K == i)
            if mine: bits[cu // 32] |= 1 << (cu % 32)
        arr = (ctypes.c_uint32 * nwords)(*bits)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), nwords, arr)
        if rc: raise RuntimeError("hipExtStreamCreateWithCUMask rc=%d" % rc)
        return st.value, st
    res = []
    for (w, h, n) in ((1920, 1080, 96), (3840, 2160, 32)):
        a, b = gen_frames.tiled_real_pair(w // 640)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        outs = [torch.empty_like(da) for _ in range(K)]
        strs = [make_stream(i) for i in range(K)]
        def worker(i, reps):
            for _ in range(reps): g.process_device(da.data_ptr(), db.data_ptr(), w, h, 0.5, outs[i].data_ptr(), strs[i][0])
            hip.hipStreamSynchronize(ctypes.c_void_p(strs[i][0]))
        fps = []
        for reps in (4, n, n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i, reps)) for i in range(K)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize(); fps.append(K * reps / (time.perf_counter() - t0))
        ref = g.process(a, b, 0.5)
        ok = all(np.array_equal(o.cpu().numpy(), ref) for o in outs)
        res.append("%dx%d: %.1f / %.1f frames/s%s" % (w, h, fps[1], fps[2], "" if ok else " WRONG FRAMES"))
    print("RESULT K=%d mask=%s grids for %s CUs | " % (K, layout, budget or "all") + " | ".join(res))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3], int(sys.argv[4])); sys.exit(0)
    os.makedirs("gpurun_out", exist_ok=True)
    log = open("gpurun_out/cumask_probe.txt", "w")
    for rnd in range(2):
        for (K, layout, budget) in ((3, "none", 0), (4, "none", 0), (4, "interleaved", 64), (8, "interleaved", 32), (8, "blocks", 32), (6, "none", 0), (8, "none", 0), (8, "none", 32)):
            p = subprocess.run([sys.executable, __file__, "--child", str(K), layout, str(budget)], capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            s = line[0][7:] if line else "K=%d %s %d FAILED: %s" % (K, layout, budget, p.stderr[-600:])
            print(s, flush=True); log.write(s + "\n"); log.flush()
