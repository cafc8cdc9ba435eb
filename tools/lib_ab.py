"""GPU box: two BUILDS of the library against each other in one call (each in its own process, alternating): per-class kernel time per pair
(one pair in flight), frames/s with three pairs in flight on resident frames, run-to-run determinism of the frames (N repeats), byte comparison
of the frames between the builds.  The child process points the Python mirror at the build under test.
    python tools/lib_ab.py libA.so libB.so [reps]      -> gpurun_out/lib_ab.txt"""
import importlib, json, os, subprocess, sys, threading, time
import numpy as np

def child(libpath, sizes, det_reps):
    sys.path.insert(0, os.getcwd())
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    amd.LIB_PATH = libpath                                  # another BUILD of the same sources, loaded instead of librife_hip.so (this tool only)
    import torch, hashlib
    from tools import gen_frames, gen_models
    g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
    out = {}
    for (w, h, npairs) in sizes:
        a, b = gen_frames.tiled_real_pair(w // 640)
        x = g.process(a, b, 0.5)
        ndiff = sum(int(not np.array_equal(x, g.process(a, b, 0.5))) for _ in range(det_reps))
        g.profile_enable(True)
        for _ in range(8): g.process(a, b, 0.5)
        prof = g.profile_read(); g.profile_enable(False)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        outs = [torch.empty_like(da) for _ in range(3)]
        def worker(i, n):
            st = torch.cuda.Stream()
            for _ in range(n): g.process_device(da.data_ptr(), db.data_ptr(), w, h, 0.5, outs[i].data_ptr(), st.cuda_stream)
            st.synchronize()
        fps = []
        for n in (4, npairs, npairs):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i, n)) for i in range(3)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize(); fps.append(3 * n / (time.perf_counter() - t0))
        out["%dx%d" % (w, h)] = {"md5": hashlib.md5(x.tobytes()).hexdigest(), "nondeterministic_repeats": ndiff, "fps3": fps[1:],
                                 "ms": {k: v["ms"] / 8 for k, v in prof.items()}}
    print("RESULT " + json.dumps(out))

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], [(1920, 1080, 64), (3840, 2160, 32)], int(sys.argv[3])); sys.exit(0)
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    reps = sys.argv[3] if len(sys.argv) > 3 else "20"
    os.makedirs("gpurun_out", exist_ok=True)
    log = open("gpurun_out/lib_ab.txt", "a")
    def say(s):
        print(s, flush=True); log.write(s + "\n"); log.flush()
    res = {}
    for rnd in range(2):
        for lp in libs:
            p = subprocess.run([sys.executable, __file__, "--child", lp, reps], capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                say("%s: FAILED\n%s" % (lp, p.stderr[-2000:])); continue
            r = json.loads(line[0][7:]); res.setdefault(lp, []).append(r)
            for size, v in r.items():
                tot = sum(v["ms"].values())
                top = ", ".join("%s %.3f" % kv for kv in sorted(v["ms"].items(), key=lambda kv: -kv[1])[:12])
                say("%s %s round %d: %.1f / %.1f frames/s (3 in flight), kernel ms/pair %.3f, md5 %s, %d of %s repeats differ | %s" %
                    (os.path.basename(lp), size, rnd, v["fps3"][0], v["fps3"][1], tot, v["md5"][:8], v["nondeterministic_repeats"], reps, top))
