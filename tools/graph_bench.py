"""1080p / 720p / 360p single-stream throughput of the plain rife-v4.6 path with and without hipGraph replay (profiler off)."""
import importlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from tools import gen_models, gen_frames
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    g = amd.RIFE(0, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
    for (w, h) in ((1920, 1080), (1280, 720), (640, 360)):
        a, b = gen_frames.smooth_pair(w, h, 5)
        fa, fb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        st = torch.cuda.Stream()
        for i in range(5): g.process_device(fa.data_ptr(), fb.data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter(); n = 200
        for i in range(n): g.process_device(fa.data_ptr(), fb.data_ptr(), w, h, 0.3 + 0.001 * i, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        print("RIFE_HIP_GRAPH=%s %dx%d  %.3f ms/pair = %.0f frames/s" % (os.environ.get("RIFE_HIP_GRAPH", "1"), w, h, dt * 1e3, 1 / dt))
else:
    for mode in ("1", "0", "1", "0"):
        env = dict(os.environ, RIFE_HIP_GRAPH=mode)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
