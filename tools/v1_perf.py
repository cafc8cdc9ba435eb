import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tools import gen_models, gen_frames
amd = importlib.import_module("rife-ncnn-vulkan_amd")
for fam in ("rife", "rife-HD"):
    for (w, h) in ((1920, 1080), (3840, 2160)):
        g = amd.RIFE(0); g.load(gen_models.ensure(None, fam))
        a, b = gen_frames.smooth_pair(w // 4, h // 4, 5)
        fa = torch.from_numpy(np.ascontiguousarray(np.kron(a, np.ones((4, 4, 1), np.uint8)))).cuda()
        fb = torch.from_numpy(np.ascontiguousarray(np.kron(b, np.ones((4, 4, 1), np.uint8)))).cuda()
        out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        st = torch.cuda.Stream()
        for i in range(3): g.process_device(fa.data_ptr(), fb.data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter(); n = 10
        for i in range(n): g.process_device(fa.data_ptr(), fb.data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        g.profile_enable(True)
        for i in range(3): g.process_device(fa.data_ptr(), fb.data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        pr = g.profile_read()
        top = sorted(pr.items(), key=lambda kv: -kv[1]["ms"])[:6]
        print(fam, w, h, "%.2f ms/pair = %.1f fps" % (dt * 1e3, 1 / dt), {k: round(v["ms"] / 3, 2) for k, v in top})
        del g
