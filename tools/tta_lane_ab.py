"""GPU box: A/B of the -x -z lane layout (RIFE_HIP_TTA_LANE_PARTS, a process-scope switch of the test build): every variant in its own child process,
alternating, in one call: outputs per second at 3840x2160 (resident frames, one caller), md5 of the output frame.
    python tools/tta_lane_ab.py [0 2 4]"""
import hashlib, importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    from tools import gen_frames, gen_models
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    t = amd.test_build()
    w, h, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    g = t.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v4=True); g.load(gen_models.ensure(None, "rife-v4.6"))
    try:
        pair = gen_frames.tiled_real_pair(w // 640)
    except Exception:
        pair = gen_frames.smooth_pair_native(w, h, 1000)
    fr = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pair]
    out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(2):
        g.process_device(fr[0].data_ptr(), fr[1].data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        g.process_device(fr[i % 2].data_ptr(), fr[(i + 1) % 2].data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.process_device(fr[0].data_ptr(), fr[1].data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    print("RESULT %.3f %s" % (n / dt, hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:10]))
    sys.exit(0)
variants = sys.argv[1:] or ["0", "2", "4"]
for size, n in (((3840, 2160), 24), ((1920, 1080), 60)):
    for rnd in range(3):
        for v in variants:
            env = dict(os.environ, RIFE_HIP_TTA_LANE_PARTS=v)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(size[0]), str(size[1]), str(n)], capture_output=True, text=True, env=env)
            r = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            print("%dx%d -x -z RIFE_HIP_TTA_LANE_PARTS=%s round %d: %s" % (size[0], size[1], v, rnd, ("%s frames/s, md5 %s" % tuple(r[0].split()[1:3])) if r else "FAILED " + p.stderr[-400:]), flush=True)
