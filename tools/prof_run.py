"""Workload driver for rocprofv3: N rife-v4.6 frame pairs, inputs resident in HBM, one stream."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import gen_frames, gen_models

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--pairs", type=int, default=6)
args = ap.parse_args()
w, h = {"4k": (3840, 2160), "1080p": (1920, 1080), "360p": (640, 360)}[args.workload]
amd = importlib.import_module("rife-ncnn-vulkan_amd")
eng = amd.RIFE(0, rife_v4=True)
eng.load(gen_models.ensure(None, "rife-v4.6"))
base = gen_frames.smooth_pair(w // 4, h // 4, 1000)
fr = [torch.from_numpy(np.ascontiguousarray(np.kron(b, np.ones((4, 4, 1), np.uint8)))).cuda() for b in base]
out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
for i in range(args.pairs):
    eng.process_device(fr[0].data_ptr(), fr[1].data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
print("done", args.pairs, "pairs", w, h)
