"""Workload driver for rocprofv3: N frame pairs of a bench.py workload, inputs resident in HBM (F1: the reference's real frame pair tiled to
size, as in bench.py), one stream."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import gen_frames, gen_models

WL = {"4k": ("rife-v4.6", 3840, 2160, {}), "1080p": ("rife-v4.6", 1920, 1080, {}), "360p": ("rife-v4.6", 640, 360, {}),
      "v23-1080p": ("rife-v2.3", 1920, 1080, {}), "4k-tta": ("rife-v4.6", 3840, 2160, {"tta_mode": True, "tta_temporal_mode": True})}
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k", choices=list(WL))
ap.add_argument("--pairs", type=int, default=6)
args = ap.parse_args()
fam, w, h, kw = WL[args.workload]
amd = importlib.import_module("rife-ncnn-vulkan_amd")
eng = amd.RIFE(0, rife_v2=fam.startswith("rife-v2"), rife_v4=fam.startswith("rife-v4"), **kw)
eng.load(gen_models.ensure(None, fam))
fr = [torch.from_numpy(f).cuda() for f in gen_frames.tiled_real_pair(w // 640)]
out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
for i in range(args.pairs):
    eng.process_device(fr[0].data_ptr(), fr[1].data_ptr(), w, h, 0.5, out.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
print("done", args.pairs, "pairs", fam, w, h, kw)
