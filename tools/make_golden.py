"""Generate the committed golden fixtures under tests/golden/ from the CPU oracle (after it has been pinned
against the PyTorch executor by tests/test_oracle_vs_torch.py).  Small seeded cases only:

    python tools/make_golden.py

Each .npz holds the two input frames, timestep, flags, the oracle's u8 output and (v4) the four flow blobs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyoracle
from tools import gen_frames, gen_models

CASES = [  # name, w, h, t, seed, tta, temporal
    ("v46_plain_96x64", 96, 64, 0.5, 101, False, False),
    ("v46_plain_100x60_ragged", 100, 60, 0.7, 102, False, False),
    ("v46_tta_64x32", 64, 32, 0.3, 103, True, False),
    ("v46_temporal_64x64", 64, 64, 0.25, 104, False, True),
]


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    d = gen_models.ensure(None, "rife-v4.6")
    for name, w, h, t, seed, tta, temporal in CASES:
        o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=temporal, rife_v4=True)
        o.set_gpu_crop(1)
        o.load(d)
        a, b = gen_frames.smooth_pair(w, h, seed)
        res = dict(in0=a, in1=b, timestep=np.float32(t), tta=tta, temporal=temporal, out=o.process(a, b, t), weights_seed=0x51FE)
        if not tta and not temporal:
            for k in range(4):
                res["flow%d" % k] = o.v4_extract(a, b, t, "flow%d" % k).astype(np.float16)   # compact; compared at 1e-2
        np.savez_compressed(os.path.join(out, name + ".npz"), **res)
        print(name, res["out"].shape)


if __name__ == "__main__":
    main()
