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

CASES = [  # name, family, w, h, t, seed, tta, temporal, uhd
    ("v46_plain_96x64", "rife-v4.6", 96, 64, 0.5, 101, False, False, False),
    ("v46_plain_100x60_ragged", "rife-v4.6", 100, 60, 0.7, 102, False, False, False),
    ("v46_tta_64x32", "rife-v4.6", 64, 32, 0.3, 103, True, False, False),
    ("v46_temporal_64x64", "rife-v4.6", 64, 64, 0.25, 104, False, True, False),
    ("v40_plain_100x60_ragged", "rife-v4", 100, 60, 0.4, 105, False, False, False),
    ("v23_plain_96x64", "rife-v2.3", 96, 64, 0.5, 106, False, False, False),
    ("v23_tta_temporal_uhd_64x64", "rife-v2.3", 64, 64, 0.5, 107, True, True, True),
    ("v31_plain_100x60_ragged", "rife-v3.1", 100, 60, 0.5, 108, False, False, False),
    ("v1_plain_100x60_ragged", "rife", 100, 60, 0.5, 109, False, False, False),
    ("hd_tta_64x64", "rife-HD", 64, 64, 0.5, 110, True, False, False),
]


def flags(family):
    return dict(rife_v2=family.startswith(("rife-v2", "rife-v3")), rife_v4=family.startswith("rife-v4"))


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only_new = "--only-new" in sys.argv
    for name, family, w, h, t, seed, tta, temporal, uhd in CASES:
        path = os.path.join(out, name + ".npz")
        if only_new and os.path.exists(path):
            continue
        d = gen_models.ensure(None, family)
        o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=temporal, uhd_mode=uhd, **flags(family))
        o.set_gpu_crop(1)
        o.load(d)
        a, b = gen_frames.smooth_pair(w, h, seed)
        res = dict(in0=a, in1=b, timestep=np.float32(t), tta=tta, temporal=temporal, uhd=uhd, family=family, out=o.process(a, b, t), weights_seed=0x51FE)
        if family == "rife-v4.6" and not tta and not temporal:
            for k in range(4):
                res["flow%d" % k] = o.v4_extract(a, b, t, "flow%d" % k).astype(np.float16)   # compact; compared at 1e-2
        np.savez_compressed(path, **res)
        print(name, res["out"].shape)


if __name__ == "__main__":
    main()
