"""Extreme frame sizes (1x1, one side below the 32-pixel pad unit, long thin strips) through every model family."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")
FAMILIES = ["rife-v4.6", "rife-v4", "rife-v2.3", "rife-v3.1", "rife", "rife-HD"]


@pytest.mark.parametrize("fam", FAMILIES)
def test_extreme_sizes_within_1_lsb(modeldirs, fam):
    kw = dict(rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    g = amd.RIFE(0, **kw); g.load(modeldirs[fam])
    o = pyoracle.OracleRIFE(**kw); o.set_gpu_crop(1); o.load(modeldirs[fam])
    for (w, h) in ((1, 1), (31, 33), (8, 300), (520, 16), (33, 32)):
        a, b = gen_frames.smooth_pair(w, h, 77)
        d = np.abs(g.process(a, b, 0.5).astype(int) - o.process(a, b, 0.5).astype(int))
        assert d.max() <= 1, (fam, w, h, int(d.max()))
