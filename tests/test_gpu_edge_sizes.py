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


@pytest.mark.parametrize("fam,kw", [("rife-v2.3", dict(tta_temporal_mode=True)), ("rife-v3.1", {}), ("rife", {}), ("rife-HD", dict(uhd_mode=True)), ("rife-v4", {}),
                                    ("rife-v4.6", dict(tta_mode=True))])
def test_process_is_reentrant_for_every_family(modeldirs, fam, kw):
    """Two proc threads share one RIFE in the reference (main.cpp:860-863): concurrent process() calls, also with different frame
    sizes in flight, give the same pixels as serial calls."""
    import threading
    fl = dict(kw, rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    g = amd.RIFE(0, **fl); g.load(modeldirs[fam])
    jobs = [gen_frames.smooth_pair(w, h, 60 + i) for i, (w, h) in enumerate([(192, 128), (128, 64), (192, 128), (256, 128)])]
    want = [g.process(a, b, 0.5) for a, b in jobs]
    got = [None] * len(jobs)

    def work(i):
        got[i] = g.process(jobs[i][0], jobs[i][1], 0.5)
    for _ in range(2):
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        [t.start() for t in th]; [t.join() for t in th]
        for i in range(len(jobs)):
            assert np.array_equal(got[i], want[i]), (fam, i)
