"""rife-v3.x (3-block 160-channel residual IFNet + the v2 ContextNet / FusionNet; models/rife-v3.1) on the HIP engine vs the
CPU oracle, through the C-ABI.  The caller passes rife_v2=True for this family (src/main.cpp:658-683: "rife-v3" in the name)."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")


def pair(modeldirs, **kw):
    d = modeldirs["rife-v3.1"]
    g = amd.RIFE(0, rife_v2=True, **kw); g.load(d)
    o = pyoracle.OracleRIFE(rife_v2=True, **kw); o.set_gpu_crop(1); o.load(d)
    return g, o


def report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d == 0).mean())


@pytest.mark.parametrize("w,h,seed", [(64, 64, 1), (160, 96, 2), (100, 60, 3), (640, 360, 4), (1920, 1080, 5)])
def test_v3_process_within_1_lsb(modeldirs, w, h, seed):
    g, o = pair(modeldirs)
    a, b = gen_frames.smooth_pair(w, h, 700 + seed)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97


def test_v3_differs_from_v23(modeldirs):
    """Same weights file names, different flownet: make sure the v3 schedule is the one that ran."""
    g, _ = pair(modeldirs)
    g2 = amd.RIFE(0, rife_v2=True); g2.load(modeldirs["rife-v2.3"])
    a, b = gen_frames.smooth_pair(160, 96, 2)
    assert not np.array_equal(g.process(a, b, 0.5), g2.process(a, b, 0.5))


@pytest.mark.parametrize("kw,w,h", [(dict(uhd_mode=True), 128, 64), (dict(tta_mode=True), 100, 60), (dict(tta_temporal_mode=True), 160, 96),
                                    (dict(tta_mode=True, tta_temporal_mode=True, uhd_mode=True), 128, 64)])
def test_v3_modes_within_1_lsb(modeldirs, kw, w, h):
    g, o = pair(modeldirs, **kw)
    a, b = gen_frames.smooth_pair(w, h, 800 + w)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97
