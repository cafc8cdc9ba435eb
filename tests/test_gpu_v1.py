"""v1 family (models/rife and models/rife-HD = rife-UHD = rife-anime; flags rife_v2 = rife_v4 = False like the reference's
dir-name sniffing, src/main.cpp:658-683) on the generic layer-wise graph executor vs the CPU oracle, through the C-ABI."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")


def pair(modeldirs, fam, **kw):
    g = amd.RIFE(0, **kw); g.load(modeldirs[fam])
    o = pyoracle.OracleRIFE(**kw); o.set_gpu_crop(1); o.load(modeldirs[fam])
    return g, o


def report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d == 0).mean())


@pytest.mark.parametrize("fam", ["rife", "rife-HD"])
@pytest.mark.parametrize("w,h,seed", [(64, 64, 1), (160, 96, 2), (100, 60, 3), (640, 360, 4)])
def test_v1_process_within_1_lsb(modeldirs, fam, w, h, seed):
    g, o = pair(modeldirs, fam)
    a, b = gen_frames.smooth_pair(w, h, 900 + seed)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97


def test_v1_1080p_within_1_lsb(modeldirs):
    g, o = pair(modeldirs, "rife-HD")
    a, b = gen_frames.smooth_pair(1920, 1080, 905)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)


@pytest.mark.parametrize("fam", ["rife", "rife-HD"])
@pytest.mark.parametrize("kw,w,h", [(dict(uhd_mode=True), 128, 64), (dict(tta_mode=True), 100, 60), (dict(tta_temporal_mode=True), 160, 96),
                                    (dict(tta_mode=True, tta_temporal_mode=True, uhd_mode=True), 128, 64)])
def test_v1_modes_within_1_lsb(modeldirs, fam, kw, w, h):
    g, o = pair(modeldirs, fam, **kw)
    a, b = gen_frames.smooth_pair(w, h, 950 + w)
    got, want = g.process(a, b, 0.5), o.process(a, b, 0.5)
    mx, f0 = report(got, want)
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97
    plain = amd.RIFE(0); plain.load(modeldirs[fam])
    assert not np.array_equal(got, plain.process(a, b, 0.5))


def test_v1_endpoints_determinism_and_size_change(modeldirs):
    g, _ = pair(modeldirs, "rife")
    a, b = gen_frames.smooth_pair(96, 64, 9)
    assert np.array_equal(g.process(a, b, 0.0), a)
    assert np.array_equal(g.process(a, b, 1.0), b)
    first = g.process(a, b, 0.5)
    a2, b2 = gen_frames.smooth_pair(160, 128, 10)          # the blob storage is re-shaped between calls
    g.process(a2, b2, 0.5)
    assert np.array_equal(g.process(a, b, 0.5), first)


def test_v1_tta_symmetries(modeldirs):
    g = amd.RIFE(0, tta_mode=True, tta_temporal_mode=True); g.load(modeldirs["rife"])
    a, b = gen_frames.smooth_pair(128, 96, 33)
    base = g.process(a, b, 0.5)
    for f in (lambda x: x[:, ::-1], lambda x: x[::-1], lambda x: x.transpose(1, 0, 2)):
        got = g.process(np.ascontiguousarray(f(a)), np.ascontiguousarray(f(b)), 0.5)
        mx, f0 = report(got, f(base))
        assert mx <= 1 and f0 > 0.99, (mx, f0)
    mx, f0 = report(g.process(b, a, 0.5), base)
    assert mx <= 1 and f0 > 0.99, (mx, f0)
