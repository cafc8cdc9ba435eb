"""rife-v2.3 (IFNet + ContextNet x2 + FusionNet, BASELINE configs 1-2) on the HIP engine vs the CPU oracle."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")


@pytest.fixture(scope="module")
def engines(modeldirs):
    d = modeldirs["rife-v2.3"]
    g = amd.RIFE(0, rife_v2=True)
    g.load(d)
    o = pyoracle.OracleRIFE(rife_v2=True)
    o.set_gpu_crop(1)
    o.load(d)
    return g, o


def report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d == 0).mean())


@pytest.mark.parametrize("w,h,seed", [(64, 64, 1), (160, 96, 2), (100, 60, 3), (640, 360, 4)])
def test_v23_process_within_1_lsb(engines, w, h, seed):
    """(640, 360) is BASELINE config 1's frame size."""
    g, o = engines
    a, b = gen_frames.smooth_pair(w, h, 300 + seed)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97


def test_v23_1080p_within_1_lsb(engines):
    """BASELINE config 2: rife-v2.3, 1920x1080, timestep 0.5."""
    g, o = engines
    a, b = gen_frames.smooth_pair(1920, 1080, 2001)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)


def test_v23_endpoints_and_determinism(engines):
    g, _ = engines
    a, b = gen_frames.smooth_pair(96, 64, 9)
    assert np.array_equal(g.process(a, b, 0.0), a)
    assert np.array_equal(g.process(a, b, 1.0), b)
    assert np.array_equal(g.process(a, b, 0.5), g.process(a, b, 0.5))


def test_v23_rejects_v4_graph(modeldirs):
    g = amd.RIFE(0, rife_v2=True)
    with pytest.raises(amd.RifeError):
        g.load(modeldirs["rife-v4.6"])


@pytest.mark.parametrize("w,h", [(128, 64), (320, 192)])
def test_v23_uhd_mode_within_1_lsb(modeldirs, w, h):
    """-u: flow estimated on half-resolution frames, upsampled x2 and doubled (rife.cpp:294-332, 928-945)."""
    d = modeldirs["rife-v2.3"]
    g = amd.RIFE(0, uhd_mode=True, rife_v2=True); g.load(d)
    o = pyoracle.OracleRIFE(uhd_mode=True, rife_v2=True); o.set_gpu_crop(1); o.load(d)
    a, b = gen_frames.smooth_pair(w, h, 77)
    got, want = g.process(a, b, 0.5), o.process(a, b, 0.5)
    mx, f0 = report(got, want)
    assert mx <= 1, (mx, f0)
    plain = amd.RIFE(0, rife_v2=True); plain.load(d)
    assert not np.array_equal(got, plain.process(a, b, 0.5))     # the mode really changes the computation


def test_v23_uhd_rejects_unsupported_size(modeldirs):
    g = amd.RIFE(0, uhd_mode=True, rife_v2=True); g.load(modeldirs["rife-v2.3"])
    a, b = gen_frames.smooth_pair(96, 64, 1)        # padded 96 -> half 48, not a multiple of 32
    with pytest.raises(amd.RifeError):
        g.process(a, b, 0.5)
