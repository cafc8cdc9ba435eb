"""rife-v4 (4.0: PReLU, plain trunk + one residual add, 5-channel deconv heads; models/rife-v4/flownet.param) on the HIP
engine vs the CPU oracle, through the C-ABI.  Same bar as rife-v4.6: <= 1 LSB per channel; stage taps at 1e-3."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")
amd_t = amd.test_build()      # stage taps live in the test build (include/rife_hip_test.h)


@pytest.fixture(scope="module")
def engines(modeldirs):
    d = modeldirs["rife-v4"]
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(d)
    return g, o


@pytest.fixture(scope="module")
def tap_engines(modeldirs):
    d = modeldirs["rife-v4"]
    g = amd_t.RIFE(0, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(d)
    return g, o


def report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d == 0).mean())


@pytest.mark.parametrize("w,h", [(64, 64), (160, 96)])
def test_v40_stage_flows_match_oracle(tap_engines, w, h):
    g, o = tap_engines
    a, b = gen_frames.smooth_pair(w, h, 41)
    for fi in range(4):
        got = g.v4_extract_flow(a, b, 0.5, fi)
        want = o.v4_extract(a, b, 0.5, "flow%d" % fi)
        assert got.shape == want.shape and got.shape[0] == 5
        assert np.abs(got - want).max() < 1e-3, fi


def test_v40_flow_injection_matches_oracle(tap_engines):
    g, o = tap_engines
    a, b = gen_frames.smooth_pair(96, 64, 42)
    rng = np.random.default_rng(2)
    inj = [(rng.standard_normal((5, 64 // s, 96 // s)) * 0.3).astype(np.float32) for s in (16, 8, 4)]
    for fi in (1, 2, 3):
        got = g.v4_extract_flow(a, b, 0.4, fi, inject=inj[:fi])
        want = o.v4_extract(a, b, 0.4, "flow%d" % fi, flows=inj[:fi])
        assert np.abs(got - want).max() < 1e-3, fi


@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1100), (256, 192, 0.125, 1101), (100, 60, 0.7, 1102), (33, 47, 0.9, 1103)])
def test_v40_process_within_1_lsb(engines, w, h, t, seed):
    g, o = engines
    a, b = gen_frames.smooth_pair(w, h, seed)
    mx, f0 = report(g.process(a, b, t), o.process(a, b, t))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97


def test_v40_1080p_within_1_lsb(engines):
    g, o = engines
    a, b = gen_frames.smooth_pair(1920, 1080, 1104)
    mx, f0 = report(g.process(a, b, 0.3), o.process(a, b, 0.3))
    assert mx <= 1, (mx, f0)


@pytest.mark.parametrize("tta,temporal,w,h", [(True, False, 100, 60), (False, True, 160, 96), (True, True, 96, 64)])
def test_v40_tta_within_1_lsb(modeldirs, tta, temporal, w, h):
    d = modeldirs["rife-v4"]
    g = amd.RIFE(0, tta_mode=tta, tta_temporal_mode=temporal, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=temporal, rife_v4=True); o.set_gpu_crop(1); o.load(d)
    a, b = gen_frames.smooth_pair(w, h, 600 + w)
    mx, f0 = report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97


def test_v40_endpoints_and_determinism(engines):
    g, _ = engines
    a, b = gen_frames.smooth_pair(96, 64, 9)
    assert np.array_equal(g.process(a, b, 0.0), a)
    assert np.array_equal(g.process(a, b, 1.0), b)
    assert np.array_equal(g.process(a, b, 0.5), g.process(a, b, 0.5))
