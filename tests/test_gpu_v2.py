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


@pytest.mark.parametrize("w,h,uhd", [(3840, 2160, False), (1920, 1080, True), (2560, 1440, False), (1000, 520, False)])
def test_v23_large_and_odd_grids_within_1_lsb(modeldirs, w, h, uhd):
    """The round-5 kernels of the v2 schedule at the sizes their grids have not seen in the other tests: stem2_fused_kernel on 4K frames (scale 1: 480 x 136 = 65,280
    tiles; 64-bit output offsets), on the UHD half-resolution float4 frames (ImgF4 instantiations, scale 1 and 2), on tile rows / columns that are not whole
    (1440 / 8 = 180, 1000 x 520), the two-tensor ContextNet launches and the batched warps at the same sizes."""
    d = modeldirs["rife-v2.3"]
    g = amd.RIFE(0, uhd_mode=uhd, rife_v2=True); g.load(d)
    o = pyoracle.OracleRIFE(uhd_mode=uhd, rife_v2=True); o.set_gpu_crop(1); o.load(d)
    a, b = gen_frames.smooth_pair(w, h, 31)
    got, want = g.process(a, b, 0.5), o.process(a, b, 0.5)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.999, (w, h, uhd, int(diff.max()), float((diff == 0).mean()))
    assert np.array_equal(g.process(a, b, 0.5), got)                      # deterministic on a used workspace


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


@pytest.mark.parametrize("tta,temporal,uhd,w,h", [(True, False, False, 100, 60), (False, True, False, 160, 96), (True, True, False, 96, 64),
                                                  (True, True, True, 128, 64), (True, False, False, 640, 360)])
def test_v23_tta_within_1_lsb(modeldirs, tta, temporal, uhd, w, h):
    """-x / -z for the v2 family: 8 orientations, forward/backward consensus (rife.cpp:459-877, CPU twin 1256-2138)."""
    d = modeldirs["rife-v2.3"]
    g = amd.RIFE(0, tta_mode=tta, tta_temporal_mode=temporal, uhd_mode=uhd, rife_v2=True); g.load(d)
    o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=temporal, uhd_mode=uhd, rife_v2=True); o.set_gpu_crop(1); o.load(d)
    a, b = gen_frames.smooth_pair(w, h, 500 + w)
    got, want = g.process(a, b, 0.5), o.process(a, b, 0.5)
    mx, f0 = report(got, want)
    assert mx <= 1, (mx, f0)
    assert f0 > 0.97
    plain = amd.RIFE(0, uhd_mode=uhd, rife_v2=True); plain.load(d)
    assert not np.array_equal(got, plain.process(a, b, 0.5))


def test_v23_tta_is_equivariant_under_the_dihedral_group(modeldirs):
    """Size-independent property of the -x ensemble: flipping / transposing the inputs flips / transposes the output
    (frames are multiples of 32, so the zero padding does not break the symmetry)."""
    g = amd.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v2=True); g.load(modeldirs["rife-v2.3"])
    a, b = gen_frames.smooth_pair(192, 128, 31)
    base = g.process(a, b, 0.5)
    for name, f in [("hflip", lambda x: x[:, ::-1]), ("vflip", lambda x: x[::-1]), ("transpose", lambda x: x.transpose(1, 0, 2))]:
        got = g.process(np.ascontiguousarray(f(a)), np.ascontiguousarray(f(b)), 0.5)
        mx, f0 = report(got, f(base))
        assert mx <= 1 and f0 > 0.99, (name, mx, f0)
    # -z: swapping the two frames gives the same middle frame
    mx, f0 = report(g.process(b, a, 0.5), base)
    assert mx <= 1 and f0 > 0.99, (mx, f0)
