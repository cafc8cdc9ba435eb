"""The persistent S16 trunk kernel of the finest IFBlock (csrc/conv_t64.h) against the per-tile kernel it replaces.

Both compute the same products in the same order (reference layers: models/rife-v4.6/flownet.param:166-201), so the two
engines must agree BIT FOR BIT on every output byte wherever the per-tile path does not split K (3840x2160), and
within summation-order noise below that, at aligned, ragged and tiny frame sizes and through the TTA schedule (whose passes
borrow scratch tensors but own their S16 tensors).  Parity against the CPU oracle
is covered by tests/test_gpu_v4.py, which runs on the new path (the default)."""
import importlib
import os

import numpy as np
import pytest

from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


def _engine(modeldir, t64, **kw):
    keys = ("RIFE_HIP_T64", "RIFE_HIP_RS", "RIFE_HIP_STEM_RS", "RIFE_HIP_TAIL_RS")
    old = tuple(os.environ.get(k) for k in keys)
    os.environ["RIFE_HIP_T64"] = "1" if t64 else "0"      # read by rife_hip_create
    os.environ["RIFE_HIP_RS"] = "0"                        # conv_t64 itself, not the row-streaming kernel that serves block 3 by default (tests below)
    os.environ["RIFE_HIP_STEM_RS"] = "0"                   # nor the row-streaming stem / tail kernels around it (last-bit differences of their own:
    os.environ["RIFE_HIP_TAIL_RS"] = "0"                   #  tests/test_gpu_stem_rs.py, test_gpu_tail_rs.py)
    try:
        g = amd.RIFE(0, rife_v4=True, **kw)
    finally:
        for k, v in zip(keys, old):
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    g.load(modeldir)
    return g


@pytest.fixture(scope="module")
def pair(modeldirs):
    d = modeldirs["rife-v4.6"]
    return _engine(d, True), _engine(d, False)


@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1), (256, 192, 0.125, 2), (100, 60, 0.7, 3), (33, 47, 0.9, 4), (1, 1, 0.5, 5),
                                        (1920, 1080, 0.5, 6), (1000, 520, 0.3, 7), (3840, 2160, 0.5, 8)])
def test_t64_output_is_bit_identical_to_the_per_tile_trunk(pair, w, h, t, seed):
    new, old = pair
    a, b = gen_frames.smooth_pair(w, h, seed) if w * h < 4000000 else gen_frames.smooth_pair_native(w, h, seed)
    def check(x, y, what):
        d = np.abs(x.astype(np.int32) - y.astype(np.int32))
        report = "%s: %d of %d bytes differ, max %d" % (what, int((d > 0).sum()), d.size, int(d.max()))
        if w * h >= 3840 * 2160:
            assert d.max() == 0, report
        else:   # below 4K the per-tile path splits K over several workgroups for the tiny grids of its coarse blocks (another summation order)
            assert d.max() <= 1 and (d > 0).mean() < 1e-3, report
    check(new.process(a, b, t), old.process(a, b, t), "first call")
    # a second call on the same workspace: the zero borders of the S16 tensors must have survived the first
    check(new.process(b, a, 1.0 - t), old.process(b, a, 1.0 - t), "second call")
    x = new.process(a, b, t)
    assert np.array_equal(x, new.process(a, b, t)), "the S16 path is not deterministic"


@pytest.mark.parametrize("w,h", [(160, 96), (100, 60)])
def test_t64_block3_flow_matches(pair, w, h):
    new, old = pair
    a, b = gen_frames.noise_pair(w, h, 9)
    assert np.abs(new.v4_extract_flow(a, b, 0.5, 3) - old.v4_extract_flow(a, b, 0.5, 3)).max() < 1e-4      # split-K in the per-tile path at this size


def test_t64_tta_passes_match(modeldirs):
    d = modeldirs["rife-v4.6"]
    new, old = _engine(d, True, tta_mode=True, tta_temporal_mode=True), _engine(d, False, tta_mode=True, tta_temporal_mode=True)
    a, b = gen_frames.smooth_pair(100, 60, 11)
    d = np.abs(new.process(a, b, 0.4).astype(np.int32) - old.process(a, b, 0.4).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


# ---- round 3: the row-streaming kernel (csrc/conv_rs.h) that serves the block-3 trunk by default, against conv_t64 (RIFE_HIP_RS=0) ----
def _engine_rs(modeldir, rs, **kw):
    old = os.environ.get("RIFE_HIP_RS")
    os.environ["RIFE_HIP_RS"] = "1" if rs else "0"      # read by rife_hip_create
    try:
        g = amd.RIFE(0, rife_v4=True, **kw)
    finally:
        if old is None:
            del os.environ["RIFE_HIP_RS"]
        else:
            os.environ["RIFE_HIP_RS"] = old
    g.load(modeldir)
    return g


@pytest.fixture(scope="module")
def pair_rs(modeldirs):
    d = modeldirs["rife-v4.6"]
    return _engine_rs(d, True), _engine_rs(d, False)


@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1), (256, 192, 0.125, 2), (100, 60, 0.7, 3), (33, 47, 0.9, 4), (1, 1, 0.5, 5), (130, 9, 0.5, 12),
                                        (1920, 1080, 0.5, 6), (1000, 520, 0.3, 7), (3840, 2160, 0.5, 8)])
def test_rs_output_matches_conv_t64(pair_rs, w, h, t, seed):
    """Same products, same epilogue; conv_rs sums the hi and the lo products of a layer in two chains (conv_rs.h header), conv_t64 in one:
    the two engines agree to summation-order noise - frames within 1 LSB with very few channels touched, block-3 flows to 1e-4 - at
    aligned, ragged and tiny sizes, on a used workspace, and conv_rs is deterministic."""
    new, old = pair_rs
    a, b = gen_frames.smooth_pair(w, h, seed) if w * h < 4000000 else gen_frames.smooth_pair_native(w, h, seed)
    for x, y, tt in ((a, b, t), (b, a, 1.0 - t)):          # the second call runs on a used workspace: zero borders of the S16 tensors intact
        d = np.abs(new.process(x, y, tt).astype(np.int32) - old.process(x, y, tt).astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, "%dx%d: %d of %d bytes differ, max %d" % (w, h, int((d > 0).sum()), d.size, int(d.max()))
    if w * h <= 1920 * 1080:
        fd = np.abs(new.v4_extract_flow(a, b, t, 3) - old.v4_extract_flow(a, b, t, 3)).max()
        assert fd < 1e-4, "block-3 flows differ by %g at %dx%d" % (fd, w, h)
    x = new.process(a, b, t)
    for _ in range(3):
        assert np.array_equal(x, new.process(a, b, t)), "the row-streaming kernel is not deterministic"


def test_rs_tta_passes_match(modeldirs):
    d = modeldirs["rife-v4.6"]
    new, old = _engine_rs(d, True, tta_mode=True, tta_temporal_mode=True), _engine_rs(d, False, tta_mode=True, tta_temporal_mode=True)
    a, b = gen_frames.smooth_pair(100, 60, 11)
    d = np.abs(new.process(a, b, 0.4).astype(np.int32) - old.process(a, b, 0.4).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


# ---- round 6: two trunk layers per launch (csrc/conv_rs2.h: layer A's rows stay in LDS) against one conv_rs launch per layer (RIFE_HIP_RS2=0) ----
def _engine_rs2(modeldir, rs2, **kw):
    old = os.environ.get("RIFE_HIP_RS2")
    os.environ["RIFE_HIP_RS2"] = "1" if rs2 else "0"      # read by rife_hip_create
    try:
        g = amd.RIFE(0, rife_v4=True, **kw)
    finally:
        if old is None:
            del os.environ["RIFE_HIP_RS2"]
        else:
            os.environ["RIFE_HIP_RS2"] = old
    g.load(modeldir)
    return g


@pytest.fixture(scope="module")
def pair_rs2(modeldirs):
    d = modeldirs["rife-v4.6"]
    return _engine_rs2(d, True), _engine_rs2(d, False)


@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1), (256, 192, 0.125, 2), (100, 60, 0.7, 3), (33, 47, 0.9, 4), (1, 1, 0.5, 5), (130, 9, 0.5, 12), (117, 250, 0.5, 13),
                                        (1920, 1080, 0.5, 6), (1000, 520, 0.3, 7), (3840, 2160, 0.5, 8)])
def test_rs2_output_is_bit_identical_to_one_launch_per_layer(pair_rs2, w, h, t, seed):
    """Per layer the products, their order and the epilogue are conv_rs_kernel's, and the LDS ring between the two layers holds the {hi, lo} pairs conv_rs
    would have stored: frames and block-3 flows are the same BYTES, at aligned, ragged and tiny sizes, on a used workspace (zero borders intact), run to run."""
    new, old = pair_rs2
    a, b = gen_frames.smooth_pair(w, h, seed) if w * h < 4000000 else gen_frames.smooth_pair_native(w, h, seed)
    for x, y, tt in ((a, b, t), (b, a, 1.0 - t)):
        got, want = new.process(x, y, tt), old.process(x, y, tt)
        assert np.array_equal(got, want), "%dx%d: %d of %d bytes differ" % (w, h, int((got != want).sum()), got.size)
    if w * h <= 1920 * 1080:
        assert np.array_equal(new.v4_extract_flow(a, b, t, 3), old.v4_extract_flow(a, b, t, 3)), "block-3 flows differ at %dx%d" % (w, h)
    x = new.process(a, b, t)
    for _ in range(3):
        assert np.array_equal(x, new.process(a, b, t)), "the depth-fused kernel is not deterministic"


def test_rs2_tta_passes_are_bit_identical(modeldirs):
    d = modeldirs["rife-v4.6"]
    new, old = _engine_rs2(d, True, tta_mode=True, tta_temporal_mode=True), _engine_rs2(d, False, tta_mode=True, tta_temporal_mode=True)
    a, b = gen_frames.smooth_pair(200, 120, 11)
    assert np.array_equal(new.process(a, b, 0.4), old.process(a, b, 0.4))
