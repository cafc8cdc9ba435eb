"""The gather code of the hot path against the oracle DIRECTLY, on injected flows that leave the frame by hundreds of pixels.

rife.Warp (src/warp.cpp:96-168, src/warp.comp:24-69) is fused into three places of the rife-v4.6 schedule: the block-input assembly inside
the fused stem kernels of blocks 1-3 (stem_fused.h; flownet.param:52-62, 107-115, 160-165) and the tail of the graph inside the last head kernel
(head_h2.h EPI_FINAL; flownet.param:202-217).  (Since round 3 the product runs block 3's stems on stem_rs_kernel and, from 4K-class frames up,
the tail on tail_rs_kernel - the same gather code in two halves; tests/test_gpu_stem_rs.py and test_gpu_tail_rs.py repeat these checks on them.
The tile kernels tested here still serve blocks 1 / 2, smaller frames, and every frame when the row-streaming kernels are switched off.)  End-to-end parity (tests/test_gpu_v4.py) sees that code only behind ~50 further layers and with the
small flows a synthetic model produces.  Here the blobs flow0..flow{b-1} are INJECTED on both sides (the reference's Extractor does the same for
its TTA passes, src/rife.cpp:2653-2669), so that the sampling positions are far outside the frame in places, and three things are compared:
  1. the 12-channel block input from the unfused assembly kernel k_assemble<S> (same assemble_pixel / warp_rgbx code): BIT FOR BIT;
  2. the same tensor read back THROUGH the product's fused stem kernel, run with one-hot weights: the split-f16 matrix path returns hi + lo of
     every value = the value to 2^-22 relative, so a wrong tap, clamp or swizzle is off by orders of magnitude more than the tolerance;
  3. the tail: blob out0 before quantisation from the unfused float tail (expf aside: 2e-6), and the u8 frame of the plain pass with three
     injected flows - blocks 3's fused stem and the fused tail running on them - within 1 LSB of the oracle with > 99.9 % of the bytes equal.
"""
import importlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


@pytest.fixture(scope="module")
def engines(modeldirs):
    d = modeldirs["rife-v4.6"]
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True); o.load(d)
    # names of the block-input blobs = tops of the three two-input Concat layers (cat_4 / cat_8 / cat_12 of the reference's flownet.param:62, 115, 165)
    names = []
    for line in open(os.path.join(d, "flownet.param")):
        f = line.split()
        if len(f) > 6 and f[0] == "Concat" and f[2] == "2" and f[3] == "1":
            names.append(f[6])
    assert len(names) == 3
    return g, o, names


@pytest.fixture(scope="module")
def fused_engine(modeldirs):
    old = os.environ.get("RIFE_HIP_FUSE_FLOW")
    os.environ["RIFE_HIP_FUSE_FLOW"] = "1"                              # read at create time
    try:
        g = amd.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    finally:
        if old is None: del os.environ["RIFE_HIP_FUSE_FLOW"]
        else: os.environ["RIFE_HIP_FUSE_FLOW"] = old
    return g


def injected_flows(w, h, seed, n):
    """blobs flow0..flow{n-1} (6 x hp/s x wp/s, s = 8, 4, 2, 1): smooth fields + noise; after the x s of the flow update the coarse one moves
    samples by up to ~300 px, the finer ones add tens of pixels; channel 4 = mask logit increments of a few units."""
    wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32
    rng = np.random.default_rng(seed)
    out = []
    for k, s in enumerate((8, 4, 2, 1)[:n]):
        H, W = hp // s, wp // s
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        f = np.empty((6, H, W), np.float32)
        amp = (40.0, 6.0, 3.0, 1.5)[k]
        for c in range(4):
            ph = rng.uniform(0, 6.28, 2)
            f[c] = amp * np.sin(xx * (rng.uniform(0.5, 3.0) * 6.28 / W) + ph[0]) * np.cos(yy * (rng.uniform(0.5, 3.0) * 6.28 / H) + ph[1]) + rng.normal(0, 0.3 * amp / 8, (H, W))
        f[4] = rng.normal(0, 1.5, (H, W)); f[5] = rng.normal(0, 1.0, (H, W))
        out.append(f.astype(np.float32))
    return out


SIZES = [(100, 60, 1), (256, 192, 2), (640, 360, 3), (333, 241, 4)]


@pytest.mark.parametrize("w,h,seed", SIZES)
@pytest.mark.parametrize("b", [1, 2, 3])
def test_block_input_bit_exact_and_through_the_fused_stem(engines, w, h, seed, b):
    g, o, names = engines
    a, c = gen_frames.noise_pair(w, h, seed) if seed % 2 else gen_frames.smooth_pair(w, h, seed)
    inj = injected_flows(w, h, 100 + seed, b)
    t = 0.3 + 0.1 * b
    want = o.v4_extract(a, c, t, names[b - 1], flows=inj)
    assert want.shape[0] == 12
    # the flows really leave the frame (channels 8..11 = F / S)
    S = (4, 2, 1)[b - 1]
    assert np.abs(want[8:12]).max() * S > 100, "injected flows too small to exercise the clamps"
    got0 = g.v4_tap(a, c, t, 0, b, inj)
    assert got0.shape == want.shape
    assert np.array_equal(got0, want), "k_assemble<%d>: %d of %d floats differ, max %g" % (S, int((got0 != want).sum()), want.size, float(np.abs(got0 - want).max()))
    got1 = g.v4_tap(a, c, t, 1, b, inj)
    err = np.abs(got1 - want) - (3e-7 * np.abs(want) + 1.2e-7)       # hi + lo of a value: 2^-22 relative, f16 subnormal floor
    assert err.max() <= 0, "fused stem kernel of block %d: worst excess %g at %s" % (b, float(err.max()), np.unravel_index(np.argmax(err), err.shape))


@pytest.mark.parametrize("w,h,seed", SIZES)
@pytest.mark.parametrize("b", [2, 3])
def test_flow_update_inside_the_stem_writes_the_same_F_M(engines, fused_engine, w, h, seed, b):
    """Blocks 2 and 3, opt-in schedule RIFE_HIP_FUSE_FLOW=1 (slower than the update kernels on MI355X, kept for A/B): the stem kernel applies
    the flow update of the block before it while it gathers (stem_fused.h UPD, flownet.param:99-105, 152-158) and writes F, M for the later
    stages.  The block input it computes from them against the oracle's blob, and the tensors it WRITES against the flow-update kernel's, bit
    for bit - every full-resolution pixel, borders included."""
    _, o, names = engines
    g = fused_engine
    a, c = gen_frames.noise_pair(w, h, seed) if seed % 2 else gen_frames.smooth_pair(w, h, seed)
    inj = injected_flows(w, h, 300 + seed, b)
    want = g.v4_tap(a, c, 0.5, 4, b, inj)
    got = g.v4_tap(a, c, 0.5, 3, b, inj)
    assert np.array_equal(got, want), "%d of %d floats differ" % (int((got != want).sum()), want.size)
    blob_in = o.v4_extract(a, c, 0.5, names[b - 1], flows=inj)
    err = np.abs(g.v4_tap(a, c, 0.5, 1, b, inj) - blob_in) - (3e-7 * np.abs(blob_in) + 1.2e-7)      # through the UPD kernel's matrix path: hi + lo of every value
    assert err.max() <= 0, float(err.max())
    if b == 3:      # scale 1: channels 7..11 of the oracle's block input ARE M and F
        blob = o.v4_extract(a, c, 0.5, names[2], flows=inj)
        assert np.array_equal(got[4], blob[7]) and np.array_equal(got[:4], blob[8:12])


def test_block3_input_at_4k(engines):
    """The product's block-3 stem (three workgroups per CU, swizzled 64-byte records) at the north-star size, F1 frames tiled 6 x 6."""
    g, o, names = engines
    a, c = gen_frames.tiled_real_pair(6)
    inj = injected_flows(3840, 2160, 77, 3)
    want = o.v4_extract(a, c, 0.5, names[2], flows=inj)
    assert np.array_equal(g.v4_tap(a, c, 0.5, 0, 3, inj), want)
    got1 = g.v4_tap(a, c, 0.5, 1, 3, inj)
    err = np.abs(got1 - want) - (3e-7 * np.abs(want) + 1.2e-7)
    assert err.max() <= 0, float(err.max())


@pytest.mark.parametrize("w,h,seed", SIZES)
def test_tail_on_injected_flows(engines, w, h, seed):
    g, o, _ = engines
    a, c = gen_frames.smooth_pair(w, h, 40 + seed)
    inj = injected_flows(w, h, 200 + seed, 4)
    want = o.v4_extract(a, c, 0.45, "out0", flows=inj)                  # 3 x hp x wp, before the postproc
    got = g.v4_tap(a, c, 0.45, 2, 0, inj)
    d = np.abs(got - want)
    assert d.max() < 2e-6, "unfused tail: max %g (expf is the only operation that may differ by an ulp)" % float(d.max())
    # the plain pass on three injected flows: block 3 (fused stem on F far outside the frame) and the fused tail run as in process()
    wantf = o.v4_extract(a, c, 0.45, "out0", flows=inj[:3])[:, :h, :w]
    want8 = np.clip((wantf * 255.0 + 0.5).astype(np.int32), 0, 255).transpose(1, 2, 0)      # rife_postproc.comp:39-62 / rife.cpp:4373-4387
    got8 = g.v4_process_injected(a, c, 0.45, inj[:3]).astype(np.int32)
    dd = np.abs(got8 - want8)
    assert dd.max() <= 1 and (dd > 0).mean() < 1e-3, "%d of %d bytes differ, max %d" % (int((dd > 0).sum()), dd.size, int(dd.max()))


@pytest.mark.parametrize("w,h", [(256, 192), (640, 360), (1920, 1080)])
def test_pass_with_fused_flow_updates_is_bit_identical_to_three_update_launches(modeldirs, w, h, monkeypatch):
    """The plain pass with the updates after blocks 1 and 2 inside the stems of blocks 2 and 3 (RIFE_HIP_FUSE_FLOW=1) against the product's
    schedule (k_flow_update after every block): same arithmetic on the same values, so the frames must be the same bytes."""
    a, c = gen_frames.smooth_pair(w, h, 11) if w < 1000 else gen_frames.tiled_real_pair(3)
    monkeypatch.setenv("RIFE_HIP_STEM_RS", "0")      # both on the tile stems: stem_rs_kernel (no UPD form) differs from them in the last bits
    monkeypatch.delenv("RIFE_HIP_FUSE_FLOW", raising=False)
    g0 = amd.RIFE(0, rife_v4=True); g0.load(modeldirs["rife-v4.6"])
    monkeypatch.setenv("RIFE_HIP_FUSE_FLOW", "1")
    g1 = amd.RIFE(0, rife_v4=True); g1.load(modeldirs["rife-v4.6"])
    for t in (0.5, 0.2):
        x0, x1 = g0.process(a, c, t), g1.process(a, c, t)
        assert np.array_equal(x0, x1), "%d bytes differ" % int((x0 != x1).sum())
    assert np.array_equal(g1.process(a, c, 0.5), g1.process(a, c, 0.5))      # F / F2 swap back: the second call starts from the same buffers
