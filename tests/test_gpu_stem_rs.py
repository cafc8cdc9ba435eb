"""stem_rs_kernel (csrc/stem_rs.h): block 3's input assembly + both stride-2 stem convolutions in one row-streaming kernel, against the
two-kernel sequence it replaces (stem0_fused_kernel<1, 1, 256> + conv_h2s2_kernel<2, true>; RIFE_HIP_STEM_RS=0) and against the oracle.

Same gather code (assemble_pixel's arithmetic in two halves), same products; stem 1 adds two K partial sums instead of running one chain over
both 16-channel chunks, so the two engines agree to summation-order noise: block-3 flows to 1e-4 (values of O(1) px), frames within 1 LSB with
very few channels touched.  Flows that leave the frame by hundreds of pixels are INJECTED for blocks 0..2 (the reference's Extractor does the
same for its TTA passes, src/rife.cpp:2653-2669) so that the clamps of the gather, the zero padding of both convolutions and the strip / range
boundaries of the kernel all see real data.  models/rife-v4.6/flownet.param:160-168."""
import importlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames
from test_gpu_gather import injected_flows

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


def _engine(d, on, **kw):
    old = os.environ.get("RIFE_HIP_STEM_RS")
    os.environ["RIFE_HIP_STEM_RS"] = "1" if on else "0"                 # read at create time
    try:
        g = amd.RIFE(0, rife_v4=True, **kw); g.load(d)
    finally:
        if old is None: del os.environ["RIFE_HIP_STEM_RS"]
        else: os.environ["RIFE_HIP_STEM_RS"] = old
    return g


@pytest.fixture(scope="module")
def pair(modeldirs):
    d = modeldirs["rife-v4.6"]
    return _engine(d, True), _engine(d, False)


SIZES = [(256, 192, 1), (640, 360, 2), (333, 241, 3), (100, 60, 4), (33, 47, 5), (1, 1, 6), (130, 9, 7), (1000, 520, 8), (1920, 1080, 9)]


@pytest.mark.parametrize("w,h,seed", SIZES)
def test_block3_flow_matches_the_two_kernel_sequence(pair, w, h, seed):
    new, old = pair
    a, c = gen_frames.noise_pair(w, h, seed) if seed % 2 else gen_frames.smooth_pair(w, h, seed)
    inj = injected_flows(w, h, 400 + seed, 3)
    f1, f0 = new.v4_extract_flow(a, c, 0.4, 3, inj), old.v4_extract_flow(a, c, 0.4, 3, inj)
    d = np.abs(f1 - f0)
    assert d.max() < 1e-4 * max(1.0, float(np.abs(f0).max())), "flow3 differs by %g (|flow| up to %g) at %s" % (float(d.max()), float(np.abs(f0).max()), np.unravel_index(np.argmax(d), d.shape))
    # and on the flows the model itself produces
    d2 = np.abs(new.v4_extract_flow(a, c, 0.6, 3) - old.v4_extract_flow(a, c, 0.6, 3))
    assert d2.max() < 1e-4, float(d2.max())


@pytest.mark.parametrize("w,h,seed", SIZES + [(3840, 2160, 10)])
def test_frames_match_and_are_deterministic(pair, w, h, seed):
    new, old = pair
    if w * h > 4000000: a, c = gen_frames.tiled_real_pair(6)
    else: a, c = gen_frames.smooth_pair(w, h, 20 + seed)
    for x, y, t in ((a, c, 0.5), (c, a, 0.3)):                          # the second call runs on a used workspace
        p1, p0 = new.process(x, y, t), old.process(x, y, t)
        d = np.abs(p1.astype(np.int32) - p0.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, "%dx%d: %d of %d bytes differ, max %d" % (w, h, int((d > 0).sum()), d.size, int(d.max()))
    x = new.process(a, c, 0.5)
    for _ in range(3):
        assert np.array_equal(x, new.process(a, c, 0.5)), "stem_rs_kernel is not run-to-run identical"


@pytest.mark.parametrize("w,h,seed", [(640, 360, 31), (333, 241, 32), (1920, 1080, 33)])
def test_against_the_oracle_on_injected_flows(pair, modeldirs, w, h, seed):
    """The plain pass on three injected flows far outside the frame: stem_rs's gather and the fused tail, within 1 LSB of the oracle."""
    new, _ = pair
    o = pyoracle.OracleRIFE(rife_v4=True); o.load(modeldirs["rife-v4.6"])
    a, c = gen_frames.smooth_pair(w, h, seed) if w < 1000 else gen_frames.tiled_real_pair(3)
    inj = injected_flows(w, h, 500 + seed, 3)
    wantf = o.v4_extract(a, c, 0.45, "out0", flows=inj)[:, :h, :w]
    want8 = np.clip((wantf * 255.0 + 0.5).astype(np.int32), 0, 255).transpose(1, 2, 0)
    got8 = new.v4_process_injected(a, c, 0.45, inj).astype(np.int32)
    dd = np.abs(got8 - want8)
    assert dd.max() <= 1 and (dd > 0).mean() < 1e-3, "%d of %d bytes differ, max %d" % (int((dd > 0).sum()), dd.size, int(dd.max()))


@pytest.mark.parametrize("w,h,seed", [(256, 192, 41), (640, 360, 42), (333, 241, 43), (100, 60, 44), (1920, 1080, 45)])
def test_block_input_through_stem_rs_is_the_oracles(pair, modeldirs, w, h, seed):
    """The gather of stem_rs_kernel directly against the oracle's blob (cat_12 of flownet.param:165): the kernel runs with one-hot weights in
    both convolutions (rife_hip_v4_tap what = 5), so its S16 output holds the 12-channel block input it assembled - every full-resolution
    pixel, through the strip / row-ring / range-boundary logic of the kernel - to the 2^-21 of two passes through the split-f16 matrix path.
    Flows leave the frame by hundreds of pixels."""
    new, _ = pair
    o = pyoracle.OracleRIFE(rife_v4=True); o.load(modeldirs["rife-v4.6"])
    name = None
    for line in open(os.path.join(modeldirs["rife-v4.6"], "flownet.param")):
        f = line.split()
        if len(f) > 6 and f[0] == "Concat" and f[2] == "2" and f[3] == "1":
            name = f[6]                                                  # the last two-input Concat = block 3's input
    a, c = gen_frames.noise_pair(w, h, seed) if seed % 2 else gen_frames.smooth_pair(w, h, seed)
    inj = injected_flows(w, h, 600 + seed, 3)
    want = o.v4_extract(a, c, 0.35, name, flows=inj)
    assert np.abs(want[8:12]).max() > 100
    got = new.v4_tap(a, c, 0.35, 5, 3, inj)
    assert got.shape == want.shape
    err = np.abs(got - want) - (6e-7 * np.abs(want) + 2.4e-7)
    assert err.max() <= 0, "worst excess %g at %s" % (float(err.max()), np.unravel_index(np.argmax(err), err.shape))


def test_row_streaming_kernels_are_run_to_run_stable(modeldirs, monkeypatch):
    """stem_rs_kernel and tail_rs_kernel keep loads in flight across their barriers and fill the register file of a CU with two workgroups (the
    condition under which the packed-fp32 build of the tile stems was unstable in round 2): 200 identical calls at 640 x 360 and 60 at
    1920 x 1080 (tail_rs forced at every size) return the same bytes every time."""
    monkeypatch.setenv("RIFE_HIP_TAIL_RS", "2")
    g = amd.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    for (w, h, n) in ((640, 360, 200), (1920, 1080, 60)):
        a, c = gen_frames.noise_pair(w, h, 5) if w < 1000 else gen_frames.tiled_real_pair(3)
        ref = g.process(a, c, 0.37)
        bad = sum(int(not np.array_equal(ref, g.process(a, c, 0.37))) for _ in range(n))
        assert bad == 0, "%d of %d calls at %dx%d differ from the first" % (bad, n, w, h)
