"""tail_rs_kernel (csrc/tail_rs.h): block 3's head (Deconvolution 4x4 s2 + PixelShuffle) + the tail of the graph + postproc in one row-streaming
kernel, against the tile kernel it replaces (head_h2_kernel<EPI_FINAL, true>; RIFE_HIP_TAIL_RS=0) and against the oracle.

Same arithmetic per pixel (k_final's, models/rife-v4.6/flownet.param:200-217, src/warp.cpp:96-168, src/rife_postproc.comp:39-62); the
deconvolution adds two K partial sums instead of running one chain over the four 16-channel chunks, so the flow deltas differ in the last bits:
frames within 1 LSB of the tile kernel's with very few channels touched, and within 1 LSB of the oracle on injected flows that leave the frame by
hundreds of pixels (the clamps of the gather), at aligned, ragged and tiny sizes (strip / range boundaries, masked rows and columns, the byte-wise
store path of widths that are not multiples of four)."""
import importlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames
from test_gpu_gather import injected_flows

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


def _engine(d, on):
    old = os.environ.get("RIFE_HIP_TAIL_RS")
    os.environ["RIFE_HIP_TAIL_RS"] = "2" if on else "0"                 # read at create time; 2 = at every frame size (the product takes it from 4K-class frames up)
    try:
        g = amd.RIFE(0, rife_v4=True); g.load(d)
    finally:
        if old is None: del os.environ["RIFE_HIP_TAIL_RS"]
        else: os.environ["RIFE_HIP_TAIL_RS"] = old
    return g


@pytest.fixture(scope="module")
def pair(modeldirs):
    d = modeldirs["rife-v4.6"]
    return _engine(d, True), _engine(d, False)


SIZES = [(256, 192, 1), (640, 360, 2), (333, 241, 3), (100, 60, 4), (33, 47, 5), (1, 1, 6), (130, 9, 7), (1000, 520, 8), (1920, 1080, 9), (257, 130, 10), (3840, 2160, 11)]


@pytest.mark.parametrize("w,h,seed", SIZES)
def test_frames_match_the_tile_kernel_and_are_deterministic(pair, w, h, seed):
    new, old = pair
    if w * h > 4000000: a, c = gen_frames.tiled_real_pair(6)
    else: a, c = gen_frames.noise_pair(w, h, seed) if seed % 3 == 0 else gen_frames.smooth_pair(w, h, 20 + seed)
    for x, y, t in ((a, c, 0.5), (c, a, 0.3)):                          # the second call runs on a used workspace
        p1, p0 = new.process(x, y, t), old.process(x, y, t)
        d = np.abs(p1.astype(np.int32) - p0.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, "%dx%d: %d of %d bytes differ, max %d" % (w, h, int((d > 0).sum()), d.size, int(d.max()))
    x = new.process(a, c, 0.5)
    for _ in range(3):
        assert np.array_equal(x, new.process(a, c, 0.5)), "tail_rs_kernel is not run-to-run identical"


@pytest.mark.parametrize("w,h,seed", [(640, 360, 31), (333, 241, 32), (100, 60, 33), (1920, 1080, 34)])
def test_against_the_oracle_on_injected_flows(pair, modeldirs, w, h, seed):
    new, _ = pair
    o = pyoracle.OracleRIFE(rife_v4=True); o.load(modeldirs["rife-v4.6"])
    a, c = gen_frames.smooth_pair(w, h, seed) if w < 1000 else gen_frames.tiled_real_pair(3)
    inj = injected_flows(w, h, 700 + seed, 3)
    wantf = o.v4_extract(a, c, 0.45, "out0", flows=inj)[:, :h, :w]
    want8 = np.clip((wantf * 255.0 + 0.5).astype(np.int32), 0, 255).transpose(1, 2, 0)
    got8 = new.v4_process_injected(a, c, 0.45, inj).astype(np.int32)
    dd = np.abs(got8 - want8)
    assert dd.max() <= 1 and (dd > 0).mean() < 1e-3, "%d of %d bytes differ, max %d" % (int((dd > 0).sum()), dd.size, int(dd.max()))
