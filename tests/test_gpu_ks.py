"""The weight-stationary K-split trunk kernel of the coarse IFBlocks (csrc/conv_ks.h, round 4) against the kernels it replaces (conv_row_kernel /
conv_t64_kernel, RIFE_HIP_KS=0): same products, its own summation order (per consumer wave: hi chain + lo chain, then the K slices in order,
then the bias), so the two engines agree to summation-order noise - frames within 1 LSB with very few channels touched, flows of the blocks it
serves to 1e-4 - at aligned, ragged and tiny sizes, on a used workspace, through the TTA schedule and through rife_hip_process_batch's lockstep
groups (one launch for the coarse trunks of two pairs).  Reference layers: models/rife-v4.6/flownet.param:14-42, 66-94, 119-147.
The kernel is OPT-IN (RIFE_HIP_KS, test / bench builds only; the product neither compiles nor selects it), so the other test files do NOT run on it:
the last test of this file holds it - and the RIFE_HIP_RS_SPLIT epilogue variant of conv_rs - within 1 LSB of the CPU oracle directly."""
import importlib
import os

import numpy as np
import pytest

from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


def _engine(modeldir, mask, **kw):
    old = os.environ.get("RIFE_HIP_KS")
    os.environ["RIFE_HIP_KS"] = str(mask)                  # read by rife_hip_create
    try:
        g = amd.RIFE(0, rife_v4=True, **kw)
    finally:
        if old is None:
            del os.environ["RIFE_HIP_KS"]
        else:
            os.environ["RIFE_HIP_KS"] = old
    g.load(modeldir)
    return g


@pytest.fixture(scope="module")
def engines(modeldirs):
    d = modeldirs["rife-v4.6"]
    return {m: _engine(d, m) for m in (0, 1, 3, 7)}


@pytest.mark.parametrize("mask", [1, 3, 7])
@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1), (256, 192, 0.125, 2), (100, 60, 0.7, 3), (33, 47, 0.9, 4), (1, 1, 0.5, 5), (130, 9, 0.5, 12),
                                        (1920, 1080, 0.5, 6), (1000, 520, 0.3, 7), (2080, 1200, 0.5, 9), (3840, 2160, 0.5, 8)])
def test_ks_output_matches_the_round3_kernels(engines, mask, w, h, t, seed):
    if w * h >= 3840 * 2160 and mask == 3:
        pytest.skip("mask 3 = mask 1 at this size (block 2 is on conv_t64)")
    new, old = engines[mask], engines[0]
    a, b = gen_frames.smooth_pair(w, h, seed) if w * h < 4000000 else gen_frames.smooth_pair_native(w, h, seed)
    for x, y, tt in ((a, b, t), (b, a, 1.0 - t)):          # the second call runs on a used workspace: zero borders of the S16 tensors intact
        d = np.abs(new.process(x, y, tt).astype(np.int32) - old.process(x, y, tt).astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, "%dx%d: %d of %d bytes differ, max %d" % (w, h, int((d > 0).sum()), d.size, int(d.max()))
    if w * h <= 1920 * 1080:
        for fi in (1, 2, 3):
            fd = np.abs(new.v4_extract_flow(a, b, t, fi) - old.v4_extract_flow(a, b, t, fi)).max()
            assert fd < 1e-4, "block-%d flows differ by %g at %dx%d" % (fi, fd, w, h)
    x = new.process(a, b, t)
    for _ in range(3):
        assert np.array_equal(x, new.process(a, b, t)), "conv_ks is not deterministic"


def test_ks_many_launches_identical(engines):
    g = engines[7]
    a, b = gen_frames.noise_pair(1920, 1080, 3)
    x = g.process(a, b, 0.5)
    for _ in range(40):
        assert np.array_equal(x, g.process(a, b, 0.5))


def test_ks_batched_groups_equal_single_calls(engines):
    """rife_hip_process_batch runs the coarse trunks of two pairs as one launch (gridDim.y = 2, half the ranges each): same bytes as single calls."""
    g = engines[7]
    fr = [gen_frames.smooth_pair(1920, 1080, 20 + i) for i in range(4)]
    a0 = [f[0] for f in fr]; a1 = [f[1] for f in fr]; ts = [0.5, 0.25, 0.7, 0.5]
    outs = g.process_batch(a0, a1, ts)
    for i in range(4):
        assert np.array_equal(outs[i], g.process(a0[i], a1[i], ts[i])), "pair %d" % i


def test_ks_tta_passes_match(modeldirs):
    d = modeldirs["rife-v4.6"]
    new, old = _engine(d, 7, tta_mode=True, tta_temporal_mode=True), _engine(d, 0, tta_mode=True, tta_temporal_mode=True)
    a, b = gen_frames.smooth_pair(100, 60, 11)
    d = np.abs(new.process(a, b, 0.4).astype(np.int32) - old.process(a, b, 0.4).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_ks_and_rs_split_variants_within_1_lsb_of_the_oracle(modeldirs):
    """1 LSB against the round-3 kernels + 1 LSB of those against the oracle does not bound the opt-in variants against the oracle at 1 LSB.
    RIFE_HIP_KS=7 (every layer conv_ks can serve) and RIFE_HIP_RS_SPLIT=1 (read once per process: a child process of the test build) against the CPU oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, importlib; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle import pyoracle\n"
        "from tools import gen_frames\n"
        "amd = importlib.import_module('rife-ncnn-vulkan_amd').test_build()\n"
        "g = amd.RIFE(0, rife_v4=True); g.load(%r)\n"
        "o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(%r)\n"
        "for (w, h, t, seed) in ((640, 360, 0.5, 3), (256, 192, 0.3, 4), (1000, 520, 0.7, 5)):\n"
        "    a, b = gen_frames.smooth_pair(w, h, seed)\n"
        "    d = np.abs(g.process(a, b, t).astype(int) - o.process(a, b, t).astype(int))\n"
        "    print('MAXLSB', w, h, int(d.max()), float((d == 0).mean()))\n") % (root, modeldirs["rife-v4.6"], modeldirs["rife-v4.6"])
    env = dict(os.environ, RIFE_HIP_KS="7", RIFE_HIP_RS_SPLIT="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-800:]
    rows = [l.split() for l in p.stdout.splitlines() if l.startswith("MAXLSB")]
    assert len(rows) == 3
    for _, w, h, mx, exact in rows:
        assert int(mx) <= 1 and float(exact) > 0.99, (w, h, mx, exact)
