"""Per-kernel parity through the C-ABI: HIP kernels vs the CPU oracle on seeded inputs (GPU box only).

Tolerance for the fp32 MFMA convolutions: the kernel accumulates the same fp32 products as the oracle in a
different order (v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain), so |diff| <= 1e-5 * (1 + sum|a*b|) is the
rounding-noise bound used.  Warp is restated operation-for-operation: required bit-exact."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd").test_build()      # librife_hip_test.so: parity taps, single-kernel entry points and kernel-selection switches (include/rife_hip_test.h)


def conv_tol(x, w):
    return 2e-5 * (1.0 + np.abs(w).sum(axis=(1, 2, 3)).max() * np.abs(x).max())


@pytest.mark.parametrize("cin,cout,stride,h,w,res", [
    (7, 96, 2, 34, 60, False),      # v4.6 block-0 stem (K = 63, Cout not a multiple of 64)
    (12, 48, 2, 40, 64, False),     # block-2 stem (Cout padded to a 64-wide tile)
    (12, 32, 2, 36, 70, False),     # block-3 stem, ragged width
    (48, 96, 2, 24, 40, False),     # stem-1, N = 96
    (64, 64, 1, 24, 64, True),      # block-3 trunk conv: residual + leaky
    (96, 96, 1, 17, 33, True),      # block-2 trunk (N = 96, CC = 8), ragged tile edges
    (128, 128, 1, 9, 31, True),     # block-1 trunk, 2 N-tiles
    (192, 192, 1, 8, 32, True),     # block-0 trunk, 3 N-tiles, 12 channel chunks
    (16, 24, 1, 5, 7, False),       # tiny: one partial tile
])
def test_conv3x3_matches_oracle(cin, cout, stride, h, w, res):
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = pyoracle.conv2d(x, wt, b, stride=stride, pad=1)
    r = rng.standard_normal(want.shape).astype(np.float32) if res else None
    if res:
        want = want + r
    want = np.where(want < 0, want * np.float32(0.2), want)
    got = amd.op_conv3x3(x, wt, b, stride=stride, residual=r, slope=np.full(cout, 0.2, np.float32))
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= conv_tol(x, wt)


def test_conv3x3_prelu_slopes_and_identity_weights():
    """Transpose-detecting check: asymmetric one-hot weights pick a known (channel, tap); per-channel PReLU slopes."""
    cin, cout, h, w = 16, 40, 12, 37
    rng = np.random.default_rng(5)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = np.zeros((cout, cin, 3, 3), np.float32)
    for o in range(cout):
        wt[o, (o * 7) % cin, (o // 3) % 3, o % 3] = 1.0
    slope = rng.uniform(-0.9, 1.2, cout).astype(np.float32)
    got = amd.op_conv3x3(x, wt, np.zeros(cout, np.float32), stride=1, slope=slope)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for o in range(cout):
        ky, kx = (o // 3) % 3, o % 3
        v = xp[(o * 7) % cin, ky:ky + h, kx:kx + w]
        want = np.where(v < 0, v * slope[o], v)
        # single product per output: exact on the fp32 path; the split-f16 path (taken here, the weights are exactly fp16)
        # reconstructs x as hi + lo/2048, which is x up to 2^-22 relative
        assert np.all(np.abs(got[o] - want) <= 3.6e-7 * np.abs(want) + 6e-8), o   # 2^-22 (split) + 2^-24 (final add); f16-subnormal lo parts: abs error <= 3e-8 each


@pytest.mark.parametrize("cin,cout,h,w", [(64, 24, 16, 40), (192, 24, 5, 9), (96, 24, 9, 33), (32, 4, 12, 20), (128, 32, 6, 10), (256, 64, 4, 6)])
def test_deconv4x4_matches_oracle(cin, cout, h, w):
    rng = np.random.default_rng(cin + cout)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = pyoracle.deconv2d(x, wt, b)
    got = amd.op_deconv4x4(x, wt, b)
    assert got.shape == want.shape == (cout, 2 * h, 2 * w)
    assert np.abs(got - want).max() <= conv_tol(x, wt)


def test_warp_bit_exact_including_out_of_frame():
    rng = np.random.default_rng(9)
    img = rng.uniform(0, 1, (3, 45, 70)).astype(np.float32)
    flow = (rng.standard_normal((2, 45, 70)) * 9).astype(np.float32)
    flow[:, :4] *= 20                                           # far outside: exercises the clamp-then-alpha quirk
    assert np.array_equal(amd.op_warp(img, flow), pyoracle.warp(img, flow))
    img32 = rng.standard_normal((32, 20, 24)).astype(np.float32)   # v2.3 context-feature shape class
    fl = (rng.standard_normal((2, 20, 24)) * 3).astype(np.float32)
    assert np.array_equal(amd.op_warp(img32, fl), pyoracle.warp(img32, fl))


@pytest.mark.parametrize("c,h,w", [(64, 24, 64), (96, 17, 33), (128, 9, 31), (192, 8, 32), (64, 40, 100)])
def test_conv3x3_split_f16_trunk_path_matches_oracle(c, h, w):
    """Weights that are exactly fp16 (like every ncnn fp16-stored model) route stride-1 trunk layers to conv_h2_kernel:
    activations split into f16 hi + lo, two f16 MFMAs per k-step, fp32 accumulate.  Must be fp32-grade."""
    rng = np.random.default_rng(c + h)
    x = (rng.standard_normal((c, h, w)) * 3).astype(np.float32)
    x[:, :2] *= 1e-4                                            # tiny activations: below the f16 normal range
    x[:, 2:4] *= 300.0                                          # large ones
    wt = (rng.standard_normal((c, c, 3, 3)) / np.sqrt(c * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    want = pyoracle.conv2d(x, wt, b, stride=1, pad=1)
    want = np.where(want < 0, want * np.float32(0.2), want)
    got = amd.op_conv3x3(x, wt, b, stride=1, slope=np.full(c, 0.2, np.float32))
    err = np.abs(got - want)
    # fp32-grade, not f16-grade: measured 1.1e-6 of the output range (the fp32-MFMA kernel gives 0.7e-6, f16 storage would give ~5e-4)
    assert err.max() <= 4e-6 * np.abs(want).max()
    assert err[:, 7:].max() <= 1e-4           # rows fed only by O(1) activations


def test_f16_mfma_keeps_subnormals():
    """conv_h2*_kernel relies on the matrix pipe preserving f16 subnormal inputs (lo = f16(a - hi) is often subnormal)."""
    import ctypes
    from tools import benchlib                  # hardware probes live in the bench build, not in the product library
    L = benchlib.lib()
    out = ctypes.c_float()
    assert L.rife_hip_probe_f16_denorm(0, ctypes.byref(out)) == 0
    assert out.value == 16 * 2.0 ** -20


@pytest.mark.parametrize("cin,cout,h,w", [(32, 64, 36, 70), (48, 96, 24, 40), (96, 192, 18, 66), (64, 128, 10, 32), (16, 32, 12, 20)])
def test_conv3x3_stride2_split_f16_path_matches_oracle(cin, cout, h, w):
    """Stride-2 stem-1 class (c/2 -> c) with fp16-exact weights -> conv_h2s2_kernel."""
    rng = np.random.default_rng(cin + cout)
    x = (rng.standard_normal((cin, h, w)) * 2).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = pyoracle.conv2d(x, wt, b, stride=2, pad=1)
    want = np.where(want < 0, want * np.float32(0.2), want)
    got = amd.op_conv3x3(x, wt, b, stride=2, slope=np.full(cout, 0.2, np.float32))
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max() + 1e-6


@pytest.mark.parametrize("cin,cout,h,w", [(64, 24, 16, 40), (1024, 256, 5, 9), (128, 32, 20, 33), (32, 4, 12, 20), (256, 64, 9, 17)])
def test_deconv4x4_split_f16_path_matches_oracle(cin, cout, h, w):
    """fp16-exact weights route transposed convs to head_h2_kernel (4 parities per workgroup, 32-channel N-tiles),
    with per-channel PReLU slopes (v2.3 FusionNet up path)."""
    rng = np.random.default_rng(cin * 3 + cout)
    x = (rng.standard_normal((cin, h, w)) * 2).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 4, 4)) / np.sqrt(cin * 4)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    slope = rng.uniform(-0.5, 1.0, cout).astype(np.float32)
    want = pyoracle.deconv2d(x, wt, b)
    want = np.where(want < 0, want * slope[:, None, None], want)
    got = amd.op_deconv4x4(x, wt, b, slope=slope)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max() + 1e-6


def test_conv3x3_32ch_split_f16_path():
    """32 -> 32 stride-1 (v2.3 ContextNet / FusionNet) now also takes the split-f16 kernel (NS = 1)."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((32, 30, 50)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    slope = rng.uniform(0, 0.5, 32).astype(np.float32)
    want = pyoracle.conv2d(x, wt, b, stride=1, pad=1)
    want = np.where(want < 0, want * slope[:, None, None], want)
    got = amd.op_conv3x3(x, wt, b, stride=1, slope=slope)
    assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max() + 1e-6


@pytest.mark.parametrize("variant,name", [(4, "block 1 (S = 4)"), (2, "block 2 (S = 2)"), (1 + 16 * 256, "block 3 (S = 1, 64-byte records)")])
def test_fused_stem_kernels_are_deterministic(variant, name):
    """Identical launches of a fused stem kernel on the same random inputs (flows leaving the frame included) must give identical bytes.
    Regression test for the round-2 finding that the kernels of blocks 1 / 2 were NOT run-to-run stable when built with SLP-vectorized
    (packed fp32) arithmetic next to 8-byte warp-tap loads (csrc/Makefile); the probe runs the kernel alone at the 4K and 1080p geometry."""
    import ctypes
    from tools import benchlib
    L = benchlib.lib()
    L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
    reps = 10
    for wp, hp in ((3840, 2176), (1920, 1088)):
        mm = (ctypes.c_longlong * reps)()
        assert L.rife_hip_probe_stem_det(0, variant, wp, hp, reps, mm) == 0, L.rife_hip_last_error()
        assert list(mm) == [0] * reps, (name, wp, hp, list(mm))
