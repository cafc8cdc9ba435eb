"""oracle/_ref: the reference's OWN src/rife.cpp and src/warp.cpp, compiled UNMODIFIED from /root/reference against the ncnn look-alike of
oracle/refbuild/ (recipe oracle/refbuild/Makefile, gpuid = -1 = the reference's `-g -1` CPU path), against the restated oracle
(oracle/rife_oracle.cpp) - BIT FOR BIT.

What is the reference's own code in these comparisons: RIFE::load (src/rife.cpp:127-379), the dispatcher (381-393), process_cpu (1214-2460) and
process_v4_cpu (3204-4401) - the /255 + zero padding, the eight TTA orientations and their flow sign algebra, the temporal merges, the UHD
half-resolution path, the v2 flow slice, the blob binding order of every Extractor, the `*255 + 0.5` flat crop (confirmed
here on a ragged width), to_pixels - and Warp::forward (src/warp.cpp:96-168), which the graph interpreter reaches through
register_custom_layer exactly like ncnn would.  What stays restated on both sides: the arithmetic of the ncnn built-in layers (Tencent/ncnn
is an un-vendored submodule of the reference)."""
import numpy as np
import pytest

from oracle import pyoracle, pyref
from tools import gen_frames, gen_models

pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref/libref_rife.so absent and /root/reference not there to build it")

FLAGS = {"rife-v4.6": dict(rife_v4=True), "rife-v4": dict(rife_v4=True), "rife-v2.3": dict(rife_v2=True), "rife-v3.1": dict(rife_v2=True),
         "rife": {}, "rife-HD": {}}


def both(fam, modeldir, threads=4, **mode):
    r = pyref.RefRIFE(num_threads=threads, **mode, **FLAGS[fam]); r.load(modeldir)
    o = pyoracle.OracleRIFE(num_threads=threads, **mode, **FLAGS[fam]); o.load(modeldir)      # gpu_crop = 0: the literal CPU behaviour
    return r, o


def test_ref_library_is_the_reference_source_not_the_restatement():
    """The recipe compiles /root/reference/src/{rife,warp}.cpp and does NOT link the restated orchestration."""
    import os
    import subprocess
    mk = open(os.path.join(os.path.dirname(pyref.__file__), "refbuild", "Makefile")).read()
    assert "$(REF)/src/rife.cpp" in mk and "$(REF)/src/warp.cpp" in mk
    so = pyref.build()
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", so], text=True)
    assert "RIFE::process_v4_cpu(" in syms and "RIFE::process_cpu(" in syms and "Warp::forward(" in syms
    assert "oracle_process" not in syms and "oracle_create" not in syms          # rife_oracle.cpp's entry points are not in this library


@pytest.mark.parametrize("fam", sorted(FLAGS))
@pytest.mark.parametrize("mode", [dict(), dict(tta_temporal_mode=True)], ids=["plain", "z"])
@pytest.mark.parametrize("size", [(128, 64), (100, 60), (101, 61)], ids=["128x64", "ragged100x60", "odd101x61"])      # 101 x 61: ncnn's 16-byte cstep alignment pads the 3-D Mats of the frame
def test_plain_and_temporal_bit_identical(modeldirs, fam, mode, size):
    r, o = both(fam, modeldirs[fam], **mode)
    a, b = gen_frames.smooth_pair(size[0], size[1], 5)
    for t in (0.5, 0.3):
        if t != 0.5 and not FLAGS[fam].get("rife_v4"):
            continue                                        # only v4 takes a timestep
        assert np.array_equal(r.process(a, b, t), o.process(a, b, t)), (fam, mode, size, t)


@pytest.mark.parametrize("fam", ["rife-v4.6", "rife-v2.3", "rife-v3.1", "rife"])
@pytest.mark.parametrize("mode", [dict(tta_mode=True), dict(tta_mode=True, tta_temporal_mode=True)], ids=["x", "xz"])
def test_spatial_tta_bit_identical(modeldirs, fam, mode):
    r, o = both(fam, modeldirs[fam], **mode)
    a, b = gen_frames.smooth_pair(100, 60, 9)                # ragged: 100 x 60 pads to 128 x 64, orientations 4-7 are 64 x 128
    assert np.array_equal(r.process(a, b, 0.5), o.process(a, b, 0.5))


@pytest.mark.parametrize("fam", ["rife-v2.3", "rife-v3.1", "rife", "rife-HD"])
@pytest.mark.parametrize("mode", [dict(), dict(tta_temporal_mode=True), dict(tta_mode=True, tta_temporal_mode=True)], ids=["u", "uz", "uxz"])
def test_uhd_mode_bit_identical(modeldirs, fam, mode):
    if mode.get("tta_mode") and fam in ("rife-v3.1", "rife-HD"):
        pytest.skip("covered by rife-v2.3 / rife (same code path; keeps the CPU suite short)")
    r, o = both(fam, modeldirs[fam], uhd_mode=True, **mode)
    a, b = gen_frames.smooth_pair(100, 60, 11)               # pads to 128 x 64: the half-resolution pass is 64 x 32
    assert np.array_equal(r.process(a, b, 0.5), o.process(a, b, 0.5))


def test_flat_crop_quirk_is_the_reference_behaviour(modeldirs):
    """the CPU non-TTA crop walks h * w contiguous floats of a plane whose pitch is w_padded (src/rife.cpp:4375-4387).  The
    reference's own compiled code shows it: on a ragged width its output equals the oracle's literal mode and differs from the pitch-correct
    crop the GPU shader (rife_postproc.comp:42) and the HIP engine implement; on a multiple of 32 the two are the same."""
    d = modeldirs["rife-v4.6"]
    r = pyref.RefRIFE(rife_v4=True); r.load(d)
    lit = pyoracle.OracleRIFE(rife_v4=True, num_threads=4); lit.load(d)
    gpu = pyoracle.OracleRIFE(rife_v4=True, num_threads=4); gpu.set_gpu_crop(1); gpu.load(d)
    a, b = gen_frames.smooth_pair(100, 60, 3)
    ref = r.process(a, b, 0.5)
    assert np.array_equal(ref, lit.process(a, b, 0.5))
    shader = gpu.process(a, b, 0.5)
    assert not np.array_equal(ref, shader)
    assert np.array_equal(ref[0], shader[0])                 # row 0 starts at the same address in both walks
    a, b = gen_frames.smooth_pair(96, 60, 3)
    assert np.array_equal(r.process(a, b, 0.5), gpu.process(a, b, 0.5))


def test_timestep_0_and_1_return_the_inputs(modeldirs):
    r, o = both("rife-v4.6", modeldirs["rife-v4.6"])
    a, b = gen_frames.smooth_pair(64, 64, 2)
    assert np.array_equal(r.process(a, b, 0.0), a) and np.array_equal(r.process(a, b, 1.0), b)
    assert np.array_equal(o.process(a, b, 0.0), a) and np.array_equal(o.process(a, b, 1.0), b)


def test_c1_real_frames_bit_identical():
    """BASELINE config 1 sizes (640 x 360) on the reference's real frame pair: rife-v4.6 (seeded weights) and rife-v2.3 with the reference's
    REAL contextnet.bin inside (tests/golden/ref)."""
    from test_ref_fixtures import real_frames
    a, b = real_frames()
    for fam, d in (("rife-v4.6", gen_models.ensure(None, "rife-v4.6")), ("rife-v2.3", gen_models.ensure_realctx("rife-v2.3"))):
        r, o = both(fam, d, threads=pyoracle.default_threads())
        assert np.array_equal(r.process(a, b, 0.5), o.process(a, b, 0.5)), fam


@pytest.mark.parametrize("fam", sorted(gen_models.GRAPH_FAMILY))
def test_real_contextnet_families_bit_identical(fam):
    """All nine model directories for which the reference ships a real contextnet.bin, on a crop of the real frames."""
    from test_ref_fixtures import real_frames
    a, b = real_frames()
    a, b = np.ascontiguousarray(a[100:196, 200:360]), np.ascontiguousarray(b[100:196, 200:360])
    g = gen_models.GRAPH_FAMILY[fam]
    r, o = both(g, gen_models.ensure_realctx(fam))
    assert np.array_equal(r.process(a, b, 0.5), o.process(a, b, 0.5))


def test_uhd_on_a_half_size_that_is_not_a_multiple_of_32_breaks_the_reference_itself(modeldirs):
    """`-u` where the padded frame's half size is not a multiple of 32 (96 x 64 -> 48 x 32; 1280 x 720 -> 640 x 368 is the same case): the HIP engine
    refuses these with RIFE_HIP_EINVAL (tests/test_gpu_v2.py::test_v23_uhd_rejects_unsupported_size).  This is NOT an input the reference handles: its own
    compiled src/rife.cpp (process_cpu, 1256-1290 / GPU twin 928-945) feeds the IFNet a 48 x 32 frame, and with ncnn's shape rules (Interp: int(w * scale);
    Convolution 3 x 3 s2 p1: (w - 1) / 2 + 1; Deconvolution 4 x 4 s2 p1: 2 w; models/rife-v2.3/flownet.param:7-27) block 0 runs 48 x 32 -> Interp 1/8: 6 x 4 ->
    3 x 2 -> 2 x 1 -> Deconvolution: 4 x 2 -> Interp x8: 32 x 16 -> Interp x2: 64 x 32, against warped frames of 48 x 32 (the chain only round-trips when
    the IFNet input is a multiple of 32), so
    Concat_57 (flownet.param:35) receives blobs of different sizes.  ncnn's Concat does not check: its CPU kernel copies each bottom blob's whole cstep *
    channels floats into a top blob sized for the first one (a heap overflow of 2,048 floats here), the Vulkan shader reads the larger blob on the smaller
    grid (misaligned garbage).  The reference build on the shape-checking look-alike fails the extraction, src/rife.cpp ignores the return value
    (`ex.extract("flow", flow)`) and walks an empty Mat: it does not survive.  Run in a child process because of that."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from oracle import pyref\n"
        "from tools import gen_frames\n"
        "r = pyref.RefRIFE(num_threads=2, uhd_mode=True, rife_v2=True); r.load(%r)\n"
        "a, b = gen_frames.smooth_pair(96, 64, 3)\n"
        "out = r.process(a, b, 0.5)\n"
        "print('SURVIVED', out.shape)\n") % (__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), modeldirs["rife-v2.3"])
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "Concat" in p.stderr and "64x32 vs 48x32" in p.stderr, p.stderr[-400:]          # the mis-sized blobs of flownet.param:35
    assert p.returncode != 0 and "SURVIVED" not in p.stdout, (p.returncode, p.stdout[-200:])
    # ... while a size whose half IS a multiple of 32 goes through the same code bit-identically (test_uhd_mode_bit_identical above: 128 x 64)
