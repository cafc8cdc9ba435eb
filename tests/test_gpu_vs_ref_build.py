"""HIP engine vs the reference's OWN compiled CPU path (oracle/_ref/libref_rife.so = /root/reference/src/rife.cpp + warp.cpp built unmodified
by oracle/refbuild/Makefile, gpuid -1; the library travels to the GPU box prebuilt).  Same bar as against the restated oracle: <= 1 LSB per
channel (BASELINE.json north_star).  Sizes are multiples of 32 in width, where the reference's CPU crop and its GPU shader
agree; the ragged case compares with the rows the two walks share."""
import importlib
import os

import numpy as np
import pytest

from oracle import pyoracle, pyref
from tools import gen_models

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pyref.available(), reason="oracle/_ref/libref_rife.so not built")]
amd = importlib.import_module("rife-ncnn-vulkan_amd")
REF = gen_models.REF_FIXTURES


def f1(tiles):
    from PIL import Image
    fr = [np.asarray(Image.open(os.path.join(REF, "images", n)).convert("RGB")) for n in ("0.png", "1.png")]
    return [np.ascontiguousarray(np.tile(f, (tiles, tiles, 1))) for f in fr]


def flags(fam):
    g = gen_models.GRAPH_FAMILY.get(fam, fam)
    return dict(rife_v2=g in ("rife-v2.3", "rife-v3.1"), rife_v4=g in ("rife-v4.6", "rife-v4"))


def check(got, want, what):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, "%s: max %d LSB, %.4f %% exact, %d channels >= 2" % (what, d.max(), 100 * (d == 0).mean(), int((d >= 2).sum()))
    assert (d == 0).mean() > 0.99, "%s: only %.3f %% of the channels exact" % (what, 100 * (d == 0).mean())


def pair_of(modeldir, **kw):
    g = amd.RIFE(0, **kw)
    g.load(modeldir)
    r = pyref.RefRIFE(num_threads=pyoracle.default_threads(), **kw)
    r.load(modeldir)
    return g, r


@pytest.mark.parametrize("tiles,t", [(1, 0.5), (1, 0.3), (3, 0.5)])
def test_v46_plain_vs_reference_build(modeldirs, tiles, t):
    g, r = pair_of(modeldirs["rife-v4.6"], rife_v4=True)
    a, b = f1(tiles)
    check(g.process(a, b, t), r.process(a, b, t), "rife-v4.6 %dx%d t=%g vs reference build" % (a.shape[1], a.shape[0], t))


@pytest.mark.parametrize("mode", [dict(tta_temporal_mode=True), dict(tta_mode=True), dict(tta_mode=True, tta_temporal_mode=True)], ids=["z", "x", "xz"])
def test_v46_tta_vs_reference_build(modeldirs, mode):
    g, r = pair_of(modeldirs["rife-v4.6"], rife_v4=True, **mode)
    a, b = f1(1)
    check(g.process(a, b, 0.5), r.process(a, b, 0.5), "rife-v4.6 640x360 %s vs reference build" % mode)


@pytest.mark.parametrize("fam", sorted(gen_models.GRAPH_FAMILY))
def test_real_contextnet_families_vs_reference_build(fam):
    g, r = pair_of(gen_models.ensure_realctx(fam), **flags(fam))
    a, b = f1(1)
    check(g.process(a, b, 0.5), r.process(a, b, 0.5), fam + " 640x360 F1 vs reference build")


@pytest.mark.parametrize("mode", [dict(uhd_mode=True), dict(tta_mode=True, tta_temporal_mode=True)], ids=["u", "xz"])
def test_v23_modes_vs_reference_build(mode):
    g, r = pair_of(gen_models.ensure_realctx("rife-v2.3"), rife_v2=True, **mode)
    a, b = f1(1)
    check(g.process(a, b, 0.5), r.process(a, b, 0.5), "rife-v2.3 640x360 %s vs reference build" % mode)
