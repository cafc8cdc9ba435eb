"""HIP engine vs CPU oracle on the reference's own data (tests/golden/ref/, SURVEY.md §8c / §8d "F1"): the real frame pair
images/0.png / images/1.png, tiled 1x1 (C1 size 640x360), 3x3 (C2 / C3 size 1920x1080) and 6x6 (C4 size 3840x2160), and the nine REAL
trained contextnet.bin files inside their model families (flownet / fusionnet stay seeded synthetic: the trained ones are absent
from the reference snapshot).  Bar: <= 1 LSB per channel (BASELINE.json north_star), PSNR and exact fraction reported on failure."""
import importlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_models

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")
REF = gen_models.REF_FIXTURES


def f1(tiles):
    from PIL import Image
    fr = [np.asarray(Image.open(os.path.join(REF, "images", n)).convert("RGB")) for n in ("0.png", "1.png")]
    return [np.ascontiguousarray(np.tile(f, (tiles, tiles, 1))) for f in fr]


def flags(fam):
    g = gen_models.GRAPH_FAMILY.get(fam, fam)
    return dict(rife_v2=g in ("rife-v2.3", "rife-v3.1"), rife_v4=g in ("rife-v4.6", "rife-v4"))     # like src/main.cpp:658-683


def check(got, want, what):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert d.max() <= 1, "%s: max %d LSB, %.4f %% exact, %d channels >= 2, PSNR %.1f dB" % (what, d.max(), 100 * (d == 0).mean(), int((d >= 2).sum()), psnr)
    assert (d == 0).mean() > 0.99, "%s: only %.3f %% of the channels exact" % (what, 100 * (d == 0).mean())


def pair_of(modeldir, **kw):
    g = amd.RIFE(0, **kw)
    g.load(modeldir)
    o = pyoracle.OracleRIFE(**kw)
    o.set_gpu_crop(1)
    o.load(modeldir)
    return g, o


@pytest.mark.parametrize("fam", sorted(gen_models.GRAPH_FAMILY))
def test_real_contextnet_inside_its_family_on_the_real_frames(fam):
    """C1 size: every family of the reference that ships a trained contextnet.bin, real PReLU slopes and all."""
    g, o = pair_of(gen_models.ensure_realctx(fam), **flags(fam))
    a, b = f1(1)
    check(g.process(a, b, 0.5), o.process(a, b, 0.5), fam + " 640x360 F1")


@pytest.mark.parametrize("fam,tiles,t", [("rife-v4.6", 1, 0.5), ("rife-v4.6", 3, 0.5), ("rife-v4.6", 3, 0.25), ("rife-v4.6", 6, 0.5)])
def test_v46_on_the_tiled_real_frames(modeldirs, fam, tiles, t):
    g, o = pair_of(modeldirs[fam], **flags(fam))
    a, b = f1(tiles)
    check(g.process(a, b, t), o.process(a, b, t), "%s %dx%d F1 t=%g" % (fam, a.shape[1], a.shape[0], t))


def test_v23_with_the_real_contextnet_at_1080p():
    g, o = pair_of(gen_models.ensure_realctx("rife-v2.3"), rife_v2=True)
    a, b = f1(3)
    check(g.process(a, b, 0.5), o.process(a, b, 0.5), "rife-v2.3 (real contextnet) 1920x1080 F1")


def test_v46_spatial_and_temporal_tta_at_1080p(modeldirs):
    """BASELINE config 5's modes (-x -z) at 1080p inside pytest (the 4K row lives in profiles/): 16 passes, F1 tiled 3x3."""
    kw = dict(tta_mode=True, tta_temporal_mode=True, rife_v4=True)
    g, o = pair_of(modeldirs["rife-v4.6"], **kw)
    a, b = f1(3)
    check(g.process(a, b, 0.5), o.process(a, b, 0.5), "rife-v4.6 -x -z 1920x1080 F1")
