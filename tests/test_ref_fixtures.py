"""The reference's own data files as fixtures (tests/golden/ref/, copied by tools/make_ref_fixtures.py): the nine REAL trained
`models/*/contextnet.bin` and the real frame pair `images/0.png`, `images/1.png` (SURVEY.md §8c "Usable fixtures", §8d "F1").

CPU side of the pinning: every real weight file must parse to exactly its last byte under the graph this repo executes for that
family, and the C++ oracle must agree with the independent PyTorch executor on the REAL ContextNet (real PReLU slopes span
-0.9 .. 1.2, far outside the synthetic U[0, 0.5]) applied to a REAL frame with rife.Warp at every pyramid level."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import pyoracle
from tools import gen_models
from torch_graph import TorchNet

REF = gen_models.REF_FIXTURES
FAMS = sorted(gen_models.GRAPH_FAMILY)


def real_frames():
    from PIL import Image
    return [np.asarray(Image.open(os.path.join(REF, "images", n)).convert("RGB")) for n in ("0.png", "1.png")]


def test_manifest_matches_files():
    man = json.load(open(os.path.join(REF, "MANIFEST.json")))
    assert len(man) == 11
    for rel, info in man.items():
        data = open(os.path.join(REF, rel), "rb").read()
        assert len(data) == info["bytes"] and hashlib.md5(data).hexdigest() == info["md5"], rel
    a, b = real_frames()
    assert a.shape == b.shape == (360, 640, 3)


@pytest.mark.parametrize("fam", FAMS)
def test_real_contextnet_bin_parses_to_eof(fam):
    d = gen_models.ensure_realctx(fam)
    v1 = gen_models.GRAPH_FAMILY[fam] in ("rife", "rife-HD")
    o = pyoracle.OracleRIFE(rife_v2=not v1)
    o.load(d)
    consumed, size = o.bin_bytes(1)
    assert size == os.path.getsize(os.path.join(REF, "models", fam, "contextnet.bin"))
    assert consumed == size


def flow_field(h, w, seed):
    """Smooth half-resolution flow of a few pixels, with a band that leaves the frame (the clamp-then-alpha rule of rife.Warp)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    f = np.stack([3.0 * np.sin(xx / 17.0 + rng.uniform(0, 6)) + 1.5 * np.cos(yy / 11.0), 2.0 * np.cos(xx / 23.0) - 2.5 * np.sin(yy / 13.0 + rng.uniform(0, 6))])
    f[:, :, :6] -= 9.0
    return f.astype(np.float32)


@pytest.mark.parametrize("fam", FAMS)
def test_real_contextnet_oracle_matches_torch_on_a_real_frame(fam):
    d = gen_models.ensure_realctx(fam)
    v1 = gen_models.GRAPH_FAMILY[fam] in ("rife", "rife-HD")
    a, _ = real_frames()
    a = a[:192, :256]                                        # a crop keeps the CPU suite fast; 32-aligned
    H, W = a.shape[:2]
    x = (a.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1).copy()
    flow = flow_field(H // 2, W // 2, 5)
    o = pyoracle.OracleRIFE(rife_v2=not v1)
    o.load(d)
    net = TorchNet(os.path.join(d, "contextnet.param"), os.path.join(d, "contextnet.bin"))
    for fname in (["flow.0", "flow.1"] if v1 else ["flow.0"]):
        want = net.run({"input.1": torch.from_numpy(x), fname: torch.from_numpy(flow)}, ["f1", "f2", "f3", "f4"])
        for k, name in enumerate(["f1", "f2", "f3", "f4"]):
            got = o.net_extract(1, {"input.1": x, fname: flow}, name, 256 * W * H)
            ref = want[k].numpy()
            assert got.shape == ref.shape
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(got - ref).max() < 2e-4 * scale, (fam, fname, name)
            assert float(np.abs(ref).max()) > 1e-3              # the real weights produce real features, not zeros
