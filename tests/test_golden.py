"""Committed golden fixtures (tests/golden/*.npz, made by tools/make_golden.py from the pinned oracle):
the oracle must keep reproducing them (CPU), and the HIP engine must match them without the oracle in the loop (GPU)."""
import glob
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def meta(g):
    """family + mode flags of a fixture (fixtures from before the family field are rife-v4.6)"""
    fam = str(g["family"]) if "family" in g.files else "rife-v4.6"
    kw = dict(tta_mode=bool(g["tta"]), tta_temporal_mode=bool(g["temporal"]), uhd_mode=bool(g["uhd"]) if "uhd" in g.files else False,
              rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    return fam, kw


def test_fixtures_exist():
    assert len(FIXTURES) >= 10


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_golden(modeldirs, path):
    from oracle import pyoracle
    g = np.load(path)
    fam, kw = meta(g)
    o = pyoracle.OracleRIFE(**kw)
    o.set_gpu_crop(1)
    o.load(modeldirs[fam])
    out = o.process(g["in0"], g["in1"], float(g["timestep"]))
    assert np.abs(out.astype(int) - g["out"].astype(int)).max() <= 1
    assert (out != g["out"]).mean() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_engine_matches_golden(modeldirs, path):
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    g = np.load(path)
    fam, kw = meta(g)
    e = amd.RIFE(0, **kw)
    e.load(modeldirs[fam])
    out = e.process(g["in0"], g["in1"], float(g["timestep"]))
    assert np.abs(out.astype(int) - g["out"].astype(int)).max() <= 1          # north_star: <= 1 LSB per channel
    if "flow3" in g.files:
        e = amd.test_build().RIFE(0, **kw)      # the stage taps live in the test build (include/rife_hip_test.h)
        e.load(modeldirs[fam])
        for k in range(4):
            f = e.v4_extract_flow(g["in0"], g["in1"], float(g["timestep"]), k)
            assert np.abs(f - g["flow%d" % k].astype(np.float32)).max() < 1e-2
