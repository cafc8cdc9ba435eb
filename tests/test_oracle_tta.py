"""TTA branches of the oracle, CPU only."""
import numpy as np

from oracle import pyoracle
from tools import gen_frames


def test_oracle_temporal_tta_close_to_plain_on_identical_frames(modeldirs):
    d = modeldirs["rife-v4.6"]
    a, _ = gen_frames.smooth_pair(64, 64, 3)
    plain = pyoracle.OracleRIFE(rife_v4=True); plain.load(d)
    temp = pyoracle.OracleRIFE(tta_temporal_mode=True, rife_v4=True); temp.load(d)
    p, t = plain.process(a, a, 0.5), temp.process(a, a, 0.5)
    assert p.shape == t.shape == a.shape
    assert np.abs(p.astype(int) - t.astype(int)).mean() < 8      # an ensemble of the same content stays close


def test_oracle_spatial_tta_runs_on_non_square(modeldirs):
    d = modeldirs["rife-v4.6"]
    a, b = gen_frames.smooth_pair(72, 40, 8)
    o = pyoracle.OracleRIFE(tta_mode=True, rife_v4=True); o.set_gpu_crop(1); o.load(d)
    out = o.process(a, b, 0.5)
    assert out.shape == a.shape and out.dtype == np.uint8
    plain = pyoracle.OracleRIFE(rife_v4=True); plain.set_gpu_crop(1); plain.load(d)
    d8 = np.abs(out.astype(int) - plain.process(a, b, 0.5).astype(int))
    assert 0 < d8.mean() < 25      # an ensemble: different from the plain pass, but the same picture


def test_oracle_v2_tta_symmetries(modeldirs):
    """v2 family -x -z (rife.cpp:1256-2138): the ensemble is equivariant under flips / transposition and symmetric in time —
    this only holds if the orientation index maps and the flow sign algebra are right."""
    o = pyoracle.OracleRIFE(tta_mode=True, tta_temporal_mode=True, rife_v2=True); o.set_gpu_crop(1); o.load(modeldirs["rife-v2.3"])
    a, b = gen_frames.smooth_pair(96, 64, 5)
    base = o.process(a, b, 0.5)
    assert base.shape == a.shape
    for f in (lambda x: x[:, ::-1], lambda x: x[::-1], lambda x: x.transpose(1, 0, 2)):
        got = o.process(np.ascontiguousarray(f(a)), np.ascontiguousarray(f(b)), 0.5)
        d = np.abs(got.astype(int) - f(base).astype(int))
        assert d.max() <= 1 and (d == 0).mean() > 0.99
    d = np.abs(o.process(b, a, 0.5).astype(int) - base.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.99
    plain = pyoracle.OracleRIFE(rife_v2=True); plain.set_gpu_crop(1); plain.load(modeldirs["rife-v2.3"])
    assert not np.array_equal(plain.process(a, b, 0.5), base)


def test_oracle_v4_tta_symmetries(modeldirs):
    o = pyoracle.OracleRIFE(tta_mode=True, tta_temporal_mode=True, rife_v4=True); o.set_gpu_crop(1); o.load(modeldirs["rife-v4.6"])
    a, b = gen_frames.smooth_pair(96, 64, 6)
    base = o.process(a, b, 0.5)
    for f in (lambda x: x[:, ::-1], lambda x: x[::-1], lambda x: x.transpose(1, 0, 2)):
        got = o.process(np.ascontiguousarray(f(a)), np.ascontiguousarray(f(b)), 0.5)
        d = np.abs(got.astype(int) - f(base).astype(int))
        assert d.max() <= 1 and (d == 0).mean() > 0.99
    d = np.abs(o.process(b, a, 0.5).astype(int) - base.astype(int))       # t = 0.5: the time-reversed ensemble is the same ensemble
    assert d.max() <= 1 and (d == 0).mean() > 0.99
