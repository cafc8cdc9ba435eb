"""rife-v4.6 on the HIP engine vs the CPU oracle, through the C-ABI (GPU box only).

Bar (BASELINE.json north_star): <= 1 LSB per RGB channel against the reference CPU path.  Stage taps are checked
at 1e-3 absolute (flows are O(1) px; fp32 reassociation noise through up to 40 stacked convs is ~1e-5)."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")
amd_t = amd.test_build()      # librife_hip_test.so: the same sources + the parity taps and kernel-selection switches (include/rife_hip_test.h)


@pytest.fixture(scope="module")
def engines(modeldirs):
    d = modeldirs["rife-v4.6"]
    g = amd.RIFE(0, rife_v4=True)
    g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True)
    o.set_gpu_crop(1)     # the HIP path crops like the reference's GPU shader (rife_postproc.comp:42); identical when w % 32 == 0
    o.load(d)
    return g, o


@pytest.fixture(scope="module")
def tap_engines(modeldirs):
    """The test build of the engine (stage taps, RIFE_HIP_* switches) next to the oracle; the frame-parity tests below run the PRODUCT library."""
    d = modeldirs["rife-v4.6"]
    g = amd_t.RIFE(0, rife_v4=True)
    g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True)
    o.set_gpu_crop(1)
    o.load(d)
    return g, o


def lsb_report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    return int(d.max()), float((d == 0).mean()), float((d == 1).mean()), psnr


@pytest.mark.parametrize("w,h", [(64, 64), (160, 96)])
def test_stage_flows_match_oracle(tap_engines, w, h):
    g, o = tap_engines
    a, b = gen_frames.smooth_pair(w, h, 21)
    for fi in range(4):
        got = g.v4_extract_flow(a, b, 0.5, fi)
        want = o.v4_extract(a, b, 0.5, "flow%d" % fi)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-3, fi


def test_flow_injection_matches_oracle(tap_engines):
    """The Extractor inject/extract contract TTA relies on (rife.cpp:2653-2669)."""
    g, o = tap_engines
    a, b = gen_frames.smooth_pair(96, 64, 22)
    rng = np.random.default_rng(1)
    inj = [(rng.standard_normal((6, 64 // s, 96 // s)) * 0.3).astype(np.float32) for s in (8, 4, 2)]
    for fi in (1, 2, 3):
        got = g.v4_extract_flow(a, b, 0.4, fi, inject=inj[:fi])
        want = o.v4_extract(a, b, 0.4, "flow%d" % fi, flows=inj[:fi])
        assert np.abs(got - want).max() < 1e-3, fi


@pytest.mark.parametrize("w,h,t,seed", [(640, 360, 0.5, 1000), (256, 192, 0.125, 1001), (100, 60, 0.7, 1002), (33, 47, 0.9, 1003)])
def test_process_within_1_lsb(engines, w, h, t, seed):
    g, o = engines
    a, b = gen_frames.smooth_pair(w, h, seed)
    got = g.process(a, b, t)
    want = o.process(a, b, t)
    mx, f0, f1, psnr = lsb_report(got, want)
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.97 and psnr > 48.0


def test_process_noise_frames_within_1_lsb(engines):
    g, o = engines
    a, b = gen_frames.noise_pair(128, 96, 7)       # F3 stress input: every pixel is an edge
    mx, f0, f1, psnr = lsb_report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0, f1, psnr)


def test_timestep_endpoints_return_inputs(engines):
    g, _ = engines
    a, b = gen_frames.smooth_pair(64, 48, 3)
    assert np.array_equal(g.process(a, b, 0.0), a)      # rife.cpp:2470-2480
    assert np.array_equal(g.process(a, b, 1.0), b)


def test_deterministic_and_reentrant(engines):
    """process() is const + re-entrant in the reference (two proc threads share one RIFE, main.cpp:860-863)."""
    import threading
    g, _ = engines
    a, b = gen_frames.smooth_pair(320, 192, 4)
    ref = g.process(a, b, 0.5)
    outs = [None] * 4
    def work(i):
        outs[i] = g.process(a, b, 0.5)
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for o in outs:
        assert np.array_equal(o, ref)


def test_1080p_within_1_lsb(engines):
    """BASELINE config 3 size (1920x1080 -> padded 1920x1088)."""
    g, o = engines
    a, b = gen_frames.smooth_pair(1920, 1080, 2000)
    mx, f0, f1, psnr = lsb_report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.97


def test_load_rejects_wrong_family(modeldirs):
    g = amd.RIFE(0, rife_v4=True)
    with pytest.raises(amd.RifeError):
        g.load(modeldirs["rife-v2.3"])       # flownet.param there is the v2.3 IFNet
    with pytest.raises(amd.RifeError):
        g.load("/nonexistent/dir")


def test_cpp_class_shim_matches_python_path(engines, modeldirs, tmp_path):
    """`class RIFE` (csrc/rife.h, the reference's surface) driven from C++ like src/main.cpp:360 does."""
    import subprocess
    from test_host_and_sharding import build_shim_demo
    g, _ = engines
    exe = build_shim_demo(tmp_path)
    a, b = gen_frames.smooth_pair(200, 120, 31)
    (tmp_path / "a.rgb").write_bytes(a.tobytes()); (tmp_path / "b.rgb").write_bytes(b.tobytes())
    r = subprocess.run([exe, modeldirs["rife-v4.6"], "200", "120", "0.25", str(tmp_path / "a.rgb"), str(tmp_path / "b.rgb"), str(tmp_path / "o.rgb")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.frombuffer((tmp_path / "o.rgb").read_bytes(), np.uint8).reshape(120, 200, 3)
    assert np.array_equal(got, g.process(a, b, 0.25))


@pytest.mark.parametrize("tta,temporal,w,h", [(True, False, 100, 60), (False, True, 96, 64), (True, True, 72, 40)])
def test_tta_modes_within_1_lsb(modeldirs, tta, temporal, w, h):
    """BASELINE config 5 (-x -z): 8 orientations x 2 directions with per-stage flow consensus
    (rife.cpp:2534-2930; CPU twin 3246-4145)."""
    d = modeldirs["rife-v4.6"]
    g = amd.RIFE(0, tta_mode=tta, tta_temporal_mode=temporal, rife_v4=True)
    g.load(d)
    o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=temporal, rife_v4=True)
    o.set_gpu_crop(1)
    o.load(d)
    a, b = gen_frames.smooth_pair(w, h, 40 + w)
    got = g.process(a, b, 0.3)
    want = o.process(a, b, 0.3)
    mx, f0, f1, psnr = lsb_report(got, want)
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.97
    assert np.array_equal(got, g.process(a, b, 0.3))      # deterministic


def test_tta_4k_within_1_lsb(modeldirs):
    """BASELINE config 5 at size: rife-v4.6 3840 x 2160 with `-x -z` (16 passes, the flow consensus after every block, the output average) on the
    reference's real frame pair tiled 6 x 6, against the oracle's 16-pass restatement (src/rife.cpp:3246-4145).  ~2 minutes of oracle time."""
    a, b = gen_frames.tiled_real_pair(6)
    d = modeldirs["rife-v4.6"]
    g = amd.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(tta_mode=True, tta_temporal_mode=True, rife_v4=True); o.set_gpu_crop(1); o.load(d)
    mx, f0, f1, psnr = lsb_report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.999, (mx, f0, f1, psnr)


def test_tta_differs_from_plain(engines, modeldirs):
    """Sanity: the ensemble really changes the result (otherwise the test above proves nothing)."""
    g, _ = engines
    gt = amd.RIFE(0, tta_mode=True, rife_v4=True)
    gt.load(modeldirs["rife-v4.6"])
    a, b = gen_frames.smooth_pair(96, 64, 5)
    assert not np.array_equal(g.process(a, b, 0.5), gt.process(a, b, 0.5))


def test_4k_within_1_lsb(engines):
    """BASELINE config 4 size (3840x2160 -> padded 3840x2176), the size of the north-star target."""
    g, o = engines
    a, b = gen_frames.smooth_pair_native(3840, 2160, 3000)      # F2 at native resolution; F1 tiled 6 x 6: tests/test_gpu_ref_fixtures.py
    mx, f0, f1, psnr = lsb_report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.999, (mx, f0, f1, psnr)


def test_process_batch_equals_single_calls(engines):
    """rife_hip_process_batch: n pairs over internal streams give the same pixels as n process() calls (incl. timestep 0 / 1 copies)."""
    g, _ = engines
    frames = [gen_frames.smooth_pair(160, 96, 70 + i)[0] for i in range(6)]
    ts = [0.5, 0.25, 0.0, 0.7, 1.0]
    want = [g.process(frames[i], frames[i + 1], ts[i]) for i in range(5)]
    got = g.process_batch(frames[:5], frames[1:6], ts)
    for i in range(5):
        assert np.array_equal(got[i], want[i]), i
    assert g.process_batch([], [], []) == []


@pytest.mark.parametrize("w,h,n", [(160, 96, 9), (640, 360, 7), (1920, 1080, 6)])
def test_process_batch_lockstep_groups_equal_single_calls(engines, tap_engines, w, h, n):
    """The lockstep-group path of rife_hip_process_batch (pairs of pairs; the coarse blocks' trunk layers are one launch for both, gridDim.y = 2;
    SURVEY 8f-2) against n single calls: identical bytes, with shared frames (a sequence), an odd pair left over and timestep 0 / 1 copies in the mix,
    and against the per-pair path of the same call (RIFE_HIP_BATCH_GROUPS=0)."""
    import os
    g, _ = engines
    gt, _ = tap_engines      # RIFE_HIP_BATCH_GROUPS is a switch of the test build
    frames = [gen_frames.smooth_pair(w, h, 500 + i)[i & 1] for i in range(n + 1)]
    ts = [(0.5, 0.25, 0.0, 0.7, 1.0, 0.125, 0.9, 0.3, 0.6)[i % 9] for i in range(n)]
    want = [g.process(frames[i], frames[i + 1], ts[i]) for i in range(n)]
    got = g.process_batch(frames[:n], frames[1:n + 1], ts)
    for i in range(n):
        assert np.array_equal(got[i], want[i]), (i, ts[i])
    os.environ["RIFE_HIP_BATCH_GROUPS"] = "0"
    try:
        got0 = gt.process_batch(frames[:n], frames[1:n + 1], ts)
    finally:
        del os.environ["RIFE_HIP_BATCH_GROUPS"]
    for i in range(n):
        assert np.array_equal(got0[i], want[i]), (i, ts[i])


def test_8k_within_1_lsb(engines):
    """Maximum-size case: 7680x4320 (4x the pixels of the north-star frame; ~6.5 GB of workspace)."""
    g, o = engines
    a, b = gen_frames.tiled_real_pair(12)       # F1: the reference's real frames, 12 x 12 tiles = 7680 x 4320 at native detail
    mx, f0, f1, psnr = lsb_report(g.process(a, b, 0.5), o.process(a, b, 0.5))
    assert mx <= 1, (mx, f0, f1, psnr)
    assert f0 > 0.999, (mx, f0, f1, psnr)


def test_device_frames_at_odd_addresses_equal_aligned_ones(engines):
    """rife_hip_process_device on frames whose device address is not 4-byte aligned (k_preproc, one pixel per lane) equals the same
    frames at aligned addresses (k_preproc4, four pixels per lane: the width is a multiple of 4) and the host-buffer call."""
    import torch
    g, _ = engines
    w, h = 128, 72
    f0, f1 = gen_frames.smooth_pair(w, h, 4242)
    want = g.process(f0, f1, 0.5)
    n = w * h * 3
    outs = []
    for off in (0, 1, 3):
        b0 = torch.zeros(n + 8, dtype=torch.uint8, device="cuda"); b1 = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
        b0[off:off + n] = torch.from_numpy(f0.reshape(-1)).cuda(); b1[off:off + n] = torch.from_numpy(f1.reshape(-1)).cuda()
        out = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()      # the engine enqueues on its own stream
        g.process_device(b0.data_ptr() + off, b1.data_ptr() + off, w, h, 0.5, out.data_ptr() + off, None)
        torch.cuda.synchronize()
        outs.append(out[off:off + n].cpu().numpy().reshape(h, w, 3))
    for o in outs:
        assert np.array_equal(o, want)


def test_4k_pass_is_run_to_run_stable(engines, tap_engines):
    """The same 3840x2160 pair four times through one engine: identical frames and identical block-1 / block-3 flows (the first stages that
    go through the fused stem kernels and the fused tail)."""
    g, _ = engines
    a, b = gen_frames.smooth_pair_native(3840, 2160, 705)
    outs = [g.process(a, b, 0.5) for _ in range(4)]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    gt, _ = tap_engines
    for fi in (1, 3):
        fl = [gt.v4_extract_flow(a, b, 0.5, fi) for _ in range(3)]
        for f in fl[1:]:
            assert np.array_equal(f, fl[0]), fi


@pytest.mark.parametrize("w,h", [(100, 60), (256, 192), (333, 241)])
def test_fused_tta_consensus_is_bit_identical_to_the_two_kernels(modeldirs, w, h, monkeypatch):
    """`-x -z`: k_v4_consensus (temporal + spatial flow consensus of a block in one pass over the sixteen flow tensors) against the two separate
    kernels (RIFE_HIP_TTA_CONSENSUS=0): the same additions in the same order, so the frames are the same bytes.
    Reference src/rife.cpp:3477-3512, 3515-3821."""
    a, b = gen_frames.smooth_pair(w, h, 77)
    monkeypatch.setenv("RIFE_HIP_TTA_CONSENSUS", "0")
    g0 = amd_t.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v4=True); g0.load(modeldirs["rife-v4.6"])
    monkeypatch.delenv("RIFE_HIP_TTA_CONSENSUS")
    g1 = amd.RIFE(0, tta_mode=True, tta_temporal_mode=True, rife_v4=True); g1.load(modeldirs["rife-v4.6"])      # the product (fused consensus)
    for t in (0.5, 0.3):
        x0, x1 = g0.process(a, b, t), g1.process(a, b, t)
        assert np.array_equal(x0, x1), "%d bytes differ" % int((x0 != x1).sum())


def test_hipgraph_replay_is_opt_in_and_bit_identical(engines, modeldirs):
    """RIFE_HIP_GRAPH=1 (a PRODUCT switch, read once per process: run_v4_replay in csrc/engine.hip) replays the plain pass <= 1920 x 1088 from a captured
    hipGraph: warm-up call, capture call, replays with new frames and timesteps - the same bytes as the default launches.  (Until round 5 a closure-type
    name collision made the switch's static initialiser read RIFE_HIP_TRUNK instead, so the replay was silently ON for every process: the default
    path below is therefore the one the earlier rounds never ran at these sizes.)"""
    import hashlib
    import os
    import subprocess
    import sys
    g, _ = engines
    code = (
        "import sys, hashlib, importlib; sys.path.insert(0, %r)\n"
        "from tools import gen_frames\n"
        "amd = importlib.import_module('rife-ncnn-vulkan_amd')\n"
        "g = amd.RIFE(0, rife_v4=True); g.load(%r)\n"
        "for (w, h) in ((160, 96), (100, 60), (160, 96), (640, 360)):\n"
        "    for i, t in enumerate((0.5, 0.25, 0.7, 0.9, 0.125)):\n"
        "        a, b = gen_frames.smooth_pair(w, h, 40 + i)\n"
        "        print('MD5', w, h, i, hashlib.md5(g.process(a, b, t).tobytes()).hexdigest())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), modeldirs["rife-v4.6"])
    env = dict(os.environ, RIFE_HIP_GRAPH="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-800:]
    got = [l.split() for l in p.stdout.splitlines() if l.startswith("MD5")]
    assert len(got) == 20
    for _, w, h, i, md5 in got:
        a, b = gen_frames.smooth_pair(int(w), int(h), 40 + int(i))
        want = hashlib.md5(g.process(a, b, (0.5, 0.25, 0.7, 0.9, 0.125)[int(i)]).tobytes()).hexdigest()
        assert md5 == want, (w, h, i)


@pytest.mark.parametrize("w,h,seed", [(640, 360, 1), (100, 60, 2), (33, 47, 3), (1920, 1080, 4), (3840, 2160, 5)])
def test_merged_first_flow_update_is_bit_identical(engines, modeldirs, w, h, seed, monkeypatch):
    """Round 5: the update after block 0 is sampled by block 1's stem from flow0 (assemble_pixel UPD = 2) and written together with block 1's update in one
    pass (k_flow_update2; flownet.param:47-58, 99-105) - against the three separate k_flow_update launches (RIFE_HIP_MERGE_FLOW0=0, test build): the same
    expressions in the same order, so the frames are the same bytes."""
    g, _ = engines
    monkeypatch.setenv("RIFE_HIP_MERGE_FLOW0", "0")           # read by every plain pass of the test build (the product ignores it)
    g0 = amd_t.RIFE(0, rife_v4=True); g0.load(modeldirs["rife-v4.6"])
    a, b = (gen_frames.noise_pair if seed % 2 else gen_frames.smooth_pair)(w, h, 60 + seed)
    for t in (0.5, 0.2):
        x0, x1 = g0.process(a, b, t), g.process(a, b, t)
        assert np.array_equal(x0, x1), "%d bytes differ" % int((x0 != x1).sum())
