"""Pin the CPU oracle (oracle/liboracle.so) against an independent PyTorch-CPU execution of the same graphs.

The reference has no tests or golden vectors and cannot be built here, so agreement between two
independently written executors — a C++ restatement of ncnn's layer semantics and the PyTorch ops the models
were exported from — is the acceptance test for the oracle itself.  Tolerance: 2e-4 absolute on blobs of
O(1) magnitude (fp32 summation-order noise through up to 40 stacked convs), and <= 1 LSB on the u8 frame."""
import os

import numpy as np
import pytest
import torch

from oracle import pyoracle
from tools import gen_frames
from torch_graph import TorchNet

torch.set_num_threads(max(1, os.cpu_count() or 1))


def chw(img_u8, wp, hp):
    h, w, _ = img_u8.shape
    x = np.zeros((3, hp, wp), np.float32)
    x[:, :h, :w] = (img_u8.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)
    return torch.from_numpy(x)


def torch_v4(net, a, b, t):
    h, w, _ = a.shape
    wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32
    ins = {"in0": chw(a, wp, hp), "in1": chw(b, wp, hp), "in2": torch.full((1, hp, wp), np.float32(t))}
    outs = net.run(ins, ["flow0", "flow1", "flow2", "flow3", "out0"])
    o = outs[4][:, :h, :w] * 255.0 + 0.5
    u8 = o.to(torch.int32).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    return [x.numpy() for x in outs[:4]], u8


@pytest.mark.parametrize("w,h,t,seed", [(96, 64, 0.5, 1), (64, 96, 0.3, 2), (100, 60, 0.7, 3)])
def test_v46_graph_matches_torch(modeldirs, w, h, t, seed):
    d = modeldirs["rife-v4.6"]
    net = TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    a, b = gen_frames.smooth_pair(w, h, seed)
    o = pyoracle.OracleRIFE(rife_v4=True)
    o.set_gpu_crop(1)                     # pitch-correct crop (== reference whenever w % 32 == 0)
    o.load(d)
    tflows, tu8 = torch_v4(net, a, b, t)
    for k in range(4):
        f = o.v4_extract(a, b, t, "flow%d" % k)
        assert f.shape == tflows[k].shape
        assert np.abs(f - tflows[k]).max() < 2e-4, k
    ou8 = o.process(a, b, t)
    diff = np.abs(ou8.astype(int) - tu8.astype(int))
    assert diff.max() <= 1
    assert (diff > 0).mean() < 0.02


def test_v46_flow_injection_matches_torch(modeldirs):
    """Extractor semantics used by TTA: inject flow0..flow2, extract flow3 (rife.cpp:2653-2669)."""
    d = modeldirs["rife-v4.6"]
    net = TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    a, b = gen_frames.smooth_pair(64, 64, 5)
    o = pyoracle.OracleRIFE(rife_v4=True)
    o.load(d)
    rng = np.random.default_rng(0)
    inj = [rng.standard_normal((6, 64 // s, 64 // s)).astype(np.float32) * 0.3 for s in (8, 4, 2)]
    got = o.v4_extract(a, b, 0.5, "flow3", flows=inj)
    ins = {"in0": chw(a, 64, 64), "in1": chw(b, 64, 64), "in2": torch.full((1, 64, 64), 0.5)}
    for k in range(3):
        ins["flow%d" % k] = torch.from_numpy(inj[k])
    (want,) = net.run(ins, ["flow3"])
    assert np.abs(got - want.numpy()).max() < 2e-4


def test_v23_nets_match_torch(modeldirs):
    d = modeldirs["rife-v2.3"]
    W, H = 96, 64
    a, b = gen_frames.smooth_pair(W, H, 11)
    o = pyoracle.OracleRIFE(rife_v2=True)
    o.load(d)
    x0, x1 = chw(a, W, H), chw(b, W, H)
    fnet = TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    (tflow,) = fnet.run({"input0": x0, "input1": x1}, ["flow"])
    oflow = o.net_extract(0, {"input0": x0.numpy(), "input1": x1.numpy()}, "flow", 4 * W * H)
    assert oflow.shape == (4, H // 2, W // 2)
    assert np.abs(oflow - tflow.numpy()).max() < 2e-4
    cnet = TorchNet(os.path.join(d, "contextnet.param"), os.path.join(d, "contextnet.bin"))
    tf = cnet.run({"input.1": x0, "flow.0": tflow[:2]}, ["f1", "f2", "f3", "f4"])
    ctx = []
    for k, name in enumerate(["f1", "f2", "f3", "f4"]):
        of = o.net_extract(1, {"input.1": x0.numpy(), "flow.0": tflow[:2].numpy()}, name, 256 * W * H)
        assert np.abs(of - tf[k].numpy()).max() < 2e-4, name
        ctx.append(of)
    unet = TorchNet(os.path.join(d, "fusionnet.param"), os.path.join(d, "fusionnet.bin"))
    ins = {"img0": x0, "img1": x1, "flow": tflow}
    for k in range(4):
        ins[str(3 + k)] = tf[k]
        ins[str(7 + k)] = tf[k] * 0.5
    (tout,) = unet.run(ins, ["output"])
    oout = o.net_extract(2, {k: v.numpy() for k, v in ins.items()}, "output", 3 * W * H)
    assert np.abs(oout - tout.numpy()).max() < 2e-4


def test_v23_process_matches_torch(modeldirs):
    d = modeldirs["rife-v2.3"]
    W, H = 64, 64
    a, b = gen_frames.smooth_pair(W, H, 12)
    o = pyoracle.OracleRIFE(rife_v2=True)
    o.load(d)
    got = o.process(a, b, 0.5)
    x0, x1 = chw(a, W, H), chw(b, W, H)
    fnet = TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    cnet = TorchNet(os.path.join(d, "contextnet.param"), os.path.join(d, "contextnet.bin"))
    unet = TorchNet(os.path.join(d, "fusionnet.param"), os.path.join(d, "fusionnet.bin"))
    (flow,) = fnet.run({"input0": x0, "input1": x1}, ["flow"])
    c0 = cnet.run({"input.1": x0, "flow.0": flow[:2]}, ["f1", "f2", "f3", "f4"])
    c1 = cnet.run({"input.1": x1, "flow.0": flow[2:]}, ["f1", "f2", "f3", "f4"])
    ins = {"img0": x0, "img1": x1, "flow": flow}
    for k in range(4):
        ins[str(3 + k)] = c0[k]
        ins[str(7 + k)] = c1[k]
    (out,) = unet.run(ins, ["output"])
    want = (out * 255.0 + 0.5).to(torch.int32).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02


@pytest.mark.parametrize("fam", ["rife", "rife-HD"])
def test_v1_family_process_matches_torch(modeldirs, fam):
    """rife / rife-HD (= rife-UHD, rife-anime) graphs: SE blocks (global mean, two InnerProducts), 5x5 convs, bias-free
    strided skip convs, UnaryOp neg; one 2-channel flow bound to "flow.0" for frame 0 and "flow.1" for frame 1 (rife.cpp:2339-2362)."""
    d = modeldirs[fam]
    W, H = 96, 64
    a, b = gen_frames.smooth_pair(W, H, 13)
    o = pyoracle.OracleRIFE()
    o.load(d)
    got = o.process(a, b, 0.5)
    x0, x1 = chw(a, W, H), chw(b, W, H)
    fnet = TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    cnet = TorchNet(os.path.join(d, "contextnet.param"), os.path.join(d, "contextnet.bin"))
    unet = TorchNet(os.path.join(d, "fusionnet.param"), os.path.join(d, "fusionnet.bin"))
    (flow,) = fnet.run({"input0": x0, "input1": x1}, ["flow"])
    assert flow.shape == (2, H // 2, W // 2)
    oflow = o.net_extract(0, {"input0": x0.numpy(), "input1": x1.numpy()}, "flow", 2 * W * H)
    assert np.abs(oflow - flow.numpy()).max() < 2e-4
    c0 = cnet.run({"input.1": x0, "flow.0": flow}, ["f1", "f2", "f3", "f4"])
    c1 = cnet.run({"input.1": x1, "flow.1": flow}, ["f1", "f2", "f3", "f4"])
    ins = {"img0": x0, "img1": x1, "flow": flow}
    for k in range(4):
        ins[str(3 + k)] = c0[k]
        ins[str(7 + k)] = c1[k]
    (out,) = unet.run(ins, ["output"])
    want = (out * 255.0 + 0.5).to(torch.int32).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02


@pytest.mark.parametrize("stride", [1, 2])
def test_single_ops_match_torch(stride):
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, 19, 23)).astype(np.float32)
    w = rng.standard_normal((7, 5, 3, 3)).astype(np.float32)
    bias = rng.standard_normal(7).astype(np.float32)
    got = pyoracle.conv2d(x, w, bias, stride=stride, pad=1, act_type=2, act_p0=0.2)
    want = F.leaky_relu(F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(bias), stride=stride, padding=1), 0.2)[0].numpy()
    assert np.abs(got - want).max() < 1e-4
    wd = rng.standard_normal((6, 5, 4, 4)).astype(np.float32)     # ncnn [oc][ic][kh][kw]
    bd = rng.standard_normal(6).astype(np.float32)
    got = pyoracle.deconv2d(x, wd, bd)
    want = F.conv_transpose2d(torch.from_numpy(x)[None], torch.from_numpy(wd).transpose(0, 1).contiguous(), torch.from_numpy(bd), stride=2, padding=1)[0].numpy()
    assert np.abs(got - want).max() < 1e-4
    for s in (0.125, 0.25, 0.5, 2.0, 4.0, 8.0):
        xi = rng.standard_normal((3, 32, 64)).astype(np.float32)
        got = pyoracle.interp(xi, s, s)
        want = F.interpolate(torch.from_numpy(xi)[None], scale_factor=s, mode="bilinear", align_corners=False)[0].numpy()
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-5, s
    xs = rng.standard_normal((24, 6, 5)).astype(np.float32)
    assert np.array_equal(pyoracle.pixelshuffle(xs, 2), F.pixel_shuffle(torch.from_numpy(xs)[None], 2)[0].numpy())


def test_warp_matches_torch_restatement_and_quirks():
    from torch_graph import warp as twarp
    rng = np.random.default_rng(4)
    img = rng.uniform(0, 1, (3, 20, 30)).astype(np.float32)
    flow = (rng.standard_normal((2, 20, 30)) * 6).astype(np.float32)      # many samples leave the frame
    got = pyoracle.warp(img, flow)
    want = twarp(torch.from_numpy(img), torch.from_numpy(flow)).numpy()
    assert np.abs(got - want).max() < 1e-6
    # zero flow is the identity, exactly
    assert np.array_equal(pyoracle.warp(img, np.zeros_like(flow)), img)
    # integer flow inside the frame is an exact shift
    f2 = np.zeros_like(flow); f2[0] = 2; f2[1] = -1
    out = pyoracle.warp(img, f2)
    assert np.array_equal(out[:, 1:, :-2], img[:, :-1, 2:])
