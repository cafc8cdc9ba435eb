"""Synthetic model directories in the reference's on-disk format (ncnn `.param` + `.bin`).

The reference ships the graphs but not the trained `flownet.bin` / `fusionnet.bin`
(`/root/reference/.MISSING_LARGE_BLOBS`), and `/root/reference` does not exist on the GPU box.
This module therefore

* re-derives the IFNet v4.6 and the v2.3 IFNet / ContextNet / FusionNet topologies from their
  architecture description and writes them as ncnn `.param` text — the test
  `tests/test_models.py` proves them structurally identical to the reference's own `.param` files
  whenever `/root/reference` is present;
* writes seeded synthetic weights in ncnn's `.bin` layout (App. D: u32 tag 0x01306B47 + fp16
  weights padded to 4 B + fp32 bias per conv/deconv, fp32 slopes per PReLU).

CLI:  python -m tools.gen_models <outdir> [rife-v4.6|rife-v2.3] [--seed N]
"""
import os
import struct
import sys

import numpy as np

FP16_TAG = 0x01306B47


# ----------------------------------------------------------------------------------------------
# tiny graph builder that emits ncnn param text (auto-inserting Split layers like ncnn's converters)
# ----------------------------------------------------------------------------------------------
class Graph:
    def __init__(self):
        self.layers = []      # dict(type,name,bottoms,tops,params(str list), weights meta)
        self.counter = 0
        self.nblob = 0

    def _blob(self, name=None):
        if name is not None:
            return name
        self.nblob += 1
        return "b%d" % self.nblob

    def add(self, typ, bottoms, params=(), ntops=1, top_names=None, meta=None):
        self.counter += 1
        tops = [self._blob(top_names[i] if top_names else None) for i in range(ntops)]
        self.layers.append(dict(type=typ, name="%s_%d" % (typ.lower().replace(".", ""), self.counter), bottoms=list(bottoms),
                                tops=tops, params=list(params), meta=meta))
        return tops[0] if ntops == 1 else tops

    # ---- layer helpers -----------------------------------------------------------------------
    def input(self, name):
        return self.add("Input", [], top_names=[name])

    def concat(self, xs):
        return self.add("Concat", xs)

    def interp(self, x, s):
        return self.add("Interp", [x], ["0=2", "1=%e" % s, "2=%e" % s])

    def conv(self, x, cin, cout, stride=1, leaky=None, kind="trunk"):
        p = ["0=%d" % cout, "1=3"]
        if stride != 1:
            p.append("3=%d" % stride)
        p += ["4=1", "5=1", "6=%d" % (cin * cout * 9)]
        if leaky is not None:
            p += ["9=2", "-23310=1,%e" % leaky]
        return self.add("Convolution", [x], p, meta=dict(w=(cout, cin, 3, 3), kind=kind))

    def deconv(self, x, cin, cout, sigmoid=False, kind="head"):
        p = ["0=%d" % cout, "1=4", "3=2", "4=1", "5=1", "6=%d" % (cin * cout * 16)]
        if sigmoid:
            p.append("9=4")
        return self.add("Deconvolution", [x], p, meta=dict(w=(cout, cin, 4, 4), kind=kind))

    def prelu(self, x, c):
        return self.add("PReLU", [x], ["0=%d" % c], meta=dict(slope=c))

    def leaky(self, x, s):
        return self.add("ReLU", [x], ["0=%e" % s])

    def pixelshuffle(self, x, r, name=None):
        return self.add("PixelShuffle", [x], ["0=%d" % r], top_names=[name] if name else None)

    def crop(self, x, c0, c1):
        return self.add("Crop", [x], ["-23309=1,%d" % c0, "-23310=1,%d" % c1, "-23311=1,0"])

    def scalar(self, x, op, b):
        return self.add("BinaryOp", [x], ["0=%d" % op, "1=1", "2=%e" % b])

    def binary(self, a, b, op, name=None):
        return self.add("BinaryOp", [a, b], ["0=%d" % op] if op else [], top_names=[name] if name else None)

    def wsum(self, a, b, ca, cb):
        return self.add("Eltwise", [a, b], ["0=1", "-23301=2,%e,%e" % (ca, cb)])

    def warp(self, img, flow, name=None):
        return self.add("rife.Warp", [img, flow], top_names=[name] if name else None)

    def sigmoid(self, x):
        return self.add("Sigmoid", [x])

    def clip(self, x, lo, hi, name=None):
        return self.add("Clip", [x], ["0=%e" % lo, "1=%e" % hi], top_names=[name] if name else None)

    def convk(self, x, cin, cout, k=3, stride=1, bias=True, act=None, kind="plain"):
        """Convolution with an explicit kernel size (pad k//2), optional bias, optional fused activation (ncnn 9=...)."""
        p = ["0=%d" % cout, "1=%d" % k]
        if stride != 1:
            p.append("3=%d" % stride)
        p.append("4=%d" % (k // 2))
        if bias:
            p.append("5=1")
        p.append("6=%d" % (cin * cout * k * k))
        if act == "sigmoid":
            p.append("9=4")
        return self.add("Convolution", [x], p, meta=dict(w=(cout, cin, k, k), kind=kind, bias=bias))

    def pool_global_avg(self, x):
        return self.add("Pooling", [x], ["0=1", "4=1"])

    def inner(self, x, cin, cout, act, slope=None):
        p = ["0=%d" % cout, "2=%d" % (cin * cout)]
        if act == "leaky":
            p += ["9=2", "-23310=1,%e" % slope]
        elif act == "sigmoid":
            p.append("9=4")
        return self.add("InnerProduct", [x], p, meta=dict(w=(cout, cin, 1, 1), kind="fc", bias=False))

    def neg(self, x, name=None):
        return self.add("UnaryOp", [x], ["0=1"], top_names=[name] if name else None)

    # ---- emit --------------------------------------------------------------------------------
    def emit(self):
        # count consumers, insert Split after any producer whose top feeds >1 consumer
        uses = {}
        for l in self.layers:
            for b in l["bottoms"]:
                uses[b] = uses.get(b, 0) + 1
        out, nsplit, alias_next = [], 0, {}
        pending = {}     # blob -> list of split outputs still unassigned
        for l in self.layers:
            bottoms = []
            for b in l["bottoms"]:
                if b in pending:
                    bottoms.append(pending[b].pop(0))
                else:
                    bottoms.append(b)
            out.append((l["type"], l["name"], bottoms, l["tops"], l["params"]))
            for t in l["tops"]:
                n = uses.get(t, 0)
                if n > 1:
                    nsplit += 1
                    outs = ["%s_s%d" % (t, i) for i in range(n)]
                    out.append(("Split", "split_%d" % nsplit, [t], outs, []))
                    pending[t] = list(outs)
        blobs = set()
        for typ, name, bottoms, tops, params in out:
            blobs.update(bottoms)
            blobs.update(tops)
        lines = ["7767517", "%d %d" % (len(out), len(blobs))]
        for typ, name, bottoms, tops, params in out:
            lines.append("%-24s %-24s %d %d %s" % (typ, name, len(bottoms), len(tops), " ".join(bottoms + tops + params)))
        return "\n".join(lines) + "\n"

    def weighted(self):
        return [l for l in self.layers if l["meta"] is not None]


# ----------------------------------------------------------------------------------------------
# topologies
# ----------------------------------------------------------------------------------------------
def ifnet_v46():
    """rife-v4.6 IFNet: 4 coarse-to-fine blocks, trunks 192/128/96/64, 8 residual convs each,
    deconv(24) + PixelShuffle(2) heads, LeakyReLU(0.2)."""
    g = Graph()
    in0, in1, in2 = g.input("in0"), g.input("in1"), g.input("in2")
    F = M = None
    for b, (c, s) in enumerate(zip((192, 128, 96, 64), (8, 4, 2, 1))):
        if b == 0:
            x = g.interp(g.concat([in0, in1, in2]), 1.0 / s)
            cin = 7
        else:
            Fd = g.scalar(g.interp(F, 1.0 / s), 3, float(s)) if s > 1 else F
            w1 = g.warp(in1, g.crop(F, 2, 4))
            w0 = g.warp(in0, g.crop(F, 0, 2))
            x = g.concat([w0, w1, in2, M])
            if s > 1:
                x = g.interp(x, 1.0 / s)
            x = g.concat([x, Fd])
            cin = 12
        x = g.conv(x, cin, c // 2, 2, leaky=0.2, kind="stem")
        x = g.conv(x, c // 2, c, 2, leaky=0.2, kind="stem")
        for _ in range(8):
            y = g.conv(x, c, c, kind="res")
            x = g.leaky(g.binary(y, x, 0), 0.2)
        flow = g.pixelshuffle(g.deconv(x, c, 24), 2, "flow%d" % b)
        u = g.interp(flow, float(s)) if s > 1 else flow
        d4 = g.crop(u, 0, 4)
        if b == 0:
            F = g.scalar(d4, 2, float(s))
            M = g.crop(u, 4, 5)
        else:
            F = g.wsum(F, d4, 1.0, float(s)) if s > 1 else g.binary(F, d4, 0)
            M = g.binary(M, g.crop(u, 4, 5), 0)
    m = g.sigmoid(M)
    rm = g.scalar(m, 7, 1.0)
    a = g.binary(g.warp(in1, g.crop(F, 2, 4)), rm, 2)
    bb = g.binary(g.warp(in0, g.crop(F, 0, 2)), m, 2)
    g.binary(bb, a, 0, "out0")
    return g


def ifnet_v40():
    """rife-v4 (4.0) IFNet (models/rife-v4/flownet.param): the v4.6 skeleton with PReLU activations, a plain 8-conv trunk closed
    by ONE residual add (no activation after it), and a 5-channel deconv head at half the block resolution that is
    bilinearly upsampled by 2 x scale (no PixelShuffle)."""
    g = Graph()
    in0, in1, in2 = g.input("in0"), g.input("in1"), g.input("in2")
    F = M = None
    for b, (c, s) in enumerate(zip((192, 128, 96, 64), (8, 4, 2, 1))):
        if b == 0:
            x = g.interp(g.concat([in0, in1, in2]), 1.0 / s)
            cin = 7
        else:
            Fd = g.scalar(g.interp(F, 1.0 / s), 2, 1.0 / s) if s > 1 else g.add("Interp", [F], ["0=2"])
            w1 = g.warp(in1, g.crop(F, 2, 4))
            w0 = g.warp(in0, g.crop(F, 0, 2))
            x = g.concat([w0, w1, in2, M])
            x = g.interp(x, 1.0 / s) if s > 1 else g.add("Interp", [x], ["0=2"])
            x = g.concat([x, Fd])
            cin = 12
        x = g.prelu(g.conv(x, cin, c // 2, 2, kind="stem"), c // 2)
        x = g.prelu(g.conv(x, c // 2, c, 2, kind="stem"), c)
        t = x
        for _ in range(8):
            t = g.prelu(g.conv(t, c, c, kind="res"), c)
        x = g.binary(t, x, 0)
        flow = g.add("Deconvolution", [x], ["0=5", "1=4", "3=2", "4=1", "5=1", "6=%d" % (c * 5 * 16)], top_names=["flow%d" % b],
                     meta=dict(w=(5, c, 4, 4), kind="head"))
        u = g.interp(flow, 2.0 * s)
        d4 = g.crop(u, 0, 4)
        if b == 0:
            F = g.scalar(d4, 2, 2.0 * s)
            M = g.crop(u, 4, 5)
        else:
            F = g.wsum(F, d4, 1.0, 2.0 * s)
            M = g.binary(M, g.crop(u, 4, 5), 0)
    m = g.sigmoid(M)
    rm = g.scalar(m, 7, 1.0)
    a = g.binary(g.warp(in1, g.crop(F, 2, 4)), rm, 2)
    bb = g.binary(g.warp(in0, g.crop(F, 0, 2)), m, 2)
    g.binary(bb, a, 0, "out0")
    return g


def ifnet_v23():
    """rife-v2.3 IFNet: 4 blocks, trunks 384/256/192/96, 6 conv+PReLU each, deconv(4) heads,
    flow kept at half resolution, output = dF0+dF1+dF2+dF3."""
    g = Graph()
    x01 = g.concat([g.input("input0"), g.input("input1")])
    deltas = []
    for b, (c, s) in enumerate(zip((384, 256, 192, 96), (8, 4, 2, 1))):
        if b == 0:
            x = x01
            cin = 6
        else:
            acc = deltas[0]
            for d in deltas[1:]:
                acc = g.binary(acc, d, 0)
            Ff = g.scalar(g.interp(acc, 2.0), 2, 2.0)
            w0 = g.warp(g.crop(x01, 0, 3), g.crop(Ff, 0, 2))
            w1 = g.warp(g.crop(x01, 3, 2147483647), g.crop(Ff, 2, 4))
            x = g.concat([w0, w1, Ff])
            cin = 10
        if s > 1:
            x = g.interp(x, 1.0 / s)
        x = g.prelu(g.conv(x, cin, c // 2, 2, kind="stem"), c // 2)
        x = g.prelu(g.conv(x, c // 2, c, 2, kind="stem"), c)
        for _ in range(6):
            x = g.prelu(g.conv(x, c, c, kind="plain"), c)
        d = g.deconv(x, c, 4)
        if s > 1:
            d = g.interp(d, float(s))
        deltas.append(d)
    acc = g.binary(deltas[0], deltas[1], 0)
    acc = g.binary(acc, deltas[2], 0)
    g.binary(acc, deltas[3], 0, "flow")
    return g


def ifnet_v3():
    """rife-v3.0 / v3.1 IFNet (models/rife-v3.1/flownet.param): 3 blocks at scales 4, 2, 1, all 160 channels wide
    (stems 6|10 -> 80 -> 160), trunk = 3 x [conv+PReLU, conv+PReLU, + skip], deconv(4) heads; the flow is kept at half
    resolution and - unlike v2.3 - rescaled with every resize (x 1/s going in, x s coming out)."""
    g = Graph()
    x01 = g.add("Interp", [g.concat([g.input("input0"), g.input("input1")])], ["0=2"])
    deltas = []
    for b, s in enumerate((4, 2, 1)):
        if b == 0:
            x = g.interp(x01, 1.0 / s)
            cin = 6
        else:
            acc = deltas[0] if b == 1 else g.binary(deltas[0], deltas[1], 0)
            Ff = g.scalar(g.interp(acc, 2.0), 2, 2.0)
            w0 = g.warp(g.crop(x01, 0, 3), g.crop(Ff, 0, 2))
            w1 = g.warp(g.crop(x01, 3, 2147483647), g.crop(Ff, 2, 4))
            xw = g.concat([w0, w1])
            xw = g.interp(xw, 1.0 / s) if s > 1 else g.add("Interp", [xw], ["0=2"])
            fd = g.interp(Ff, 1.0 / s) if s > 1 else g.add("Interp", [Ff], ["0=2"])
            x = g.concat([xw, g.scalar(fd, 2, 1.0 / s)])
            cin = 10
        x = g.prelu(g.conv(x, cin, 80, 2, kind="stem"), 80)
        x = g.prelu(g.conv(x, 80, 160, 2, kind="stem"), 160)
        for _ in range(3):
            y = g.prelu(g.conv(x, 160, 160, kind="res"), 160)
            y = g.prelu(g.conv(y, 160, 160, kind="res"), 160)
            x = g.binary(y, x, 0)
        d = g.deconv(x, 160, 4)
        if s > 1:
            d = g.scalar(g.interp(d, float(s)), 2, float(s))
        deltas.append(d)
    g.binary(g.binary(deltas[0], deltas[1], 0), deltas[2], 0, "flow")
    return g


def contextnet_v23():
    g = Graph()
    x, f = g.input("input.1"), g.input("flow.0")
    cin = 3
    for lvl, c in enumerate((32, 64, 128, 256)):
        if lvl == 0:
            x = g.prelu(g.conv(x, cin, c, 2, kind="plain"), c)
            x = g.prelu(g.conv(x, c, c, 1, kind="plain"), c)
            x = g.prelu(g.conv(x, c, c, 2, kind="plain"), c)
            x = g.prelu(g.conv(x, c, c, 1, kind="plain"), c)
        else:
            x = g.prelu(g.conv(x, cin, c, 2, kind="plain"), c)
            x = g.prelu(g.conv(x, c, c, 1, kind="plain"), c)
        f = g.scalar(g.interp(f, 0.5), 2, 0.5)
        g.warp(x, f, "f%d" % (lvl + 1))
        cin = c
    return g


def fusionnet_v23():
    g = Graph()
    img0, img1, flow = g.input("img0"), g.input("img1"), g.input("flow")
    c0 = [g.input(str(i)) for i in (3, 4, 5, 6)]
    c1 = [g.input(str(i)) for i in (7, 8, 9, 10)]
    Ff = g.scalar(g.interp(flow, 2.0), 2, 2.0)
    w0 = g.warp(img0, g.crop(Ff, 0, 2))
    w1 = g.warp(img1, g.crop(Ff, 2, 4))
    x = g.concat([w0, w1, Ff])

    def down(x, cin, c):
        x = g.prelu(g.conv(x, cin, c, 2, kind="plain"), c)
        return g.prelu(g.conv(x, c, c, 1, kind="plain"), c)

    x = down(x, 10, 32)
    s0 = down(x, 32, 64)
    s1 = down(g.concat([s0, c0[0], c1[0]]), 128, 128)
    s2 = down(g.concat([s1, c0[1], c1[1]]), 256, 256)
    s3 = down(g.concat([s2, c0[2], c1[2]]), 512, 512)
    x = g.prelu(g.deconv(g.concat([s3, c0[3], c1[3]]), 1024, 256, kind="plain"), 256)
    x = g.prelu(g.deconv(g.concat([x, s2]), 512, 128, kind="plain"), 128)
    x = g.prelu(g.deconv(g.concat([x, s1]), 256, 64, kind="plain"), 64)
    x = g.prelu(g.deconv(g.concat([x, s0]), 128, 32, kind="plain"), 32)
    o = g.deconv(x, 32, 4, sigmoid=True, kind="head")
    res = g.scalar(g.scalar(g.crop(o, 0, 3), 2, 2.0), 1, 1.0)
    m = g.crop(o, 3, 4)
    a = g.binary(w0, m, 2)
    b = g.binary(w1, g.scalar(m, 7, 1.0), 2)
    g.clip(g.binary(g.binary(a, b, 0), res, 0), 0.0, 1.0, "output")
    return g


# ----------------------------------------------------------------------------------------------
# the v1 family: models/rife (3 x 3 convs) and models/rife-HD = rife-UHD = rife-anime (5 x 5 convs, one more level).
# Everything is built from one squeeze-and-excitation residual block:  y = conv(PReLU(conv(x)));  s = sigmoid(fc(leaky(fc(mean(y)))));
# out = PReLU(y * s + skip).  The negative slopes of the 16-wide bottleneck are trained constants stored in the .param itself
# (e.g. models/rife/flownet.param:15); here they come from the seeded generator like every other weight.
# ----------------------------------------------------------------------------------------------
def _se_tail(g, y, skip, c, rng):
    m = g.pool_global_avg(y)
    m = g.inner(m, c, 16, "leaky", float(rng.uniform(-0.05, 0.6)))
    m = g.inner(m, 16, c, "sigmoid")
    return g.prelu(g.binary(g.binary(y, m, 2), skip, 0), c)


def _se_res(g, x, c, k, rng):
    """same-resolution block of the IFNets (models/rife/flownet.param:10-21, models/rife-HD/flownet.param:10-21)"""
    y = g.prelu(g.convk(x, c, c, k, kind="res"), c)
    y = g.convk(y, c, c, 3, kind="res")
    return _se_tail(g, y, x, c, rng)


def _se_down(g, x, cin, c, rng):
    """stride-2 block of the ContextNet / FusionNet: bias-free strided conv as the skip path (models/rife/contextnet.param:6-16)"""
    skip = g.convk(x, cin, c, 3, 2, bias=False, kind="plain")
    y = g.prelu(g.convk(x, cin, c, 3, 2, kind="plain"), c)
    y = g.convk(y, c, c, 3, kind="res")
    return _se_tail(g, y, skip, c, rng)


def _ifnet_v1(widths, scales, k, rng):
    g = Graph()
    x01 = g.interp(g.concat([g.input("input0"), g.input("input1")]), 0.5)
    deltas = []

    def total():
        acc = deltas[0]
        for d in deltas[1:]:
            acc = g.binary(acc, d, 0)
        return acc
    for b, (c, s) in enumerate(zip(widths, scales)):
        if b == 0:
            x = g.interp(x01, 1.0 / s)
            cin = 6
        else:
            F = total()
            w0 = g.warp(g.crop(x01, 0, 3), F)
            w1 = g.warp(g.crop(x01, 3, 2147483647), g.neg(F))
            x = g.concat([w0, w1, F])
            if s > 1:
                x = g.interp(x, 1.0 / s)
            cin = 8
        x = g.prelu(g.convk(x, cin, c, k, 2, kind="stem"), c)
        for _ in range(6):
            x = _se_res(g, x, c, k, rng)
        d = g.pixelshuffle(g.convk(x, c, 8, 3, kind="head"), 2)
        if s > 1:
            d = g.interp(d, float(s))
        deltas.append(d)
    acc = deltas[0]
    for d in deltas[1:-1]:
        acc = g.binary(acc, d, 0)
    g.binary(acc, deltas[-1], 0, "flow")
    return g


def ifnet_v1(rng=None):
    """models/rife/flownet.param: 3 blocks (scales 4, 2, 1 of the half-resolution pair; 192 / 128 / 64 channels), 2-channel flow."""
    return _ifnet_v1((192, 128, 64), (4, 2, 1), 3, rng or np.random.default_rng(7))


def ifnet_hd(rng=None):
    """models/rife-HD/flownet.param: 4 blocks (scales 8, 4, 2, 1; 192 / 128 / 96 / 48 channels), 5 x 5 convs."""
    return _ifnet_v1((192, 128, 96, 48), (8, 4, 2, 1), 5, rng or np.random.default_rng(7))


def _contextnet_v1(widths, stem, rng):
    g = Graph()
    x = g.input("input.1")
    f = g.neg(g.input("flow.1"), "flow.0")
    cin = 3
    if stem:                                   # rife-HD: a plain strided conv first, so the first warp already needs flow / 2
        x = g.prelu(g.convk(x, 3, stem, 3, 2, kind="plain"), stem)
        cin = stem
    for i, c in enumerate(widths):
        x = _se_down(g, x, cin, c, rng)
        if i > 0 or stem:
            f = g.scalar(g.interp(f, 0.5), 2, 0.5)
        g.warp(x, f, "f%d" % (i + 1))
        cin = c
    return g


def contextnet_v1(rng=None):
    return _contextnet_v1((16, 32, 64, 128), 0, rng or np.random.default_rng(8))


def contextnet_hd(rng=None):
    return _contextnet_v1((32, 64, 128, 256), 32, rng or np.random.default_rng(8))


def _fusionnet_v1(hd, rng):
    g = Graph()
    img0, img1, flow = g.input("img0"), g.input("img1"), g.input("flow")
    c0 = [g.input(str(i)) for i in (3, 4, 5, 6)]
    c1 = [g.input(str(i)) for i in (7, 8, 9, 10)]
    Ff = g.scalar(g.interp(flow, 2.0), 2, 2.0)
    w0 = g.warp(img0, Ff)
    w1 = g.warp(img1, g.neg(Ff))
    x = g.concat([w0, w1, Ff])
    if hd:
        x = g.prelu(g.convk(x, 8, 32, 3, 2, kind="plain"), 32)
        s = [x]
        widths, cin = (64, 128, 256, 512), 32
        s.append(_se_down(g, s[-1], cin, widths[0], rng))
        for i in range(3):
            s.append(_se_down(g, g.concat([s[-1], c0[i], c1[i]]), 2 * widths[i], widths[i + 1], rng))
        x = g.prelu(g.deconv(g.concat([s[4], c0[3], c1[3]]), 1024, 256, kind="plain"), 256)
        x = g.prelu(g.deconv(g.concat([x, s[3]]), 512, 128, kind="plain"), 128)
        x = g.prelu(g.deconv(g.concat([x, s[2]]), 256, 64, kind="plain"), 64)
        x = g.prelu(g.deconv(g.concat([x, s[1]]), 128, 32, kind="plain"), 32)
        o = g.sigmoid(g.pixelshuffle(g.convk(x, 32, 16, 3, kind="head"), 2))
    else:
        s = [_se_down(g, x, 8, 32, rng)]
        widths = (32, 64, 128, 256)
        for i in range(3):
            s.append(_se_down(g, g.concat([s[-1], c0[i], c1[i]]), 2 * widths[i], widths[i + 1], rng))
        x = g.prelu(g.deconv(g.concat([s[3], c0[3], c1[3]]), 512, 128, kind="plain"), 128)
        x = g.prelu(g.deconv(g.concat([x, s[2]]), 256, 64, kind="plain"), 64)
        x = g.prelu(g.deconv(g.concat([x, s[1]]), 128, 32, kind="plain"), 32)
        x = g.prelu(g.deconv(g.concat([x, s[0]]), 64, 16, kind="plain"), 16)
        o = g.convk(x, 16, 4, 3, act="sigmoid", kind="head")
    res = g.scalar(g.scalar(g.crop(o, 0, 3), 2, 2.0), 1, 1.0)
    m = g.crop(o, 3, 4)
    a = g.binary(w0, m, 2)
    b = g.binary(w1, g.scalar(m, 7, 1.0), 2)
    g.clip(g.binary(g.binary(a, b, 0), res, 0), 0.0, 1.0, "output")
    return g


def fusionnet_v1(rng=None):
    return _fusionnet_v1(False, rng or np.random.default_rng(9))


def fusionnet_hd(rng=None):
    return _fusionnet_v1(True, rng or np.random.default_rng(9))


FAMILIES = {
    "rife-v4.6": {"flownet": ifnet_v46},
    "rife-v4": {"flownet": ifnet_v40},
    "rife-v2.3": {"flownet": ifnet_v23, "contextnet": contextnet_v23, "fusionnet": fusionnet_v23},
    "rife-v3.1": {"flownet": ifnet_v3, "contextnet": contextnet_v23, "fusionnet": fusionnet_v23},
    "rife": {"flownet": ifnet_v1, "contextnet": contextnet_v1, "fusionnet": fusionnet_v1},
    "rife-HD": {"flownet": ifnet_hd, "contextnet": contextnet_hd, "fusionnet": fusionnet_hd},
}


# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------
def synth_weights(graph, rng, head_gain, res_gain=1.0):
    """He-style init so that activations stay O(1) through the stacks and flows come out a few px.
    Returns list of (kind, weight fp16 array | None, bias fp32 | None, slope fp32 | None) in .bin order."""
    out = []
    for l in graph.weighted():
        m = l["meta"]
        if "slope" in m:
            out.append(("prelu", None, None, rng.uniform(0.0, 0.5, m["slope"]).astype(np.float32)))
            continue
        oc, ic, kh, kw = m["w"]
        if l["type"] == "Deconvolution":
            fan_in = ic * kh * kw / 4.0      # each output pixel of a k4 s2 deconv sees 2x2 taps
        else:
            fan_in = ic * kh * kw
        std = np.sqrt(2.0 / (fan_in * 1.04))
        if m["kind"] == "head":
            std *= head_gain
        elif m["kind"] == "res":
            std *= res_gain                  # residual branch: keep x + conv(x) from doubling the variance
        w = (rng.standard_normal((oc, ic, kh, kw)) * std).astype(np.float16)
        b = (rng.standard_normal(oc) * 0.01).astype(np.float32) if m.get("bias", True) else None
        out.append((l["type"], w, b, None))
    return out


def write_bin(path, weights):
    with open(path, "wb") as f:
        for kind, w, b, s in weights:
            if kind == "prelu":
                f.write(s.astype("<f4").tobytes())
                continue
            f.write(struct.pack("<I", FP16_TAG))
            raw = w.astype("<f2").tobytes()
            f.write(raw)
            f.write(b"\0" * ((-len(raw)) % 4))
            if b is not None:
                f.write(b.astype("<f4").tobytes())


def generate(outdir, family="rife-v4.6", seed=0x51FE, real_contextnet=None, flow_gain=1.0):
    """Write <outdir>/{flownet,...}.{param,bin}.  `real_contextnet`: optional path to the reference's real
    rife-v2.3 contextnet.bin (present in the reference tree); copied verbatim when given.  `flow_gain` multiplies the gain of the flow heads
    (rife-v4.6: flows and mask logits `flow_gain` times larger - the parity-margin tests, tests/test_gpu_margin.py)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    for net, builder in FAMILIES[family].items():
        g = builder()
        with open(os.path.join(outdir, net + ".param"), "w") as f:
            f.write(g.emit())
        if net == "contextnet" and real_contextnet and os.path.exists(real_contextnet):
            with open(real_contextnet, "rb") as src, open(os.path.join(outdir, net + ".bin"), "wb") as dst:
                dst.write(src.read())
            continue
        if family == "rife-v4.6":
            w = synth_weights(g, rng, head_gain=0.25 * flow_gain, res_gain=0.5)
        elif family == "rife-v4":
            w = synth_weights(g, rng, head_gain=0.25)
        elif family in ("rife", "rife-HD"):
            w = synth_weights(g, rng, head_gain=0.25 if net == "flownet" else 0.15, res_gain=0.7)
        elif family == "rife-v3.1":      # small fusion residual: keeps the synthetic output away from the 0 / 255 clip
            w = synth_weights(g, rng, head_gain=0.25 if net == "flownet" else 0.15)
        else:
            w = synth_weights(g, rng, head_gain=0.25 if net == "flownet" else 1.0)
        write_bin(os.path.join(outdir, net + ".bin"), w)
    return outdir


def default_dir(family="rife-v4.6"):
    """<repo>/_synth_models/<family>: git-ignored, regenerated on demand (deterministic), travels with gpurun snapshots."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(os.environ.get("RIFE_SYNTH_MODELS", os.path.join(root, "_synth_models")), family)


def ensure(outdir=None, family="rife-v4.6", seed=0x51FE, flow_gain=1.0):
    """Idempotent: (re)generate only if the directory is incomplete."""
    if outdir is None:
        outdir = default_dir(family if flow_gain == 1.0 and seed == 0x51FE else "%s-gain%g-seed%x" % (family, flow_gain, seed))
    need = [os.path.join(outdir, n + e) for n in FAMILIES[family] for e in (".param", ".bin")]
    if not all(os.path.exists(p) for p in need):
        generate(outdir, family, seed, flow_gain=flow_gain)
    return outdir


# reference model directory -> the generator family with the same three graphs (identical .param files; tests/test_models.py
# proves the structural equivalence against /root/reference whenever it exists)
GRAPH_FAMILY = {"rife": "rife", "rife-HD": "rife-HD", "rife-UHD": "rife-HD", "rife-anime": "rife-HD", "rife-v2": "rife-v2.3",
                "rife-v2.3": "rife-v2.3", "rife-v2.4": "rife-v2.3", "rife-v3.0": "rife-v3.1", "rife-v3.1": "rife-v3.1"}
REF_FIXTURES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref")


def ensure_realctx(ref_family, seed=0x51FE):
    """Model directory `<ref_family>-realctx`: the graphs of `ref_family` with the reference's REAL trained contextnet.bin
    (tests/golden/ref/models/<ref_family>/contextnet.bin, copied from the reference by tools/make_ref_fixtures.py) between a
    seeded synthetic flownet and fusionnet (the trained ones are absent from the reference snapshot)."""
    fam = GRAPH_FAMILY[ref_family]
    real = os.path.join(REF_FIXTURES, "models", ref_family, "contextnet.bin")
    if not os.path.exists(real):
        raise FileNotFoundError(real)
    outdir = default_dir(ref_family + "-realctx")
    need = [os.path.join(outdir, n + e) for n in FAMILIES[fam] for e in (".param", ".bin")]
    if not all(os.path.exists(p) for p in need) or os.path.getsize(os.path.join(outdir, "contextnet.bin")) != os.path.getsize(real):
        generate(outdir, fam, seed, real_contextnet=real)
    return outdir


if __name__ == "__main__":
    out = sys.argv[1]
    fam = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "rife-v4.6"
    seed = int(sys.argv[sys.argv.index("--seed") + 1], 0) if "--seed" in sys.argv else 0x51FE
    generate(out, fam, seed)
    print(out)
