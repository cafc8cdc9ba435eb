"""Independent second oracle: execute an ncnn `.param`/`.bin` graph with PyTorch-CPU fp32 ops
(SURVEY.md §8c "Independent second oracle").  The RIFE graphs were exported from PyTorch, so
`conv2d / conv_transpose2d / pixel_shuffle / interpolate(bilinear, align_corners=False) / prelu / leaky_relu`
are the ground truth that ncnn's layers mirror; only `rife.Warp` has no torch twin and is restated from
reference src/warp.cpp:96-168 with torch tensor ops.  Used to pin oracle/liboracle.so."""
import struct

import numpy as np
import torch
import torch.nn.functional as F

from tools import ncnn_param


def load_bin(layers, path):
    raw = open(path, "rb").read()
    pos = 0
    for l in layers:
        p = l["params"]
        if l["type"] in ("Convolution", "Deconvolution"):
            n, oc = int(p[6]), int(p[0])
            (tag,) = struct.unpack_from("<I", raw, pos)
            pos += 4
            if tag == 0x01306B47:
                w = np.frombuffer(raw, "<f2", n, pos).astype(np.float32)
                pos += (n * 2 + 3) // 4 * 4
            else:
                assert tag == 0
                w = np.frombuffer(raw, "<f4", n, pos).copy()
                pos += n * 4
            l["weight"] = torch.from_numpy(w.copy())
            if int(p.get(5, 0)):
                l["bias"] = torch.from_numpy(np.frombuffer(raw, "<f4", oc, pos).copy())
                pos += oc * 4
            else:
                l["bias"] = None
        elif l["type"] == "InnerProduct":
            n, oc = int(p[2]), int(p[0])
            (tag,) = struct.unpack_from("<I", raw, pos)
            pos += 4
            assert tag == 0x01306B47
            l["weight"] = torch.from_numpy(np.frombuffer(raw, "<f2", n, pos).astype(np.float32).copy()).reshape(oc, -1)
            pos += (n * 2 + 3) // 4 * 4
            l["bias"] = None
            if int(p.get(1, 0)):
                l["bias"] = torch.from_numpy(np.frombuffer(raw, "<f4", oc, pos).copy())
                pos += oc * 4
        elif l["type"] == "PReLU":
            n = int(p[0])
            l["slope"] = torch.from_numpy(np.frombuffer(raw, "<f4", n, pos).copy())
            pos += n * 4
    assert pos == len(raw), (pos, len(raw))


def warp(img, flow):
    """img (C,H,W), flow (2,H,W): literal restatement of the reference's CPU Warp::forward."""
    C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    sx, sy = xs + flow[0], ys + flow[1]
    x0 = torch.floor(sx).to(torch.int64)
    y0 = torch.floor(sy).to(torch.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0c, x1c = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0c, y1c = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    alpha, beta = sx - x0c.to(torch.float32), sy - y0c.to(torch.float32)
    flat = img.reshape(C, -1)
    v0 = flat[:, (y0c * W + x0c).reshape(-1)].reshape(C, H, W)
    v1 = flat[:, (y0c * W + x1c).reshape(-1)].reshape(C, H, W)
    v2 = flat[:, (y1c * W + x0c).reshape(-1)].reshape(C, H, W)
    v3 = flat[:, (y1c * W + x1c).reshape(-1)].reshape(C, H, W)
    v4 = v0 * (1 - alpha) + v1 * alpha
    v5 = v2 * (1 - alpha) + v3 * alpha
    return v4 * (1 - beta) + v5 * beta


class TorchNet:
    def __init__(self, param_path, bin_path):
        self.layers = ncnn_param.parse(param_path)
        load_bin(self.layers, bin_path)
        self.producer = {}
        for l in self.layers:
            for t in l["tops"]:
                self.producer[t] = l

    @torch.no_grad()
    def run(self, inputs, want):
        """inputs: {blob: (C,H,W) tensor}; want: list of blob names.  Evaluates every layer whose bottoms are
        available, in file order (sufficient for these graphs)."""
        blobs = dict(inputs)
        for l in self.layers:
            if all(t in blobs for t in l["tops"]):
                continue
            if not all(b in blobs for b in l["bottoms"]) or l["type"] == "Input":
                continue
            x = [blobs[b] for b in l["bottoms"]]
            p, a, t = l["params"], l["arrays"], l["type"]
            if t == "Split":
                for o in l["tops"]:
                    blobs[o] = x[0]
                continue
            if t == "Concat":
                y = torch.cat(x, 0)
            elif t == "Crop":
                c0 = int(a[9][0]); c1 = a[10][0]
                c1 = x[0].shape[0] if c1 >= 2147483647 else int(c1)
                y = x[0][c0:c1]
            elif t == "Interp":
                y = F.interpolate(x[0][None], scale_factor=(p[1], p[2]), mode="bilinear", align_corners=False,
                                  recompute_scale_factor=False)[0]
            elif t == "Convolution":
                oc, k = int(p[0]), int(p[1])
                w = l["weight"].reshape(oc, -1, k, k)
                y = F.conv2d(x[0][None], w, l["bias"], stride=int(p.get(3, 1)), padding=int(p.get(4, 0)))[0]
                y = self._act(y, p, a)
            elif t == "Deconvolution":
                oc, k = int(p[0]), int(p[1])
                w = l["weight"].reshape(oc, -1, k, k).transpose(0, 1).contiguous()   # ncnn [oc][ic] -> torch [ic][oc]
                y = F.conv_transpose2d(x[0][None], w, l["bias"], stride=int(p.get(3, 1)), padding=int(p.get(4, 0)))[0]
                y = self._act(y, p, a)
            elif t == "PixelShuffle":
                y = F.pixel_shuffle(x[0][None], int(p[0]))[0]
            elif t == "ReLU":
                y = F.leaky_relu(x[0], p.get(0, 0.0))
            elif t == "PReLU":
                y = F.prelu(x[0][None], l["slope"])[0]
            elif t == "Sigmoid":
                y = torch.sigmoid(x[0])
            elif t == "Clip":
                y = x[0].clamp(p[0], p[1])
            elif t == "Pooling":
                assert int(p.get(0, 0)) == 1 and int(p.get(4, 0)) == 1
                y = x[0].mean(dim=(1, 2))                       # (C,) like ncnn's 1-D blob
            elif t == "InnerProduct":
                y = F.linear(x[0].reshape(-1), l["weight"], l["bias"])
                y = self._act(y, p, a)
            elif t == "UnaryOp":
                assert int(p.get(0, 0)) == 1
                y = -x[0]
            elif t == "BinaryOp":
                op = int(p.get(0, 0))
                b = x[1] if len(x) > 1 else torch.tensor(np.float32(p[2]))
                if len(x) > 1 and b.dim() == 1 and x[0].dim() == 3:
                    b = b[:, None, None]                        # per-channel operand (SE scale)
                y = {0: lambda: x[0] + b, 1: lambda: x[0] - b, 2: lambda: x[0] * b, 3: lambda: x[0] / b, 7: lambda: b - x[0]}[op]()
            elif t == "Eltwise":
                c = a[1]
                y = x[0] * np.float32(c[0]) + x[1] * np.float32(c[1])
            elif t == "rife.Warp":
                y = warp(x[0], x[1])
            else:
                raise NotImplementedError(t)
            blobs[l["tops"][0]] = y
        return [blobs[w] for w in want]

    @staticmethod
    def _act(y, p, a):
        act = int(p.get(9, 0))
        if act == 2:
            return F.leaky_relu(y, a[10][0])
        if act == 4:
            return torch.sigmoid(y)
        assert act == 0
        return y
