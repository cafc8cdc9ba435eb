"""CPU simulation of reduced-precision operand schemes for the trunk convolutions of rife-v4.6.

The HIP trunk kernels feed the f16 matrix pipe with activations split as a = hi + lo (hi = f16(a), lo = f16(a - hi)) and f16 weights
(exact: the weights are stored as fp16 on disk).  This script asks what happens to the final u8 frame when the `lo` product runs
at lower precision (fp8 e4m3 / e5m2 operands for BOTH lo and the weights of the lo product), or is dropped, by evaluating the whole
graph in PyTorch with the operands of every 3x3 stride-1 Cin == Cout trunk convolution rounded accordingly and fp32 accumulation.

usage: python tools/sim_split_precision.py [w h]      (default 640 360: the reference's images tiled / cropped to that size)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_graph  # noqa: E402
from tools import gen_frames, gen_models  # noqa: E402

torch.set_num_threads(max(1, os.cpu_count() or 1))
SCHEME = {"name": "exact"}
LO_SCALE = 2.0 ** float(os.environ.get("LO_SCALE_LOG2", "11"))          # lo <= 2^-11 |a|, so lo * 2^11 <= |a|: only |a| > 448 saturates e4m3


def q8(x, dtype, scale):
    lim = 448.0 if dtype == torch.float8_e4m3fn else 57344.0
    return (x * scale).clamp(-lim, lim).to(dtype).float() / scale


def trunk_conv(x, w, b, stride=1, padding=0):
    s = SCHEME["name"]
    is_trunk = w.shape[0] == w.shape[1] and w.shape[2] == 3 and stride == 1 and w.shape[0] >= 64
    if s == "exact" or not is_trunk:
        return REAL_CONV(x, w, b, stride=stride, padding=padding)
    hi = x.half().float()
    if s == "hi":
        return REAL_CONV(hi, w, b, stride=stride, padding=padding)
    lo = x - hi
    if s.startswith("wino"):
        return wino_conv(hi + lo.half().float(), w, b, int(s[4:] or 3))
    if s == "rs_fp8":
        # conv_rs_kernel's fp8 form (block 3 only, 64 channels): the tensor between the layers is {hi f16, e4m3(lo * 2^9)}, so the skip
        # connection sees hi + lo_q too; the lo product runs on e4m3(w * 2^kw), kw = the largest power of two that keeps max |w| <= 448
        if w.shape[0] != 64:
            return REAL_CONV(x, w, b, stride=stride, padding=padding)
        kw = int(np.floor(np.log2(448.0 / max(float(w.abs().max()), 1e-30))))
        lo_q = q8(lo, torch.float8_e4m3fn, 2.0 ** 9)
        w_q = w if os.environ.get("RS_W_EXACT") else q8(w, torch.float8_e4m3fn, 2.0 ** kw)
        y = REAL_CONV(hi, w, b, stride=stride, padding=padding) + REAL_CONV(lo_q, w_q, None, stride=stride, padding=padding)
        if not os.environ.get("RS_SKIP_EXACT"):
            x.copy_(hi + lo_q)
        return y
    if s == "gpu_fp8":
        # what conv_h2c does: f16 weights carry 2^k (k <= 15 so that the identity tap 2^k stays finite), lo * 2^9 and w * 2^(k-9) go
        # through e4m3, one shared accumulator; the folded skip tap sees hi + lo_q, so the residual stream is rounded too
        k = min(15, int(np.floor(np.log2(65504.0 / max(float(w.abs().max()), 1.0)))))
        lo_q = q8(lo, torch.float8_e4m3fn, 2.0 ** 9)
        w_q = q8(w, torch.float8_e4m3fn, 2.0 ** (k - 9))
        y = REAL_CONV(hi, w, b, stride=stride, padding=padding) + REAL_CONV(lo_q, w_q, None, stride=stride, padding=padding)
        x.copy_(hi + lo_q)            # the Split alias that feeds the residual add
        return y
    if s == "hi+lo_f16":
        lo_q, w_q = lo.half().float(), w
    else:
        dt = torch.float8_e4m3fn if "e4m3" in s else torch.float8_e5m2
        lo_q = q8(lo, dt, LO_SCALE)
        wmax = float(w.abs().max())
        ws = 2.0 ** np.floor(np.log2((448.0 if dt == torch.float8_e4m3fn else 57344.0) / max(wmax, 1e-30)))     # one power-of-two scale per layer
        w_q = q8(w, dt, ws) if "w8" in s else w
    return REAL_CONV(hi, w, b, stride=stride, padding=padding) + REAL_CONV(lo_q, w_q, None, stride=stride, padding=padding)


_G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
_BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
_AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv(x, w, b, nprod):
    """Winograd F(2x2, 3x3) on the split-f16 matrix pipe (round 4 study, DESIGN (f)-0): 3x3 pad-1 stride-1 convolution of x (the S16 tensor's value hi + lo).
    Transformed weights U = G g G^T and transformed input tiles V = B^T d B are fp32 and are each split into f16 hi + lo; the element-wise GEMMs
    over the input channels run as nprod products with fp32 accumulation: 2 = Uhi Vhi + Uhi Vlo (weights rounded to f16), 3 = + Ulo Vhi, 4 = + Ulo Vlo.
    Matrix instructions per output pixel: 16 / 4 x nprod against 9 x 2 for the direct split-f16 form."""
    C, H, W = x.shape[-3], x.shape[-2], x.shape[-1]
    xb = x.reshape(1, C, H, W)
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(xb, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [1, C, th, tw, 4, 4]
    V = torch.einsum("ij,bcthjk,lk->bcthil", _BT, d, _BT)          # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", _G, w, _G)                # G g G^T
    Uh = U.half().float(); Ul = (U - Uh).half().float()
    Vh = V.half().float(); Vl = (V - Vh).half().float()
    M = torch.einsum("ocil,bcthil->bothil", Uh, Vh) + torch.einsum("ocil,bcthil->bothil", Uh, Vl)
    if nprod >= 3:
        M = M + torch.einsum("ocil,bcthil->bothil", Ul, Vh)
    if nprod >= 4:
        M = M + torch.einsum("ocil,bcthil->bothil", Ul, Vl)
    Y = torch.einsum("ij,bothjk,lk->bothil", _AT, M, _AT)          # [1, O, th, tw, 2, 2]
    O, th, tw = Y.shape[1], Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(1, O, th * 2, tw * 2)[:, :, :H, :W]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y.reshape(x.shape[:-3] + (O, H, W))


REAL_CONV = F.conv2d
torch_graph.F.conv2d = trunk_conv


def run(net, a, b, t):
    h, w, _ = a.shape
    wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32

    def chw(img):
        x = np.zeros((3, hp, wp), np.float32)
        x[:, :h, :w] = (img.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)
        return torch.from_numpy(x)
    outs = net.run({"in0": chw(a), "in1": chw(b), "in2": torch.full((1, hp, wp), np.float32(t))}, ["flow3", "out0"])
    o = outs[1][:, :h, :w] * 255.0 + 0.5
    return outs[0].numpy(), o.to(torch.int32).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 360)
    d = gen_models.ensure(None, "rife-v4.6")
    net = torch_graph.TorchNet(os.path.join(d, "flownet.param"), os.path.join(d, "flownet.bin"))
    only = os.environ.get("ONLY")
    pairs = {"F3 noise": gen_frames.noise_pair(w, h, 7)} if only == "F3" else {"F2 smooth": gen_frames.smooth_pair(w, h, 1000), "F3 noise": gen_frames.noise_pair(w, h, 7)}
    try:
        pairs["F1 real"] = gen_frames.real_pair(w, h)
    except Exception:
        pass
    schemes = os.environ.get("SCHEMES", "hi,hi+lo_f16,hi+lo_e4m3_w8,gpu_fp8").split(",")
    for name, (a, b) in pairs.items():
        for t in ((0.5,) if only else (0.5, 0.25)):
            SCHEME["name"] = "exact"
            f_ref, u_ref = run(net, a, b, t)
            for s in schemes:
                SCHEME["name"] = s
                f, u = run(net, a, b, t)
                dd = np.abs(u.astype(int) - u_ref.astype(int))
                print("%-10s t=%.2f %-16s flow max err %.2e   u8: max %d, ==1: %.5f %%, >=2: %d channels" %
                      (name, t, s, np.abs(f - f_ref).max(), dd.max(), 100.0 * (dd == 1).mean(), int((dd >= 2).sum())), flush=True)


if __name__ == "__main__":
    main()
