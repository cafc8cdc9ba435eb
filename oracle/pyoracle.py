"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/liboracle.so (PARITY UNPINNED, see
oracle/ncnn_graph.h).  Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """(Re)build liboracle.so with the committed Makefile when sources are newer than the binary."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ncnn_graph.cpp", "ncnn_graph.h", "rife_oracle.cpp", "conv_cpu.cpp", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-j4"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_create.restype = ctypes.c_void_p
        L.oracle_create.argtypes = [ctypes.c_int] * 6
        L.oracle_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_set_gpu_crop.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_load.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.oracle_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        L.oracle_v4_extract.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.oracle_net_extract.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.oracle_bin_bytes.restype = ctypes.c_size_t
        L.oracle_bin_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def default_threads():
    """Usable cores (cgroup/affinity aware), capped: the conv loops parallelise over <= 192 output channels."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


class OracleRIFE:
    """Mirror of the reference's `RIFE(gpuid=-1, ...)` (src/rife.h:11-52) on the CPU restatement."""

    def __init__(self, tta_mode=False, tta_temporal_mode=False, uhd_mode=False, num_threads=None, rife_v2=False, rife_v4=False):
        if num_threads is None:
            num_threads = default_threads()
        self.num_threads = num_threads
        self.h = lib().oracle_create(int(tta_mode), int(tta_temporal_mode), int(uhd_mode), int(num_threads), int(rife_v2), int(rife_v4))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_destroy(self.h)
            self.h = None

    def set_gpu_crop(self, v):
        lib().oracle_set_gpu_crop(self.h, int(v))

    def load(self, modeldir):
        rc = lib().oracle_load(self.h, modeldir.encode())
        if rc:
            raise RuntimeError("oracle_load(%s) failed: %d" % (modeldir, rc))
        return 0

    def process(self, in0, in1, timestep):
        in0 = np.ascontiguousarray(in0, dtype=np.uint8)
        in1 = np.ascontiguousarray(in1, dtype=np.uint8)
        h, w, c = in0.shape
        assert c == 3 and in1.shape == in0.shape
        out = np.empty_like(in0)
        rc = lib().oracle_process(self.h, _p(in0), _p(in1), w, h, float(timestep), _p(out))
        if rc:
            raise RuntimeError("oracle_process failed: %d" % rc)
        return out

    def v4_extract(self, in0, in1, timestep, blob, flows=()):
        """Plain v4 graph: return blob `blob` as a (C,H,W) float32 array; `flows` injects flow0.."""
        in0 = np.ascontiguousarray(in0, dtype=np.uint8)
        in1 = np.ascontiguousarray(in1, dtype=np.uint8)
        h, w, _ = in0.shape
        wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32
        cap = 16 * wp * hp
        out = np.empty(cap, dtype=np.float32)
        flows = [np.ascontiguousarray(f, dtype=np.float32) for f in flows]
        arr = (ctypes.c_void_p * max(1, len(flows)))(*[f.ctypes.data for f in flows])
        ow, oh = ctypes.c_int(), ctypes.c_int()
        c = lib().oracle_v4_extract(self.h, _p(in0), _p(in1), w, h, float(timestep), blob.encode(), arr, len(flows), _p(out), cap,
                                    ctypes.byref(ow), ctypes.byref(oh))
        if c < 0:
            raise RuntimeError("oracle_v4_extract(%s) failed: %d" % (blob, c))
        return out[: c * ow.value * oh.value].reshape(c, oh.value, ow.value).copy()

    def net_extract(self, which, inputs, blob, cap_elems):
        """Generic tap: which = 0 flownet / 1 contextnet / 2 fusionnet; inputs = {name: (C,H,W) float32}."""
        names = list(inputs.keys())
        arrs = [np.ascontiguousarray(inputs[n], dtype=np.float32) for n in names]
        cn = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        ca = (ctypes.c_void_p * len(names))(*[a.ctypes.data for a in arrs])
        dims = np.array([[a.shape[2], a.shape[1], a.shape[0]] for a in arrs], dtype=np.int32)
        out = np.empty(cap_elems, dtype=np.float32)
        ow, oh = ctypes.c_int(), ctypes.c_int()
        c = lib().oracle_net_extract(self.h, which, len(names), cn, ca, _p(dims), blob.encode(), _p(out), cap_elems, ctypes.byref(ow), ctypes.byref(oh))
        if c < 0:
            raise RuntimeError("oracle_net_extract(%s) failed: %d" % (blob, c))
        return out[: c * ow.value * oh.value].reshape(c, oh.value, ow.value).copy()

    def bin_bytes(self, which=0):
        return lib().oracle_bin_bytes(self.h, which, 0), lib().oracle_bin_bytes(self.h, which, 1)


# ---- single ops (planar CHW float32) ---------------------------------------------------------
def conv2d(x, weight, bias, stride=1, pad=1, act_type=0, act_p0=0.0, num_threads=8):
    x = np.ascontiguousarray(x, np.float32); weight = np.ascontiguousarray(weight, np.float32); bias = np.ascontiguousarray(bias, np.float32)
    c, h, w = x.shape
    oc, ic, k, _ = weight.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = np.empty((oc, oh, ow), np.float32)
    L = lib()
    L.oracle_conv2d.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
    L.oracle_conv2d(_p(x), w, h, c, _p(weight), _p(bias), oc, k, stride, pad, act_type, act_p0, _p(out), num_threads)
    return out


def deconv2d(x, weight, bias, stride=2, pad=1, act_type=0, act_p0=0.0, num_threads=8):
    x = np.ascontiguousarray(x, np.float32); weight = np.ascontiguousarray(weight, np.float32); bias = np.ascontiguousarray(bias, np.float32)
    c, h, w = x.shape
    oc, ic, k, _ = weight.shape
    oh, ow = (h - 1) * stride + k - 2 * pad, (w - 1) * stride + k - 2 * pad
    out = np.empty((oc, oh, ow), np.float32)
    L = lib()
    L.oracle_deconv2d.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
    L.oracle_deconv2d(_p(x), w, h, c, _p(weight), _p(bias), oc, k, stride, pad, act_type, act_p0, _p(out), num_threads)
    return out


def warp(image, flow, num_threads=8):
    image = np.ascontiguousarray(image, np.float32); flow = np.ascontiguousarray(flow, np.float32)
    c, h, w = image.shape
    out = np.empty_like(image)
    L = lib()
    L.oracle_warp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.oracle_warp(_p(image), _p(flow), w, h, c, _p(out), num_threads)
    return out


def interp(x, hscale, wscale):
    x = np.ascontiguousarray(x, np.float32)
    c, h, w = x.shape
    out = np.empty((c, int(h * hscale), int(w * wscale)), np.float32)
    L = lib()
    L.oracle_interp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    L.oracle_interp(_p(x), w, h, c, hscale, wscale, _p(out))
    return out


def pixelshuffle(x, r):
    x = np.ascontiguousarray(x, np.float32)
    c, h, w = x.shape
    out = np.empty((c // (r * r), h * r, w * r), np.float32)
    L = lib()
    L.oracle_pixelshuffle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.oracle_pixelshuffle(_p(x), w, h, c, r, _p(out))
    return out
