"""rife-ncnn-vulkan_amd — MI355X-native RIFE frame interpolation behind the reference's `RIFE` surface.

Python host-side mirror of the reference's C++ class (nihui/rife-ncnn-vulkan src/rife.h:11-52):

    r = RIFE(gpuid, tta_mode=False, tta_temporal_mode=False, uhd_mode=False, num_threads=1, rife_v2=False, rife_v4=False)
    r.load(modeldir)                       # RIFE::load,    src/rife.cpp:127-379
    out = r.process(in0, in1, timestep)    # RIFE::process, src/rife.cpp:381-1212 / 2462-3202

Everything is computed by the hand-written HIP kernels in csrc/ through the C-ABI of include/rife_hip.h
(librife_hip.so).  There is no CPU or PyTorch fallback: importing works without a GPU, computing does not.
The package name contains '-' and '.', so import it with importlib.import_module("rife-ncnn-vulkan_amd").
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RIFE_HIP_LIB: load another build of the same sources as "the product" (tools/*.py A/B two builds with it; the tests never set it)
LIB_PATH = os.environ.get("RIFE_HIP_LIB") or os.path.join(_HERE, "librife_hip.so")
# the TEST build: same sources + -DRIFE_HIP_TEST_BUILD = include/rife_hip_test.h's parity taps / single-kernel entry points and the kernel-selection
# switches (RIFE_HIP_T64, _RS, _KS, _STEM_RS, _TAIL_RS, _FUSE_FLOW, ...) the kernel-vs-kernel tests flip.  The product exports include/rife_hip.h only.
TEST_LIB_PATH = os.path.join(_HERE, "librife_hip_test.so")
_lib = None
_testlib = None

# every symbol include/rife_hip.h declares (= everything the product library exports)
C_ABI_SYMBOLS = [
    "rife_hip_device_count", "rife_hip_create", "rife_hip_destroy", "rife_hip_load", "rife_hip_process",
    "rife_hip_process_device", "rife_hip_process_device_batch", "rife_hip_stream_create", "rife_hip_stream_destroy", "rife_hip_process_batch", "rife_hip_frame_upload", "rife_hip_process_frames", "rife_hip_frame_release",
    "rife_hip_last_error", "rife_hip_profile_enable", "rife_hip_profile_read",
    "rife_hip_host_alloc", "rife_hip_host_free", "rife_hip_host_register", "rife_hip_host_unregister",
    "rife_hip_graph_check", "rife_hip_param_hash",
]
# include/rife_hip_test.h: exported by librife_hip_test.so (and the bench build) only
TEST_ABI_SYMBOLS = ["rife_hip_v4_extract_flow", "rife_hip_v4_flow_dims", "rife_hip_v4_tap", "rife_hip_v4_process_injected", "rife_hip_op_conv3x3", "rife_hip_op_deconv4x4", "rife_hip_op_warp", "rife_hip_pool_state"]


def build(force=False):
    """Compile librife_hip.so (+ the test build, the class shim and the CLI) for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(_HERE, "..", "include", "rife_hip.h"), os.path.join(_HERE, "..", "include", "rife_hip_test.h")]
    outs = [os.path.join(_HERE, "librife_hip.so"), TEST_LIB_PATH]
    stale = any(not os.path.exists(o) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in srcs) for o in outs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", csrc] + (["-B"] if force else []))
    return os.path.join(_HERE, "librife_hip.so")


def _load(path, with_test_surface):
    if not os.path.exists(path):
        raise RuntimeError("%s is not built (run __graft_entry__.build() or make -C rife-ncnn-vulkan_amd/csrc); there is no fallback path" % os.path.basename(path))
    try:
        import torch  # noqa: F401   (if PyTorch-ROCm is importable it is imported first so that both share one libamdhip64)
    except Exception:
        pass
    L = ctypes.CDLL(path)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.rife_hip_device_count.restype = ci
    L.rife_hip_create.restype = vp
    L.rife_hip_create.argtypes = [ci] * 7
    L.rife_hip_destroy.argtypes = [vp]
    L.rife_hip_load.argtypes = [vp, ctypes.c_char_p]
    L.rife_hip_process.argtypes = [vp, vp, vp, ci, ci, cf, vp]
    L.rife_hip_process_device.argtypes = [vp, vp, vp, ci, ci, cf, vp, vp]
    L.rife_hip_process_device_batch.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, vp]
    L.rife_hip_stream_create.argtypes = [vp, ci, ci, ctypes.POINTER(vp)]
    L.rife_hip_stream_destroy.argtypes = [vp, vp]
    L.rife_hip_last_error.restype = ctypes.c_char_p
    L.rife_hip_profile_enable.argtypes = [vp, ci]
    L.rife_hip_profile_read.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, vp, vp, vp, ci]
    L.rife_hip_graph_check.argtypes = [ctypes.c_char_p]
    L.rife_hip_process_batch.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci]
    L.rife_hip_frame_upload.argtypes = [vp, vp, ci, ci, vp]
    L.rife_hip_process_frames.argtypes = [vp, vp, vp, cf, vp]
    L.rife_hip_frame_release.argtypes = [vp]
    L.rife_hip_param_hash.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    L.rife_hip_host_alloc.restype = vp
    L.rife_hip_host_alloc.argtypes = [ctypes.c_size_t]
    L.rife_hip_host_free.argtypes = [vp]
    L.rife_hip_host_register.argtypes = [vp, ctypes.c_size_t]
    L.rife_hip_host_unregister.argtypes = [vp]
    if with_test_surface:
        L.rife_hip_v4_extract_flow.argtypes = [vp, vp, vp, ci, ci, cf, ci, vp, ci, vp]
        L.rife_hip_v4_flow_dims.argtypes = [vp, ci, ci, ci, vp, vp, vp]
        L.rife_hip_v4_tap.argtypes = [vp, vp, vp, ci, ci, cf, ci, ci, vp, ci, vp]
        L.rife_hip_v4_process_injected.argtypes = [vp, vp, vp, ci, ci, cf, vp, ci, vp]
        L.rife_hip_op_conv3x3.argtypes = [ci, vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp]
        L.rife_hip_op_deconv4x4.argtypes = [ci, vp, ci, ci, ci, vp, vp, ci, vp, vp]
        L.rife_hip_op_warp.argtypes = [ci, vp, vp, ci, ci, ci, vp]
        L.rife_hip_pool_state.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    return L


def lib():
    """The PRODUCT C-ABI library (include/rife_hip.h)."""
    global _lib
    if _lib is None:
        surface = False
        if os.environ.get("RIFE_HIP_LIB") and os.path.exists(LIB_PATH):      # a test / bench build standing in for the product (tools/*.py): bind the taps it exports
            surface = hasattr(ctypes.CDLL(LIB_PATH), "rife_hip_v4_tap")
        _lib = _load(LIB_PATH, surface)
    return _lib


def testlib():
    """The TEST build (include/rife_hip.h + include/rife_hip_test.h + kernel-selection switches)."""
    global _testlib
    if _testlib is None:
        _testlib = _load(TEST_LIB_PATH, True)
    return _testlib


class _TestBuild:
    """`amd.test_build()`: the same Python surface on librife_hip_test.so - RIFE(...) engines with the parity taps and the RIFE_HIP_* kernel-selection switches,
    op_conv3x3 / op_deconv4x4 / op_warp.  Everything else resolves to the package itself."""

    def __init__(self, mod):
        self._mod = mod

    def RIFE(self, *a, **kw):
        return RIFE(*a, test_build=True, **kw)

    def lib(self):
        return testlib()

    def __getattr__(self, name):
        return getattr(self._mod, name)


def test_build():
    import sys
    testlib()
    return _TestBuild(sys.modules[__name__])


class RifeError(RuntimeError):
    pass


def _check(rc, what, L=None):
    if rc != 0:
        raise RifeError("%s failed (%d): %s" % (what, rc, (L or lib()).rife_hip_last_error().decode()))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def graph_check(param_base):
    """Raise RifeError if the generic graph executor has no kernel for some layer of <param_base>.param (CPU-only check)."""
    _check(lib().rife_hip_graph_check(param_base.encode()), "graph_check")


def device_count():
    return lib().rife_hip_device_count()


class _PinnedOwner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.rife_hip_host_free(self.ptr)
        self.ptr = None


def pinned_empty(shape, dtype=np.uint8):
    """A numpy array in page-locked host memory (rife_hip_host_alloc): copies to / from it are asynchronous DMA at PCIe rate.
    The memory is released when the last array that views it is collected."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib().rife_hip_host_alloc(max(1, nbytes))
    if not ptr:
        raise RifeError("rife_hip_host_alloc(%d) failed: %s" % (nbytes, lib().rife_hip_last_error().decode()))
    buf = (ctypes.c_uint8 * max(1, nbytes)).from_address(ptr)
    buf._owner = _PinnedOwner(ptr)        # arr.base -> buf -> owner: freed when the last view is collected
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class Frame:
    """A frame resident in device memory (rife_hip_frame_t): upload once, use as either side of any number of pairs."""

    def __init__(self, handle, w, h, L=None):
        self._f, self.w, self.h, self._L = handle, w, h, L or lib()

    def release(self):
        if getattr(self, "_f", None) and getattr(self, "_L", None) is not None:
            self._L.rife_hip_frame_release(self._f)
        self._f = None

    __del__ = release


class RIFE:
    """Same constructor arguments, in the same order, as the reference's `RIFE` (src/rife.h:14)."""

    def __init__(self, gpuid, tta_mode=False, tta_temporal_mode=False, uhd_mode=False, num_threads=1, rife_v2=False, rife_v4=False, test_build=False):
        # test_build (not a reference argument): the engine lives in librife_hip_test.so - parity taps (v4_*) and kernel-selection switches
        self._L = testlib() if test_build else lib()
        self._taps = test_build or hasattr(self._L, "rife_hip_v4_tap")
        self._h = self._L.rife_hip_create(int(gpuid), int(tta_mode), int(tta_temporal_mode), int(uhd_mode), int(num_threads),
                                          int(rife_v2), int(rife_v4))
        if not self._h:
            raise RifeError("rife_hip_create: " + self._L.rife_hip_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_L", None) is not None:       # at interpreter shutdown the module globals may be gone already
            try:
                self._L.rife_hip_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def load(self, modeldir):
        _check(self._L.rife_hip_load(self._h, os.fspath(modeldir).encode()), "load", self._L)
        return 0

    def process(self, in0image, in1image, timestep, outimage=None):
        """in0image / in1image: (h, w, 3) uint8 RGB arrays (the ncnn::Mat the CLI builds, src/main.cpp:187)."""
        a = np.ascontiguousarray(in0image, dtype=np.uint8)
        b = np.ascontiguousarray(in1image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3 or a.shape != b.shape:
            raise ValueError("frames must be (h, w, 3) uint8 arrays of equal size")
        h, w, _ = a.shape
        out = outimage if outimage is not None else np.empty_like(a)
        if not isinstance(out, np.ndarray) or out.shape != a.shape or out.dtype != np.uint8 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("outimage must be a writable contiguous (h, w, 3) uint8 array of the frames' size")
        _check(self._L.rife_hip_process(self._h, _p(a), _p(b), w, h, float(timestep), _p(out)), "process", self._L)
        return out

    def upload(self, image):
        """Stream mode: copy one (h, w, 3) uint8 frame to the device and keep it there."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("frame must be an (h, w, 3) uint8 array")
        f = ctypes.c_void_p()
        _check(self._L.rife_hip_frame_upload(self._h, _p(a), a.shape[1], a.shape[0], ctypes.byref(f)), "frame_upload", self._L)
        return Frame(f, a.shape[1], a.shape[0], self._L)

    def process_frames(self, frame0, frame1, timestep, outimage=None):
        """process() between two resident frames; same pixels as process() on the host arrays they were uploaded from."""
        if frame0._f is None or frame1._f is None:
            raise ValueError("frame was released")
        out = outimage if outimage is not None else np.empty((frame0.h, frame0.w, 3), np.uint8)
        if out.shape != (frame0.h, frame0.w, 3) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("outimage must be a contiguous (h, w, 3) uint8 array of the frames' size")
        _check(self._L.rife_hip_process_frames(self._h, frame0._f, frame1._f, float(timestep), _p(out)), "process_frames", self._L)
        return out

    def process_device(self, d_in0, d_in1, w, h, timestep, d_out, stream=None):
        """Device pointers (ints) to tightly packed u8 HWC RGB frames; enqueues on `stream` (hipStream_t as int)."""
        _check(self._L.rife_hip_process_device(self._h, d_in0, d_in1, w, h, float(timestep), d_out, stream), "process_device", self._L)

    def stream_create(self, part, nparts):
        """A hipStream_t (int) that owns the compute units i with i % nparts == part (rife_hip_stream_create); for process_device."""
        st = ctypes.c_void_p()
        _check(self._L.rife_hip_stream_create(self._h, int(part), int(nparts), ctypes.byref(st)), "stream_create", self._L)
        return st.value

    def stream_destroy(self, stream):
        _check(self._L.rife_hip_stream_destroy(self._h, stream), "stream_destroy", self._L)

    def process_device_batch(self, d_in0, d_in1, w, h, timesteps, d_out, stream=None):
        """n resident pairs in one call (rife_hip_process_device_batch): lists of device pointers; enqueued relative to `stream`."""
        n = len(d_in0)
        if len(d_in1) != n or len(d_out) != n or len(timesteps) != n:
            raise ValueError("one in1 / out / timestep per pair")
        pa = (ctypes.c_void_p * n)(*[int(x) for x in d_in0])
        pb = (ctypes.c_void_p * n)(*[int(x) for x in d_in1])
        po = (ctypes.c_void_p * n)(*[int(x) for x in d_out])
        ts = (ctypes.c_float * n)(*[float(t) for t in timesteps])
        _check(self._L.rife_hip_process_device_batch(self._h, n, pa, pb, ts, po, w, h, stream), "process_device_batch", self._L)

    # ---- measurement / parity taps ----
    def process_batch(self, in0images, in1images, timesteps, outimages=None):
        """n independent pairs in one call (rife_hip_process_batch); returns the list of interpolated frames
        (`outimages`: optional preallocated (h, w, 3) uint8 arrays to write into)."""
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in in0images]
        b = [np.ascontiguousarray(x, dtype=np.uint8) for x in in1images]
        n = len(a)
        if n == 0:
            return []
        h, w, _ = a[0].shape
        if any(x.shape != (h, w, 3) for x in a + b) or len(b) != n or len(timesteps) != n:
            raise ValueError("all frames of a batch must share one size, and there must be one timestep per pair")
        outs = list(outimages) if outimages is not None else [np.empty((h, w, 3), np.uint8) for _ in range(n)]
        if len(outs) != n or any(o.shape != (h, w, 3) or o.dtype != np.uint8 or not o.flags.c_contiguous for o in outs):
            raise ValueError("outimages must be n contiguous (h, w, 3) uint8 arrays")
        pa = (ctypes.c_void_p * n)(*[x.ctypes.data for x in a])
        pb = (ctypes.c_void_p * n)(*[x.ctypes.data for x in b])
        po = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
        ts = (ctypes.c_float * n)(*[float(t) for t in timesteps])
        _check(self._L.rife_hip_process_batch(self._h, n, pa, pb, ts, po, w, h), "process_batch", self._L)
        return outs

    def profile_enable(self, on=True):
        _check(self._L.rife_hip_profile_enable(self._h, int(on)), "profile_enable", self._L)

    def profile_read(self):
        names = ctypes.create_string_buffer(4096)
        ms = np.zeros(64, np.float64); n = np.zeros(64, np.int64); fl = np.zeros(64, np.float64)
        k = self._L.rife_hip_profile_read(self._h, names, 4096, _p(ms), _p(n), _p(fl), 64)
        nm = names.value.decode().split("\n")
        return {nm[i]: dict(ms=float(ms[i]), launches=int(n[i]), flops=float(fl[i])) for i in range(k)}

    def _need_taps(self):
        if not self._taps:
            raise RifeError("the parity taps (include/rife_hip_test.h) live in the test build: create the engine with amd.test_build().RIFE(...)")

    def pool_state(self):
        """(pooled, leased, high_water) of the workspace pool of the host-buffer entry points (test build; include/rife_hip_test.h)."""
        self._need_taps()
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self._L.rife_hip_pool_state(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "pool_state", self._L)
        return a.value, b.value, c.value

    def v4_extract_flow(self, in0image, in1image, timestep, fi, inject=()):
        self._need_taps()
        a = np.ascontiguousarray(in0image, dtype=np.uint8); b = np.ascontiguousarray(in1image, dtype=np.uint8)
        h, w, _ = a.shape
        nc, fh, fw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self._L.rife_hip_v4_flow_dims(self._h, w, h, fi, ctypes.byref(nc), ctypes.byref(fh), ctypes.byref(fw)), "v4_flow_dims", self._L)
        out = np.empty((nc.value, fh.value, fw.value), np.float32)
        inj = [np.ascontiguousarray(f, dtype=np.float32) for f in inject]
        arr = (ctypes.c_void_p * max(1, len(inj)))(*[f.ctypes.data for f in inj])
        _check(self._L.rife_hip_v4_extract_flow(self._h, _p(a), _p(b), w, h, float(timestep), fi, arr, len(inj), _p(out)), "v4_extract_flow", self._L)
        return out


    def v4_tap(self, in0image, in1image, timestep, what, b, inject):
        """what 0 / 1: 12-channel input of IFBlock b (unfused kernel / through the fused stem kernel); 2: blob out0 before the postproc;
        4 / 3: F (4 channels) and M as block b's stem finds them, after k_flow_update / as written by the stem that applies the last update itself;
        5 (b = 3): the block input through the row-streaming stem kernel of the product."""
        self._need_taps()
        a = np.ascontiguousarray(in0image, dtype=np.uint8); bb = np.ascontiguousarray(in1image, dtype=np.uint8)
        h, w, _ = a.shape
        wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32
        s = {1: 4, 2: 2, 3: 1}.get(b, 1)
        out = np.empty((3, hp, wp) if what == 2 else (5, hp, wp) if what in (3, 4) else (12, hp // s, wp // s), np.float32)
        inj = [np.ascontiguousarray(f, dtype=np.float32) for f in inject]
        arr = (ctypes.c_void_p * max(1, len(inj)))(*[f.ctypes.data for f in inj])
        _check(self._L.rife_hip_v4_tap(self._h, _p(a), _p(bb), w, h, float(timestep), int(what), int(b), arr, len(inj), _p(out)), "v4_tap", self._L)
        return out

    def v4_process_injected(self, in0image, in1image, timestep, inject):
        self._need_taps()
        a = np.ascontiguousarray(in0image, dtype=np.uint8); bb = np.ascontiguousarray(in1image, dtype=np.uint8)
        h, w, _ = a.shape
        out = np.empty((h, w, 3), np.uint8)
        inj = [np.ascontiguousarray(f, dtype=np.float32) for f in inject]
        arr = (ctypes.c_void_p * max(1, len(inj)))(*[f.ctypes.data for f in inj])
        _check(self._L.rife_hip_v4_process_injected(self._h, _p(a), _p(bb), w, h, float(timestep), arr, len(inj), _p(out)), "v4_process_injected", self._L)
        return out


# ---- single-kernel entry points (planar CHW float32 numpy arrays): include/rife_hip_test.h, test build ----
def op_conv3x3(x, weight, bias, stride=1, residual=None, slope=None, gpuid=0):
    x = np.ascontiguousarray(x, np.float32); weight = np.ascontiguousarray(weight, np.float32); bias = np.ascontiguousarray(bias, np.float32)
    c, h, w = x.shape
    oc = weight.shape[0]
    out = np.empty((oc, (h - 1) // stride + 1, (w - 1) // stride + 1), np.float32)
    res = None if residual is None else np.ascontiguousarray(residual, np.float32)
    sl = None if slope is None else np.ascontiguousarray(slope, np.float32)
    _check(testlib().rife_hip_op_conv3x3(gpuid, _p(x), c, h, w, _p(weight), _p(bias), oc, stride, _p(res), _p(sl), _p(out)), "op_conv3x3", testlib())
    return out


def op_deconv4x4(x, weight, bias, slope=None, gpuid=0):
    x = np.ascontiguousarray(x, np.float32); weight = np.ascontiguousarray(weight, np.float32); bias = np.ascontiguousarray(bias, np.float32)
    c, h, w = x.shape
    oc = weight.shape[0]
    out = np.empty((oc, 2 * h, 2 * w), np.float32)
    sl = None if slope is None else np.ascontiguousarray(slope, np.float32)
    _check(testlib().rife_hip_op_deconv4x4(gpuid, _p(x), c, h, w, _p(weight), _p(bias), oc, _p(sl), _p(out)), "op_deconv4x4", testlib())
    return out


def op_warp(image, flow, gpuid=0):
    image = np.ascontiguousarray(image, np.float32); flow = np.ascontiguousarray(flow, np.float32)
    c, h, w = image.shape
    out = np.empty_like(image)
    _check(testlib().rife_hip_op_warp(gpuid, _p(image), _p(flow), c, h, w, _p(out)), "op_warp", testlib())
    return out
