"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/_ref/libref_rife.so = the reference's OWN src/rife.cpp + src/warp.cpp,
compiled unmodified against the ncnn look-alike of oracle/refbuild/ (recipe: oracle/refbuild/Makefile).  gpuid is -1 throughout: the
reference's `-g -1` CPU path (RIFE::process_cpu / process_v4_cpu).  It exists to PIN the restated oracle (oracle/rife_oracle.cpp):
tests/test_ref_build.py compares the two bit for bit.  /root/reference is needed to BUILD the library (this container); the built .so travels
to the GPU box inside oracle/_ref/ and is only ever loaded there.  Never imported by the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libref_rife.so")
REFERENCE = os.environ.get("RIFE_REFERENCE_DIR", "/root/reference")
_LIB = None


def can_build():
    return os.path.exists(os.path.join(REFERENCE, "src", "rife.cpp"))


def build():
    """make -C oracle/refbuild when the reference sources are present; otherwise the prebuilt library (or None)."""
    if can_build():
        os.makedirs(os.path.join(_HERE, "_ref"), exist_ok=True)
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "refbuild"), "-j4", "REF=" + REFERENCE])
    return SO if os.path.exists(SO) else None


def available():
    return os.path.exists(SO) or can_build()


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        if so is None:
            raise RuntimeError("oracle/_ref/libref_rife.so is absent and %s is not there to build it from" % REFERENCE)
        L = ctypes.CDLL(so)
        L.ref_rife_create.restype = ctypes.c_void_p
        L.ref_rife_create.argtypes = [ctypes.c_int] * 6
        L.ref_rife_load.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.ref_rife_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        L.ref_rife_destroy.argtypes = [ctypes.c_void_p]
        _LIB = L
    return _LIB


class RefRIFE:
    """The reference's `RIFE(-1, tta_mode, tta_temporal_mode, uhd_mode, num_threads, rife_v2, rife_v4)` (src/rife.h:14), compiled from its own source."""

    def __init__(self, tta_mode=False, tta_temporal_mode=False, uhd_mode=False, num_threads=4, rife_v2=False, rife_v4=False):
        self.h = lib().ref_rife_create(int(tta_mode), int(tta_temporal_mode), int(uhd_mode), int(num_threads), int(rife_v2), int(rife_v4))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_rife_destroy(self.h)
            self.h = None

    def load(self, modeldir):
        for f in ("flownet.param", "flownet.bin"):      # the reference ignores a failed fopen (src/rife.cpp:119-120) and would run an empty net
            if not os.path.exists(os.path.join(modeldir, f)):
                raise FileNotFoundError(os.path.join(modeldir, f))
        return lib().ref_rife_load(self.h, modeldir.encode())

    def process(self, in0, in1, timestep):
        in0 = np.ascontiguousarray(in0, dtype=np.uint8)
        in1 = np.ascontiguousarray(in1, dtype=np.uint8)
        h, w, c = in0.shape
        assert c == 3 and in1.shape == in0.shape
        out = np.empty_like(in0)
        rc = lib().ref_rife_process(self.h, in0.ctypes.data_as(ctypes.c_void_p), in1.ctypes.data_as(ctypes.c_void_p), w, h, float(timestep),
                                    out.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError("ref_rife_process failed: %d" % rc)
        return out
