"""Print the engine's structural hash (rife_hip_param_hash) of a blob of a .param file.
Used to (re)derive the constants in rife-ncnn-vulkan_amd/csrc/model_hashes.h."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def param_hash(param_path, blob, lib=None):
    lib = lib or ctypes.CDLL(os.path.join(ROOT, "rife-ncnn-vulkan_amd", "librife_hip.so"))
    lib.rife_hip_param_hash.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    out = ctypes.c_uint64()
    rc = lib.rife_hip_param_hash(param_path.encode(), blob.encode(), ctypes.byref(out))
    if rc:
        raise RuntimeError("rife_hip_param_hash failed: %d" % rc)
    return out.value


if __name__ == "__main__":
    print("0x%016xull" % param_hash(sys.argv[1], sys.argv[2]))
