"""GPU box: which STATE does the SLP build of the fused stem kernels depend on?  (after tools/probes/stem_bisect.py and stem_poison.py: not wait states,
not memory waits, not register residue)  1. are the input buffers intact after the launches (an out-of-bounds store)?  2. does the result follow
what a scrub kernel leaves in the CUs' LDS between the launches (a read of LDS the kernel did not write)?
    python tools/probes/stem_state.py      -> gpurun_out/stem_state.txt"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
OUT = "gpurun_out/stem_state"
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
log = open("gpurun_out/stem_state.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
s_ = os.path.join(OUT, "slp.s"); o_ = os.path.join(OUT, "slp.o"); h_ = os.path.join(OUT, "slp.hsaco")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", s_, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_, "-o", o_])
subprocess.check_call([LLVM + "/ld.lld", "-shared", o_, "-o", h_])
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
L.rife_hip_probe_set_stem_hsaco(h_.encode())
N = 12
def run(S, label):
    mm = (ctypes.c_longlong * N)()
    rc = L.rife_hip_probe_stem_det(0, S, 1920, 1088, N, mm)
    ex = (ctypes.c_longlong * 3)(); L.rife_hip_probe_last_extra(ex)
    say("%-34s S=%d rc=%d differing floats vs launch 0: %s | launch 0 vs library kernel %d, NaN %d, input buffers modified %d" % (label, S, rc, list(mm), ex[0], ex[1], ex[2]))
for S in (2, 4):
    os.environ.pop("RIFE_HIP_PROBE_SCRUB", None)
    run(S, "no scrub")
    for pat in ("0", "7fc00000", "3f800000", "alt"):
        os.environ["RIFE_HIP_PROBE_SCRUB"] = pat
        run(S, "LDS scrubbed with %s before each" % pat)
os.environ.pop("RIFE_HIP_PROBE_SCRUB", None)
