"""GPU box: the LIBRARY-flags (no SLP) assembly of stem0_fused_kernel<2, 2>, whose 115 VGPRs are allocated as 120, with its register allocation
raised to 128 in the kernel descriptor (four waves x 128 = the whole 512-entry register file of a SIMD, like the SLP build): stable or not?
    python tools/probes/stem_alloc128.py      -> gpurun_out/stem_alloc128.txt"""
import ctypes, os, re, subprocess, sys
sys.path.insert(0, os.getcwd())
from tools.probes import stem_unpack as U
from tools import benchlib
log = open("gpurun_out/stem_alloc128.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
s_ = os.path.join(U.OUT, "noslp.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize", "-S", "--cuda-device-only", "-o", s_, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
lines = open(s_).read().split("\n")
# descriptor values of the stem kernels
vals = sorted(set(re.findall(r"\.amdhsa_next_free_vgpr\s+(\d+)", "\n".join(l for l in lines))))
say("next_free_vgpr values in the translation unit:", vals)
def with_alloc(n):
    out = []; inside = False
    for l in lines:
        if ".amdhsa_kernel _ZN4rife18stem0_fused_kernel" in l: inside = True
        if ".end_amdhsa_kernel" in l: inside = False
        if inside:
            l = re.sub(r"(\.amdhsa_next_free_vgpr)\s+\d+", r"\1 %d" % n, l); l = re.sub(r"(\.amdhsa_accum_offset)\s+\d+", r"\1 %d" % n, l)
        out.append(l)
    return out
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
N = 40
for rep in range(2):
    for tag, ls in (("library flags as compiled", lines), ("library flags, allocation 120", with_alloc(120)), ("library flags, allocation 128", with_alloc(128)), ("library flags, allocation 124 (-> 128)", with_alloc(124))):
        h = U.assemble(ls, "a_" + re.sub(r"\W+", "_", tag))
        L.rife_hip_probe_set_stem_hsaco(h.encode())
        for S, wp, hp in ((2, 1920, 1088), (2, 3840, 2176), (4, 3840, 2176)):
            mm = (ctypes.c_longlong * N)()
            rc = L.rife_hip_probe_stem_det(0, S, wp, hp, N, mm)
            ex = (ctypes.c_longlong * 3)(); L.rife_hip_probe_last_extra(ex)
            say("%-40s S=%d %dx%d rc=%d: %2d of %d launches differ from launch 0 (%d floats); launch 0 vs the library kernel %d floats" % (tag, S, wp, hp, rc, sum(1 for v in mm if v), N, sum(mm), ex[0]))
