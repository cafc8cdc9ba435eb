"""GPU box: run-to-run stability of the fused stem kernels of blocks 1 / 2 under BOTH sets of compiler flags, old and new staging code.

    python tools/probes/stem_det_both.py [launches]      -> gpurun_out/stem_det_both.txt
Four builds of stem0_fused_kernel<4, 2> and <2, 2> (csrc/stem_fused.h), `launches` (default 200) identical launches each at 3840x2176 and
1920x1088, floats that differ from launch 0 summed over the launches, and launch 0 against the library's kernel:
  library flags (-fno-slp-vectorize), round-3 staging     = the product
  library flags,                      round-2 staging     (ABL 4096: second halo pixel stored under a lane-divergent branch)
  SLP vectorizer on (the round-2 flags), round-3 staging  <- must be stable too: the source fix
  SLP vectorizer on,                  round-2 staging     <- the instability of DESIGN.md (d)-8
The SLP builds are compiled here from tools/probes/stem_tu.hip and loaded as a code object (rife_hip_probe_set_stem_hsaco)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd())
from tools import benchlib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OUT = "gpurun_out/stem_det_both"
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
log = open("gpurun_out/stem_det_both.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
s_ = os.path.join(OUT, "slp.s"); o_ = os.path.join(OUT, "slp.o"); h_ = os.path.join(OUT, "slp.hsaco")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", s_, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_, "-o", o_])
subprocess.check_call([LLVM + "/ld.lld", "-shared", o_, "-o", h_])
say("SLP build: %d packed fp32 instructions in the assembly" % sum(1 for l in open(s_) if l.startswith("\tv_pk_") and "_f32" in l))
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
bad_product = 0
for flags, hs in (("library flags", None), ("SLP vectorizer on", h_)):
    for staging, abl in (("round-3 staging", 0), ("round-2 staging", 4096)):
        L.rife_hip_probe_set_stem_hsaco(hs.encode() if hs else None)
        for S in (4, 2):
            for wp, hp, n in ((3840, 2176, min(N, 60)), (1920, 1088, N)):       # 4K: 33 MB of output per launch
                mm = (ctypes.c_longlong * n)()
                rc = L.rife_hip_probe_stem_det(0, S + 16 * abl, wp, hp, n, mm)
                ex = (ctypes.c_longlong * 2)()
                if hs: L.rife_hip_probe_last_extra(ex)
                tot = sum(mm); nbad = sum(1 for v in mm if v)
                say("%-18s %-16s S=%d %dx%d: rc=%d %d launches, %d differ from launch 0 (%d floats in total)%s" % (flags, staging, S, wp, hp, rc, n, nbad, tot,
                    "; launch 0 vs the library kernel: %d floats, %d NaN" % (ex[0], ex[1]) if hs else ""))
                if abl == 0 and (tot or rc): bad_product += 1
say("round-3 staging: %s" % ("stable under both flag sets" if not bad_product else "%d UNSTABLE cases" % bad_product))
sys.exit(1 if bad_product else 0)
