"""GPU box: does the SLP-vectorized fused stem kernel read registers it never wrote?  (follow-up of tools/probes/stem_bisect.py)

tools/probes/stem_bisect.py showed that no wait state anywhere cures the run-to-run instability of the SLP build, that its differences are large (1 - 3 % of
the value, whole groups of 16 halo pixels) and that launches 1, 2, ... of a burst agree with each other while launch 0 differs: the result depends
on what the previous kernel left behind in the CU (registers or LDS), not on timing.  This script assembles variants of the compiler's assembly
whose first instructions fill VGPRs v1 .. v127 (v0 = work-item id) with a poison value - a quiet NaN, or a large number - and compares launch 0
with the library's own (non-SLP) kernel on the same inputs: if poison reaches the output, the kernel consumes an uninitialised register; a
bisection over the register set names it, and the first read of that register in the assembly is printed.

    python tools/probes/stem_poison.py [budget seconds]      -> gpurun_out/stem_poison.txt
"""
import ctypes, os, re, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from tools import benchlib
T0 = time.time()
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
OUT = "gpurun_out/stem_poison"
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
log = open("gpurun_out/stem_poison.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

base_s = os.path.join(OUT, "base.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                       "-o", base_s, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
lines = open(base_s).read().split("\n")
entry = {}      # kernel variant -> line index of its label
for i, l in enumerate(lines):
    if l.startswith("_ZN4rife18stem0_fused_kernelILi4ELi2ELi0") and l.split(";")[0].rstrip().endswith(":"): entry[4] = i
    if l.startswith("_ZN4rife18stem0_fused_kernelILi2ELi2ELi0") and l.split(";")[0].rstrip().endswith(":"): entry[2] = i
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
REPS = 4

def test(regs, value, tag, sregs=()):
    """poison VGPRs `regs` (and SGPRs `sregs`) at the entry of both kernels; returns per kernel (launch-to-launch mismatches, launch 0 vs built-in, NaNs)"""
    ins = ["\tv_mov_b32 v%d, 0x%08x" % (r, value) for r in regs] + ["\ts_mov_b32 s%d, 0x%08x" % (r, value) for r in sregs]
    out = []
    for i, l in enumerate(lines):
        out.append(l)
        if i in (entry[4], entry[2]): out.extend(ins)
    s = os.path.join(OUT, tag + ".s"); o = os.path.join(OUT, tag + ".o"); h = os.path.join(OUT, tag + ".hsaco")
    open(s, "w").write("\n".join(out))
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", h])
    L.rife_hip_probe_set_stem_hsaco(h.encode())
    res = {}
    for v in (4, 2):
        mm = (ctypes.c_longlong * REPS)()
        rc = L.rife_hip_probe_stem_det(0, v, 1920, 1088, REPS, mm)
        ex = (ctypes.c_longlong * 2)(); L.rife_hip_probe_last_extra(ex)
        res[v] = (rc, list(mm), ex[0], ex[1])
    return res

def left():
    return BUDGET - (time.time() - T0)

NAN, BIG = 0x7fc00000, 0x7149f2ca      # quiet NaN, 1e30
say("per kernel S: (rc, mismatches of launches 0..%d vs launch 0, launch 0 vs the library's non-SLP kernel, NaNs in launch 0)" % (REPS - 1))
say("unmodified SLP build          :", test([], 0, "p_base"))
say("v1..v127 = 0                  :", test(range(1, 128), 0, "p_zero"))
say("v1..v127 = NaN                :", test(range(1, 128), NAN, "p_nan"))
say("v1..v127 = 1e30               :", test(range(1, 128), BIG, "p_big"))
# SGPRs above the preloaded ones (kernarg pointer s[0:1], workgroup id s2 .. per the kernel descriptor): s8 .. s99
say("s16..s95 = NaN bits           :", test([], NAN, "p_snan", sregs=range(16, 96)))
r = test(range(1, 128), NAN, "p_nan")
bad = [v for v in (4, 2) if r[v][3] > 0 or r[v][2] > 0]
if not bad:
    say("poisoned registers never reach the output: the kernels do not consume uninitialised VGPRs (look at LDS / memory next)")
    sys.exit(0)
# bisect the register set on the kernel that shows it
k = bad[0]
cur = list(range(1, 128))
def hit(regs):
    rr = test(regs, NAN, "p_dd")
    return rr[k][3] > 0 or rr[k][2] > 0
while len(cur) > 1 and left() > 30:
    half = cur[:len(cur) // 2]
    if hit(half): cur = half
    elif hit(cur[len(cur) // 2:]): cur = cur[len(cur) // 2:]
    else:
        say("  neither half of %d registers alone: several registers involved; keeping the set" % len(cur)); break
    say("  narrowed to v%d .. v%d (%d registers), %.0f s" % (cur[0], cur[-1], len(cur), time.time() - T0))
say("kernel S=%d consumes poison placed in: %s" % (k, ", ".join("v%d" % r for r in cur[:16])))
# first instructions that READ such a register before any write, in program order from the kernel entry (linear scan: good enough for the prologue)
for rg in cur[:4]:
    pat = re.compile(r"\bv%d\b|v\[(\d+):(\d+)\]" % rg)
    n = 0
    for i in range(entry[k] + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"): break
        if not l.startswith("\t") or l.strip().startswith((".", ";")): continue
        ops = l.split(None, 1)[1] if len(l.split(None, 1)) > 1 else ""
        use = False
        for m in pat.finditer(ops):
            if m.group(1) is None or int(m.group(1)) <= rg <= int(m.group(2)): use = True
        if use:
            say("   v%d line %d: %s" % (rg, i, l.strip()))
            n += 1
            if n >= 6: break
