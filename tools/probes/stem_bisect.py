"""GPU box: where does the run-to-run instability of the SLP-vectorized fused stem kernels come from?  (round-2 review item 3)

The library is built with -fno-slp-vectorize because stem0_fused_kernel<4, 2> / <2, 2> built WITH the SLP vectorizer (v_pk_mul_f32 / v_pk_add_f32 in
the gather arithmetic) gave different bytes on identical launches.  This script rebuilds those two kernels the old way into assembly, then
assembles VARIANTS of that assembly - wait states (s_nop 7) inserted after chosen instructions - into code objects which the bench library's
determinism probe launches instead of its built-in kernels (rife_hip_probe_set_stem_hsaco).  Stage 1 tries instruction classes, stage 2 delta-
debugs the smallest class that cures the instability down to a 1-minimal set of sites and prints them with their neighbourhood.

    python tools/probes/stem_bisect.py [budget seconds]      -> gpurun_out/stem_bisect.txt
"""
import ctypes, os, re, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from tools import benchlib
T0 = time.time()
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
OUT = "gpurun_out/stem_bisect"
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
log = open("gpurun_out/stem_bisect.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

base_s = os.path.join(OUT, "base.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                       "-o", base_s, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
lines = open(base_s).read().split("\n")
# instruction lines inside the two kernels
sites = []       # (line index, mnemonic)
inside = False
for i, l in enumerate(lines):
    if l.startswith("_ZN4rife18stem0_fused_kernel") and l.split(";")[0].rstrip().endswith(":"):
        inside = True
    elif l.startswith(".Lfunc_end") or l.strip().startswith(".section") or l.strip().startswith(".amdhsa_kernel"):
        inside = False
    elif inside and l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";"):
        m = l.split()[0]
        if m not in ("s_endpgm", "s_branch") and not m.startswith("s_cbranch") and m != "s_barrier" and not m.startswith("s_waitcnt") and m != "s_nop":
            sites.append((i, m))
say("assembly: %d instruction sites in the two kernels; v_pk_*: %d" % (len(sites), sum(1 for _, m in sites if m.startswith("v_pk_"))))

L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
REPS = 10
ntest = 0

def test(after=(), before=(), tag="v", ins="s_nop 7"):
    """assemble the variant with s_nop 7 after / before the given line indices; returns mismatching floats summed over REPS launches of both kernels at 4K"""
    global ntest
    ntest += 1
    A, B = set(after), set(before)
    out = []
    for i, l in enumerate(lines):
        if i in B: out.append("\t" + ins)
        out.append(l)
        if i in A: out.append("\t" + ins)
    s = os.path.join(OUT, tag + ".s"); o = os.path.join(OUT, tag + ".o"); h = os.path.join(OUT, tag + ".hsaco")
    open(s, "w").write("\n".join(out))
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", h])
    L.rife_hip_probe_set_stem_hsaco(h.encode())
    tot = 0; per = []
    for v in (4, 2):
        mm = (ctypes.c_longlong * REPS)()
        rc = L.rife_hip_probe_stem_det(0, v, 3840, 2176, REPS, mm)
        if rc:
            say("probe rc", rc, L.rife_hip_last_error().decode()); return -1
        per.append(sum(mm)); tot += sum(mm)
    return tot, per

def left():
    return BUDGET - (time.time() - T0)

r0 = test(tag="base")
say("unmodified SLP build: mismatching floats over %d launches (S=4, S=2): %s" % (REPS, r0))
r0b = test(tag="base")
say("unmodified SLP build, again: %s" % (r0b,))
# what do the differences look like?  (the probe prints the first mismatching floats of channels 0 and 63 per launch to stderr)
del os.environ["RIFE_HIP_PROBE_QUIET"]
L.rife_hip_probe_set_stem_hsaco(os.path.join(OUT, "base.hsaco").encode())
mm3 = (ctypes.c_longlong * 3)(); L.rife_hip_probe_stem_det(0, 2, 1920, 1088, 3, mm3)
say("S=2 at 1920x1088, 3 launches: %s (values on stderr)" % list(mm3))
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
if len(sys.argv) > 2 and sys.argv[2] == "values-only":
    sys.exit(0)
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
L.rife_hip_probe_set_stem_hsaco(None)
mm = (ctypes.c_longlong * REPS)(); L.rife_hip_probe_stem_det(0, 4, 3840, 2176, REPS, mm)
say("built-in kernel (library flags, no SLP): %s" % (sum(mm),))
if r0[0] == 0 and r0b[0] == 0:
    say("the instability did not reproduce on this box with this toolchain: nothing to bisect")
    sys.exit(0)

# ---- stage W: full waits for memory instead of wait states (is a counted s_waitcnt of the compiler too weak somewhere?)
vm = [i for i, m in sites if m.startswith("global_load")]
ds = [i for i, m in sites if m.startswith("ds_")]
sm = [i for i, m in sites if m.startswith("s_load")]
rw = test(after=vm, tag="w_vm", ins="s_waitcnt vmcnt(0)")
say("s_waitcnt vmcnt(0) after every global_load (%d sites): %s" % (len(vm), rw))
rl = test(after=ds + sm, tag="w_lgkm", ins="s_waitcnt lgkmcnt(0)")
say("s_waitcnt lgkmcnt(0) after every ds_* / s_load (%d sites): %s" % (len(ds) + len(sm), rl))
if rw != -1 and rw[0] == 0 and test(after=vm, tag="w_vm", ins="s_waitcnt vmcnt(0)")[0] == 0:
    say("a full vmcnt wait behind every load cures it -> which load's consumer is under-waited?  delta debugging over the %d loads" % len(vm))
    cur = list(vm)
    def okw(sub):
        r = test(after=sub, tag="ddw", ins="s_waitcnt vmcnt(0)")
        if r != -1 and r[0] == 0: r = test(after=sub, tag="ddw", ins="s_waitcnt vmcnt(0)")
        return r != -1 and r[0] == 0
    n = 2
    while len(cur) >= 2 and left() > 40:
        chunk = max(1, len(cur) // n)
        subsets = [cur[k:k + chunk] for k in range(0, len(cur), chunk)]
        reduced = False
        for sub in subsets:
            if left() < 40: break
            if okw(sub): cur = sub; n = 2; reduced = True; break
        if not reduced:
            for sub in subsets:
                if left() < 40: break
                comp = [x for x in cur if x not in set(sub)]
                if comp and okw(comp): cur = comp; n = max(n - 1, 2); reduced = True; break
        if not reduced:
            if n >= len(cur): break
            n = min(len(cur), 2 * n)
        say("  %d loads left (%d tests, %.0f s)" % (len(cur), ntest, time.time() - T0))
    say("1-minimal (or budget-limited) set of loads whose full wait cures the instability: %d" % len(cur))
    for i in cur[:12]:
        say("---- line %d" % i)
        for k in range(max(0, i - 3), min(len(lines), i + 40)):
            say(("  >> " if k == i else "     ") + lines[k])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "waits-only":
    sys.exit(0)
idx = [i for i, _ in sites]
cls = {
    "every instruction": idx,
    "VALU (v_*)": [i for i, m in sites if m.startswith("v_")],
    "packed fp32 (v_pk_*), after": [i for i, m in sites if m.startswith("v_pk_")],
    "vector memory (global_*)": [i for i, m in sites if m.startswith("global_")],
    "LDS (ds_*)": [i for i, m in sites if m.startswith("ds_")],
    "scalar (s_*)": [i for i, m in sites if m.startswith("s_")],
    "conversions (v_cvt_*)": [i for i, m in sites if m.startswith("v_cvt_")],
    "selects / compares (v_cndmask, v_cmp)": [i for i, m in sites if m.startswith("v_cndmask") or m.startswith("v_cmp")],
    "matrix (v_mfma)": [i for i, m in sites if m.startswith("v_mfma")],
}
res = {}
for name, ix in cls.items():
    if left() < 60: break
    r = test(after=ix, tag="c%d" % len(res))
    res[name] = (r, ix)
    say("s_nop 7 after %-40s (%4d sites): %s" % (name, len(ix), r))
if left() > 60:
    ix = [i for i, m in sites if m.startswith("v_pk_")]
    r = test(before=ix, tag="pkb")
    say("s_nop 7 BEFORE every v_pk_* (%d sites): %s" % (len(ix), r))
    res["packed fp32 (v_pk_*), before"] = (r, ix)
cures = sorted([(len(ix), name) for name, (r, ix) in res.items() if r != -1 and r[0] == 0])
if not cures:
    say("no class of wait states cures it: not a wait-state hazard between adjacent instructions (timing-independent cause or memory side)")
    sys.exit(0)
name = cures[0][1]
say("smallest curing class: %s (%d sites) -> delta debugging" % (name, cures[0][0]))
before_mode = name.endswith("before")
cur = list(res[name][1])
def ok(sub):
    r = test(before=sub, tag="dd") if before_mode else test(after=sub, tag="dd")
    if r != -1 and r[0] == 0:      # confirm a cure once more: the fault is intermittent
        r = test(before=sub, tag="dd") if before_mode else test(after=sub, tag="dd")
    return r != -1 and r[0] == 0
n = 2
while len(cur) >= 2 and left() > 40:
    chunk = max(1, len(cur) // n)
    subsets = [cur[k:k + chunk] for k in range(0, len(cur), chunk)]
    reduced = False
    for sub in subsets:                       # a subset alone cures
        if left() < 40: break
        if ok(sub):
            cur = sub; n = 2; reduced = True; break
    if not reduced:
        for sub in subsets:                   # the complement of a subset still cures
            if left() < 40: break
            comp = [x for x in cur if x not in set(sub)]
            if comp and ok(comp):
                cur = comp; n = max(n - 1, 2); reduced = True; break
    if not reduced:
        if n >= len(cur): break
        n = min(len(cur), 2 * n)
    say("  %d sites left (%d tests, %.0f s)" % (len(cur), ntest, time.time() - T0))
say("1-minimal (or budget-limited) curing set: %d sites, %s them" % (len(cur), "before" if before_mode else "after"))
for i in cur[:24]:
    say("---- line %d" % i)
    for k in range(max(0, i - 6), min(len(lines), i + 7)):
        say(("  >> " if k == i else "     ") + lines[k])
