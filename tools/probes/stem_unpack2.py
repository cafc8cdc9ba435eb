"""GPU box: WHICH packed-fp32 instructions of the SLP build of stem0_fused_kernel<2, 2> misbehave?  (tools/probes/stem_unpack.py: unpacking all of them
cures the instability.)  Classes by opcode / operand modifiers first, then delta debugging inside the smallest curing class.
    python tools/probes/stem_unpack2.py [budget seconds]      -> gpurun_out/stem_unpack2.txt"""
import ctypes, os, re, sys, time
sys.path.insert(0, os.getcwd())
from tools.probes import stem_unpack as U
from tools import benchlib
T0 = time.time(); BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
log = open("gpurun_out/stem_unpack2.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
lines = U.build_asm()
# sites = packed fp32 instructions inside stem0_fused_kernel<2, 2, 0>
sites = []; inside = False
for i, l in enumerate(lines):
    if l.startswith("_ZN4rife18stem0_fused_kernelILi2ELi2ELi0") and l.split(";")[0].rstrip().endswith(":"): inside = True
    elif l.startswith(".Lfunc_end"): inside = False
    elif inside and (l.startswith("\tv_pk_mul_f32") or l.startswith("\tv_pk_add_f32")): sites.append(i)
say("%d packed fp32 instructions in stem0_fused_kernel<2, 2, 0>" % len(sites))
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
N = 16; ntest = 0
def test(unp):
    """unpack the given sites (line indices); returns floats differing from launch 0 over N launches of S = 2 at 1920x1088"""
    global ntest; ntest += 1
    S = set(unp); out = []
    for i, l in enumerate(lines):
        u = U.unpack(l) if i in S else None
        if u is None:
            l2 = re.sub(r"(\.amdhsa_next_free_vgpr)\s+128\b", r"\1 136", l); l2 = re.sub(r"(\.vgpr_count:\s+)128\b", r"\g<1>136", l2); l2 = re.sub(r"(\.amdhsa_accum_offset)\s+128\b", r"\1 136", l2)
            out.append(l2)
        else: out.extend(u)
    h = U.assemble(out, "dd")
    L.rife_hip_probe_set_stem_hsaco(h.encode())
    mm = (ctypes.c_longlong * N)()
    rc = L.rife_hip_probe_stem_det(0, 2, 1920, 1088, N, mm)
    return -1 if rc else sum(mm)
def left(): return BUDGET - (time.time() - T0)
say("none unpacked: %d   all unpacked: %d" % (test([]), test(sites)))
def has(i, pat): return re.search(pat, lines[i]) is not None
cls = {
    "v_pk_mul_f32": [i for i in sites if "v_pk_mul" in lines[i]],
    "v_pk_add_f32": [i for i in sites if "v_pk_add" in lines[i]],
    "with op_sel / op_sel_hi": [i for i in sites if has(i, r"op_sel")],
    "with neg_lo / neg_hi": [i for i in sites if has(i, r"neg_")],
    "without any modifier": [i for i in sites if not has(i, r"op_sel|neg_")],
    "with an inline constant operand": [i for i in sites if has(i, r",\s*-?\d+(\.\d+)?(\s|$)")],
    "with an SGPR operand": [i for i in sites if has(i, r"s\[\d+:\d+\]")],
    "destination overlaps a source": [i for i in sites if (lambda m: m and (m.group(1) in m.group(2)))(re.match(r"\tv_pk_\w+\s+(v\[\d+:\d+\]),(.*)", lines[i]))],
}
res = {}
for name, ix in cls.items():
    if left() < 60 or not ix: continue
    r = test(ix); res[name] = (r, ix)
    say("unpack only %-34s (%4d sites): %d floats differ" % (name, len(ix), r))
cures = sorted((len(ix), name) for name, (r, ix) in res.items() if r == 0)
if not cures:
    say("no single class cures it"); cur = list(sites)
else:
    say("smallest curing class: %s" % cures[0][1]); cur = list(res[cures[0][1]][1])
def ok(sub):
    r = test(sub)
    if r == 0: r = test(sub)
    return r == 0
n = 2
while len(cur) >= 2 and left() > 30:
    chunk = max(1, len(cur) // n)
    subsets = [cur[k:k + chunk] for k in range(0, len(cur), chunk)]
    reduced = False
    for sub in subsets:
        if left() < 30: break
        if ok(sub): cur = sub; n = 2; reduced = True; break
    if not reduced:
        for sub in subsets:
            if left() < 30: break
            comp = [x for x in cur if x not in set(sub)]
            if comp and ok(comp): cur = comp; n = max(n - 1, 2); reduced = True; break
    if not reduced:
        if n >= len(cur): break
        n = min(len(cur), 2 * n)
    say("  %d sites left (%d tests, %.0f s)" % (len(cur), ntest, time.time() - T0))
say("1-minimal (or budget-limited) set of packed instructions whose unpacking cures the kernel: %d" % len(cur))
for i in cur[:16]:
    say("---- line %d" % i)
    for k in range(max(0, i - 8), min(len(lines), i + 5)):
        say(("  >> " if k == i else "     ") + lines[k])
