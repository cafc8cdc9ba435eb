"""GPU box: are the packed-fp32 INSTRUCTIONS the culprit of the SLP build's instability, or something else in that build's instruction stream?

Takes the compiler's SLP assembly of the fused stem kernels (tools/probes/stem_tu.hip, round-2 flags) and rewrites every v_pk_mul_f32 /
v_pk_add_f32 into two scalar v_mul_f32 / v_add_f32 (through two scratch registers v128 / v129, so that overlapping operands cannot
interfere) - same registers, same schedule, same everything else - and runs the determinism probe on both.
    python tools/probes/stem_unpack.py      -> gpurun_out/stem_unpack.txt"""
import ctypes, os, re, subprocess, sys
sys.path.insert(0, os.getcwd())
OUT = "gpurun_out/stem_unpack"
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
log = open("gpurun_out/stem_unpack.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

def build_asm():
    s_ = os.path.join(OUT, "slp.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", s_, "tools/probes/stem_tu.hip"], stderr=subprocess.DEVNULL)
    return open(s_).read().split("\n")

MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[(\d),(\d)\]")
def half(op, sel):
    """dword `sel` (0 low / 1 high) of a packed operand"""
    op = op.strip()
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", op)
    if m:
        return "%s%d" % (m.group(1), int(m.group(2)) + sel)
    return op      # inline constant / literal: the same value in both halves

def unpack(line):
    """v_pk_{mul,add}_f32 vD, A, B mods -> scalar pair; None if the line is not such an instruction"""
    t = line.strip()
    m = re.match(r"^v_pk_(mul|add)_f32\s+(.*)$", t)
    if not m:
        return None
    opn = m.group(1); rest = m.group(2)
    mods = {"op_sel": (0, 0), "op_sel_hi": (1, 1), "neg_lo": (0, 0), "neg_hi": (0, 0)}
    for mm in MOD.finditer(rest):
        mods[mm.group(1)] = (int(mm.group(2)), int(mm.group(3)))
    ops = MOD.sub("", rest).strip().rstrip(",").strip()
    # split the three operands at top-level commas
    parts = [p.strip() for p in re.split(r",\s*(?![^\[]*\])", ops) if p.strip()]
    assert len(parts) == 3, line
    d, a, b = parts
    dm = re.match(r"^v\[(\d+):(\d+)\]$", d); assert dm, line
    dlo, dhi = int(dm.group(1)), int(dm.group(1)) + 1
    def src(op, sel, neg):
        h = half(op, sel)
        return ("-" + h) if neg else h
    out = []
    out.append("\tv_%s_f32_e64 v128, %s, %s" % (opn, src(a, mods["op_sel"][0], mods["neg_lo"][0]), src(b, mods["op_sel"][1], mods["neg_lo"][1])))
    out.append("\tv_%s_f32_e64 v129, %s, %s" % (opn, src(a, mods["op_sel_hi"][0], mods["neg_hi"][0]), src(b, mods["op_sel_hi"][1], mods["neg_hi"][1])))
    out.append("\tv_mov_b32 v%d, v128" % dlo)
    out.append("\tv_mov_b32 v%d, v129" % dhi)
    return out

def unpack_in_place(line):
    """the same without scratch registers (the kernel keeps its 128 VGPRs): the half whose destination is not read by the other half goes
    second; None if both orders would clobber a source (the instruction stays packed)"""
    u = unpack(line)
    if u is None:
        return None
    lo, hi, mlo, mhi = u
    dlo = mlo.split()[1].rstrip(","); dhi = mhi.split()[1].rstrip(",")
    def reads(ins, reg):
        return re.search(r"[\s,-]%s\b" % reg, " " + ins.split(None, 2)[2]) is not None
    a = lo.replace("v128", dlo); b = hi.replace("v129", dhi)
    if not reads(b, dlo): return [a, b]
    if not reads(a, dhi): return [b, a]
    return False

def transform(lines):
    out = []; n = 0
    for l in lines:
        u = unpack(l) if l.startswith("\tv_pk_") else None
        if u is None:
            # the two scratch registers: raise the kernels' VGPR budget 128 -> 136
            l2 = re.sub(r"(\.amdhsa_next_free_vgpr)\s+128\b", r"\1 136", l)
            l2 = re.sub(r"(\.vgpr_count:\s+)128\b", r"\g<1>136", l2)
            l2 = re.sub(r"(\.amdhsa_accum_offset)\s+128\b", r"\1 136", l2)
            out.append(l2)
        else:
            out.extend(u); n += 1
    return out, n

def assemble(lines, tag):
    s = os.path.join(OUT, tag + ".s"); o = os.path.join(OUT, tag + ".o"); h = os.path.join(OUT, tag + ".hsaco")
    open(s, "w").write("\n".join(lines))
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", h])
    return h

if __name__ == "__main__":
    lines = build_asm()
    un, n = transform(lines)
    left = sum(1 for l in un if l.startswith("\tv_pk_") and "_f32" in l)
    say("rewrote %d packed fp32 instructions into scalar pairs; packed fp32 instructions left: %d" % (n, left))
    h0 = assemble(lines, "packed"); h1 = assemble(un, "unpacked")
    if len(sys.argv) > 1 and sys.argv[1] == "dry":
        sys.exit(0)
    from tools import benchlib
    L = benchlib.lib()
    L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
    L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
    L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
    os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
    N = 40
    for tag, h in (("packed (the compiler's SLP assembly)", h0), ("unpacked (same assembly, scalar pairs)", h1), ("packed again", h0), ("unpacked again", h1)):
        L.rife_hip_probe_set_stem_hsaco(h.encode())
        for S, wp, hp in ((2, 1920, 1088), (4, 3840, 2176), (2, 3840, 2176)):
            mm = (ctypes.c_longlong * N)()
            rc = L.rife_hip_probe_stem_det(0, S, wp, hp, N, mm)
            ex = (ctypes.c_longlong * 3)(); L.rife_hip_probe_last_extra(ex)
            say("%-42s S=%d %dx%d rc=%d: %d of %d launches differ from launch 0 (%d floats); launch 0 vs the library kernel %d floats, %d NaN" % (tag, S, wp, hp, rc, sum(1 for v in mm if v), N, sum(mm), ex[0], ex[1]))
