"""GPU box: packed fp32 or occupancy?  stem0_fused_kernel<2, 2> from the SLP assembly, four ways (S = 2, 1920x1088 and 3840x2176, 40 launches each):
  E1 packed,   128 VGPRs, 65 KB of LDS  -> two workgroups per CU (the product's geometry)
  E2 packed,   128 VGPRs, 100 KB of LDS -> one workgroup per CU, same registers
  E3 packed,   136 VGPRs                -> one workgroup per CU (three waves per SIMD allowed, a 512-thread workgroup needs two)
  E4 unpacked IN PLACE (no scratch registers), 128 VGPRs, 65 KB -> two workgroups per CU, scalar fp32 instead of v_pk_*
    python tools/probes/stem_occupancy.py      -> gpurun_out/stem_occupancy.txt"""
import ctypes, os, re, sys
sys.path.insert(0, os.getcwd())
from tools.probes import stem_unpack as U
from tools import benchlib
log = open("gpurun_out/stem_occupancy.txt", "w")
def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()
lines = U.build_asm()
def bump(ls):
    out = []
    for l in ls:
        l2 = re.sub(r"(\.amdhsa_next_free_vgpr)\s+128\b", r"\1 136", l); l2 = re.sub(r"(\.vgpr_count:\s+)128\b", r"\g<1>136", l2); l2 = re.sub(r"(\.amdhsa_accum_offset)\s+128\b", r"\1 136", l2)
        out.append(l2)
    return out
inpl = []; n_ok = n_kept = 0
for l in lines:
    u = U.unpack_in_place(l) if l.startswith("\tv_pk_") else None
    if u is None: inpl.append(l)
    elif u is False: inpl.append(l); n_kept += 1
    else: inpl.extend(u); n_ok += 1
say("in-place unpacking: %d instructions rewritten, %d kept packed (both orders would clobber a source)" % (n_ok, n_kept))
h_packed = U.assemble(lines, "o_packed"); h_136 = U.assemble(bump(lines), "o_packed136"); h_inpl = U.assemble(inpl, "o_inplace")
L = benchlib.lib()
L.rife_hip_probe_stem_det.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
L.rife_hip_probe_set_stem_hsaco.argtypes = [ctypes.c_char_p]
L.rife_hip_probe_last_extra.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
os.environ["RIFE_HIP_PROBE_QUIET"] = "1"
N = 40
for rep in range(2):
    for tag, h, lds in (("E1 packed, 128 VGPRs, 2 workgroups / CU", h_packed, None), ("E2 packed, 128 VGPRs, 100 KB LDS: 1 / CU", h_packed, 100 * 1024),
                        ("E3 packed, 136 VGPRs: 1 / CU", h_136, None), ("E4 scalar in place, 128 VGPRs, 2 / CU", h_inpl, None)):
        if lds: os.environ["RIFE_HIP_PROBE_LDS"] = str(lds)
        else: os.environ.pop("RIFE_HIP_PROBE_LDS", None)
        L.rife_hip_probe_set_stem_hsaco(h.encode())
        for wp, hp in ((1920, 1088), (3840, 2176)):
            mm = (ctypes.c_longlong * N)()
            rc = L.rife_hip_probe_stem_det(0, 2, wp, hp, N, mm)
            ex = (ctypes.c_longlong * 3)(); L.rife_hip_probe_last_extra(ex)
            say("%-44s %dx%d rc=%d: %2d of %d launches differ from launch 0 (%d floats); launch 0 vs the library kernel %d floats" % (tag, wp, hp, rc, sum(1 for v in mm if v), N, sum(mm), ex[0]))
os.environ.pop("RIFE_HIP_PROBE_LDS", None)
