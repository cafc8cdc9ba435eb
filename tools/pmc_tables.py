"""Turn the four per-kernel PMC summaries of tools/profile_round.sh into the files profiles/<tag>/ keeps:

    python tools/pmc_tables.py gpurun_out/profile_r4 profiles/r4 [pairs] [workload]      (workload: 4k (default), 1080p, v23_1080p, 4k_tta)

  pmc_all_kernels_4k_{mfma_lds,wait,fetch,write}.txt   copies of the per-kernel summaries (tools/pmc_summary.py output)
  pmc_trunk_kernels_4k.txt                            the two persistent trunk kernels: counters of all passes + derived HBM bytes and matrix-pipe busy
  bandwidth_kernels_4k.txt                            per kernel: mean duration, HBM MB per launch, TB/s, share of the pair's kernel time
  pmc_4k.json                                         the dominant kernel's HBM bytes per launch in the form bench.py reads (roofline.traffic)
HBM bytes per launch = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (KiB counters; the x2 on FETCH_SIZE is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md).  Matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import json
import os
import shutil
import sys

PASSES = {"mfma_lds": "SQ_VALU_MFMA_BUSY_CYCLES", "wait": "SQ_WAIT_ANY", "fetch": "FETCH_SIZE", "write": "WRITE_SIZE"}
# the dominant kernel per workload: the first of these symbols that has dispatches (round 6: conv_rs2_kernel, two block-3 trunk layers per launch; round 3 - 5:
# conv_rs_kernel, which still serves segments too short for conv_rs2 and the summaries committed by earlier rounds; round 2: "conv_t64_kernel<3, 2>")
DOMINANT = ("conv_rs2_kernel", "conv_rs_kernel")
DOMINANT_OF = {"4k": DOMINANT, "1080p": DOMINANT, "4k_tta": DOMINANT, "v23_1080p": ("conv_h2_kernel<3, 9, 0>",)}
TRUNKS = ("conv_rs2_kernel", "conv_rs_kernel", "conv_t64_kernel", "conv_row_kernel", "conv_ks_kernel", "conv_h2_kernel", "conv_h2b_kernel")


def parse(path):
    d, cur = {}, None
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        if not line.startswith(" "):
            cur = line.strip()
            d[cur] = {}
        else:
            k, v = line.split()
            d[cur][k] = float(v)
    return d


def main(src, dst, pairs=3, wl="4k"):
    os.makedirs(dst, exist_ok=True)
    tabs = {}
    sfx = "" if wl == "4k" else "_" + wl
    dominant_names = DOMINANT_OF.get(wl, DOMINANT)
    for short, first in PASSES.items():
        s = os.path.join(src, "pmc_%s%s_all.txt" % (first, sfx))
        shutil.copy(s, os.path.join(dst, "pmc_all_kernels_%s_%s.txt" % (wl, short)))
        tabs[short] = parse(s)
    f, w, m = tabs["fetch"], tabs["write"], tabs["mfma_lds"]
    dominant = next((d for d in dominant_names if any(d in k for k in f)), dominant_names[0])
    # ---- trunk kernels
    with open(os.path.join(dst, "pmc_trunk_kernels_%s.txt" % wl), "w") as o:
        o.write("# bash tools/profile_round.sh: rocprofv3 --kernel-trace --pmc <pass> -- python tools/prof_run.py --workload %s --pairs %d   (MI355X)\n" % (wl.replace("_", "-"), pairs))
        o.write("# per-dispatch means for the trunk kernels (rife-v4.6: trunk_b3 = conv_rs2_kernel<0> (two layers per launch; conv_rs_kernel<0> before round 6), 64 channels; trunk_b2 = conv_t64_kernel<2, 3> / conv_row_kernel<96>, 96 channels; coarse blocks conv_row / conv_h2b), 4 separate PMC passes\n")
        for short in PASSES:
            for k, c in tabs[short].items():
                if any(t in k for t in TRUNKS):
                    o.write(k + "\n")
                    for n, v in sorted(c.items()):
                        o.write("    %-32s %18.1f\n" % (n, v))
        o.write("# derived\n")
        for k in f:
            if any(t in k for t in TRUNKS) and k in w and k in m:
                hb = f[k]["FETCH_SIZE"] * 2 * 1024 + w[k]["WRITE_SIZE"] * 1024
                busy = m[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * m[k]["GRBM_GUI_ACTIVE"] / 8)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                o.write("%s: HBM bytes per launch %.1f MB (fetch %.1f + write %.1f), %.2f TB/s over the mean %.1f us; matrix pipe busy %.1f %%; LDS bank conflicts %.0f\n" % (
                    k.split("(")[0], hb / 1e6, f[k]["FETCH_SIZE"] * 2 * 1024 / 1e6, w[k]["WRITE_SIZE"] * 1024 / 1e6, hb / f[k]["_avg_ns"] / 1e3, f[k]["_avg_ns"] / 1e3,
                    100 * busy, m[k].get("SQ_LDS_BANK_CONFLICT", 0)))
                if dominant in k:
                    json.dump({"hbm_bytes_per_launch": int(hb), "kernel": k.split("(")[0],
                               "source": "%s/pmc_trunk_kernels_%s.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md gfx950 correction" % (dst, wl)},
                              open(os.path.join(dst, "pmc_%s.json" % wl), "w"))
    # ---- bandwidth table
    rows = []
    for k in f:
        if k in w and not k.startswith("__amd_rocclr"):       # runtime copy / fill kernels: weight uploads and workspace memsets of the engine's start
            hb = f[k].get("FETCH_SIZE", 0) * 2 * 1024 + w[k].get("WRITE_SIZE", 0) * 1024
            ns = (f[k]["_avg_ns"] + w[k]["_avg_ns"]) / 2
            rows.append((ns * f[k]["_dispatches"] / pairs, k, ns, hb, f[k]["_dispatches"] / pairs))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    with open(os.path.join(dst, "bandwidth_kernels_%s.txt" % wl), "w") as o:
        o.write("# workload " + wl.replace("_", "-") + " of bench.py / tools/prof_run.py\n")
        o.write("# per kernel of the pass (%d pairs, one in flight): launches per pair, mean duration, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes\n" % pairs)
        o.write("# (x2 gfx950 correction on FETCH_SIZE), achieved TB/s, kernel time per pair.  Sum of kernel time per pair: %.0f us\n" % (total / 1e3))
        o.write("%-78s %6s %9s %9s %7s %9s\n" % ("kernel", "n/pair", "mean us", "MB", "TB/s", "us/pair"))
        for t, k, ns, hb, n in rows:
            if t / 1e3 < 2:
                continue
            o.write("%-78s %6.1f %9.1f %9.1f %7.2f %9.1f\n" % (k[:78], n, ns / 1e3, hb / 1e6, hb / ns / 1e3, t / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3, sys.argv[4] if len(sys.argv) > 4 else "4k")
