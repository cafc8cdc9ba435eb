"""Per-dispatch timeline of ONE frame pair of a bench.py workload, on the whole chip or on a CU-masked stream (rife_hip_stream_create).

    runner (under rocprofv3 --kernel-trace):  python tools/layer_trace.py run --workload v23-1080p --parts 4 --pairs 3
    summary of the trace:                     python tools/layer_trace.py sum <kernel_trace.csv> [--pairs 3]

The summary lists the dispatches of the LAST pair in launch order (kernel, grid, workgroup, duration, gap to the previous kernel), i.e. what
every layer of the schedule costs on that part of the chip - the per-class HIP-event profile of bench.py is too coarse for the v2 family."""
import argparse, csv, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WL = {"4k": ("rife-v4.6", 3840, 2160, {}), "1080p": ("rife-v4.6", 1920, 1080, {}),
      "v23-1080p": ("rife-v2.3", 1920, 1080, {}), "4k-tta": ("rife-v4.6", 3840, 2160, {"tta_mode": True, "tta_temporal_mode": True})}


def run(args):
    import torch
    from tools import gen_frames, gen_models
    fam, w, h, kw = WL[args.workload]
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    eng = amd.RIFE(0, rife_v2=fam.startswith("rife-v2"), rife_v4=fam.startswith("rife-v4"), **kw)
    eng.load(gen_models.ensure(None, fam))
    fr = [torch.from_numpy(f).cuda() for f in gen_frames.tiled_real_pair(w // 640)]
    out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    st = eng.stream_create(0, args.parts) if args.parts > 1 else torch.cuda.Stream().cuda_stream
    for i in range(args.pairs):
        eng.process_device(fr[0].data_ptr(), fr[1].data_ptr(), w, h, 0.5, out.data_ptr(), st)
    torch.cuda.synchronize()
    print("done", args.pairs, "pairs", fam, w, h, "parts", args.parts)


def summarize(args):
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]),
                         int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0)))
    rows.sort()
    rows = [r for r in rows if "rife::" in r[2]]
    n = len(rows) // args.pairs
    last = rows[-n:]
    t_prev = None
    tot = 0.0
    print("# last of %d pairs: %d dispatches" % (args.pairs, n))
    print("%4s %8s %8s %7s %5s %6s %5s  %s" % ("#", "us", "gap us", "wgs", "thr", "lds", "vgpr", "kernel"))
    for i, (s, e, name, grid, wg, lds, vg) in enumerate(last):
        gap = 0.0 if t_prev is None else (s - t_prev) / 1e3
        t_prev = e
        tot += (e - s) / 1e3
        print("%4d %8.1f %8.1f %7d %5d %6d %5d  %s" % (i, (e - s) / 1e3, gap, grid // max(wg, 1), wg, lds, vg, name.replace("rife::", "")[:110]))
    print("# kernel time %.1f us, wall %.1f us" % (tot, (last[-1][1] - last[0][0]) / 1e3))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("run")
    a.add_argument("--workload", default="v23-1080p", choices=list(WL))
    a.add_argument("--parts", type=int, default=1)
    a.add_argument("--pairs", type=int, default=3)
    b = sub.add_parser("sum")
    b.add_argument("csv")
    b.add_argument("--pairs", type=int, default=3)
    args = ap.parse_args()
    run(args) if args.cmd == "run" else summarize(args)
