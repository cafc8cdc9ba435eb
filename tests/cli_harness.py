#!/usr/bin/env python3
"""Command line of the reference (`rife-ncnn-vulkan`, src/main.cpp:102-121, 442-917) on top of the HIP engine.

    tests/cli_harness.py -0 in0.png -1 in1.png -o out.png [options]
    tests/cli_harness.py -i indir -o outdir [options]

Same flags, defaults, validation and frame/timestep schedule as the reference; the three-stage pipeline
(load -> proc -> save, bounded queues of 8, `-j load:proc[,proc..]:save`, one RIFE replica per `-g` id,
src/main.cpp:248-436, 819-904) is re-hosted on Python threads (decode/encode via PIL; the GPU call releases the GIL).
This is host glue ("next" row): all arithmetic stays in librife_hip.so.

ROLE: the TEST HARNESS of the command-line contract.  The product's command line is the C++ binary `rife-hip` (csrc/main.cpp, `make -C csrc`);
this module restates the same schedule / validation logic in Python so that tests/test_cli.py can check both against the reference's rules and
against each other (same frames from the same arguments) without a subprocess per case.
"""
import getopt
import math
import os
import queue
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))


def usage():
    sys.stderr.write("""Usage: rife-ncnn-vulkan -0 infile -1 infile1 -o outfile [options]...
       rife-ncnn-vulkan -i indir -o outdir [options]...

  -h                   show this help
  -v                   verbose output
  -0 input0-path       input image0 path (jpg/png/webp)
  -1 input1-path       input image1 path (jpg/png/webp)
  -i input-path        input image directory (jpg/png/webp)
  -o output-path       output image path (jpg/png/webp) or directory
  -n num-frame         target frame count (default=N*2)
  -s time-step         time step (0~1, default=0.5)
  -m model-path        rife model path (default=rife-v2.3)
  -g gpu-id            gpu device to use (default=0) can be 0,1,2 for multi-gpu (-1=cpu is not available in the HIP build)
  -j load:proc:save    thread count for load/proc/save (default=1:2:2) can be 1:2,2,2:2 for multi-gpu
  -x                   enable spatial tta mode
  -z                   enable temporal tta mode
  -u                   enable UHD mode
  -f pattern-format    output image filename pattern format (%08d.jpg/png/webp, default=ext/%08d.png)
""")


def list_directory(path):
    """Sorted regular files (src/filesystem_utils.h:list_directory)."""
    return sorted(f for f in os.listdir(path) if os.path.isfile(os.path.join(path, f)))


def build_schedule(count, numframe):
    """Directory mode frame schedule, src/main.cpp:705-731: output i takes frames (sx, sx+1) at timestep fx.
    Returns [(sx, timestep)] of length numframe (numframe == 0 -> 2*count)."""
    if numframe == 0:
        numframe = count * 2
    scale = float(count) / numframe
    out = []
    for i in range(numframe):
        fx = float(i * scale)
        fx = float(__import__("numpy").float32(fx))           # the reference computes `float fx = i * scale`
        sx = int(math.floor(fx))
        fx -= sx
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= count - 1:
            sx, fx = count - 2, 1.0
        out.append((sx, fx))
    return out


def model_family(model):
    """Family flags from the directory *name* (src/main.cpp:658-683). Returns (rife_v2, rife_v4) or None."""
    if "rife-v2" in model or "rife-v3" in model:
        return True, False
    if "rife-v4" in model:
        return False, True
    if "rife" in model:
        return False, False
    return None


def parse_jobs(spec):
    """-j load:proc[,proc...]:save  (src/main.cpp:549-550)"""
    parts = spec.split(":")
    if len(parts) != 3:
        raise ValueError("invalid -j")
    return int(parts[0]), [int(x) for x in parts[1].split(",")], int(parts[2])


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        opts, _ = getopt.getopt(argv, "0:1:i:o:n:s:m:g:j:f:vxzuh")
    except getopt.GetoptError:
        usage()
        return -1
    input0 = input1 = inputpath = outputpath = ""
    numframe, timestep, model = 0, 0.5, "rife-v2.3"
    gpuid, jobs_load, jobs_proc, jobs_save = [], 1, [], 2
    verbose = tta = tta_temporal = uhd = False
    pattern_format = "%08d.png"
    for o, a in opts:
        if o == "-0": input0 = a
        elif o == "-1": input1 = a
        elif o == "-i": inputpath = a
        elif o == "-o": outputpath = a
        elif o == "-n": numframe = int(a)
        elif o == "-s": timestep = float(a)
        elif o == "-m": model = a
        elif o == "-g": gpuid = [int(x) for x in a.split(",")]
        elif o == "-j":
            try:
                jobs_load, jobs_proc, jobs_save = parse_jobs(a)
            except ValueError:
                sys.stderr.write("invalid thread count argument\n"); return -1
        elif o == "-f": pattern_format = a
        elif o == "-v": verbose = True
        elif o == "-x": tta = True
        elif o == "-z": tta_temporal = True
        elif o == "-u": uhd = True
        elif o == "-h": usage(); return -1

    # ---- validation, in the reference's order (src/main.cpp:575-689) ----
    if ((not input0 or not input1) and not inputpath) or not outputpath:
        usage(); return -1
    if not inputpath and (timestep <= 0.0 or timestep >= 1.0):
        sys.stderr.write("invalid timestep argument, must be 0~1\n"); return -1
    if inputpath and numframe < 0:
        sys.stderr.write("invalid numframe argument, must not be negative\n"); return -1
    if jobs_load < 1 or jobs_save < 1:
        sys.stderr.write("invalid thread count argument\n"); return -1
    if jobs_proc and len(jobs_proc) != (len(gpuid) if gpuid else 1):
        sys.stderr.write("invalid jobs_proc thread count argument\n"); return -1
    if any(j < 1 for j in jobs_proc):
        sys.stderr.write("invalid jobs_proc thread count argument\n"); return -1
    pattern, ext = os.path.splitext(pattern_format)
    fmt = ext[1:]
    if not fmt:
        pattern, fmt = "%08d", pattern_format
    if not pattern:
        pattern = "%08d"
    if not os.path.isdir(outputpath):
        e = os.path.splitext(outputpath)[1][1:]
        if e in ("png", "PNG"): fmt = "png"
        elif e in ("webp", "WEBP"): fmt = "webp"
        elif e in ("jpg", "JPG", "jpeg", "JPEG"): fmt = "jpg"
        else:
            sys.stderr.write("invalid outputpath extension type\n"); return -1
    if fmt not in ("png", "webp", "jpg"):
        sys.stderr.write("invalid format argument\n"); return -1
    fam = model_family(model)          # substring match on the whole argument, like the reference
    if fam is None:
        sys.stderr.write("unknown model dir type\n"); return -1
    rife_v2, rife_v4 = fam
    if not rife_v4 and (numframe != 0 or timestep != 0.5):
        sys.stderr.write("only rife-v4 model support custom numframe and timestep\n"); return -1

    # ---- task list (src/main.cpp:692-766) ----
    tasks = []      # (in0path, in1path, timestep, outpath)
    if inputpath and os.path.isdir(inputpath) and os.path.isdir(outputpath):
        names = list_directory(inputpath)
        if len(names) < 2:
            return -1
        for i, (sx, fx) in enumerate(build_schedule(len(names), numframe)):
            tasks.append((os.path.join(inputpath, names[sx]), os.path.join(inputpath, names[sx + 1]), fx,
                          os.path.join(outputpath, (pattern % (i + 1)) + "." + fmt)))        # ffmpeg numbering starts at 1
    elif not inputpath and not os.path.isdir(input0) and not os.path.isdir(input1) and not os.path.isdir(outputpath):
        tasks.append((input0, input1, timestep, outputpath))
    else:
        sys.stderr.write("input0path, input1path and outputpath must be file at the same time\n")
        sys.stderr.write("inputpath and outputpath must be directory at the same time\n")
        return -1

    modeldir = model if os.path.isdir(model) else os.path.join(_HERE, "..", "models", model)   # sanitize_dirpath: exe-relative fallback

    # ---- devices (src/main.cpp:774-828) ----
    import importlib
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.dirname(_HERE))
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    if not gpuid:
        gpuid = [0]
    if not jobs_proc:
        jobs_proc = [2] * len(gpuid)
    ndev = amd.device_count()
    for g in gpuid:
        if g < 0 or g >= ndev:
            sys.stderr.write("invalid gpu device\n"); return -1
    rifes = []
    for g in gpuid:
        r = amd.RIFE(g, tta, tta_temporal, uhd, 1, rife_v2, rife_v4)
        r.load(modeldir)
        rifes.append(r)

    # ---- load -> proc -> save (src/main.cpp:309-436, 830-904) ----
    END = object()
    toproc, tosave = queue.Queue(maxsize=8), queue.Queue(maxsize=8)
    jobs = queue.Queue()
    for t in tasks:
        jobs.put(t)

    def decode(path):
        try:
            return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
        except Exception:
            sys.stderr.write("decode image %s failed\n" % path)
            return None

    class SharedFrame:
        """A decoded frame and its copy in device memory per engine (uploaded by the first proc thread that needs it there): consecutive
        tasks use the same files, so each is decoded once and crosses PCIe once (the C++ CLI does the same)."""

        def __init__(self, px):
            self.px, self.lock, self.resident = px, threading.Lock(), {}

        def on(self, r):
            with self.lock:
                if id(r) not in self.resident:
                    self.resident[id(r)] = r.upload(self.px)
                return self.resident[id(r)]

    cache, cache_lock = [], threading.Lock()          # the last few decoded frames: [(path, SharedFrame)]

    def frame(path):
        with cache_lock:
            for p, f in cache:
                if p == path:
                    return f
        px = decode(path)
        if px is None:
            return None
        with cache_lock:
            for p, f in cache:
                if p == path:
                    return f
            f = SharedFrame(px)
            cache.append((path, f))
            del cache[:-6]
            return f

    def load():
        while True:
            try:
                t = jobs.get_nowait()
            except queue.Empty:
                return
            a, b = frame(t[0]), frame(t[1])
            if a is None or b is None or a.px.shape != b.px.shape:
                continue
            toproc.put((t, a, b))

    def proc(r):
        while True:
            item = toproc.get()
            if item is END:
                return
            t, a, b = item
            try:
                if t[2] == 0.0 or t[2] == 1.0:                  # rife.cpp:2470-2480: an input frame, unchanged
                    tosave.put((t, a.px if t[2] == 0.0 else b.px))
                else:
                    tosave.put((t, r.process_frames(a.on(r), b.on(r), t[2])))
            except Exception as e:                              # log and go on like csrc/main.cpp: a dead proc thread would stall the loaders on the bounded queue
                sys.stderr.write("process %s failed: %s\n" % (t[3], e))

    def save():
        while True:
            item = tosave.get()
            if item is END:
                return
            t, out = item
            im = Image.fromarray(out, "RGB")
            try:
                if fmt == "png": im.save(t[3], "PNG")
                elif fmt == "jpg": im.save(t[3], "JPEG", quality=100)
                else: im.save(t[3], "WEBP", lossless=True)
            except Exception as e:
                sys.stderr.write("encode image %s failed: %s\n" % (t[3], e))
                continue
            if verbose:
                sys.stderr.write("%s %s %f -> %s done\n" % (t[0], t[1], t[2], t[3]))

    loaders = [threading.Thread(target=load) for _ in range(jobs_load)]
    procs = [threading.Thread(target=proc, args=(rifes[i],)) for i in range(len(gpuid)) for _ in range(jobs_proc[i])]
    savers = [threading.Thread(target=save) for _ in range(jobs_save)]
    for th in loaders + procs + savers:
        th.start()
    for th in loaders:
        th.join()
    for _ in procs:
        toproc.put(END)
    for th in procs:
        th.join()
    for _ in savers:
        tosave.put(END)
    for th in savers:
        th.join()
    del cache[:]                                           # resident frames go before their engines
    return 0


if __name__ == "__main__":
    sys.exit(main())
