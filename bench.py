#!/usr/bin/env python3
"""Throughput of the RIFE hot path (`RIFE::process`, rife-v4.6) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p] [--streams S]
    N > 1: either launched as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE come from the launcher) or plainly as `python bench.py --gpus N`, which re-executes
    itself under that launcher on 127.0.0.1.  The communicator size must equal --gpus (checked; printed as `rccl_ranks`).
    --dry-run: launcher / sharding / barrier / JSON plumbing on CPU (gloo), no HIP work: what the CPU tests exercise.

One "step" = one frame pair (two resident u8 RGB frames -> one interpolated frame) through
`rife_hip_process_device`, per rank.  Frame pairs shard embarrassingly (the reference runs one RIFE replica
per device, src/main.cpp:819-866): every rank processes its own pairs, there is no data-path collective;
RCCL is used only for the start/stop barrier.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (model family, w, h, GFLOP per pair [SURVEY.md §8(d) / App. E: 2 x MAC over Convolution + Deconvolution],
#          roofline ms per pair [SURVEY §8(d): rife-v4.6 = the fused-minimum HBM time (it exceeds the matrix time); rife-v2.3 = FLOPs / dense f16 matrix peak, the
#          only bound the survey gives for it], tta, tta_temporal)
WORKLOADS = {
    "4k": ("rife-v4.6", 3840, 2160, 701.0, 0.701, False, False),          # BASELINE config 4 on one GPU (-u is a no-op for v4)
    "1080p": ("rife-v4.6", 1920, 1080, 175.2, 0.176, False, False),       # BASELINE config 3
    "v23-1080p": ("rife-v2.3", 1920, 1080, 597.5, 0.239, False, False),   # BASELINE config 2; the survey gives FLOPs only: 597.5 G / 2.5 PFLOP/s = 0.239 ms (matrix roofline, 4,180 frames/s)
    "4k-tta": ("rife-v4.6", 3840, 2160, 16 * 701.0, 16 * 0.701, True, True),   # BASELINE config 5 (-x -z) on one GPU
}
DEFAULT_PARTS = {"4k": 2, "1080p": 2, "v23-1080p": 4, "4k-tta": 0}      # CU partitions of the streams in the timed region (0 = ordinary streams), see main()
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense v_mfma_f32_32x32x2_f32 peak
F16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
HBM_PEAK_TBPS = 8.0               # MI355X_MICROARCH.md: HBM3E spec peak
# What the f16 matrix pipe SUSTAINS on random operands on this pool's MI355X boxes (tools/mfma_power_peak.py, profiles/r6/mfma_power_peak.txt: nothing but
# v_mfma_f32_32x32x16_f16 on register operands, bursts of seconds): 1,743 - 1,756 TFLOP/s at 1,320 W and 1.80 GHz with the pipe 94 % busy (one wave per SIMD, two
# or four accumulation chains; four waves per SIMD x four chains: 1,710 - 1,744) - 0.70 of the 2,500 TFLOP/s figure, which the same loop reaches on all-zero
# operands (2,461 - 2,477 at 2.40 GHz, 860 - 890 W).  Real data clocks the matrix pipe down (conv_rs2 on random tensors: 1.63 GHz, 144 - 154 us per launch; on a
# tensor that has converged to constants: 2.26 GHz, 110 us; profiles/r6/rs2_bench_real_data.txt).  `roofline.peak` stays the guide's figure; the fields
# *_of_sustained price the same rates against this measured one.
F16_MFMA_SUSTAINED_TFLOPS = 1750.0
# dominant kernel per family: (profile class, kernel symbol, channels C of the C->C 3x3 trunk conv, MFMA instructions issued per
# algorithmic product: 2 for the split-f16 scheme (hi and lo) x the identity tap of the folded skip connection)
DOMINANT = {"rife-v4.6": ("trunk_b3", "conv_rs2_kernel (IFNet block-3 trunk: TWO 3x3 convs 64->64 + skip + LeakyReLU per launch, the first layer's rows LDS-resident; row-streaming, specialised waves; split-f16 MFMA, S16 {hi, lo} tensors)", 64, 2.0 * 38 / 36),
            "rife-v2.3": ("v2_flow_trunk_b3", "conv_h2_kernel<3,9,0> (IFNet block-3 trunk: 3x3 conv 96->96 + PReLU, split-f16)", 96, 2.0)}


def roofline_of(dom, family, w, h, f32_mode):
    """Roofline object for the dominant kernel from the live HIP-event timing of its launches.

    Two byte models per launch, both reported:
      * `bytes_per_launch` = ALGORITHMIC bytes on SURVEY.md 8(d) / App. E-2's basis: fp16 input + fp16 output of the C->C trunk conv
        at 1/4 of the padded resolution + fp16 weights (4K block 3: 133.8 MB).  `achieved` and `frac` use these.
      * `bytes_per_launch_stored` = what the kernel really moves: activations are stored as {hi, lo} f16 pairs (4 B per element, the
        fp32-equivalent precision the <= 1 LSB bar needs) -> twice the bytes; `frac_stored_bytes` uses these.
    Algorithmic flops = 2*C*C*9 per output pixel (38.50 GFLOP at 4K); the split scheme issues 2 x 38/36 as many MFMA flops.
    Round 6: a launch of conv_rs2_kernel processes TWO layers (`layers_per_launch`, from the flops the engine books per launch): the algorithmic bytes
    per launch are SURVEY 8(d)'s per-layer figure x 2 units, the bytes it really moves are ONE input and ONE output tensor (the tensor between the two
    layers stays in LDS) - `bytes_per_launch_stored` - so on the survey basis `traffic` can now be BELOW `bytes_per_launch`."""
    cls, name, C, mfma_factor = DOMINANT[family]
    if not dom["launches"]:
        return None
    wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32
    pix = (hp // 4) * (wp // 4)
    flops_launch = dom["flops"] / dom["launches"]
    layers = max(1, int(round(flops_launch / (2.0 * C * C * 9 * pix))))
    if layers == 1:
        name = name.replace("conv_rs2_kernel", "conv_rs_kernel").replace("TWO 3x3 convs", "3x3 conv").replace(" per launch, the first layer's rows LDS-resident", "")
    bytes_alg = layers * (2.0 * pix * C * 2 + C * C * 9 * 2)
    bytes_stored = 2.0 * pix * C * 4 + layers * C * C * 9 * 2
    avg_ms = dom["ms"] / dom["launches"]
    tflops = flops_launch / (avg_ms * 1e-3) / 1e12
    tbps = bytes_alg / (avg_ms * 1e-3) / 1e12
    tbps_stored = bytes_stored / (avg_ms * 1e-3) / 1e12
    if f32_mode:
        return {"bound": "mfma", "kernel": name.replace("split-f16", "fp32 MFMA"), "achieved": round(tflops, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tflops / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                "launches": dom["launches"], "flops_per_launch": flops_launch, "bytes_per_launch": bytes_alg}
    return {"bound": "hbm", "kernel": name, "achieved": round(tbps * 1e3, 1), "peak": HBM_PEAK_TBPS * 1e3, "unit": "GB/s",
            "frac": round(tbps / HBM_PEAK_TBPS, 4), "frac_survey_basis": round(tbps / HBM_PEAK_TBPS, 4), "traffic": None,
            "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"], "layers_per_launch": layers,
            "flops_per_launch": flops_launch, "bytes_per_launch": bytes_alg, "bytes_basis": "algorithmic: (fp16 in + fp16 out + fp16 weights) per layer (SURVEY 8(d), App. E-2) x layers_per_launch",
            "bytes_per_launch_stored": bytes_stored, "achieved_stored_GBps": round(tbps_stored * 1e3, 1), "frac_stored_bytes": round(tbps_stored / HBM_PEAK_TBPS, 4),
            "algorithmic_tflops": round(tflops, 1), "mfma_frac_survey_basis": round(tflops / F16_MFMA_PEAK_TFLOPS, 4),
            "mfma_frac_issued": round(tflops * mfma_factor / F16_MFMA_PEAK_TFLOPS, 4),
            "mfma_sustained_tflops": F16_MFMA_SUSTAINED_TFLOPS, "mfma_frac_issued_of_sustained": round(tflops * mfma_factor / F16_MFMA_SUSTAINED_TFLOPS, 4),
            "mfma_sustained_basis": "dense f16 matrix rate sustained on random operands by a bare v_mfma_f32_32x32x16_f16 loop on this pool's MI355X (tools/mfma_power_peak.py, profiles/r6/mfma_power_peak.txt): 1,750 of the nominal 2,500 TFLOP/s, at 1,320 W and 1.80 GHz"}


DOMINANT_SYMBOL = {"rife-v4.6": ("conv_rs2_kernel", "conv_rs_kernel"), "rife-v2.3": ("conv_h2_kernel<3, 9, 0>",)}      # first symbol with dispatches in the trace


def live_traffic(workload, family, timeout_s=150):
    """HBM bytes per launch of the dominant kernel, measured IN THIS RUN on this box: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    counters alone, never mixed with other trace domains) over tools/prof_run.py on the same workload, one pair in flight.  KiB counters;
    FETCH_SIZE x 2 is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md.  None (with the reason) if rocprofv3 is not usable here."""
    import glob, shutil, subprocess, tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    from tools import pmc_summary
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rife_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "prof_run.py"), "--workload", workload, "--pairs", "2"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            rows = None
            for sym in DOMINANT_SYMBOL[family]:
                rows = pmc_summary.summarize(files[0], sym)
                if rows:
                    break
            if not rows:
                return None, "no dispatch of %s in the %s pass" % (" / ".join(DOMINANT_SYMBOL[family]), ctr)
            vals[ctr] = sum(dd[ctr] * dd["_dispatches"] for _, dd in rows) / sum(dd["_dispatches"] for _, dd in rows)
        except Exception as e:
            return None, "%s pass: %s" % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(vals["FETCH_SIZE"] * 2 * 1024 + vals["WRITE_SIZE"] * 1024), ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes of tools/prof_run.py --workload %s --pairs 2 inside "
                                                                             "this bench.py run; FETCH_SIZE x 2 = gfx950 correction)" % workload)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def percentiles(xs):
    xs = sorted(xs)
    n = len(xs)
    pick = lambda q: xs[min(n - 1, max(0, int(round(q * (n - 1)))))]
    return {"n": n, "median": round(pick(0.5), 3), "p10": round(pick(0.1), 3), "p90": round(pick(0.9), 3), "min": round(xs[0], 3), "max": round(xs[-1], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="4k", choices=list(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="frame pairs in flight per GPU in the timed region; 0 = the workload's default: 4 where the chip is partitioned (--cu-parts; the reference's -j knob, src/main.cpp:849-866), "
                    "2 otherwise (the reference's default, -j 1:2:2)")
    ap.add_argument("--cu-parts", type=int, default=-1, help="partition the compute units between the pairs in flight: every stream owns 1 / N of them (rife_hip_stream_create; "
                    "include/rife_hip.h).  -1 = the workload's default: 4 for the 1080p workloads (rife-v4.6: measured 1,690 vs 1,450 - 1,590 frames/s; rife-v2.3: 478 vs 448 - 470), 2 for 4k (two pairs per half: 473 - 479 vs 465 - 469 from three ordinary streams, same call), none for 4k-tta")
    ap.add_argument("--no-extra", action="store_true", help="skip the second region (1 pair in flight, clean per-launch kernel timing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive legs (rife_hip_process from pageable host buffers)")
    ap.add_argument("--frames", default="f1", choices=["f1", "f2"], help="synthetic frame content (SURVEY 8(d)): f1 = the reference's real frame pair tiled to size, f2 = smooth synthetic at native resolution")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (then the committed figure is used)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not restrict the rank to the CPUs of its GPU's NUMA node (read from /sys/class/drm/card*/device/numa_node)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST MODE for a 1-GPU box: all ranks use device 0 and rendezvous over gloo (RCCL refuses two ranks on one device); "
                    "exercises the N-rank code paths (sharding, barriers, the all-ranks host-buffer leg) on real HIP work.  The line is labelled and is not a scaling measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configs (extra.configs: 1080p, v23-1080p, 4k-tta) that the default call appends")
    ap.add_argument("--no-sustained", action="store_true", help="skip the ~ 4 s sustained leg with socket power / shader clock sampling (extra.sustained)")
    ap.add_argument("--dry-run", action="store_true", help="CPU plumbing check (gloo): launcher, sharding, barrier, MAX over ranks, JSON; no HIP work")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s); the two must agree (n_gpus is the communicator size)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.share_gpu:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d device(s) visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    sh0 = importlib.import_module("rife-ncnn-vulkan_amd.sharding")
    numa = None
    if not args.no_numa_pin and world > 1:
        # one rank per GPU: its caller threads, the batch workers and the runtime's staging copies stay on the socket the GPU hangs off
        props = torch.cuda.get_device_properties(local)
        pci = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
        numa = sh0.pin_to_gpu_numa(pci)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        rccl_ranks = dist.get_world_size()
    coll_dev = "cpu" if args.share_gpu else "cuda"          # where the 1-element barrier / MAX tensors live

    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    from tools import gen_frames, gen_models
    family, w, h, gflop_pair, roofline_ms, tta, tta_temporal = WORKLOADS[args.workload]
    if rank == 0:
        modeldir = gen_models.ensure(None, family)       # one writer; the other ranks wait, then find it complete
    if dist is not None:
        dist.barrier()
    modeldir = gen_models.ensure(None, family)
    eng = amd.RIFE(local, tta_mode=tta, tta_temporal_mode=tta_temporal, rife_v2=family.startswith("rife-v2"), rife_v4=family.startswith("rife-v4"))
    eng.load(modeldir)

    # a short synthetic stream of distinct pairs at NATIVE resolution, resident in HBM (consecutive pairs share a frame like a video):
    # F1 of SURVEY 8(d) - the reference's real 640x360 frame pair tiled to the workload's size (6 x 6 = 3840x2160) - or, --frames f2, the smooth
    # synthetic pair generated at full size; frames 2, 3 = frames 0, 1 shifted by a few pixels (no upsampled content: the matrix pipe clocks with
    # the toggle rate of its operands)
    nfr = 4
    base = None
    if args.frames == "f1" and w % 640 == 0 and h % 360 == 0 and w // 640 == h // 360:
        try:
            base = gen_frames.tiled_real_pair(w // 640)
            frame_kind = "F1: images/0.png, 1.png of the reference tiled %d x %d" % (w // 640, w // 640)
        except Exception as e:                               # no PIL / no fixture PNGs on this box: the synthetic frames instead (labelled)
            sys.stderr.write("bench.py: F1 frames unavailable (%s); using F2\n" % e)
    if base is None:
        base = gen_frames.smooth_pair_native(w, h, 1000 + rank)
        frame_kind = "F2: smooth synthetic pair generated at %dx%d" % (w, h)
    frames = []
    for i in range(nfr):
        f = np.roll(base[i % 2], (2 * (i // 2), 5 * (i // 2)), axis=(0, 1))
        frames.append(torch.from_numpy(np.ascontiguousarray(f)).cuda())
    timesteps = [0.5, 0.125, 0.25, 0.7, 0.9]
    # default layouts (same-call sweeps: profiles/r4/cumask_probe.txt, profiles/r5/stream_layouts.txt, profiles/r6/layout_sweep.txt): rife-v4.6 at 4K and (since round 6)
    # at 1080p four pairs in flight, two per half of the compute units (1080p: 1,807-1,838 against 1,796-1,807 frames/s from one per quarter); rife-v2.3 one per quarter
    cu_parts = args.cu_parts if args.cu_parts >= 0 else DEFAULT_PARTS.get(args.workload, 0)
    nstreams = args.streams if args.streams > 0 else (4 if cu_parts > 1 else 2)

    class PartStream:                                        # a stream of rife_hip_stream_create, with torch.cuda.Stream's attribute
        def __init__(self, part, nparts):
            self.cuda_stream = eng.stream_create(part, nparts)
    streams = [PartStream(i % cu_parts, cu_parts) if cu_parts > 1 else torch.cuda.Stream() for i in range(nstreams)]
    outs = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(nstreams)]

    def step(i):
        s = i % len(streams)
        a, b = frames[i % nfr], frames[(i + 1) % nfr]
        eng.process_device(a.data_ptr(), b.data_ptr(), w, h, timesteps[i % len(timesteps)], outs[s].data_ptr(), streams[s].cuda_stream)

    import threading

    def run_steps(first, count):
        """`count` steps starting at index `first`; with S > 1 pairs in flight each stream is driven by its own host thread,
        exactly like the reference's proc threads (src/main.cpp:849-866; the C-ABI call releases the GIL)."""
        S = len(streams)
        if S == 1:
            for i in range(first, first + count):
                step(i)
            return

        def worker(s):
            torch.cuda.set_device(local)
            for i in range(first, first + count):
                if i % S == s:
                    step(i)
        th = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
        [t.start() for t in th]
        [t.join() for t in th]

    sh = importlib.import_module("rife-ncnn-vulkan_amd.sharding")
    # untimed set-up: one pair per stream (workspace allocation, kernel attributes), then exactly the W warm-up steps the contract asks for
    SETUP = nstreams
    run_steps(0, SETUP)
    for i in range(args.warmup):
        step(i)
    sh.barrier(dist, torch.cuda.synchronize)
    # THE timed region (`value`): K steps, nothing but the product path (the per-launch HIP events of the profiler cost 5 % at
    # 4K and 20 % at 1080p, so they are kept out of it) ...
    elapsed = sh.timed_steps(lambda i: run_steps(i, args.steps), 1, first_index=args.warmup, dist=dist, device_sync=torch.cuda.synchronize,
                             make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
    # the same K-step region repeated (>= 50 pairs, median / p10 / p90): `value` stays the FIRST region, exactly K steps
    reps = max(3, -(-150 // max(1, args.steps)))
    region_fps = [world * args.steps / elapsed]
    for _ in range(reps - 1):
        el = sh.timed_steps(lambda i: run_steps(i, args.steps), 1, first_index=args.warmup, dist=dist, device_sync=torch.cuda.synchronize,
                            make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
        region_fps.append(world * args.steps / el)
    # sustained leg (rank 0 of a 1-GPU run): the same steps for ~ 4 s while a thread reads the socket power and the shader clock (rocm-smi).  Every BASELINE
    # workload runs at the board's power cap (profiles/r6/power_workloads.txt: 1,337 - 1,400 W of 1,400, 1.8 - 2.05 GHz): the clock the matrix pipe really gets is
    # part of the roofline story, and the sustained rate is a few per cent below the first K steps of a cool chip.  Never `value`.
    sustained = None
    if world == 1 and not args.no_sustained:
        sustained = sustained_leg(lambda n: run_steps(args.warmup, n), region_fps[0], torch.cuda.synchronize)
    # PCIe-inclusive legs: the boundary call the reference CLI makes (RIFE::process on host frames: H2D x2 + pass + D2H inside the
    # call, src/rife.cpp:2522-2530, 3176-3186) from pageable host memory, 1 and 2 caller threads (the reference's default -j 1:2:2)
    host = None
    if not args.no_host_path and not tta:
        host = {}
        pageable = ([f.cpu().numpy() for f in frames], [np.empty((h, w, 3), np.uint8) for _ in range(3)])       # one output per caller thread
        pinned = ([amd.pinned_empty((h, w, 3)) for _ in frames], [amd.pinned_empty((h, w, 3)) for _ in range(3)])     # rife_hip_host_alloc
        for dst, src in zip(pinned[0], pageable[0]):
            dst[...] = src

        HK = max(args.steps, 96)       # pairs per host leg: K = 30 pairs are over before the caller threads and the workspace pool have settled (tools/host_path_bench2.py, 96 pairs
                                       # after an untimed round: process_batch 475 frames/s where a 30-pair leg read 425); every leg runs one untimed round first

        def host_run(bufs, nthreads):
            hin, hout = bufs

            def host_step(i, slot):
                eng.process(hin[i % nfr], hin[(i + 1) % nfr], timesteps[i % len(timesteps)], outimage=hout[slot])

            def region(_):
                if nthreads == 1:
                    for i in range(HK):
                        host_step(i, 0)
                    return

                errs = []

                def worker(s):
                    try:
                        torch.cuda.set_device(local)
                        for i in range(HK):
                            if i % nthreads == s:
                                host_step(i, s)
                    except Exception as e:       # a dead caller thread would inflate the rate: fail the leg instead
                        errs.append(e)
                th = [threading.Thread(target=worker, args=(s,)) for s in range(nthreads)]
                [t.start() for t in th]
                [t.join() for t in th]
                if errs:
                    raise errs[0]
            region(0)                                      # untimed: caller threads started once, pool workspaces of this caller count allocated
            return sh.timed_steps(region, 1, dist=dist, device_sync=torch.cuda.synchronize,
                                  make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
        for kind, bufs in (("pageable", pageable), ("page_locked", pinned)):
            for nt in (1, 2, 3, 4):
                if nt == 4 and len(bufs[1]) < 4:
                    bufs[1].append(amd.pinned_empty((h, w, 3)) if kind == "page_locked" else np.empty((h, w, 3), np.uint8))
                host["%s_caller_threads_%d" % (kind, nt)] = round(world * HK / host_run(bufs, nt), 3)

        def batch_run(bufs):
            """ONE caller thread, rife_hip_process_batch over the K pairs of the region (internal workers overlap copies and passes)"""
            hin, _ = bufs
            bouts = [np.empty((h, w, 3), np.uint8) for _ in range(HK)]
            a0 = [hin[i % nfr] for i in range(HK)]; a1 = [hin[(i + 1) % nfr] for i in range(HK)]
            ts = [timesteps[i % len(timesteps)] for i in range(HK)]
            eng.process_batch(a0, a1, ts, bouts)           # untimed round
            return sh.timed_steps(lambda _: eng.process_batch(a0, a1, ts, bouts), 1, dist=dist, device_sync=torch.cuda.synchronize,
                                  make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
        host["process_batch_one_thread_pageable"] = round(world * HK / batch_run(pageable), 3)
        host["process_batch_one_thread_page_locked"] = round(world * HK / batch_run(pinned), 3)
        host["pairs_per_leg"] = HK
    # N > 1: the host-buffer path on ALL ranks at once - what can stop a node from scaling is not on the GPUs (pairs are independent, no collective)
    # but under them: every rank pulls 3 x w x h x 3 bytes per pair through host DRAM and its PCIe root.  One caller per rank, rife_hip_process_batch over
    # the K pairs, all ranks inside one barrier-bracketed region; whole-job frames/s and the aggregate host <-> device traffic.
    host_all = None
    if world > 1 and not args.no_host_path and not tta:
        host_all = {}
        per_pair_bytes = 3.0 * w * h * 3
        for kind in ("pageable", "page_locked"):
            if kind == "pageable":
                hin = [f.cpu().numpy() for f in frames]
                bouts = [np.empty((h, w, 3), np.uint8) for _ in range(args.steps)]
            else:
                hin = [amd.pinned_empty((h, w, 3)) for _ in frames]
                for dst, src in zip(hin, frames):
                    dst[...] = src.cpu().numpy()
                bouts = [amd.pinned_empty((h, w, 3)) for _ in range(min(args.steps, 8))]
                bouts = [bouts[i % len(bouts)] for i in range(args.steps)]
            a0 = [hin[i % nfr] for i in range(args.steps)]; a1 = [hin[(i + 1) % nfr] for i in range(args.steps)]
            ts = [timesteps[i % len(timesteps)] for i in range(args.steps)]
            eng.process_batch(a0[:3], a1[:3], ts[:3], bouts[:3])
            rate, el = sh.all_ranks_rate(args.steps, lambda: eng.process_batch(a0, a1, ts, bouts), dist=dist, device_sync=torch.cuda.synchronize,
                                         make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
            host_all[kind] = {"frames_per_s": round(rate, 3), "host_device_GBps": round(rate * per_pair_bytes / 1e9, 2)}
    # ... and the same K steps again with HIP events around every launch on its stream (`roofline_in_timed_region`)
    sh.barrier(dist, torch.cuda.synchronize)
    eng.profile_enable(True)
    elapsed_instr = sh.timed_steps(lambda i: run_steps(i, args.steps), 1, first_index=args.warmup, dist=dist, device_sync=torch.cuda.synchronize,
                                   make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
    prof = eng.profile_read()
    eng.profile_enable(False)
    # second region, same K steps with ONE pair in flight: kernels of different pairs no longer overlap, so the HIP-event time of
    # a launch is that kernel's own duration (this is what rocprofv3 --kernel-trace of tools/prof_run.py measures too)
    prof1, fps1 = None, None
    if not args.no_extra and nstreams > 1:
        saved = streams[:]
        streams[:] = [torch.cuda.Stream()]                  # ONE ordinary stream: the kernel alone on the WHOLE chip (also when the timed region partitions the CUs)
        for i in range(2):
            step(i)
        sh.barrier(dist, torch.cuda.synchronize)
        eng.profile_enable(True)
        el1 = sh.timed_steps(step, args.steps, first_index=args.warmup, dist=dist, device_sync=torch.cuda.synchronize,
                             make_tensor=lambda v, dtype: torch.tensor(v, dtype=dtype, device=coll_dev))
        prof1 = eng.profile_read()
        eng.profile_enable(False)
        fps1 = world * args.steps / el1
        streams[:] = saved
    if rank == 0:
        # dominant kernel: the block-3 trunk conv of the IFNet (one shape per class, so flops per launch are well defined)
        f32_mode = os.environ.get("RIFE_HIP_TRUNK", "") == "f32"
        dom = prof.get(DOMINANT[family][0], dict(ms=0.0, launches=0, flops=0.0))
        pclean = prof1 or prof
        conv_ms = sum(v["ms"] for k, v in pclean.items() if v["flops"] > 0)
        all_ms = sum(v["ms"] for v in pclean.values())
        roof_timed = roofline_of(dom, family, w, h, f32_mode)
        roof = roof_timed
        if prof1 is not None:
            roof = roofline_of(prof1.get(DOMINANT[family][0], dict(ms=0.0, launches=0, flops=0.0)), family, w, h, f32_mode)
            if roof is not None:
                roof["measured_in"] = "HIP events on the launch stream over a region of the same %d steps with 1 pair in flight (non-overlapping launches); roofline_in_timed_region = the timed region repeated with the events on" % args.steps
        # roofline.traffic: measured live in this run (two rocprofv3 --pmc passes in child processes, after the timed regions); if rocprofv3 cannot
        # run here, the committed figure of the newest round that has one for THIS workload, labelled as such
        if roof is not None and world == 1 and not args.no_live_traffic:
            tb, src = live_traffic(args.workload, family)
            if tb is not None:
                roof["traffic"], roof["traffic_source"] = tb, src
            else:
                roof["traffic_live_failed"] = src
        traffic_file = next((f for f in (os.path.join(ROOT, "profiles", r, "pmc_%s.json" % args.workload.replace("-", "_")) for r in ("r6", "r5", "r4", "r3", "r2")) if os.path.exists(f)), "")
        if roof is not None and roof.get("traffic") is None and traffic_file:
            tf = json.load(open(traffic_file))                   # from the committed rocprofv3 --pmc passes (not live)
            roof["traffic"] = tf["hbm_bytes_per_launch"]
            roof["traffic_source"] = "committed, not live: " + tf["source"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(modeldir, family, w * h, 16 if tta and tta_temporal else 1, (w, h))
        fps = world * args.steps / elapsed
        # roofline.e2e_frac: the whole pair against the fused-minimum HBM roofline of SURVEY 8(d) / App. E-3 (every tensor of the graph read and written once
        # at fp16, 0.701 ms per 4K pair at 8 TB/s) - roofline time / measured time per pair PER GPU, in the timed region (4 pairs in flight): the one number
        # that compares across rounds
        if roof is not None and roofline_ms is not None:
            roof["e2e_frac"] = round(roofline_ms / (elapsed / args.steps * 1e3), 5)
            # the same pair against the matrix work it cannot avoid at the rate the chip sustains: 2 products (hi, lo) x the algorithmic flops / 1,750 TFLOP/s
            roof["e2e_frac_of_sustained_mfma"] = round(2.0 * gflop_pair / F16_MFMA_SUSTAINED_TFLOPS / (elapsed / args.steps * 1e3), 5)
            roof["e2e_basis"] = ("%s per pair (%.3f ms, SURVEY 8(d)) / ms_per_step"
                                 % ("matrix-roofline time (597.5 GFLOP / 2.5 PFLOP/s)" if family == "rife-v2.3" else "fused-minimum HBM time", roofline_ms))
        # the default call certifies EVERY BASELINE config: short legs of the other three workloads, same stream policy as their --workload runs
        configs = None
        if args.workload == "4k" and world == 1 and not args.no_configs and not f32_mode:
            configs = other_configs(amd, torch, sh, local, eng, base, None if args.no_cpu_baseline else cpu_baseline_small_frames())
        boundary = None
        if host:
            kbest = max((k for k in host if isinstance(host[k], float)), key=lambda k: host[k])
            boundary = {"value": host[kbest], "unit": "frames/s", "leg": kbest,
                        "note": "best PCIe-inclusive figure of this run: host frames in, host frame out through RIFE::process / rife_hip_process_batch (the boundary the reference exposes); never `value`"}
        line = {
            "metric": "interpolated frames/sec (%s, %dx%d%s)" % (family, w, h, " -x -z" if tta else ""), "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if f32_mode else "f16x2-split MFMA + f32 accumulate (fp32-equivalent; activations stored f32 or as {hi, lo} f16 pairs)", "data": "synthetic",
            "config": {"workload": "%s %dx%d%s frame pairs resident in HBM, timestep sweep %s, synthetic seeded weights" % (family, w, h, " -x -z (TTA)" if tta else "", timesteps),
                       "pairs_in_flight_per_gpu": nstreams,
                       "cu_partition": "none (ordinary streams)" if cu_parts <= 1 else "every stream owns 1 / %d of the compute units (rife_hip_stream_create)" % cu_parts,
                       "frames": frame_kind, "untimed_setup_pairs": SETUP, "parallelism": "frame pairs sharded over ranks, no data-path collective" + (" [--share-gpu TEST MODE: all ranks on ONE device over gloo - not a scaling measurement]" if args.share_gpu else "")},
            "roofline": roof,
            "roofline_in_timed_region": roof_timed if prof1 is not None else None,
            "cpu_baseline": cpu,
            "extra": {"frames_per_s_repeated_regions": dict(percentiles(region_fps), pairs_measured=reps * args.steps * world,
                                                            note="the K-step timed region repeated %d times back to back; `value` is the first" % reps),
                      "frames_per_s_host_buffers": None if host is None else dict(host, note="rife_hip_process on host frames: 2 x H2D + pass + D2H inside the call (PCIe-inclusive; never `value`); page_locked = frames from rife_hip_host_alloc; process_batch_one_thread = ONE caller, rife_hip_process_batch over the region's K pairs"),
                      "frames_per_s_host_buffers_all_ranks": None if host_all is None else dict(host_all, note="every rank at once: one caller per rank, rife_hip_process_batch over its K host-frame pairs, one barrier-bracketed region; whole-job rate = pairs of all ranks / MAX elapsed (PCIe-inclusive; never `value`)"),
                      "numa": numa,
                      "sustained": sustained,
                      "frames_per_s_same_region_with_per_launch_events": round(world * args.steps / elapsed_instr, 3),
                      "frames_per_s_with_1_pair_in_flight": None if fps1 is None else round(fps1, 3), "kernel_ms_per_pair": round(all_ms / args.steps, 4), "conv_ms_per_pair": round(conv_ms / args.steps, 4),
                      "conv_tflops_overall": round(gflop_pair / max(conv_ms / args.steps, 1e-9), 2),
                      "frac_of_fused_hbm_roofline_e2e": None if roofline_ms is None else round(roofline_ms / (elapsed / args.steps * 1e3), 5),
                      "per_class_ms_per_pair": {k: round(v["ms"] / args.steps, 4) for k, v in sorted((prof1 or prof).items(), key=lambda kv: -kv[1]["ms"])},
                      "frames_per_s_process_boundary": boundary,
                      "configs": configs},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# short legs of the other BASELINE configs, appended to the default call as extra.configs: (steps, warm-up, profiled steps)
CONFIG_LEGS = {"1080p": (60, 8, 4), "v23-1080p": (40, 8, 4), "4k-tta": (4, 1, 1)}      # 12 steps of 0.6 ms under-read the 1080p rate by 10 % (thread start-up inside the region)


def leg_fps(torch, sh, eng, frames, w, h, nstreams, cu_parts, steps, warmup, local):
    """One barrier-bracketed region of `steps` resident pairs, `nstreams` pairs in flight (one host thread per stream), the workload's stream
    policy.  Returns (frames/s, the streams) - the same procedure as the main region, on one rank."""
    import threading
    timesteps = [0.5, 0.125, 0.25, 0.7, 0.9]
    streams = [eng.stream_create(i % cu_parts, cu_parts) if cu_parts > 1 else torch.cuda.Stream().cuda_stream for i in range(nstreams)]
    outs = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
    nfr = len(frames)

    def step(i):
        s = i % nstreams
        eng.process_device(frames[i % nfr].data_ptr(), frames[(i + 1) % nfr].data_ptr(), w, h, timesteps[i % len(timesteps)], outs[s].data_ptr(), streams[s])

    def run(first, count):
        def worker(s):
            torch.cuda.set_device(local)
            for i in range(first, first + count):
                if i % nstreams == s:
                    step(i)
        th = [threading.Thread(target=worker, args=(s,)) for s in range(nstreams)]
        [t.start() for t in th]
        [t.join() for t in th]
    run(0, nstreams)                                         # untimed set-up: workspaces, kernel attributes
    for i in range(warmup):
        step(i)
    el = sh.timed_steps(lambda i: run(i, steps), 1, first_index=warmup, device_sync=torch.cuda.synchronize)
    return steps / el, outs


def other_configs(amd, torch, sh, local, eng46, base4k, oracle_small):
    """extra.configs: {name: {value, ms_per_step, steps, roofline_frac, max_lsb, ...}} for BASELINE configs 3, 2 and 5 (the headline `value` stays config 4's).
    max_lsb = the engine against the CPU oracle on a 256x160 smooth pair in the same mode (the full-size comparisons are the -m gpu tests and
    profiles/*/parity_report.txt; the oracle needs seconds to minutes per full-size pair)."""
    from tools import gen_frames, gen_models
    res = {}
    t_all = time.perf_counter()
    small = gen_frames.smooth_pair(*SMALL_PAIR)
    for name, (steps, warmup, psteps) in CONFIG_LEGS.items():
        t0 = time.perf_counter()
        try:
            family, w, h, gflop_pair, roofline_ms, tta, tta_temporal = WORKLOADS[name]
            v2, v4 = family.startswith("rife-v2"), family.startswith("rife-v4")
            modeldir = gen_models.ensure(None, family)
            if family == "rife-v4.6" and not tta:
                eng = eng46
            else:
                eng = amd.RIFE(local, tta_mode=tta, tta_temporal_mode=tta_temporal, rife_v2=v2, rife_v4=v4)
                eng.load(modeldir)
            # parity first (small workspaces), then the timed legs at full size
            max_lsb = None
            if oracle_small is not None and name in oracle_small:
                got = eng.process(small[0], small[1], 0.5)
                max_lsb = int(np.abs(got.astype(np.int32) - oracle_small[name].astype(np.int32)).max())
            if w == 3840:
                pair = base4k
            else:
                try:
                    pair = gen_frames.tiled_real_pair(w // 640)
                except Exception:                                    # no PIL / fixture PNGs on this box
                    pair = gen_frames.smooth_pair_native(w, h, 1000)
            frames = []
            for i in range(4):
                f = np.roll(pair[i % 2], (2 * (i // 2), 5 * (i // 2)), axis=(0, 1))
                frames.append(torch.from_numpy(np.ascontiguousarray(f)).cuda())
            cu_parts = DEFAULT_PARTS.get(name, 0)
            nstreams = 2 if tta else 4
            fps, _ = leg_fps(torch, sh, eng, frames, w, h, nstreams, cu_parts, steps, warmup, local)
            # dominant kernel of the leg: HIP events per launch, ONE pair in flight on the whole chip
            st = torch.cuda.Stream().cuda_stream
            out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
            eng.process_device(frames[0].data_ptr(), frames[1].data_ptr(), w, h, 0.5, out.data_ptr(), st)
            torch.cuda.synchronize()
            eng.profile_enable(True)
            for i in range(psteps):
                eng.process_device(frames[i % 4].data_ptr(), frames[(i + 1) % 4].data_ptr(), w, h, 0.5, out.data_ptr(), st)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile_enable(False)
            roof = roofline_of(prof.get(DOMINANT[family][0], dict(ms=0.0, launches=0, flops=0.0)), family, w, h, False)
            res[name] = {"metric": "interpolated frames/sec (%s, %dx%d%s)" % (family, w, h, " -x -z" if tta else ""), "value": round(fps, 3), "unit": "frames/s",
                         "ms_per_step": round(1e3 / fps, 4), "steps": steps, "warmup": warmup, "pairs_in_flight": nstreams,
                         "cu_partition": "none" if cu_parts <= 1 else "1 / %d of the compute units per stream" % cu_parts,
                         "roofline_frac": None if roof is None else roof["frac"], "roofline_kernel": None if roof is None else roof["kernel"].split(" (")[0],
                         "roofline_avg_launch_ms": None if roof is None else roof["avg_launch_ms"],
                         "e2e_frac": None if roofline_ms is None else round(roofline_ms * fps / 1e3, 5),
                         "e2e_basis": "matrix-roofline time 0.239 ms (597.5 GFLOP / 2.5 PFLOP/s)" if family == "rife-v2.3" else "fused-minimum HBM time %.3f ms" % roofline_ms,
                         "mfma_frac_issued": None if roof is None else roof["mfma_frac_issued"],
                         "max_lsb": max_lsb, "max_lsb_on": "256x160 smooth pair vs the CPU oracle, same mode", "leg_seconds": round(time.perf_counter() - t0, 2)}
            if eng is not eng46:
                del eng
            torch.cuda.synchronize()
        except Exception as e:                                       # a failing leg must not take the headline line with it - but it must be visible
            res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    res["_seconds"] = round(time.perf_counter() - t_all, 2)
    return res


def dry_run(args, rank, world):
    """Everything of the N-rank bench but the GPU: rendezvous (gloo), pair -> rank sharding, barrier + MAX timing, one JSON line."""
    import importlib as il
    import torch
    sh = il.import_module("rife-ncnn-vulkan_amd.sharding")
    dist = None
    ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = dist.get_world_size()
    mine = sh.shard_pairs(args.steps * world, rank, world)            # weak scaling: K pairs per rank
    assert len(mine) == args.steps
    elapsed = sh.timed_steps(lambda i: time.sleep(0.001), args.steps, first_index=args.warmup, dist=dist)
    # the all-ranks host-buffer leg (N > 1): same plumbing, a sleep instead of rife_hip_process_batch; rank r "processes" K pairs in (r + 1) x 10 ms
    rate_all, el_all = sh.all_ranks_rate(args.steps, lambda: time.sleep(0.01 * (rank + 1)), dist=dist)
    numa = sh.pin_to_gpu_numa("0000:00:00.0", setaffinity=lambda cpus: None)      # lookup only (no such card on a CPU box): must not raise
    if rank == 0:
        print(json.dumps({"metric": "dry run: launcher / sharding / barrier plumbing (no GPU work)", "value": round(world * args.steps / elapsed, 3), "unit": "steps/s",
                          "extra": {"all_ranks_leg": {"units_per_s": round(rate_all, 3), "max_elapsed_s": round(el_all, 4), "units": world * args.steps}, "numa": numa},
                          "n_gpus": world, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run",
                          "config": {"workload": "dry run on CPU (gloo)", "parallelism": "frame pairs sharded over ranks, no data-path collective"}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


SMALL_PAIR = (256, 160, 77)       # w, h, seed of the smooth pair extra.configs' max_lsb is taken on


def cpu_baseline_small_frames():
    """Part of the cpu_baseline leg (the only place of bench.py that touches oracle/): the oracle's frames for the 256x160 smooth pair in the mode of
    every extra.configs leg, so that each leg can state max_lsb of the engine against them."""
    from oracle import pyoracle
    from tools import gen_frames, gen_models
    a, b = gen_frames.smooth_pair(*SMALL_PAIR)
    out = {}
    for name in CONFIG_LEGS:
        family, _, _, _, _, tta, tta_temporal = WORKLOADS[name]
        try:
            o = pyoracle.OracleRIFE(tta_mode=tta, tta_temporal_mode=tta_temporal, rife_v2=family.startswith("rife-v2"), rife_v4=family.startswith("rife-v4"),
                                    num_threads=min(len(os.sched_getaffinity(0)), 64))
            o.load(gen_models.ensure(None, family))
            out[name] = o.process(a, b, 0.5)
        except Exception as e:
            sys.stderr.write("bench.py: oracle frame for %s unavailable (%s)\n" % (name, e))
    return out


def sustained_leg(run_n, fps_guess, device_sync, seconds=4.0):
    """~ `seconds` of the timed region's steps while a thread samples `rocm-smi --showpower --showclocks` (Current Socket Graphics Package Power, sclk):
    {frames_per_s, seconds, power_w / sclk_mhz medians over the second half of the leg, power_cap_w}.  Figures are None where rocm-smi is not usable."""
    import subprocess
    import threading
    n = max(8, int(fps_guess * seconds))
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = [float(l.split(":")[-1]) for l in t.splitlines() if "Socket Graphics Package Power" in l]
                fq = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk clock level" in l]
                if pw and fq:
                    samples.append((time.perf_counter(), pw[0], fq[0]))
            except Exception:
                time.sleep(0.2)
    cap = None
    try:
        t = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
        capv = [float(l.split(":")[-1]) for l in t.splitlines() if "Max Graphics Package Power" in l]
        cap = capv[0] if capv else None
    except Exception:
        pass
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    device_sync()
    t0 = time.perf_counter()
    run_n(n)
    device_sync()
    t1 = time.perf_counter()
    stop[0] = True
    th.join(timeout=15)
    tail = [(p, f) for (t, p, f) in samples if t0 + 0.5 * (t1 - t0) <= t <= t1]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    return {"frames_per_s": round(n / (t1 - t0), 3), "seconds": round(t1 - t0, 2), "steps": n, "power_w": med([p for p, _ in tail]), "sclk_mhz": med([f for _, f in tail]),
            "power_cap_w": cap, "samples": len(tail), "f16_mfma_peak_at_that_clock_tflops": None if not tail else round(F16_MFMA_PEAK_TFLOPS * med([f for _, f in tail]) / 2400.0, 1),
            "note": "the timed region's steps for ~ %.0f s, rocm-smi sampled by a second thread (second half of the leg); dense f16 matrix peak scaled from 2,400 MHz; never `value`" % seconds}


def cpu_model():
    """`lscpu`'s model name (recorded next to the core count); /proc/cpuinfo where lscpu is missing."""
    try:
        import subprocess
        for l in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if l.lower().startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    try:
        for l in open("/proc/cpuinfo"):
            if l.lower().startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def cpu_baseline(modeldir, family, pixels, passes, size=None):
    """The reference's `-g -1` BINARY cannot be built here (ncnn/Vulkan absent), so the CPU leg is the oracle
    (kind "port"; `reference_build` next to it = the reference's own src/rife.cpp + warp.cpp compiled against the ncnn look-alike of
    oracle/refbuild/, whose convolutions are the port's: same speed, it is there to show that), timed on a bounded sample (>= 10 s of wall time on the host's cores): plain pairs of the same model at
    the workload's own frame size (1920x1080 for the TTA workload; the result is then scaled by the pixel ratio and the
    x16 passes of -x -z: the work is linear in both)."""
    from oracle import pyoracle
    from tools import gen_frames
    cores = min(len(os.sched_getaffinity(0)), 64)
    o = pyoracle.OracleRIFE(rife_v2=family.startswith("rife-v2"), rife_v4=family.startswith("rife-v4"), num_threads=cores)
    o.load(modeldir)
    w, h = size if (size and passes == 1) else (1920, 1080)
    a, b = gen_frames.smooth_pair(w, h, 1000)
    n, t0 = 0, time.perf_counter()
    while True:
        o.process(a, b, 0.5)
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= 10.0 and n >= 2) or dt >= 30.0:
            break
    per_pair = dt / n
    scale = pixels / float(w * h) * passes
    res = {"value": round(1.0 / (per_pair * scale), 5), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(), "nproc": os.cpu_count(),
           "sample": "%d plain %s pair(s) at %dx%d in %.2f s with %d OpenMP threads; scaled by x%.3g (pixels x TTA passes, work is linear in both)"
                     % (n, family, w, h, dt, cores, 1.0 / scale)}
    # the reference's own default is `-j 1:2:2`: TWO proc threads, i.e. num_threads = 2 on its CPU device (src/main.cpp:783-786, 807-810, 823): the same port with
    # 2 OpenMP threads on 640x360 pairs (>= 2 pairs, <= 12 s), scaled by the pixel ratio (the work is linear in the pixels)
    try:
        o2 = pyoracle.OracleRIFE(rife_v2=family.startswith("rife-v2"), rife_v4=family.startswith("rife-v4"), num_threads=2)
        o2.load(modeldir)
        sa, sb = gen_frames.smooth_pair(640, 360, 1000)
        n2, t2 = 0, time.perf_counter()
        while True:
            o2.process(sa, sb, 0.5)
            n2 += 1
            d2 = time.perf_counter() - t2
            if (d2 >= 6.0 and n2 >= 2) or d2 >= 12.0:
                break
        sc2 = pixels / float(640 * 360) * passes
        res["threads_2"] = {"value": round(1.0 / (d2 / n2 * sc2), 6), "unit": "frames/s", "cores": 2,
                            "sample": "%d plain %s pair(s) at 640x360 in %.2f s with 2 OpenMP threads (the reference's default -j 1:2:2); scaled by x%.3g (pixels x TTA passes)" % (n2, family, d2, 1.0 / sc2)}
        del o2
    except Exception as e:
        res["threads_2"] = {"error": str(e)[:200]}
    try:                                                             # the reference build (oracle/_ref; prebuilt on the GPU box): one pair, same frames, same threads
        from oracle import pyref
        if pyref.available():
            rr = pyref.RefRIFE(rife_v2=family.startswith("rife-v2"), rife_v4=family.startswith("rife-v4"), num_threads=cores)
            rr.load(modeldir)
            t2 = time.perf_counter()
            out_ref = rr.process(a, b, 0.5)
            dtr = time.perf_counter() - t2
            res["reference_build"] = {"value": round(1.0 / (dtr * scale), 5), "unit": "frames/s", "cores": cores,
                                      "same_bytes_as_the_port": (bool(np.array_equal(out_ref, o.process(a, b, 0.5))) if w % 32 == 0 else
                                                                 "not compared (ragged width: the reference's CPU crop quirk, SURVEY App. F-1; tests/test_ref_build.py compares the literal mode)"),
                                      "sample": "1 pair at %dx%d in %.2f s: /root/reference/src/rife.cpp + warp.cpp compiled unmodified (oracle/refbuild), layer arithmetic = the port's" % (w, h, dtr)}
    except Exception as e:
        res["reference_build"] = {"error": str(e)[:200]}
    # secondary, labelled proxy: the same graph through PyTorch-CPU (oneDNN convolutions) as a stand-in for the optimised
    # x86 kernels of the reference's ncnn CPU path, which the naive direct convolutions of the oracle do not represent
    if family.startswith("rife-v4"):
        try:
            import torch
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from torch_graph import TorchNet
            nth = min(len(os.sched_getaffinity(0)), 64)
            torch.set_num_threads(nth)
            net = TorchNet(os.path.join(modeldir, "flownet.param"), os.path.join(modeldir, "flownet.bin"))
            wp, hp = (w + 31) // 32 * 32, (h + 31) // 32 * 32

            def chw(img):
                x = np.zeros((3, hp, wp), np.float32)
                x[:, :h, :w] = (img.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)
                return torch.from_numpy(x)
            ins = {"in0": chw(a), "in1": chw(b), "in2": torch.full((1, hp, wp), np.float32(0.5))}
            net.run(ins, ["out0"])                                   # warm-up (thread pool, oneDNN primitive cache)
            m, t1 = 0, time.perf_counter()
            while True:
                net.run(ins, ["out0"])
                m += 1
                dt2 = time.perf_counter() - t1
                if dt2 >= 5.0 or m >= 8:
                    break
            res["proxy_torch_onednn"] = {"value": round(1.0 / (dt2 / m * scale), 5), "unit": "frames/s", "cores": nth,
                                         "sample": "%d pair(s) at %dx%d in %.2f s, graph only (no u8 pre/post), tests/torch_graph.py" % (m, w, h, dt2)}
        except Exception as e:                                       # the proxy is optional; never let it break the bench line
            res["proxy_torch_onednn"] = {"error": str(e)[:200]}
    return res


if __name__ == "__main__":
    main()
