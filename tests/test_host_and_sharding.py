"""CPU-side checks: the C-ABI library loads and exports every declared symbol; host logic; the N>1 path
(world_size 2, gloo) of the sharding helpers bench.py uses."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REFERENCE, ROOT

amd = importlib.import_module("rife-ncnn-vulkan_amd")


def _exports(path):
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(set(re.findall(r"\b(rife_hip_[A-Za-z0-9_]+)$", nm, flags=re.M)))


def test_product_library_exports_exactly_the_product_header():
    """librife_hip.so == include/rife_hip.h, both directions: no parity tap, single-kernel entry point, bench / probe / ablation hook rides in the product."""
    hdr = open(os.path.join(ROOT, "include", "rife_hip.h")).read()
    assert "rife_hip_test.h" not in hdr and not re.search(r"rife_hip_(op_|v4_)", hdr)
    declared = sorted(set(re.findall(r"\b(rife_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(amd.C_ABI_SYMBOLS)
    product = os.path.join(ROOT, "rife-ncnn-vulkan_amd", "librife_hip.so")
    L = ctypes.CDLL(product)
    for s in declared:
        assert hasattr(L, s), s
    exported = _exports(product)
    assert exported == declared, sorted(set(exported) ^ set(declared))
    assert "bench" not in " ".join(exported) and "probe" not in " ".join(exported)


def test_test_build_exports_exactly_both_headers():
    """librife_hip_test.so (same sources, -DRIFE_HIP_TEST_BUILD) == include/rife_hip.h + include/rife_hip_test.h."""
    hdr = open(os.path.join(ROOT, "include", "rife_hip.h")).read()
    test_hdr = open(os.path.join(ROOT, "include", "rife_hip_test.h")).read()
    only_test = sorted(set(re.findall(r"\b(rife_hip_[a-z0-9_]+)\s*\(", test_hdr)))
    assert only_test == sorted(amd.TEST_ABI_SYMBOLS)
    declared = sorted(set(re.findall(r"\b(rife_hip_[a-z0-9_]+)\s*\(", hdr + test_hdr)))
    exported = _exports(amd.TEST_LIB_PATH)
    assert exported == declared, sorted(set(exported) ^ set(declared))


def _library_sources():
    """The sources of librife_hip*.so: engine.hip, its engine_*.h sections and the kernel headers (not the shim / CLI: rife.h, ncnn_mat.h, jpeg_codec.h, main.cpp)."""
    csrc = os.path.join(ROOT, "rife-ncnn-vulkan_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")) and f not in ("jpeg_codec.h", "ncnn_mat.h", "rife.h"):
            yield f, open(os.path.join(csrc, f)).read()


def test_env_switches_live_in_one_table():
    """ONE switch table (csrc/engine_switches.h: struct Switches, read_switches()): no other library source calls getenv, and no switch is a static object
    initialised from the environment anywhere (round 5's closure-name collision had hipGraph replay silently on for four rounds because the
    switches were namespace-scope statics initialised by lambdas spread over a 3,500-line file)."""
    files = dict(_library_sources())
    assert "engine_switches.h" in files
    for f, src in files.items():
        code = "\n".join(l.split("//")[0] for l in src.splitlines())            # comments may name the function
        n = len(re.findall(r"(?<![A-Za-z_])getenv\s*\(", code))
        if f == "engine_switches.h":
            assert n >= 4, n
            # every call sits inside read_switches()
            body = code[code.index("static Switches read_switches()"):code.index("static const Switches& process_switches()")]
            assert len(re.findall(r"(?<![A-Za-z_])getenv\s*\(", body)) == n
            # exactly one function-local static holds the process-scope values
            assert len(re.findall(r"static\s+const\s+Switches\s+\w+\s*=\s*read_switches\(\)", code)) == 1
        else:
            assert n == 0, (f, "getenv outside the switch table")
        # no static / namespace-scope object initialised from the environment or from the table's parser helpers, and no immediately-invoked lambda initialisers
        for pat in (r"static[^;\n(]*=\s*(?:env_\w+|ab_getenv|getenv)\s*\(", r"static[^;\n]*=\s*\[[^\]]*\]\s*\([^)]*\)\s*(?:->[^{]*)?\{"):
            m = re.search(pat, code)
            assert m is None or f == "engine_switches.h" and "read_switches()" in m.group(0), (f, m.group(0))
        if f != "engine_switches.h":
            assert re.search(r"static[^;\n]*=\s*(?:read_switches|process_switches)\s*\(", code) is None, (f, "a static copy of a switch")
    # the engine is split into sections of one translation unit; none of them is the 3,500-line file of round 5
    sections = [f for f in files if f.startswith("engine_")]
    assert {"engine_switches.h", "engine_layers.h", "engine_dispatch.h", "engine_ctx.h", "engine_v4.h", "engine_v2.h", "engine_v1.h", "engine_abi.h"} <= set(sections)
    master = files["engine.hip"]
    for sct in sections:
        assert master.count('#include "%s"' % sct) == 1, sct
    assert len(master.splitlines()) < 120


def test_product_ignores_the_kernel_selection_switches():
    """The A/B and kernel-selection environment switches are compiled out of the product: in the switch table every direct getenv names one of the four
    documented product variables, everything else goes through ab(), which returns null unless RIFE_HIP_TEST_BUILD is defined; the product binary holds
    neither the long switch names nor the opt-in K-split kernel."""
    allowed = {"RIFE_HIP_TRUNK", "RIFE_HIP_GRAPH", "RIFE_HIP_BATCH_WORKERS", "RIFE_HIP_PROFILE_FINE"}
    src = dict(_library_sources())["engine_switches.h"]
    seen = set()
    for m in re.finditer(r"(?<![A-Za-z_])getenv\(\s*\"?([A-Za-z0-9_]*)", "\n".join(l.split("//")[0] for l in src.splitlines())):
        if m.group(1) == "name":
            before = src[max(0, src.index("return getenv(name)") - 200):src.index("return getenv(name)")]
            assert "#ifdef RIFE_HIP_TEST_BUILD" in before                       # the body of ab() itself
            continue
        assert m.group(1) in allowed, m.group(0)
        seen.add(m.group(1))
    assert seen == allowed
    blob = open(os.path.join(ROOT, "rife-ncnn-vulkan_amd", "librife_hip.so"), "rb").read()
    for name in (b"RIFE_HIP_FUSE_FLOW", b"RIFE_HIP_POOL_PARTS", b"RIFE_HIP_BATCH_GROUPS", b"RIFE_HIP_TTA_CONSENSUS", b"RIFE_HIP_V2_FUSED_STEM", b"RIFE_HIP_RS2", b"conv_ks_kernel"):
        assert name not in blob, name
    test_blob = open(amd.TEST_LIB_PATH, "rb").read()
    assert b"RIFE_HIP_FUSE_FLOW" in test_blob and b"conv_ks_kernel" in test_blob


def test_no_cpu_fallback_without_a_device():
    """The product path must fail loudly, not fall back, when there is no GPU."""
    if amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(amd.RifeError):
        amd.RIFE(0, rife_v4=True)
    with pytest.raises(amd.RifeError):
        amd.test_build().RIFE(0, rife_v4=True)
    with pytest.raises(amd.RifeError):
        amd.op_warp(np.zeros((3, 4, 4), np.float32), np.zeros((2, 4, 4), np.float32))


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rife-ncnn-vulkan_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only in the build container")
def test_engine_accepts_the_reference_param_and_rejects_others():
    from tools.param_hash import param_hash
    hdr = open(os.path.join(ROOT, "rife-ncnn-vulkan_amd", "csrc", "model_hashes.h")).read()
    want = int(re.search(r"RIFE_V46_HASH_OUT0\s+(0x[0-9a-f]+)ull", hdr).group(1), 16)
    assert param_hash(os.path.join(REFERENCE, "models", "rife-v4.6", "flownet.param"), "out0") == want
    assert param_hash(os.path.join(REFERENCE, "models", "rife-v4", "flownet.param"), "out0") != want


def test_shard_pairs_partitions_the_work():
    from importlib import import_module
    sh = import_module("rife-ncnn-vulkan_amd.sharding")
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += sh.shard_pairs(37, r, world)
        assert sorted(seen) == list(range(37))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sh = importlib.import_module("rife-ncnn-vulkan_amd.sharding")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import time
    mine = sh.shard_pairs(10, rank, world)
    done = []
    def step(i):
        done.append(mine[i % len(mine)])
        time.sleep(0.01 * (rank + 1))          # rank 1 is slower: MAX must report its time on both ranks
    el = sh.timed_steps(step, 5, dist=dist)
    q.put((rank, mine, el))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]
    assert abs(res[0][2] - res[1][2]) < 1e-9          # both ranks report the same (max) elapsed time
    assert res[0][2] >= 5 * 0.02 * 0.9                # ... which is the slow rank's


def build_shim_demo(outdir):
    """Compile the C++ driver that uses `class RIFE` the way the reference's proc thread does."""
    import subprocess
    amd.build()
    exe = os.path.join(str(outdir), "shim_demo")
    pkg = os.path.join(ROOT, "rife-ncnn-vulkan_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(pkg, "csrc"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"),
                           "-o", exe, "-L" + pkg, "-lrife", "-lrife_hip", "-Wl,-rpath," + pkg])
    return exe


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=240, env=e)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must run 2 ranks and say so: n_gpus = rccl_ranks = 2 (the flag was dead)."""
    import json
    r = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    r1 = _bench("--steps", "3", "--warmup", "0", "--dry-run")
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_bench_dry_run_covers_the_all_ranks_host_leg():
    """N > 1: the host-buffer leg runs on every rank inside ONE barrier-bracketed region (bench.py, sharding.all_ranks_rate): the whole-job rate is
    the pairs of all ranks over the SLOWEST rank's time.  Dry run: rank r sleeps (r + 1) x 10 ms for its K pairs, so 2 ranks x 4 pairs take >= 20 ms."""
    import json
    r = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    leg = d["extra"]["all_ranks_leg"]
    assert leg["units"] == 8 and leg["max_elapsed_s"] >= 0.02 * 0.9
    assert abs(leg["units_per_s"] - 8 / leg["max_elapsed_s"]) / leg["units_per_s"] < 0.02
    assert d["extra"]["numa"] == {"pci": "0000:00:00.0", "numa_node": None, "cpus": None, "pinned": False}


def test_numa_node_of_a_gpu_comes_from_sysfs(tmp_path):
    """Rank placement reads the GPU's NUMA node from /sys/class/drm/card*/device/numa_node - no hard-coded topology (fake sysfs tree here)."""
    sh = importlib.import_module("rife-ncnn-vulkan_amd.sharding")
    sysfs = tmp_path / "sys"
    for card, addr, node in ((0, "0000:05:00.0", 0), (1, "0000:c1:00.0", 1), (2, "0000:d5:00.0", -1)):
        pci = sysfs / "devices" / "pci" / addr
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text("%d\n" % node)
        drm = sysfs / "class" / "drm" / ("card%d" % card)
        drm.mkdir(parents=True)
        os.symlink(str(pci), str(drm / "device"))
    (sysfs / "class" / "drm" / "card1-DP-1").mkdir()                      # connectors sit next to the cards: must be ignored
    for node, cpus in ((0, "0-3,16-19"), (1, "4-7,20-23")):
        nd = sysfs / "devices" / "system" / "node" / ("node%d" % node)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cpus + "\n")
    assert sh.numa_node_of_pci("0000:C1:00.0", str(sysfs)) == 1
    assert sh.numa_node_of_pci("0000:05:00.0", str(sysfs)) == 0
    assert sh.numa_node_of_pci("0000:d5:00.0", str(sysfs)) is None        # the kernel says -1: unknown
    assert sh.numa_node_of_pci("0000:ff:00.0", str(sysfs)) is None
    assert sh.cpus_of_node(1, str(sysfs)) == {4, 5, 6, 7, 20, 21, 22, 23}
    pinned = []
    rep = sh.pin_to_gpu_numa("0000:c1:00.0", str(sysfs), setaffinity=pinned.append, allowed=range(0, 22))
    assert rep["numa_node"] == 1 and rep["pinned"] and rep["cpus"] == 6 and pinned == [{4, 5, 6, 7, 20, 21}]
    pinned.clear()
    rep = sh.pin_to_gpu_numa("0000:d5:00.0", str(sysfs), setaffinity=pinned.append, allowed=range(64))
    assert not rep["pinned"] and pinned == []


def test_bench_refuses_a_launcher_world_that_differs_from_gpus():
    r = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_cpp_class_shim_compiles_and_fails_cleanly_without_gpu(tmp_path):
    import subprocess
    exe = build_shim_demo(tmp_path)
    if amd.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    (tmp_path / "a.rgb").write_bytes(bytes(4 * 4 * 3))
    r = subprocess.run([exe, "/nonexistent", "4", "4", "0.5", str(tmp_path / "a.rgb"), str(tmp_path / "a.rgb"), str(tmp_path / "o.rgb")],
                       capture_output=True, text=True)
    assert r.returncode == 3                      # load() reported failure, no crash, no fallback
    assert "RIFE" in r.stderr


def test_graph_executor_has_a_kernel_for_every_layer_of_the_generated_graphs(modeldirs):
    """CPU-only dry run of the generic graph executor's loader (rife_hip_graph_check) on every generated graph."""
    import os
    for fam, d in modeldirs.items():
        for net in ("flownet", "contextnet", "fusionnet"):
            if os.path.exists(os.path.join(d, net + ".param")):
                amd.graph_check(os.path.join(d, net))
    with pytest.raises(amd.RifeError):
        amd.graph_check(os.path.join(modeldirs["rife-v4"], "no_such_net"))


def test_graph_executor_covers_the_reference_model_zoo():
    import os
    from conftest import REFERENCE
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference tree only exists in the build container")
    for fam in sorted(os.listdir(os.path.join(REFERENCE, "models"))):
        for net in ("flownet", "contextnet", "fusionnet"):
            base = os.path.join(REFERENCE, "models", fam, net)
            if os.path.exists(base + ".param"):
                amd.graph_check(base)


def test_malformed_param_files_are_refused_not_fatal(modeldirs, tmp_path):
    """A damaged .param must come back as an error through the C-ABI (never an exception or a crash across it): cyclic graphs, layer lines
    that end early, shifted fields, absurd counts - found by fuzzing rife_hip_graph_check / NcnnModel under ASan; seeded mutations here."""
    import random
    ok = refused = 0
    for fam, net in (("rife", "flownet"), ("rife-HD", "fusionnet"), ("rife-v4.6", "flownet"), ("rife-v2.3", "contextnet")):
        txt = open(os.path.join(modeldirs[fam], net + ".param"), "rb").read()
        rng = random.Random(len(txt))
        cases = []
        for _ in range(60):
            b = bytearray(txt)
            for _ in range(rng.choice((1, 2, 4, 8))):
                b[rng.randrange(len(b))] = rng.choice(b"0123456789 -=\n,.ae")
            cases.append(bytes(b))
        lines = txt.split(b"\n")
        cases.append(b"\n".join(lines[:2] + [lines[5].replace(lines[5].split()[-1 - sum(b"=" in t for t in lines[5].split())], lines[5].split()[4])] + lines[2:]))   # a layer that feeds itself
        cases.append(b"\n".join(l[: len(l) // 2] if i == 7 else l for i, l in enumerate(lines)))                                                             # a line cut in half
        cases.append(txt.replace(b" 1 1 ", b" 1000000000 1 ", 1))                                                                                             # absurd input count
        for c in cases:
            (tmp_path / "g.param").write_bytes(c)
            try:
                amd.graph_check(str(tmp_path / "g"))
                ok += 1
            except amd.RifeError:
                refused += 1
    assert refused > 100 and ok + refused == 4 * 63


def test_pmc_tables_reads_the_committed_summaries(tmp_path):
    """tools/pmc_tables.py on the per-kernel PMC summaries kept in profiles/r4 and profiles/r6, for every bench workload: the derived files are complete and the
    dominant kernel's traffic is what bench.py falls back to for roofline.traffic when it cannot measure it live."""
    import json, shutil
    from tools import pmc_tables
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # round 4: conv_rs_kernel, one trunk layer per launch; round 6: conv_rs2_kernel, two layers per launch (one input + one output tensor: the same stored bytes)
    for rnd, wl, dom, floor in (("r4", "4k", "conv_rs_kernel", 2.6e8), ("r4", "1080p", "conv_rs_kernel", 6.6e7), ("r4", "v23_1080p", "conv_h2_kernel<3, 9, 0>", 5e7),
                                ("r4", "4k_tta", "conv_rs_kernel", 2.6e8), ("r6", "4k", "conv_rs2_kernel", 2.6e8), ("r6", "1080p", "conv_rs2_kernel", 6.6e7),
                                ("r6", "v23_1080p", "conv_h2_kernel<3, 9, 0>", 5e7), ("r6", "4k_tta", "conv_rs2_kernel", 2.6e8)):
        src = tmp_path / ("src_%s_%s" % (rnd, wl)); src.mkdir()
        sfx = "" if wl == "4k" else "_" + wl
        for short, first in pmc_tables.PASSES.items():
            shutil.copy(os.path.join(root, "profiles", rnd, "pmc_all_kernels_%s_%s.txt" % (wl, short)), src / ("pmc_%s%s_all.txt" % (first, sfx)))
        dst = tmp_path / ("dst_%s_%s" % (rnd, wl))
        pmc_tables.main(str(src), str(dst), 1 if wl == "4k_tta" else 3, wl)
        j = json.load(open(dst / ("pmc_%s.json" % wl)))
        committed = json.load(open(os.path.join(root, "profiles", rnd, "pmc_%s.json" % wl)))
        assert j["hbm_bytes_per_launch"] == committed["hbm_bytes_per_launch"] > floor and dom in j["kernel"]      # at least what the layer stores
        assert committed["source"].startswith("profiles/%s/" % rnd)
        table = open(dst / ("bandwidth_kernels_%s.txt" % wl)).read()
        assert dom.split("<")[0] in table
        derived = [l for l in open(dst / ("pmc_trunk_kernels_%s.txt" % wl)) if "matrix pipe busy" in l]
        assert len(derived) >= 2


def test_live_traffic_degrades_to_a_reason_without_rocprofv3(monkeypatch):
    """bench.py measures roofline.traffic with rocprofv3 child passes; where the profiler is not usable it must come back with (None, reason),
    never raise (the bench line then carries the committed figure, labelled)."""
    import bench
    monkeypatch.setenv("PATH", "/nonexistent")
    tb, why = bench.live_traffic("4k", "rife-v4.6", timeout_s=5)
    assert tb is None and "rocprofv3" in why


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_exercise_the_all_ranks_host_leg():
    """`bench.py --gpus 2 --share-gpu`: the N-rank code paths on real HIP work (both ranks on device 0, gloo rendezvous): sharding, barriers, the
    all-ranks host-buffer leg with NUMA pinning from sysfs.  A plumbing check - the line says so - not a scaling measurement."""
    import json
    r = _bench("--gpus", "2", "--share-gpu", "--workload", "1080p", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic", "--no-extra")
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "TEST MODE" in d["config"]["parallelism"]
    leg = d["extra"]["frames_per_s_host_buffers_all_ranks"]
    assert leg["pageable"]["frames_per_s"] > 0 and leg["page_locked"]["frames_per_s"] > 0
    assert abs(leg["pageable"]["host_device_GBps"] - leg["pageable"]["frames_per_s"] * 3 * 1920 * 1080 * 3 / 1e9) < 0.05
    assert d["extra"]["numa"] is not None and d["extra"]["numa"]["pci"].count(":") == 2
