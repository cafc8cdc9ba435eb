"""Frame-pair sharding across the GPUs of one node.

The reference's multi-GPU mode is N independent replicas pulling frame pairs from one queue
(src/main.cpp:248-295, 819-866): no tensor crosses devices.  Here that is one process per GPU
(torch.distributed, RCCL over xGMI when on GPUs, gloo in the CPU tests); pairs are assigned statically
(pair i -> rank i mod world) and the only collectives are the start/stop barrier and a MAX over the ranks'
wall-clock — nothing on the data path.
"""
import glob
import os
import time


def shard_pairs(n_pairs, rank, world):
    """Indices of the frame pairs rank `rank` of `world` processes: i with i % world == rank."""
    return list(range(rank, n_pairs, world))


def barrier(dist, device_sync=None):
    if device_sync is not None:
        device_sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        if device_sync is not None:
            device_sync()


def timed_steps(step_fn, steps, first_index=0, dist=None, device_sync=None, make_tensor=None):
    """Run `steps` calls of step_fn(i) bracketed by barrier + device sync on both sides; returns the MAX over
    ranks of the elapsed seconds (every rank gets the same number)."""
    barrier(dist, device_sync)
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(first_index + i)
    if device_sync is not None:
        device_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        import torch
        t = (make_tensor or torch.tensor)([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier(dist, device_sync)
    return elapsed


def all_ranks_rate(units_this_rank, run_fn, dist=None, device_sync=None, make_tensor=None):
    """One barrier-bracketed region in which EVERY rank runs run_fn() concurrently (e.g. rife_hip_process_batch over host frames): returns
    (units of all ranks / MAX elapsed over ranks, MAX elapsed) - the whole-job rate of a leg that can contend for a shared resource (host DRAM,
    PCIe root complexes), which N independent single-rank measurements cannot show."""
    barrier(dist, device_sync)
    t0 = time.perf_counter()
    run_fn()
    if device_sync is not None:
        device_sync()
    elapsed = time.perf_counter() - t0
    total = float(units_this_rank)
    if dist is not None and dist.is_initialized():
        import torch
        mk = make_tensor or torch.tensor
        t = mk([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = mk([total], dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total = float(u.item())
    barrier(dist, device_sync)
    return total / elapsed, elapsed


# ---- NUMA placement of a rank: the host threads that feed a GPU (callers, staging copies) belong on the NUMA node its PCIe root hangs off.
# Nothing is hard-coded: the node comes from /sys/class/drm/card*/device/numa_node of the card with the GPU's PCI address.
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_node_of_pci(pci_address, sysfs="/sys"):
    """NUMA node of the DRM card whose PCI address is `pci_address` ("0000:c1:00.0", case-insensitive); None if unknown (no such card,
    no numa_node file, or the kernel reports -1)."""
    want = pci_address.strip().lower()
    for dev in sorted(glob.glob(os.path.join(sysfs, "class", "drm", "card[0-9]*", "device"))):
        try:
            addr = os.path.basename(os.path.realpath(dev)).lower()
            if addr != want:
                continue
            node = int(open(os.path.join(dev, "numa_node")).read().strip())
            return node if node >= 0 else None
        except (OSError, ValueError):
            continue
    return None


def cpus_of_node(node, sysfs="/sys"):
    try:
        return _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")).read())
    except OSError:
        return set()


def pin_to_gpu_numa(pci_address, sysfs="/sys", setaffinity=None, allowed=None):
    """Restrict the calling process to the CPUs of the GPU's NUMA node (intersected with what it may already use).  Returns a small report
    dict for the bench line; never raises: on any doubt the affinity is left alone."""
    rep = {"pci": pci_address, "numa_node": None, "cpus": None, "pinned": False}
    try:
        node = numa_node_of_pci(pci_address, sysfs)
        rep["numa_node"] = node
        if node is None:
            return rep
        allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
        cpus = cpus_of_node(node, sysfs) & allowed
        rep["cpus"] = len(cpus)
        if cpus:
            (setaffinity or (lambda c: os.sched_setaffinity(0, c)))(cpus)
            rep["pinned"] = True
    except Exception as e:          # affinity is an optimisation, not a requirement
        rep["error"] = str(e)
    return rep
