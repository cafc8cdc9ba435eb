"""Frame-pair sharding across the GPUs of one node.

The reference's multi-GPU mode is N independent replicas pulling frame pairs from one queue
(src/main.cpp:248-295, 819-866): no tensor crosses devices.  Here that is one process per GPU
(torch.distributed, RCCL over xGMI when on GPUs, gloo in the CPU tests); pairs are assigned statically
(pair i -> rank i mod world) and the only collectives are the start/stop barrier and a MAX over the ranks'
wall-clock — nothing on the data path.
"""
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
