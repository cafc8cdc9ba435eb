"""Stress version of tests/test_gpu_edge_sizes.py::test_process_is_reentrant_for_every_family for one family: concurrent process() calls with mixed
frame sizes on one engine, many rounds; prints how many (round, job) results differ from the serial ones.  usage: reentrancy_stress.py [family] [rounds]"""
import importlib, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_frames, gen_models
amd = importlib.import_module("rife-ncnn-vulkan_amd")
fam = sys.argv[1] if len(sys.argv) > 1 else "rife-v4"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = amd.RIFE(0, rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
g.load(gen_models.ensure(None, fam))
jobs = [gen_frames.smooth_pair(w, h, 60 + i) for i, (w, h) in enumerate([(192, 128), (128, 64), (192, 128), (256, 128)])]
want = [g.process(a, b, 0.5) for a, b in jobs]
again = [g.process(a, b, 0.5) for a, b in jobs]
print(fam, "serial repeat identical:", all(np.array_equal(x, y) for x, y in zip(want, again)))
got = [None] * len(jobs)
def work(i):
    got[i] = g.process(jobs[i][0], jobs[i][1], 0.5)
bad = []
for r in range(rounds):
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(len(jobs)):
        if not np.array_equal(got[i], want[i]):
            d = np.abs(got[i].astype(int) - want[i].astype(int))
            bad.append((r, i, int(d.max()), int((d > 0).sum())))
print(fam, "rounds", rounds, "mismatches", len(bad), bad[:8])
