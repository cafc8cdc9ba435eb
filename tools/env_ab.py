"""GPU box: A/B of a PROCESS-LEVEL environment switch of the engine (read once per process, e.g. RIFE_HIP_RS_SPLIT): every variant runs in its own
child process, alternating, in one call: per-class kernel time per pair (one pair in flight), frames/s with three pairs in flight on resident
frames, md5 of the frame, run-to-run determinism.
    python tools/env_ab.py VAR=a VAR=b [...]      -> gpurun_out/env_ab.txt"""
import json, os, subprocess, sys
if __name__ == "__main__":
    variants = sys.argv[1:]
    os.makedirs("gpurun_out", exist_ok=True)
    log = open("gpurun_out/env_ab.txt", "a")
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), "rife-ncnn-vulkan_amd", "librife_hip_test.so")      # the switches live in the test build; the product ignores them
    for rnd in range(2):
        for v in variants:
            env = dict(os.environ)
            for kv in v.split(","):
                k, _, val = kv.partition("=")
                env[k] = val
            p = subprocess.run([sys.executable, os.path.join(here, "lib_ab.py"), "--child", lib, "10"], capture_output=True, text=True, env=env)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                s = "%s: FAILED %s" % (v, p.stderr[-1500:])
            else:
                r = json.loads(line[0][7:])
                s = ""
                for size, d in r.items():
                    top = ", ".join("%s %.3f" % kv for kv in sorted(d["ms"].items(), key=lambda kv: -kv[1])[:8])
                    s += "%s %s round %d: %.1f / %.1f frames/s (3 in flight), kernel ms/pair %.3f, md5 %s, %d of 10 repeats differ | %s\n" % (
                        v, size, rnd, d["fps3"][0], d["fps3"][1], sum(d["ms"].values()), d["md5"][:8], d["nondeterministic_repeats"], top)
            print(s.rstrip(), flush=True); log.write(s.rstrip() + "\n"); log.flush()
