"""GPU box: socket power and shader clock while each bench.py workload runs (resident frames, the stream layout of its bench line).
The parent samples the hwmon files every 50 ms while a child runs `bench.py --workload W --steps N` long enough for a few seconds of timed region; the samples
of the loaded phase (power within 15 % of the run's maximum: the hwmon figure is a slow average, its plateau is the steady state) are summarised.  Answers one question per workload: is it running at the board's power cap?
    python tools/power_workloads.py [seconds per workload]"""
import glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def find(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


PW = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
FQ = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
CAP = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")
cap = int(open(CAP).read()) / 1e6 if CAP else None
print("power file %s, clock file %s, power cap %s W" % (PW, FQ, cap), flush=True)
# steps per second of timed region, roughly (so that the region lasts `secs`)
RATE = {"4k": 480, "1080p": 1750, "v23-1080p": 570, "4k-tta": 26}
for wl in ("4k", "1080p", "v23-1080p", "4k-tta"):
    steps = max(20, int(RATE[wl] * secs))
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", str(steps), "--warmup", "8", "--no-cpu-baseline", "--no-extra",
                          "--no-live-traffic", "--no-configs", "--no-host-path", "--no-sustained"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples = []
    while p.poll() is None:      # rocm-smi (~0.4 s per call): "Current Socket Graphics Package Power" and sclk, the figures profiles/r4 quoted; the hwmon power1_input file is a slower average
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw_ = [float(l.split(":")[-1]) for l in t.splitlines() if "Socket Graphics Package Power" in l]
            fq_ = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk clock level" in l]
            if pw_ and fq_:
                samples.append((time.perf_counter(), pw_[0], fq_[0]))
        except Exception:
            pass
    out = p.stdout.read()
    try:
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        val = "%.1f %s, %d pairs in flight" % (d["value"], d["unit"], d.get("config", {}).get("pairs_in_flight", -1))
    except Exception:
        val = "bench line not parsed (rc %s)" % p.returncode
    if not samples:
        print("%-10s no samples" % wl); continue
    pmax = max(s[1] for s in samples)
    hot = sorted(s for s in samples if s[1] >= 0.85 * pmax)      # the hwmon power is a slow average: its plateau is the steady state
    pw = sorted(s[1] for s in hot); fq = sorted(s[2] for s in hot)
    med = lambda v: v[len(v) // 2]
    print("%-10s %s | loaded phase: %d of %d samples, power median %.0f W (p10 %.0f, p90 %.0f, max %.0f), shader clock median %.0f MHz (p10 %.0f, p90 %.0f)"
          % (wl, val, len(hot), len(samples), med(pw), pw[len(pw) // 10], pw[len(pw) * 9 // 10], pmax, med(fq), fq[len(fq) // 10], fq[len(fq) * 9 // 10]), flush=True)
