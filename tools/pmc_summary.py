"""Aggregate rocprofv3 counter_collection CSVs per kernel: mean counter value per dispatch (+ mean duration)."""
import collections
import csv
import sys


def summarize(path, match=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for k in acc:
        d = {c: sum(v) / len(v) for c, v in acc[k].items()}
        d["_dispatches"] = len(dur[k])
        d["_avg_ns"] = sum(dur[k]) / len(dur[k])
        out.append((k, d))
    return out


if __name__ == "__main__":
    for k, d in summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None):
        print(k[:110])
        for c, v in sorted(d.items()):
            print("    %-32s %18.1f" % (c, v))
