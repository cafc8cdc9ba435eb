"""Turn a rocprofv3 results .db (kernel trace and/or PMC) into a small text summary for profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    lines = []
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "top_kernels" in tabs:
        lines.append("# kernel trace: name | calls | total_us | avg_us | pct")
        for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append("%-150s | %6d | %12.1f | %10.2f | %6.2f" % (name[:150], calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if avg > 1e5 else avg, pct))
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event") and t.count("_") == 2] or [t for t in tabs if t == "pmc_events"]
    if "counters_collection" in tabs:
        cur = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in cur.description]
        lines.append("# counters_collection columns: " + ",".join(cols))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
