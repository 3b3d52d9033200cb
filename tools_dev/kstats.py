"""Per-kernel summary of a rocprofv3 run.  Accepts either the `*_kernel_stats.csv` written by
`rocprofv3 --kernel-trace --stats --output-format csv` or the rocpd `*_results.db` that is
ROCm 7.2's default output, and prints / writes the same table (name, calls, total, avg, min, max, %).

    python tools_dev/kstats.py <file> [--out profiles/xxx.csv] [--top 40]
"""
import argparse
import csv
import sqlite3


def rows_from_db(path):
    c = sqlite3.connect(path)
    q = ("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         "from kernels group by name order by 3 desc")
    return [dict(Name=r[0], Calls=r[1], TotalDurationNs=r[2], AverageNs=r[3], MinNs=r[4], MaxNs=r[5])
            for r in c.execute(q)]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append(dict(Name=r["Name"], Calls=int(r["Calls"]), TotalDurationNs=float(r["TotalDurationNs"]),
                        AverageNs=float(r["AverageNs"]), MinNs=float(r["MinNs"]), MaxNs=float(r["MaxNs"])))
    return sorted(out, key=lambda r: -r["TotalDurationNs"])


def rows_from_trace(path, last_frames, skip_last=0):
    """Steady state only: the dispatches of the last `last_frames` frames of a `*_kernel_trace.csv`
    (a frame starts with k_nchw3_to_nhwc4) before the final `skip_last` ones, i.e. without the plans' autotuning passes."""
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
    starts = [i for i, e in enumerate(ev) if "nchw3_to_nhwc4" in e[2]]
    ev = ev[starts[-last_frames - skip_last - 1]:starts[-skip_last - 1]]
    agg = {}
    for s, e, n in ev:
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = float(e - s)
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    out = [dict(Name=n, Calls=a[0], TotalDurationNs=a[1], AverageNs=a[1] / a[0], MinNs=a[2], MaxNs=a[3]) for n, a in agg.items()]
    return sorted(out, key=lambda r: -r["TotalDurationNs"]), last_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--last-frames", type=int, default=0, help="with a *_kernel_trace.csv: only the last N frames")
    ap.add_argument("--skip-last", type=int, default=0, help="... ending this many frames before the end of the trace")
    a = ap.parse_args()
    if a.last_frames:
        rows, nf = rows_from_trace(a.path, a.last_frames, a.skip_last)
        print(f"steady state: {nf} frames of the trace, ending {a.skip_last} frames before its end")
    else:
        rows = rows_from_db(a.path) if a.path.endswith(".db") else rows_from_csv(a.path)
    tot = sum(r["TotalDurationNs"] for r in rows) or 1.0
    for r in rows:
        r["Percentage"] = 100.0 * r["TotalDurationNs"] / tot
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r['Calls'] for r in rows)} dispatches")
    for r in rows[:a.top]:
        print(f"{r['TotalDurationNs'] / 1e3:10.1f} us {r['Percentage']:5.1f}% x{r['Calls']:>5} "
              f"avg {r['AverageNs'] / 1e3:8.2f} min {r['MinNs'] / 1e3:8.2f} max {r['MaxNs'] / 1e3:8.2f}  {r['Name'][:90]}")
    if a.out:
        with open(a.out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            w.writeheader()
            for r in rows:
                w.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == "__main__":
    main()
