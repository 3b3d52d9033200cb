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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
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
