"""Per-kernel time of ONE steady-state training step from a rocprofv3 kernel trace of `bench.py --train`:
the window between the last two optimiser launches.  python tools_dev/train_step_stats.py <kernel_trace.csv> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'lookahead_radam' in r['Kernel_Name']]
# several shard launches per step: group launches closer than 2 ms
groups = []
for i in opt:
    t = int(rows[i]['Start_Timestamp'])
    if not groups or t - groups[-1][-1][1] > 2e6:
        groups.append([])
    groups[-1].append((i, t))
a, b = groups[-2][-1][0] + 1, groups[-1][-1][0] + 1
span = (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
cat = {}
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'].replace('void ', '')[:90]
    c = cat.setdefault(k, [0, 0.0])
    c[0] += 1; c[1] += d; tot += d
print(f"one step: {b - a} dispatches, kernel time {tot / 1e3:.2f} ms, span {span / 1e3:.2f} ms")
for k, (n, d) in sorted(cat.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{d:9.1f} us {100 * d / tot:5.1f}% x{n:4d}  {k}")
