"""Concurrency analysis of a rocprofv3 kernel trace (streaming bench): per kernel name, mean duration, and the
mean number of OTHER kernels running at the same time; plus the distribution of concurrency over wall time.
    python tools_dev/overlap.py <kernel_trace.csv> [t_from_frac t_to_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
       int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) for r in rows]
ev.sort()
T0, T1 = ev[0][0], max(e[1] for e in ev)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
a, b = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
win = [e for e in ev if e[0] >= a and e[1] <= b]
# sweep
pts = []
for s, e, n, q, g in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter(); cur = 0; last = pts[0][0]
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("wall-time share by number of kernels running:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
def short(n):
    return n.split('(')[0].replace('void ', '').replace('fpc::', '')[:44]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, g in win:
    k = (short(n), g)
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
print(f"{'kernel':46s} {'WGs':>7s} {'calls':>6s} {'mean us':>9s} {'total us':>10s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:46s} {k[1]:7d} {v[0]:6d} {v[1]/v[0]:9.2f} {v[1]:10.1f}")
nfr = sum(1 for e in win if 'nchw3_to_nhwc4' in e[2])
print(f"frames in window: {nfr}, window {1e-3*(b-a):.0f} us -> {1e-3*(b-a)/max(1,nfr):.1f} us/frame; sum of kernel time per frame {sum(v[1] for v in agg.values())/max(1,nfr):.1f} us")
