"""Per-dispatch timeline of ONE bench frame from a rocprofv3 kernel trace CSV.
    python tools_dev/frame_timeline.py gpurun_out/<run>/prof/b_kernel_trace.csv [--all]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw3_to_nhwc4' in r['Kernel_Name']]
s, e = idx[-3], idx[-2]
t0 = int(rows[s]['Start_Timestamp'])
tot = 0; cat = {}
for r in rows[s:e]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    nm = r['Kernel_Name']
    key = nm.split('(')[0].replace('void ', '').replace('fpc::', '')[:40]
    cat[key] = cat.get(key, 0) + d
    if '--all' in sys.argv:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.2f} grid=({int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {nm[:70]}")
print('sum of kernels %.1f us, span %.1f us, %d dispatches' % (tot, (int(rows[e]['Start_Timestamp']) - t0) / 1e3, e - s))
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print(f"{v:9.1f} us  {k}")
