import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/*/runc/*kernel_stats.csv'))[-1]
tot = 0
for r in csv.DictReader(open(f)):
    print(f"{float(r['AverageNs'])/1e3:9.2f} us x{r['Calls']:>5} min {float(r['MinNs'])/1e3:8.2f}  {r['Name'][:70]}")
