"""Mean duration per (kernel, grid) of a rocprofv3 --kernel-trace CSV (first launch of each group dropped as warm-up)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
g = collections.OrderedDict()
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg, last = 0, None
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    if pat and pat not in name:
        continue
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if (name, grid) != last:      # consecutive launches of one (kernel, grid) form a segment
        seg += 1
        last = (name, grid)
    g.setdefault((name[:70], grid, wg, seg), []).append(d)
for (name, grid, wg, _), ds in g.items():
    use = ds[1:] if len(ds) > 1 else ds
    print(f"{sum(use) / len(use):10.1f} us  x{len(ds):3d}  grid {grid:>9s} wg {wg:>4s}  {name}")
