"""Per (kernel, grid) mean duration from a rocprofv3 --kernel-trace CSV: which launches of a block step run on a fraction of
the chip.   python tools/kernel_by_grid.py <dir with *_kernel_trace.csv> [steps traced]"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(wg, 1)
    a = agg[(name, grid, wg)]
    a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print(f"{'kernel':60s} {'wgs':>7s} {'thr':>5s} {'calls/step':>10s} {'us/launch':>10s} {'us/step':>9s}")
for (name, grid, wg), (t, n) in rows[:70]:
    print(f"{name:60s} {grid:7d} {wg:5d} {n / steps:10.2f} {t / n:10.1f} {t / steps:9.1f}")
