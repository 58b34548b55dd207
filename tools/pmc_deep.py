"""Per-kernel instruction mix from tools/pmc_deep.sh: counters per launch, then per wave."""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_deep*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        per[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
want = [k for k in per if k.startswith(("samble::", "void samble::"))]
def avg(k, c):
    v = per[k].get(c)
    return sum(v) / len(v) if v else float("nan")
want.sort(key=lambda k: -avg(k, "SQ_BUSY_CYCLES"))
for k in want[:8]:
    waves = avg(k, "SQ_WAVES")
    print(f"== {k}   waves/launch {waves:.0f}")
    for c in sorted(per[k]):
        a = avg(k, c)
        print(f"   {c:32s} {a:16.0f}   per wave {a / waves if waves == waves and waves else float('nan'):12.1f}")
