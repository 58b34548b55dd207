#!/bin/bash
# quick per-kernel timing of one bench run: tools/kstats.sh <tag> [bench args]
tag=${1:-k}; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- python3 bench.py --steps 20 --warmup 5 --prewarm-steps 0 --no-cpu-baseline --no-breakdown "$@" > "$out/${tag}_stats.log" 2>&1
python3 - "$out/${tag}_stats" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"].split("(")[0][:60]:60s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
print("total per step (25 steps): %.3f ms" % (tot / 25 / 1e6))
PY
