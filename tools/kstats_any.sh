#!/bin/bash
# per-kernel stats of an arbitrary python script: tools/kstats_any.sh <tag> <script.py> [args]
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- python3 "$@" > "$out/${tag}_stats.log" 2>&1
python3 - "$out/${tag}_stats" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:30]:
    print(f'{r["Name"].split("(")[0][:72]:72s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
PY
