#!/bin/bash
# per-kernel timing of the whole FeatureLearningBlock step (tools/bench_block.py)
tag=${1:-kb}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- python3 tools/bench_block.py > "$out/${tag}_stats.log" 2>&1
python3 - "$out/${tag}_stats" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f'{r["Name"].split("(")[0][:70]:70s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
print("total kernel time: %.1f ms" % (tot / 1e6))
PY
