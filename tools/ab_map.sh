#!/bin/bash
# A/B on one box: map-free forward (default) vs the logit-map pipeline, metric and stress workloads
mkdir -p gpurun_out
for w in metric stress; do
  for f in "" "--logit-map"; do
    echo "== $w $f"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --workload $w $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['value'])"
  done
done
