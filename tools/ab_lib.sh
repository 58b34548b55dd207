#!/bin/bash
# same-box alternating A/B of scratch libraries on the metric step: tools/ab_lib.sh <rounds> lib_a.so lib_b.so ...
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    python3 tools/bench_with_lib.py $lib --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --no-extra-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
