#!/bin/bash
# same-box alternating A/B of scratch libraries on the metric step WITH the per-kernel breakdown:
#   tools/ab_kernels.sh <rounds> lib_a.so lib_b.so ...
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    python3 tools/bench_with_lib.py $lib --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_us',{})
print('$lib', 'ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), {n: k[n] for n in ('attn_rows','bwd_dq','bwd_dv','bwd_dk','attn_stats','knn') if n in k})"
  done
done
