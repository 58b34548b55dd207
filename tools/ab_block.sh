#!/bin/bash
# same-box A/B of scratch libraries on the two block workloads: tools/ab_block.sh <rounds> <family substring> lib_a.so lib_b.so ...
rounds=$1; fam=$2; shift 2
for r in $(seq $rounds); do for lib in "$@"; do for w in block_cls block_seg; do
python3 tools/bench_with_lib.py $lib --workload $w --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', d['ms_per_step'], {k:v for k,v in d['kernel_family_ms_per_step'].items() if '$fam' in k})"
done; done; done
