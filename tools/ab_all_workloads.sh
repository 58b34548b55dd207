for r in 1 2; do for lib in ${LIBS:-tools/scratch/nt/lib_a.so tools/scratch/nt/lib_b.so}; do
for w in metric stress; do python3 tools/bench_with_lib.py $lib --workload $w --steps 10 --warmup 4 --no-cpu-baseline --no-breakdown --no-extra-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', d['ms_per_step'])"; done
for w in block_cls block_seg; do python3 tools/bench_with_lib.py $lib --workload $w --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', d['ms_per_step'])"; done
done; done
