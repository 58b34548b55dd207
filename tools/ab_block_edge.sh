for r in 1 2; do for lib in tools/scratch/nt/lib_edgef32.so tools/scratch/nt/lib_ship.so; do for w in block_cls block_seg; do
python3 tools/bench_with_lib.py $lib --workload $w --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', d['ms_per_step'], {k:v for k,v in d['kernel_family_ms_per_step'].items() if 'edge' in k})"
done; done; done
