#!/bin/bash
# The two block workloads of BASELINE.json (configs[1], configs[2]): bench line + rocprofv3 kernel table of the same
# command, into profiles/<tag>_block_{cls,seg}_{bench.json,kernel_stats.csv}: tools/profile_block.sh r03
tag=${1:-r03}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p "$out" profiles
for w in cls seg; do
  python3 bench.py --workload block_$w --steps 10 --warmup 4 > "$out/${tag}_block_${w}_bench.json" 2> "$out/${tag}_block_${w}_bench.err"
  tail -c 600 "$out/${tag}_block_${w}_bench.json"; echo
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_block_${w}_stats" -o run -- python3 bench.py --workload block_$w --steps 10 --warmup 4 --no-cpu-baseline > "$out/${tag}_block_${w}_stats.log" 2>&1
  f=$(find "$out/${tag}_block_${w}_stats" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$out/${tag}_block_${w}_kernel_stats.csv"
  head -12 "$f" | cut -c1-150
  # HBM / fabric bytes and the matrix-pipe counter, separate passes as the counters require (no trace flags beside --pmc)
  P="python3 bench.py --workload block_$w --steps 2 --warmup 2 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_block_${w}_pmc_fetch" -o run -- $P > "$out/${tag}_block_${w}_pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_block_${w}_pmc_write" -o run -- $P > "$out/${tag}_block_${w}_pmc_write.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$out/${tag}_block_${w}_pmc_sq" -o run -- $P > "$out/${tag}_block_${w}_pmc_sq.log" 2>&1
done
