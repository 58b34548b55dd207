#!/bin/bash
# stress workload (BASELINE.json configs[4]: B=16, N=8192 -> 4096): bench line + kernel trace + HBM counters
set -u
tag=${1:-r02_stress}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --workload stress --steps 10 --warmup 3 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -c 600 "$out/${tag}_bench.json"
B="python3 bench.py --workload stress --steps 10 --warmup 3 --no-breakdown --no-graph --prewarm-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- $B > "$out/${tag}_stats.log" 2>&1
P="python3 bench.py --workload stress --steps 2 --warmup 1 --no-breakdown --no-graph --prewarm-steps 0"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$out/${tag}_pmc_sq" -o run -- $P > "$out/${tag}_pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -o run -- $P > "$out/${tag}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -o run -- $P > "$out/${tag}_pmc_write.log" 2>&1
