#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>       e.g. r01_c
# 1. bench.py default run (JSON line)            -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench   -> gpurun_out/<tag>_stats/
# 3. three separate --pmc passes (SQ_*, FETCH_SIZE, WRITE_SIZE), no trace flags, as the counters
#    require                                     -> gpurun_out/<tag>_pmc_{sq,fetch,write}/
# tools/pmc_summary.py then folds 2+3 into profiles/<tag>_*.  The program follows `--` directly
# (no env/bash hop after the profiler has initialised the GPU).
set -u
tag=${1:-r01_x}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -1 "$out/${tag}_bench.json"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --no-extra-workloads --no-graph --prewarm-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- $B > "$out/${tag}_stats.log" 2>&1
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-extra-workloads --no-graph --prewarm-steps 0"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$out/${tag}_pmc_sq" -o run -- $P > "$out/${tag}_pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -o run -- $P > "$out/${tag}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -o run -- $P > "$out/${tag}_pmc_write.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/${tag}_pmc_lds" -o run -- $P > "$out/${tag}_pmc_lds.log" 2>&1
ls "$out"/${tag}_*/ 2>/dev/null | head -40
