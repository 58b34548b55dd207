#!/bin/bash
# Instruction-mix / stall counters of the bench's kernels (three separate --pmc passes, no tracing):
#   tools/pmc_deep.sh <tag>      -> gpurun_out/<tag>_deep{1,2,3}/, table printed by tools/pmc_deep.py
set -u
tag=${1:-d}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
P=${PMC_CMD:-"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --prewarm-steps 0"}
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$out/${tag}_deep1" -o run -- $P > "$out/${tag}_deep1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$out/${tag}_deep2" -o run -- $P > "$out/${tag}_deep2.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES \
  --output-format csv -d "$out/${tag}_deep3" -o run -- $P > "$out/${tag}_deep3.log" 2>&1
python3 tools/pmc_deep.py "$tag"
