#!/bin/bash
# stall counters of the attention kernels under tools/dev_tri.py (separate --pmc passes, no tracing)
set -u
tag=${1:-t}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
P=${PMC_CMD:-"python3 tools/dev_tri.py"}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d "$out/${tag}_deep1" -o run -- $P > "$out/${tag}_deep1.log" 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA \
  --output-format csv -d "$out/${tag}_deep2" -o run -- $P > "$out/${tag}_deep2.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES \
  --output-format csv -d "$out/${tag}_deep3" -o run -- $P > "$out/${tag}_deep3.log" 2>&1
python3 tools/pmc_deep.py "$tag"
