#!/bin/bash
# Developer tool: per-kernel PMC picture of one model forward in four separate rocprofv3 --pmc passes (issue / instruction mix /
# LDS + memory queues / L1), summarised per kernel name.   gpurun -- 'bash tools/pmc_block.sh [model] [batch] [tag]'
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
MODEL=${1:-efficientnet_b3a}; B=${2:-256}; TAG=${3:-blk}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/a -- python3 $ROOT/tools/pmc_run.py $MODEL $B > $OUT/a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/b -- python3 $ROOT/tools/pmc_run.py $MODEL $B > $OUT/b.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/c -- python3 $ROOT/tools/pmc_run.py $MODEL $B > $OUT/c.log 2>&1 &&
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/d -- python3 $ROOT/tools/pmc_run.py $MODEL $B > $OUT/d.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/e -- python3 $ROOT/tools/pmc_run.py $MODEL $B > $OUT/e.log 2>&1
cd $ROOT
python tools/pmc_kernels.py $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e > $OUT/summary.json 2> $OUT/summary.err
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "mbconv_block" in k or "sweep" in k or "gemm" in k:
        print(k[:90]); print("   ", {c: (round(x,1) if isinstance(x,float) else x) for c,x in v.items()})
PY
