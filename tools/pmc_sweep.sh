set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r2/pmcsw; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 > $OUT/p2.log 2>&1
cd $ROOT
python tools/pmc_kernels.py $OUT/p1 $OUT/p2 > $OUT/kernels.json
