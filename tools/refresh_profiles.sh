#!/bin/bash
# Re-generate the measured artifacts under profiles/ on a GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02'
# Writes into gpurun_out/refresh/; copy the files listed at the end into profiles/ afterwards (gpurun_out is scratch).
# rocprofv3 always gets the program itself after "--" (no env / bash -c hops), PMC passes are separate from kernel-trace.
set -o pipefail
R=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p $OUT
cd $ROOT
SHA=$(python3 -c "import bench; print(bench.csrc_sha16())" 2>/dev/null | tail -1)
echo "csrc fingerprint $SHA"
for m in "efficientnet_b3a 256 ${R}_effnet_per_op" "rexnet_150 256 ${R}_rexnet150_per_op" "rexnet_200 256 ${R}_rexnet200_per_op" "swin_base_patch4_window7_224 128 ${R}_swin_base_per_op_b128"; do
  set -- $m; python tools/profile_ops.py $1 $2 > $OUT/$3.txt 2>&1
done
python tools/bench_models.py > $OUT/${R}_bench_models.txt 2>&1
python tools/bench_rank.py > $OUT/${R}_bench_rank.txt 2>&1
python tools/bench_latency.py > $OUT/${R}_bench_latency.txt 2>&1
python tools/check_block.py > $OUT/${R}_block_kernel_phases.txt 2>&1
python tools/tune_sweep.py 0 0 0 stamps > $OUT/${R}_sweep_kernel_phases.txt 2>&1
python tools/bench_small_batch.py > $OUT/${R}_bench_small_batch.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 $OUT/pmc_labels.json > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 $OUT/pmc_labels.json > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 $OUT/pmc_labels.json > $OUT/pmc_sq.log 2>&1
cd $ROOT
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${R}_bench_n1_kernel_stats.csv
SQ=""; ls $OUT/pmc_sq/*/*counter_collection.csv > /dev/null 2>&1 && SQ="--sq $OUT/pmc_sq"
python tools/pmc_aggregate.py --fetch $OUT/pmc_fetch --write $OUT/pmc_write $SQ --labels $OUT/pmc_labels.json --sha "$SHA" \
  --note "efficientnet_b3a bf16 B=256 forward (tools/pmc_run.py), rocprofv3 --pmc in separate passes: FETCH_SIZE | WRITE_SIZE | SQ busy/instruction counters" \
  > $OUT/${R}_pmc_traffic_effnet_b256.json
# rank kernel: HBM traffic + MFMA busy, ONE shape per run (the headline one: 256 queries x 100k rows) so that the per-kernel
# averages are not a mix of gallery sizes (the variable is exported here, never through an `env` hop behind rocprofv3)
cd /tmp
export CASES=256x100000
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rank_fetch -- python3 $ROOT/tools/bench_rank.py > $OUT/pmc_rank_fetch.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_rank_sq -- python3 $ROOT/tools/bench_rank.py > $OUT/pmc_rank_sq.log 2>&1
unset CASES
cd $ROOT
python tools/pmc_kernels.py --sha "$SHA" --note "tools/bench_rank.py with CASES=256x100000 (Q=256, G=100000, D=1536, k=3), rocprofv3 --pmc in separate passes; FETCH_SIZE is in KiB per dispatch, RAW (the gallery arrives as 64-byte-per-row LDS-DMA pieces: the x2 wide-read correction of fetch_bytes_corrected does not apply to it)" $OUT/pmc_rank_fetch $OUT/pmc_rank_sq > $OUT/${R}_pmc_rank_kernels.json 2> $OUT/pmc_rank.err
# RexNet-150 / -200 and Swin-B: HBM traffic + pipe busy per kernel name
cd /tmp
for m in "rexnet_150 256 rexnet150" "rexnet_200 256 rexnet200" "swin_base_patch4_window7_224 128 swin_base"; do
  set -- $m
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$3_fetch -- python3 $ROOT/tools/pmc_run.py $1 $2 > $OUT/pmc_$3_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$3_write -- python3 $ROOT/tools/pmc_run.py $1 $2 > $OUT/pmc_$3_write.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_$3_sq -- python3 $ROOT/tools/pmc_run.py $1 $2 > $OUT/pmc_$3_sq.log 2>&1
  (cd $ROOT && python tools/pmc_kernels.py $OUT/pmc_$3_fetch $OUT/pmc_$3_write $OUT/pmc_$3_sq > $OUT/${R}_pmc_$3_kernels.json 2>> $OUT/pmc_rank.err)
done
cd $ROOT
# the bench line LAST: bench.py reports `traffic` from profiles/${R}_pmc_*.json when their csrc fingerprint matches the sources it
# runs, so the summaries just made go into this (scratch) copy of profiles/ first
cp $OUT/${R}_pmc_traffic_effnet_b256.json $OUT/${R}_pmc_rank_kernels.json profiles/
python bench.py > $OUT/${R}_bench_n1.json 2> $OUT/bench.err; tail -c 400 $OUT/${R}_bench_n1.json; echo
ls $OUT/${R}_*
