#!/bin/bash
# Re-generate the measured artifacts under profiles/ on a GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r01'
# Writes into gpurun_out/refresh/; copy the files listed at the end into profiles/ afterwards (gpurun_out is scratch).
set -o pipefail
R=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; tail -1 $OUT/tests.txt
python bench.py > $OUT/${R}_bench_n1.json 2> $OUT/bench.err
for m in "efficientnet_b3a 256 ${R}_effnet_per_op" "rexnet_200 256 ${R}_rexnet200_per_op" "swin_base_patch4_window7_224 128 ${R}_swin_base_per_op_b128"; do
  set -- $m; python tools/profile_ops.py $1 $2 > $OUT/$3.txt 2>&1
done
python tools/bench_models.py > $OUT/${R}_bench_models.txt 2>&1
python tools/bench_rank.py > $OUT/${R}_bench_rank.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_run.py efficientnet_b3a 256 > $OUT/pmc_write.log 2>&1
cd $ROOT
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${R}_bench_n1_kernel_stats.csv
python tools/pmc_aggregate.py $OUT/pmc_fetch $OUT/pmc_write 3 "efficientnet_b3a bf16 B=256 forward (tools/pmc_run.py), 3 forwards, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes" > $OUT/${R}_pmc_traffic_effnet_b256.json
ls $OUT/${R}_*
