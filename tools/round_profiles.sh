#!/bin/bash
# Everything the round's profiles/ directory holds, in one GPU call (run on the GPU box): tools/round_profiles.sh TAG
#   kernel stats + PMC traffic + bench lines (profile_round.sh, bench_configs.sh), per-op tables, counter passes,
#   the cost of the batch-invariant plans, the micro-benchmarks, the in-kernel stamps.  Output: gpurun_out/prof_TAG/
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
[ -f $OUT/kernel_stats.csv ] || { bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?"; }
bash tools/bench_configs.sh $TAG > $OUT/bench_configs.log 2>&1; cat $OUT/bench_configs.log
python tools/per_op_profile.py 4 256 > $OUT/per_op_b4_256.txt 2>&1; head -2 $OUT/per_op_b4_256.txt | tail -1
python tools/per_op_profile.py 8 256 > $OUT/per_op_b8_256.txt 2>&1; head -2 $OUT/per_op_b8_256.txt | tail -1
python tools/per_op_profile.py 4 512 > $OUT/per_op_b4_512.txt 2>&1; head -2 $OUT/per_op_b4_512.txt | tail -1
bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc_passes.log 2>&1; echo "pmc_passes rc=$?"
bash tools/traffic_per_op.sh $TAG 4 > $OUT/traffic_per_op_b4.txt 2>&1
bash tools/inv_bench.sh > $OUT/batch_invariant_cost.txt 2>&1; cat $OUT/batch_invariant_cost.txt
./tools/mb/hbm_rate > $OUT/mb_hbm_rate.txt 2>&1
./tools/mb/pw_abl_0 > $OUT/mb_pointwise.txt 2>&1; cat $OUT/mb_pointwise.txt
if [ -f libmidd_timing.so ]; then MIDD_SPLIT=1 MIDD_PLAN_AS_SIDE=1 python tools/conv_timing.py 4 256 > $OUT/conv_timing_b4.txt 2>&1; fi
echo done
