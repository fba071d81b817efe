#!/bin/bash
# Instruction-mix counter passes (continues tools/pmc_passes.sh): tools/pmc_passes2.sh OUTDIR
OUT=$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { n=$1; shift
  MIDD_SPLIT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --inference-steps 2 > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_summary.py $f > $OUT/$n.txt; else echo "no csv for $n" > $OUT/$n.txt; tail -3 $OUT/$n.log >> $OUT/$n.txt; fi; }
run p5 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32
run p6 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES
run p7 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU
