#!/bin/bash
# Counter passes for the conv kernels (one rocprofv3 run per counter group; --kernel-trace only).
# usage (on the GPU box): tools/pmc_passes.sh OUTDIR
OUT=$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { n=$1; shift
  MIDD_SPLIT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --inference-steps 2 > $OUT/$n.log 2>&1
  python3 tools/pmc_summary.py $(find $OUT/$n -name "*counter_collection.csv" | head -1) > $OUT/$n.txt; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES
run p3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run p5 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32
