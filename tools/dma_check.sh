#!/bin/bash
# Hardware check of the hand-counted vmcnt protocol of conv_mfma_f16x3_kernel (run on the GPU box).
#   1. libmidd_dmabreak.so (a wait made one weight step too permissive on purpose) must be REPORTED (status bit 4 -> MiddError);
#   2. libmidd_dmacheck.so (the shipped waits) runs the whole GPU suite without a single report.
# Build both first:  tools/variant.sh dmacheck conv_mfma_f16x3.hip "-DMIDD_DMA_CHECK"
#                    tools/variant.sh dmabreak conv_mfma_f16x3.hip "-DMIDD_DMA_CHECK -DMIDD_DMA_CHECK_BREAK"
OUT=${1:-gpurun_out/dma_check}; mkdir -p $OUT
MIDD_LIBRARY=$PWD/libmidd_dmabreak.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sampler or denoise or forward" > $OUT/break.log 2>&1
echo "deliberately short wait: $(grep -c 'status word 0x' $OUT/break.log) report(s) of status bit 4; pytest: $(tail -1 $OUT/break.log)"
[ -n "$SKIP_SUITE" ] || MIDD_LIBRARY=$PWD/libmidd_dmacheck.so timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/check.log 2>&1
echo "shipped waits, whole GPU suite on the checking build: $(tail -1 $OUT/check.log); reports: $(grep -c 'status word 0x' $OUT/check.log)"
