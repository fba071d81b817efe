#!/bin/bash
# One-object variant of the library for same-box A/B runs: tools/variant.sh NAME file.hip "-DFLAG=1 ..."  -> libmidd_NAME.so at the
# repo root (git-ignored; travels to the GPU box).  Every other object comes from the current default build (run make first).
set -e
NAME=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/medical-image-denoising-using-diffusion_amd/csrc
mkdir -p $CSRC/build_$NAME
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $CSRC/$SRC -o $CSRC/build_$NAME/${SRC%.hip}.o
OBJS=""
for f in midd_api conv_mfma_f32 conv_mfma_f16x3 conv1x1_f16x3 groupnorm attention_f32 attention_f16x3 pointwise prepost; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS $CSRC/build_$NAME/$f.o"; else OBJS="$OBJS $CSRC/build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/libmidd_$NAME.so $OBJS
echo "built $ROOT/libmidd_$NAME.so"
