#!/bin/bash
# Build libpmoe_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/pmoe_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p ../../build
pids=()
for f in conv_igemm conv_dma conv_c16 conv_c1x1 conv_res conv_wgrad gemm_skinny elementwise stem_tail heads punet stage1 optim preprocess api; do
  if [ ! -f ../../build/$f.o ] || [ $f.hip -nt ../../build/$f.o ] || [ common.h -nt ../../build/$f.o ] || [ kernels.h -nt ../../build/$f.o ] || [ conv_common.h -nt ../../build/$f.o ] || [ conv_dma_epilogue.inc -nt ../../build/$f.o ] || [ ../../include/pmoe_hip.h -nt ../../build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o ../../build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libpmoe_hip.so ../../build/conv_igemm.o ../../build/conv_dma.o ../../build/conv_c16.o ../../build/conv_c1x1.o ../../build/conv_res.o ../../build/conv_wgrad.o ../../build/gemm_skinny.o ../../build/elementwise.o ../../build/stem_tail.o ../../build/heads.o ../../build/punet.o ../../build/stage1.o ../../build/optim.o ../../build/preprocess.o ../../build/api.o
echo "built pmoe_amd/libpmoe_hip.so"
