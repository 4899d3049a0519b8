#!/bin/bash
# Container: a development build of the library with some of the training step's matrix-core objects recompiled with extra flags.
#   tools/dev/r6_variant.sh NAME "FLAGS" K [K ...]     K: NTX_TRAIN_KERNEL numbers (0-3 forward chain builds, 4 the chain back, 5 the weight gradients)
# -> build_dev/libntx_NAME.so (run with NERFTEX_LIB=...).  The other objects are the shipped ones of nerf_tex_amd/csrc/_obj.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/nerf_tex_amd/csrc; N=$1; F=$2; shift 2
O=$R/build_dev/obj_$N; mkdir -p $O
for K in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$C -Wall -Wno-unused-function $F -DNTX_TRAIN_KERNEL=$K -c $C/ntx_train_chain.hip -o $O/train_chain_$K.o &
done
wait
OBJS=""
for f in $C/_obj/*.o; do b=$(basename $f); if [ -f $O/$b ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/build_dev/libntx_$N.so -ldl
echo built build_dev/libntx_$N.so
