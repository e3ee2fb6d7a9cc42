#!/bin/bash
# usage: tools/build_variant.sh NAME [extra -D flags...]  -> build/variants/libboundmpc_NAME.so (experiments only; the product
# library is built by __graft_entry__.build()).  Only the pipeline kernels are rebuilt with the flags; run with BMPC_LIB=<that .so>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/boundplanner_amd/csrc
OUT=$ROOT/build/variants
NAME=$1; shift
mkdir -p $OUT/$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on"
for o in bmpc_fk.o bmpc_loop.o; do [ -f $CS/$o ] || { echo "build the product library first (python -c 'import __graft_entry__ as g; g.build()')"; exit 1; }; done
hipcc $FL "$@" -c $CS/bmpc_pipeline.hip -o $OUT/$NAME/pipe.o
hipcc $FL "$@" -c $CS/bmpc_capi.hip -o $OUT/$NAME/capi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libboundmpc_$NAME.so $OUT/$NAME/capi.o $OUT/$NAME/pipe.o $CS/bmpc_fk.o $CS/bmpc_loop.o
echo built $OUT/libboundmpc_$NAME.so
