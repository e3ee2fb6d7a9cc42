#!/bin/bash
# usage: tools/build_variant.sh NAME [extra -D flags...]  -> build/variants/libboundmpc_NAME.so (experiments only)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/boundplanner_amd/csrc
OUT=$ROOT/build/variants
NAME=$1; shift
mkdir -p $OUT/$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
hipcc $FL "$@" -c $CS/bmpc_pipeline.hip -o $OUT/$NAME/pipe.o
[ -f $CS/bmpc_capi.o ] || hipcc $FL -c $CS/bmpc_capi.hip -o $CS/bmpc_capi.o
hipcc $FL -c $CS/bmpc_capi.hip -o $OUT/$NAME/capi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libboundmpc_$NAME.so $OUT/$NAME/capi.o $OUT/$NAME/pipe.o $CS/bmpc_kernels_nt64.o $CS/bmpc_kernels_nt128.o $CS/bmpc_kernels_nt256.o $CS/bmpc_loop.o
echo built $OUT/libboundmpc_$NAME.so
