#!/bin/bash
# usage (on the GPU box): tools/prof_variant.sh NAME [bench args]  -> gpurun_out/kstat_NAME.csv (rocprofv3 kernel stats)
NAME=$1; shift
export BMPC_LIB=$GRAFT_REPO_ROOT/build/variants/libboundmpc_$NAME.so
[ "$NAME" = "default" ] && unset BMPC_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$NAME
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/bench_$NAME.json 2> gpurun_out/bench_$NAME.err
f=$(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/kstat_$NAME.csv
echo "== $NAME: $(python3 -c "import json;d=json.load(open('gpurun_out/bench_$NAME.json'));print('%.0f solves/s  %.1f ms  iters %.2f conv %.4f'%(d['value'],d['ms_per_step'],d['solver']['iters_mean'],d['solver']['converged_frac']))")"
head -6 gpurun_out/kstat_$NAME.csv | cut -d, -f1-4
rm -rf gpurun_out/prof_$NAME
