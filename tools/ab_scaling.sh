#!/bin/bash
# On the GPU box: per-kernel time against live instances for the product library and for build/variants/libboundmpc_<NAME>.so, same box
#   tools/ab_scaling.sh TAG NAME [NAME...]   -> gpurun_out/TAG/kernel_time_<name>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
for NAME in default "$@" default; do
  if [ "$NAME" = "default" ]; then unset BMPC_LIB; else export BMPC_LIB=$GRAFT_REPO_ROOT/build/variants/libboundmpc_$NAME.so; fi
  rm -rf $O/trace
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/scaling_trace.py > $O/run_$NAME.log 2>&1 || { tail $O/run_$NAME.log; exit 1; }
  python3 tools/scaling_trace.py --parse $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_time_$NAME.txt
  echo "== $NAME"; cat $O/kernel_time_$NAME.txt
done
rm -rf $O/trace
