#!/bin/bash
# On the GPU box: A/B of BMPC_EVAL_SPLIT_WGS (two-wavefront k_eval up to that many groups of pairs): kernel times + bench value
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 256 100000000 256 100000000; do
  export BMPC_EVAL_SPLIT_WGS=$v
  O=gpurun_out/r04ba_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/scaling_trace.py > $O/run.log 2>&1 || { tail $O/run.log; exit 1; }
  python3 tools/scaling_trace.py --parse $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_time.txt; rm -rf $O/trace
  echo "== BMPC_EVAL_SPLIT_WGS=$v"; grep -E "^B|^  8192|^  4096|^  1024" $O/kernel_time.txt
  timeout -k 10 300 python3 bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']))"
done
