#!/bin/bash
# On the GPU box: one synchronous batch (configs[2], 8192 instances) per setting of an environment knob, same box
#   tools/ab_env.sh VAR v1 v2 ...      -> value_single_batch-style solves/s per value
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 200 python3 bench.py --pool 0 --steps 3 --warmup 1 --depth 1 --merge 1 --same-batch --no-cpu-baseline --no-extra --gen-workers 0 2>/dev/null > /tmp/ab_env.json || { echo "$VAR=$v failed"; exit 1; }
  python3 -c "import json; d=json.loads(open('/tmp/ab_env.json').read().strip().splitlines()[-1]); print('$VAR=$v', 'solves/s', round(d['value']), 'ms per batch', round(d['ms_per_step'],1), 'ric_lat', d['roofline'].get('latency_variant',{}))"
done
