#!/bin/bash
# On the GPU box: closed loop without lock step (configs[4]), sweep of the two-lane knobs.  usage: tools/lane_sweep.sh STEPS "ENV1" "ENV2" ...
# (each ENVi a space-separated list of VAR=value; "-" = defaults).  One JSON line per variant in gpurun_out/lane_sweep.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/lane_sweep.jsonl; : > $O
STEPS=$1; shift
for v in "$@"; do
  [ "$v" == "-" ] && v=""
  echo "== $v" >&2
  ( export $v GPU_MAX_HW_QUEUES=8; timeout -k 10 300 python3 tools/closed_loop_device.py --async --steps $STEPS --chunk $STEPS 2> gpurun_out/lane_sweep.err | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); print(json.dumps({'env': '$v', 'solves_per_s': round(d['solves_per_s']), 'wall_s': round(d['wall_s'], 2), 'lanes': d.get('lanes'), 'q': d.get('iters_per_rollout_total_quantiles'), 'max': d['iters_per_rollout_total_max']}))" ) >> $O || { echo "variant failed: $v" >&2; tail -5 gpurun_out/lane_sweep.err >&2; exit 1; }
  tail -1 $O
done
