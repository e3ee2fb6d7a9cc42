#!/bin/bash
# bench lines of the default schedule for other batch sizes and horizons (SURVEY 8(d) generator) -> gpurun_out/sweep.jsonl
rm -f gpurun_out/sweep.jsonl
for cfg in "1024 20" "2048 20" "4096 20" "8192 20" "16384 20" "8192 10" "8192 15" "8192 30"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-extra --batch $1 --horizon $2 2> gpurun_out/sweep.err | tail -n 1 >> gpurun_out/sweep.jsonl || echo "failed $cfg"
  python -c "
import json
d=json.loads(open('gpurun_out/sweep.jsonl').read().strip().splitlines()[-1]); print('$1 $2', round(d['value']), round(d['ms_per_step'],1), d['solver']['iters_mean'], d['solver']['converged_frac'])"
done
