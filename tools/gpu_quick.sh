#!/bin/bash
# On the GPU box (round 4): a short check + measurement of the current build.
#   tools/gpu_quick.sh TAG [tests|notests] [pmc|nopmc]
# -> gpurun_out/TAG/: pytest log of the bitwise / parity subset, per-kernel time against live instances (scaling_trace),
#    the bench line without the extra legs, FETCH_SIZE / WRITE_SIZE per kernel of one synchronous batch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-quick}; O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
if [ "$2" != "notests" ]; then
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ric_variants or solve_matches_oracle or pool_at_bench or trial_repeats or multipliers" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
  tail -3 $O/pytest.log
fi
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/scaling_trace.py > $O/scaling_run.log 2>&1 || { tail $O/scaling_run.log; exit 1; }
python3 tools/scaling_trace.py --parse $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_time_vs_live_instances.txt; rm -rf $O/trace
cat $O/kernel_time_vs_live_instances.txt
timeout -k 10 400 python3 bench.py --no-extra --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('solver',{}))"
if [ "$3" == "pmc" ]; then
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$set -- python3 bench.py --pool 0 --steps 1 --warmup 0 --depth 1 --merge 1 --same-batch --no-cpu-baseline --no-extra --gen-workers 0 > $O/pmc_$set.json 2> $O/pmc_$set.err || echo "pmc pass $set failed"
done
python3 tools/summarize_pmc.py $O/pmc_*/ --traffic-json $O/pmc_traffic.json > $O/pmc_summary.csv
rm -rf $O/pmc_*/
python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json')); t=0
for k,v in d['per_kernel_GB'].items():
    print(f'{k:22s} fetch_raw {v[\"fetch_raw\"]:7.2f}  write {v[\"write\"]:7.2f}  2F+W {2*v[\"fetch_raw\"]+v[\"write\"]:7.2f} GB'); t+=2*v['fetch_raw']+v['write']
print('total 2F+W', round(t,1), 'GB')"
fi
