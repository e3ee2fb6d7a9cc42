cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_iterate_parity.py -x -q -m gpu -k "ric_variants or solve_matches_oracle or pool_at_bench or trial_repeats or multipliers or iterates_agree or config4_generator" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/scaling_trace.py > $O/scaling_run.log 2>&1 || { tail $O/scaling_run.log; exit 1; }
python3 tools/scaling_trace.py --parse $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_time_vs_live_instances.txt; rm -rf $O/trace
cat $O/kernel_time_vs_live_instances.txt
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print({k:d.get(k) for k in ('value','value_single_batch','value_pcie_inclusive_pipelined')}, {k:r[k] for k in ('achieved','frac','launches','launch_ms_avg','instance_iterations')}, r['latency_variant'], d['closed_loop_configs4'].get('solves_per_s'), d['solver'])"
