cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; tail -4 $O/pytest_full.log | cut -c1-300
timeout -k 10 400 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err || tail $O/bench_full.err
python3 -c "
import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','value_single_batch','value_pcie_inclusive','value_pcie_inclusive_pipelined')})); print({k:r.get(k) for k in ('bound','achieved','frac','launch_ms_avg','traffic','traffic_stale','traffic_frac_of_peak','hbm_alg_frac')}); print(d['closed_loop_configs4']); print(d['solver']); c=d['cpu_baseline']; print({k:c[k] for k in ('value','cores','ipopt_on_box','iters_equal_frac_of_both_converged','max_abs_dx_vs_gpu_by_block_same_iters')})"
timeout -k 10 200 python3 tools/closed_loop_device.py --async > $O/cl_async.json 2> $O/cl_async.err; python3 -c "
import json; d=json.loads(open('$O/cl_async.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('solves_per_s','iters_sum_of_per_step_max','iters_per_rollout_total_max','iters_per_rollout_total_mean')})"
