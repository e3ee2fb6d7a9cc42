cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_iterate_parity.py tests/test_device_loop_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-400
timeout -k 10 500 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err || tail $O/bench_full.err
python3 -c "
import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline','value_single_batch','value_pcie_inclusive_pipelined','closed_loop_configs4') if k in d})[:3000]); print(d['solver']); print({k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','ipopt_on_box','iters_equal_frac_of_both_converged','max_abs_dx_vs_gpu_by_block_same_iters')})"
