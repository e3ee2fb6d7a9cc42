#!/bin/bash
# On the GPU box: bitwise tests of the kernel variants, tail trace of one synchronous batch, the full bench line.
#   tools/tail_check.sh TAG   -> gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-tail}; O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ric_variants or trial_repeats or pool_at_bench" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --pool 0 --steps 1 --warmup 0 --depth 1 --merge 1 --same-batch --no-cpu-baseline --no-extra --gen-workers 0 > $O/bench_depth1.json 2> $O/bench_depth1.err || { tail $O/bench_depth1.err; exit 1; }
python3 tools/tail_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/tail_trace_depth1.txt; rm -rf $O/trace
cat $O/tail_trace_depth1.txt
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err || { tail $O/bench_full.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'value_single_batch', 'value_pcie_inclusive_pipelined')}, {k: d['closed_loop_configs4'].get(k) for k in ('solves_per_s', 'ms_per_step')} if 'closed_loop_configs4' in d else None, d['roofline'].get('frac'))"
