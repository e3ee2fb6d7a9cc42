cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 500 python3 tests/diag/diag_iterate_parity.py 20 1024 8192 1 12 > $O/diag_iterate.txt 2>&1; tail -50 $O/diag_iterate.txt | cut -c1-600
timeout -k 10 500 python3 -m pytest tests/test_device_loop_gpu.py -x -q -m gpu -k "async or tracks_host" > $O/pytest_async.log 2>&1; tail -5 $O/pytest_async.log
timeout -k 10 300 python3 tools/closed_loop_device.py --async > $O/cl_async.json 2> $O/cl_async.err; tail -3 $O/cl_async.json | cut -c1-700
timeout -k 10 300 python3 tools/closed_loop_device.py --groups 3 > $O/cl_g3.json 2> $O/cl_g3.err; tail -3 $O/cl_g3.json | cut -c1-700
