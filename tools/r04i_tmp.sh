cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; tail -15 $O/pytest_full.log | cut -c1-300
