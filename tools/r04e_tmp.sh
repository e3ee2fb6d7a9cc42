cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
BMPC_LIB=tools/_variants/libboundmpc_prof.so BMPC_RIC_LAT_BELOW=0 timeout -k 10 300 python3 tests/diag/diag_ric_phases.py 4096 > $O/ric_phases.txt 2>&1; cat $O/ric_phases.txt
timeout -k 10 600 python3 -m pytest tests/test_iterate_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
