#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench line + PMC passes of one synchronous batch.
# Outputs under gpurun_out/prof_round/ (copy the summaries to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 1 --warmup 0 --depth 1 --no-cpu-baseline > $O/pmc_$n.json 2> $O/pmc_$n.err || echo "pmc pass $n failed"
done
python3 tools/summarize_pmc.py $O/pmc_*/ --traffic-json $O/pmc_traffic.json > $O/pmc_summary.csv
rm -rf $O/stats $O/pmc_*/
cat $O/kernel_stats.csv | cut -d, -f1-5 | head -14; cat $O/pmc_summary.csv | head -80; cat $O/bench_default.json
