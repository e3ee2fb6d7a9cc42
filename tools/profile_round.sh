#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench line and of one batch at a time (no overlap between
# solver calls: clean per-kernel durations) + PMC passes of one synchronous batch.
# Outputs under gpurun_out/prof_round/ (copy the summaries to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; rm -rf $O; mkdir -p $O
ONE="--pool 0 --steps 2 --warmup 1 --depth 1 --merge 1 --same-batch --no-cpu-baseline --no-extra --gen-workers 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gen-workers 0 --no-extra > $O/bench_default.json 2> $O/bench_default.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py $ONE > $O/bench_depth1.json 2> $O/bench_depth1.err
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_depth1_merge1.csv
if [ "$1" != "nopmc" ]; then
export BMPC_SPLIT_LAUNCHES=1      # the counter passes keep k_points / k_pose / k_eval / k_curv as launches of their own (per-kernel tables)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$n -- python3 bench.py --pool 0 --steps 1 --warmup 0 --depth 1 --merge 1 --same-batch --no-cpu-baseline --no-extra --gen-workers 0 > $O/pmc_$n.json 2> $O/pmc_$n.err || echo "pmc pass $n failed"
done
python3 tools/summarize_pmc.py $O/pmc_*/ --traffic-json $O/pmc_traffic.json > $O/pmc_summary.csv
unset BMPC_SPLIT_LAUNCHES
fi
rm -rf $O/stats $O/stats1 $O/pmc_*/
cat $O/kernel_stats.csv | cut -d, -f1-5 | head -16; cat $O/kernel_stats_depth1_merge1.csv | cut -d, -f1-5 | head -16; [ -f $O/pmc_summary.csv ] && head -80 $O/pmc_summary.csv; cat $O/bench_default.json
