#!/bin/bash
# usage (on the GPU box): tools/prof_variant.sh NAME [bench args]  -> gpurun_out/kstat_NAME.csv (rocprofv3 kernel stats)
NAME=$1; shift
export BMPC_LIB=$GRAFT_REPO_ROOT/build/variants/libboundmpc_$NAME.so
[ "$NAME" = "default" ] && unset BMPC_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$NAME
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/bench_$NAME.json 2> gpurun_out/bench_$NAME.err
f=$(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/kstat_$NAME.csv
t=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
python3 - "$t" > gpurun_out/ktrace_$NAME.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in per.items():
    if len(v) > 20:
        print(n, "calls", len(v), "total_ms %.1f" % (sum(v) / 1e3), "first40_ms %.1f" % (sum(v[:40]) / 1e3), "rest_ms %.1f" % (sum(v[40:]) / 1e3),
              "per-call us:", " ".join("%.0f" % x for x in v[:6]), "...", " ".join("%.0f" % x for x in v[40:46]), "...", " ".join("%.0f" % x for x in v[-4:]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("span_ms %.1f busy_ms %.1f" % ((t1 - t0) / 1e6, busy / 1e6))
PY
echo "== $NAME: $(python3 -c "import json;d=json.load(open('gpurun_out/bench_$NAME.json'));print('%.0f solves/s  %.1f ms  iters %.2f conv %.4f'%(d['value'],d['ms_per_step'],d['solver']['iters_mean'],d['solver']['converged_frac']))")"
head -7 gpurun_out/kstat_$NAME.csv | cut -d, -f1-4
cat gpurun_out/ktrace_$NAME.txt
rm -rf gpurun_out/prof_$NAME
