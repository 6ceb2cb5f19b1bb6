set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c4; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o s -- python3 $ROOT/bench.py --workload C4 --steps 100 --warmup 5 --cpu-seconds 0 --in-flight 1 > $OUT/log.txt 2>&1
cd $ROOT
cp $(find $OUT/s -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/s
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c4/kernel_stats.csv')))
for r in rows[:14]: print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):6.2f}%  {r['Name'][:100]}")
PY
