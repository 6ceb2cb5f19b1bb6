set -u
mkdir -p gpurun_out/r3m
( time timeout 900 python bench.py > gpurun_out/r3m/bench_full.json 2> gpurun_out/r3m/bench_full.err ) 2> gpurun_out/r3m/time.txt; echo "rc=$?"; tail -3 gpurun_out/r3m/time.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3m/bench_full.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","latency","roofline","full_unet","cpu_baseline","cpu_baseline_1thread")}, indent=1)[:3000])
print([(o['kernel'],o['avg_us'],o['avg_us_in_flight'],o['frac']) for o in d['roofline_others']])
print(d['stages'].get('splat_plus_slice'), d['stages'].get('splat_plus_slice_in_flight'))
PY
