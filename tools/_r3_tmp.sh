for k in 3 4; do for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --in-flight $k --extras 0 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 in-flight $k', d['value'], d['ms_per_step'])"
done; done
timeout 300 python bench.py --in-flight 4 --extras 1 --steps 1200 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight 4 extras', d['value'], d['stages']['splat_plus_slice_in_flight'], d['latency']['us_per_scan_median'])"
for w in C4 C5; do timeout 300 python bench.py --workload $w --in-flight 4 --extras 0 --steps 600 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w in-flight 4', d['value'], d['ms_per_step'])"; done
