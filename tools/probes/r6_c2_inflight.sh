for q in 4 8 16; do for f in 4 8 16; do
  v=$(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --workload C2 --steps 4000 --warmup 100 --extras 0 --cpu-seconds 0 --in-flight $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "queues $q in-flight $f -> $v"
done; done
