set -u
for k in 2 3 4; do timeout 200 python bench.py --in-flight $k --extras 0 --steps 900 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight', d['config']['scans_in_flight'], d['value'], d['ms_per_step'])"; done
