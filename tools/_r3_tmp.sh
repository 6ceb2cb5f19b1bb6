for i in 1 2; do
LATTICE_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2973$i timeout 300 python bench.py --extras 0 --steps 1500 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nccl one rank', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --extras 0 --steps 1500 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no dist', d['value'], d['ms_per_step'])"
for q in 5 6; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --extras 0 --steps 1500 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --extras 1 --steps 600 --cpu-seconds 0 --full-unet 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extras', d['value'], d['stages']['splat_plus_slice_in_flight']['frac_of_hbm_peak'], d['stages']['splat_plus_slice']['frac_of_hbm_peak'])"
