set -u
mkdir -p gpurun_out/r3base
python bench.py --full-unet 0 --cpu-seconds 0 > gpurun_out/r3base/bench_default.json 2> gpurun_out/r3base/bench_default.err
BENCH_ARGS="--in-flight 1" bash tools/ab.sh "if1:" > gpurun_out/r3base/ab_if1.txt 2>&1
bash tools/ab.sh "if3:" > gpurun_out/r3base/ab_if3.txt 2>&1
tail -3 gpurun_out/r3base/bench_default.json; cat gpurun_out/r3base/ab_if1.txt gpurun_out/r3base/ab_if3.txt
