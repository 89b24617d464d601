mkdir -p gpurun_out/r2p
run() { echo "variant [$1] [$2]" >> gpurun_out/r2p/variants.log; env $1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>> gpurun_out/r2p/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/r2p/variants.log; }
run "X=1" "--no-lane-grouping"; run "X=1" ""; run "X=1" ""
cat gpurun_out/r2p/variants.log
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/phase_timing.py > gpurun_out/r2p/phase.log 2>&1; head -16 gpurun_out/r2p/phase.log
