mkdir -p gpurun_out/final_c
python tools/vecenv_rate.py > gpurun_out/final_c/vecenv_rate.json 2> gpurun_out/final_c/err.log
python -m pytest tests -m gpu -q -rA > gpurun_out/final_c/gpu_tests.log 2>&1; tail -2 gpurun_out/final_c/gpu_tests.log
BRS_BENCH_ONE_DEVICE=1 BRS_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final_c/bench_2ranks_one_gpu.json 2>> gpurun_out/final_c/err.log; echo "rc=$?"; cut -c1-250 gpurun_out/final_c/bench_2ranks_one_gpu.json
WORLD_SIZE=1 python bench.py --gpus 2 --steps 5 > /dev/null 2>> gpurun_out/final_c/err.log; echo "mismatch rc=$? (expect 2)"
python __graft_entry__.py smoke 2>&1 | tail -2
