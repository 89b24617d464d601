mkdir -p gpurun_out/final_d
python -m pytest tests -m gpu -q > gpurun_out/final_d/gpu_tests.log 2>&1; tail -2 gpurun_out/final_d/gpu_tests.log
bash tools/profile_r02.sh gpurun_out/prof_r02_env01 Env01-v2 > gpurun_out/final_d/prof_env01.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/prof_r02_env01/summary.json")); b=d["bench"]
print("env01", b["value"], b["ms_per_step"], d.get("kernel_trace_ms",{}).get("timed_300_mean"), d.get("valu"))
PY
run() { echo "variant [$1] [$2]" >> gpurun_out/final_d/sizes.log; env $1 python bench.py --no-cpu-baseline $2 2>> gpurun_out/final_d/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/final_d/sizes.log; }
for n in 65536 131072 262144 524288; do run "X=1" "--env Env03-v2 --envs $n --steps 60 --warmup 10"; done
cat gpurun_out/final_d/sizes.log
