mkdir -p gpurun_out/final_e
python -m pytest tests -m gpu -q > gpurun_out/final_e/gpu_tests.log 2>&1; tail -2 gpurun_out/final_e/gpu_tests.log
bash tools/profile_r02.sh gpurun_out/prof_r02_env03 Env03-v2 > gpurun_out/final_e/prof_env03.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/prof_r02_env03/summary.json")); b=d["bench"]
print("env03", b["value"], b["ms_per_step"], d.get("kernel_trace_ms"), d.get("valu"), d.get("hbm_traffic"))
PY
python tools/vecenv_rate.py > gpurun_out/final_e/vecenv_rate.json 2>> gpurun_out/final_e/err.log; cut -c1-330 gpurun_out/final_e/vecenv_rate.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final_e/bench_20_5.json 2>> gpurun_out/final_e/err.log; cut -c1-160 gpurun_out/final_e/bench_20_5.json
python tools/phase_timing.py > gpurun_out/final_e/phase.log 2>&1; head -16 gpurun_out/final_e/phase.log
