mkdir -p gpurun_out/final_b
bash tools/profile_r02.sh gpurun_out/prof_r02_env03 Env03-v2 > gpurun_out/final_b/prof_env03.log 2>&1
bash tools/profile_r02.sh gpurun_out/prof_r02_env01 Env01-v2 > gpurun_out/final_b/prof_env01.log 2>&1
python tools/parity_report.py --scratch > gpurun_out/final_b/parity_report.log 2>&1; tail -2 gpurun_out/final_b/parity_report.log | cut -c1-300
du -sh gpurun_out
