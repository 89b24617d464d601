mkdir -p gpurun_out/r2v
run() { echo "variant [$1]" >> gpurun_out/r2v/variants.log; python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>> gpurun_out/r2v/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/r2v/variants.log; }
build() { BRS_EXTRA_HIPCC_FLAGS="$1" python -c "
from balance_robot_mujoco_rl_amd import _lib
_lib.build(force=True)" 2>> gpurun_out/r2v/err.log; }
for v in "" "-Xarch_device -fslp-vectorize" "-Xarch_device -mllvm=-amdgpu-schedule-metric-bias=0" "-Xarch_device -mllvm=-amdgpu-use-amdgpu-trackers=1" "-Xarch_device -mllvm=-enable-misched=0" "-Xarch_device -mllvm=-amdgpu-enable-max-ilp-scheduling-strategy=1" "-Xarch_device -O2" ""; do
  build "$v" && run "$v"
done
cat gpurun_out/r2v/variants.log; tail -5 gpurun_out/r2v/err.log | cut -c1-300
