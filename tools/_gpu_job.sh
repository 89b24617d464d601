mkdir -p gpurun_out/r2r
run() { echo "variant [$1] [$2]" >> gpurun_out/r2r/variants.log; python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>> gpurun_out/r2r/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/r2r/variants.log; }
build() { BRS_EXTRA_HIPCC_FLAGS="$1" python -c "
from balance_robot_mujoco_rl_amd import _lib
_lib.build(force=True)" 2>> gpurun_out/r2r/err.log; }
build "-DBRS_CLASS_V1"; run "-DBRS_CLASS_V1" ""; run "-DBRS_CLASS_V1" ""
build ""; run "" ""; run "" ""
cat gpurun_out/r2r/variants.log
python tools/phase_timing.py > gpurun_out/r2r/phase.log 2>&1; sed -n 12,16p gpurun_out/r2r/phase.log
