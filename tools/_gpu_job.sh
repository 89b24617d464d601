mkdir -p gpurun_out/r2u
run() { echo "variant [$1] [$2]" >> gpurun_out/r2u/variants.log; python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>> gpurun_out/r2u/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/r2u/variants.log; }
build() { BRS_EXTRA_HIPCC_FLAGS="$1" python -c "
from balance_robot_mujoco_rl_amd import _lib
_lib.build(force=True)" 2>> gpurun_out/r2u/err.log; }
build "-DBRS_H_CHAINED"; run "-DBRS_H_CHAINED" ""; run "-DBRS_H_CHAINED" ""
build ""; run "" ""; run "" ""
cat gpurun_out/r2u/variants.log
