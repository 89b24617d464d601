mkdir -p gpurun_out/r2t
run() { echo "variant [$1] [$2]" >> gpurun_out/r2t/variants.log; python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>> gpurun_out/r2t/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_min_median_max'])" >> gpurun_out/r2t/variants.log; }
build() { BRS_EXTRA_HIPCC_FLAGS="$1" python -c "
from balance_robot_mujoco_rl_amd import _lib
_lib.build(force=True)" 2>> gpurun_out/r2t/err.log; }
build "-DBRS_CLASS_V2"; run "-DBRS_CLASS_V2" ""; run "-DBRS_CLASS_V2" ""
build ""; run "" ""; run "" ""
cat gpurun_out/r2t/variants.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "determinism or config4" 2>&1 | tail -2
python tools/phase_timing.py > gpurun_out/r2t/phase.log 2>&1; sed -n 12,15p gpurun_out/r2t/phase.log
