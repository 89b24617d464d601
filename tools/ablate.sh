R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for V in "base:" "nocpl:-DBRS_NO_COUPLED" "nocpl_noblk:-DBRS_NO_COUPLED -DBRS_NO_BLOCKFLOOR"; do
  NAME=${V%%:*}; FL=${V#*:}
  BRS_EXTRA_HIPCC_FLAGS="$FL" python3 -c "import sys; sys.path.insert(0,'$R'); from balance_robot_mujoco_rl_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1
  python3 $R/bench.py --steps 60 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$NAME', 'env-steps/s %.3g ms/step %.3f'%(d['value'], d['ms_per_step']))"
  rm -rf $R/gpurun_out/abl_$NAME
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/abl_$NAME -- python3 $R/bench.py --steps 20 --warmup 40 --no-cpu-baseline > /dev/null 2>&1
done
