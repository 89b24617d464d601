# run on the GPU box: speed and teacher-forced parity for several active-set flip deadbands (BRS_FLIP_TOL)
R=$GRAFT_REPO_ROOT
cd $R
for T in 1e-6 1e-5 1e-4 1e-3; do
  BRS_EXTRA_HIPCC_FLAGS="-DBRS_FLIP_TOL=$T" python3 -c "from balance_robot_mujoco_rl_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1
  python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tol $T', 'env-steps/s %.4g ms/step %.3f'%(d['value'], d['ms_per_step']))"
  python3 tools/parity_report.py --only3 --envs 2048 --steps 400 --out gpurun_out/parity_tol_$T.json > /dev/null 2>&1
  python3 -c "
import json; d=json.load(open('gpurun_out/parity_tol_$T.json'))['config3_autoreset']
print('   parity: max %.3g p999 %.3g median %.3g over1e-4 %d of %d' % (d['max_dqpos'], d['p999_dqpos'], d['median_dqpos'], d['env_steps_over_1e_4'], d['env_steps']))"
done
