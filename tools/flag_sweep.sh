#!/bin/bash
# diagnostic: try compiler flags on the step kernel (run on the GPU box); not part of the product build
R=$GRAFT_REPO_ROOT
cd $R
while IFS= read -r FL; do
  BRS_EXTRA_HIPCC_FLAGS="$FL" python3 -c "from balance_robot_mujoco_rl_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1 || { echo "BUILD FAILED: $FL"; continue; }
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-80s %.4g env-steps/s' % ('''$FL''', d['value']))"
done <<'LIST'

-mllvm -misched-topdown
-mllvm -misched-bottomup
-mllvm -greedy-regclass-priority-trumps-globalness=1
-mllvm -enable-local-reassign=1
-mllvm -amdgpu-schedule-metric-bias=0
-mllvm -amdgpu-schedule-metric-bias=100
-mllvm -disable-machine-sink
-mllvm -inline-threshold=100000
-mllvm -amdgpu-disable-unclustered-high-rp-reschedule
LIST
