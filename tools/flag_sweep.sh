#!/bin/bash
# diagnostic: try compiler flags on the step kernel (run on the GPU box); not part of the product build
R=$GRAFT_REPO_ROOT
cd $R
while IFS= read -r FL; do
  BRS_EXTRA_HIPCC_FLAGS="$FL" python3 -c "from balance_robot_mujoco_rl_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1 || { echo "BUILD FAILED: $FL"; continue; }
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-80s %.4g env-steps/s' % ('''$FL''', d['value']))"
done <<'LIST'

-Xarch_device -fno-vectorize
-mllvm -amdgpu-schedule-relaxed-occupancy=true
-Xarch_device -fno-unroll-loops
-mllvm -amdgpu-use-divergent-register-indexing
-mllvm -amdgpu-dpp-combine=false
LIST
