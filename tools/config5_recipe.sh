#!/bin/bash
# BASELINE config 5 on ONE GPU, ONE fixed recipe following the reference's curriculum (README.md:54-60: train on Env01-v2, then
# resume on Env03-v2), with the on-device rollout kernels (brs_policy_act / brs_step / brs_rollout_bootstrap / brs_gae):
#   stage 1  Env01-v2, 150 iterations of 16,384 envs x 64 steps (157 M env-steps): PPO clip 0.2, 4 epochs, minibatch 8,192,
#            lr 3e-4, gamma 0.999, lambda 0.95, learner-side reward clip at 1.0 (the env's pitch x wheel-speed bonus is unbounded),
#            initial first-layer weights on obs[1] x 0.02 (that channel carries +-10 rad/s of injected noise)
#   stage 2  Env03-v2 (12.5 % of the envs stay in Env01-v2), 400 iterations: lr 1e-4, log-std reset to -1 and frozen,
#            10 critic-only iterations, returns normalised, target KL 0.02
# then deterministic evaluation on 4,096 fresh envs for 2,500 steps (two full Env03-v2 episodes).  SURVEY.md App. D criterion:
# mean eval episode length 1200 and return >= 1100 on Env03-v2.
OUT=${1:-gpurun_out/config5}
mkdir -p $OUT
python tools/train_ppo_torch.py --env Env01-v2 --then Env03-v2 --envs 16384 --n-steps 64 --minibatch 8192 --iters 150 --iters2 400 \
  --gamma 0.999 --reward-clip 1.0 --obs-init-scale 1,0.02,1,1,1,1 --lr2 1e-4 --log-std2 -1.0 --fix-log-std2 --critic-warmup2 10 \
  --norm-returns2 --target-kl2 0.02 --mix2 0.125 --device-rollout --eval-steps 2500 --save $OUT/policy.pt --out $OUT/config5.json 2>&1 | tee $OUT/config5.log
