mkdir -p gpurun_out/ppo_r02
# config 5 smoke: ONE fixed recipe (round 1's Env01-v1 recipe), rollouts on the HIP policy / bootstrap / GAE kernels
python tools/train_ppo_torch.py --env Env01-v1 --envs 16384 --iters 60 --n-steps 64 --epochs 4 --minibatch 8192 --lr 3e-4 --gamma 0.999 --reward-clip 1.0 --device-rollout --eval-steps 1500 --out gpurun_out/ppo_r02/env01_v1_device_rollout.json > gpurun_out/ppo_r02/device.log 2>&1
tail -4 gpurun_out/ppo_r02/device.log | cut -c1-400
# the same recipe with the torch rollout (round 1's path) for comparison of wall time
python tools/train_ppo_torch.py --env Env01-v1 --envs 16384 --iters 60 --n-steps 64 --epochs 4 --minibatch 8192 --lr 3e-4 --gamma 0.999 --reward-clip 1.0 --eval-steps 1500 --out gpurun_out/ppo_r02/env01_v1_torch_rollout.json > gpurun_out/ppo_r02/torch.log 2>&1
tail -3 gpurun_out/ppo_r02/torch.log | cut -c1-400
