#!/usr/bin/env python3
"""On-device PPO on the batched simulator (policy inference + rollout buffer on the GPU).

The reference trains with SB3 PPO on one CPU env (src/sb_rl.py:63-71, "several hours", README.md:129).  SB3 is not
installable here, so this is a minimal torch PPO with the same network shape (MlpPolicy: 6 -> 64 -> 64 tanh, separate
actor / critic towers, state-independent log-std) consuming BatchedSim tensors directly: no numpy, no per-env Python.

    python tools/train_ppo_torch.py --env Env01-v2 --envs 16384 --iters 60 [--then Env03-v2 --iters2 60] [--out log.json]
"""
import argparse, json, os, sys, time
import torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim


class ActorCritic(nn.Module):
    def __init__(self, log_std_init=-0.5):
        super().__init__()
        mk = lambda o: nn.Sequential(nn.Linear(6, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, o))
        self.pi, self.v = mk(2), mk(1)
        self.log_std = nn.Parameter(torch.full((2,), float(log_std_init)))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())


def train(sim, model, opt, iters, n_steps, epochs, minibatch, gamma, lam, clip, log, tag, ent=0.0):
    n = sim.n
    dev = sim.device
    obs = sim.reset().clone()
    ep_len = torch.zeros(n, device=dev); ep_ret = torch.zeros(n, device=dev)
    done_len_sum = done_ret_sum = done_cnt = 0.0
    B = {k: torch.zeros((n_steps, n) + s, device=dev) for k, s in dict(obs=(6,), act=(2,), logp=(), rew=(), val=(), done=(), boot=()).items()}
    t_start = time.time(); total = 0
    for it in range(iters):
        with torch.no_grad():
            for t in range(n_steps):
                d = model.dist(obs)
                a = d.sample()
                B["obs"][t] = obs; B["act"][t] = a; B["logp"][t] = d.log_prob(a).sum(-1); B["val"][t] = model.v(obs).squeeze(-1)
                o, r, te, tr, to = sim.step(a.clamp(-1, 1).contiguous())   # SB3 clips actions to the Box before env.step
                done = (te | tr).bool()
                # time-limit truncation (not a failure): bootstrap from the terminal observation, like SB3
                boot = torch.where(tr.bool() & ~te.bool(), model.v(to).squeeze(-1), torch.zeros_like(r))
                B["rew"][t] = r; B["done"][t] = done.float(); B["boot"][t] = boot
                ep_len += 1; ep_ret += r
                if done.any():
                    done_len_sum += ep_len[done].sum().item(); done_ret_sum += ep_ret[done].sum().item(); done_cnt += done.sum().item()
                    ep_len[done] = 0; ep_ret[done] = 0
                obs = o.clone()
            last_v = model.v(obs).squeeze(-1)
            adv = torch.zeros_like(B["rew"]); g = torch.zeros(n, device=dev)
            for t in reversed(range(n_steps)):
                nv = last_v if t == n_steps - 1 else B["val"][t + 1]
                nonterm = 1.0 - B["done"][t]
                delta = B["rew"][t] + gamma * (nv * nonterm + B["boot"][t]) - B["val"][t]
                g = delta + gamma * lam * nonterm * g
                adv[t] = g
            ret = adv + B["val"]
        flat = {k: v.reshape((-1,) + v.shape[2:]) for k, v in B.items()}
        fadv = adv.reshape(-1); fret = ret.reshape(-1)
        N = fadv.numel()
        for _ in range(epochs):
            perm = torch.randperm(N, device=dev)
            for s in range(0, N, minibatch):
                idx = perm[s:s + minibatch]
                d = model.dist(flat["obs"][idx])
                logp = d.log_prob(flat["act"][idx]).sum(-1)
                ratio = (logp - flat["logp"][idx]).exp()
                a_ = fadv[idx]; a_ = (a_ - a_.mean()) / (a_.std() + 1e-8)
                pl = -torch.min(ratio * a_, ratio.clamp(1 - clip, 1 + clip) * a_).mean()
                vl = 0.5 * (model.v(flat["obs"][idx]).squeeze(-1) - fret[idx]).pow(2).mean()
                loss = pl + 0.5 * vl - ent * d.entropy().sum(-1).mean()
                opt.zero_grad(set_to_none=True); loss.backward(); nn.utils.clip_grad_norm_(model.parameters(), 0.5); opt.step()
        total += n * n_steps
        if done_cnt > 0:
            row = dict(tag=tag, iter=it, env_steps=total, wall_s=round(time.time() - t_start, 2),
                       mean_ep_len=done_len_sum / done_cnt, mean_ep_ret=done_ret_sum / done_cnt, episodes=int(done_cnt))
            log.append(row); print(json.dumps(row), flush=True) if (it % 10 == 0 or it == iters - 1) else None
            done_len_sum = done_ret_sum = done_cnt = 0.0
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env01-v2"); ap.add_argument("--then", default="")
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--iters", type=int, default=60); ap.add_argument("--iters2", type=int, default=60)
    ap.add_argument("--n-steps", type=int, default=32); ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatch", type=int, default=65536)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--log-std-init", type=float, default=-0.5); ap.add_argument("--ent", type=float, default=0.0)
    ap.add_argument("--gamma", type=float, default=0.99); ap.add_argument("--lam", type=float, default=0.95)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = ActorCritic(a.log_std_init).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    log = []
    for env_id, iters in ((a.env, a.iters), (a.then, a.iters2)):
        if not env_id:
            continue
        sim = BatchedSim(env_id, a.envs, device=0, seed=0, auto_reset=True)
        train(sim, model, opt, iters, a.n_steps, a.epochs, a.minibatch, a.gamma, a.lam, 0.2, log, env_id, a.ent)
        sim.close()
    if a.out:
        json.dump(dict(args=vars(a), log=log), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
