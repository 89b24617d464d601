#!/usr/bin/env python3
"""On-device PPO on the batched simulator (policy inference + rollout buffer on the GPU).

The reference trains with SB3 PPO on one CPU env (src/sb_rl.py:63-71, "several hours", README.md:129).  SB3 is not
installable here, so this is a minimal torch PPO with the same network shape (MlpPolicy: 6 -> 64 -> 64 tanh, separate
actor / critic towers, state-independent log-std) consuming BatchedSim tensors directly: no numpy, no per-env Python.

    python tools/train_ppo_torch.py --env Env01-v2 --envs 16384 --iters 60 [--then Env03-v2 --iters2 60] [--out log.json]
"""
import argparse, json, os, sys, time
import torch
import torch.distributed as dist
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import BatchedSim


def flat_params(model):
    """the ActorCritic's parameters in the order of include/brs_policy.h (one device tensor, no host copy); the critic's
    output unit `ret_scale` is folded into its last layer"""
    p, v = model.pi, model.v
    parts = [p[0].weight, p[0].bias, p[2].weight, p[2].bias, p[4].weight, p[4].bias,
             v[0].weight, v[0].bias, v[2].weight, v[2].bias, v[4].weight * model.ret_scale, v[4].bias * model.ret_scale, model.log_std]
    return torch.cat([t.detach().reshape(-1) for t in parts]).contiguous()


class ActorCritic(nn.Module):
    def __init__(self, log_std_init=-0.5):
        super().__init__()
        mk = lambda o: nn.Sequential(nn.Linear(6, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, o))
        self.pi, self.v = mk(2), mk(1)
        self.log_std = nn.Parameter(torch.full((2,), float(log_std_init)))
        self.register_buffer("ret_scale", torch.ones(()))   # critic output unit (running std of the returns)

    def value(self, obs):
        return self.v(obs).squeeze(-1) * self.ret_scale

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_grads(model):
    """data-parallel learner: every rank steps its own env shard; the only exchange is ONE all-reduce
    of the flattened gradient (~10 k floats) per optimiser step -- RCCL over xGMI under backend "nccl", gloo in the CPU test"""
    w = _world()
    if w == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= w
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()


def _allreduce_mean_(t):
    if _world() > 1:
        dist.all_reduce(t); t /= _world()
    return t


class MixedSim:
    """several BatchedSims stepped back to back and presented as one batch (second phase: keep part of the envs in the
    first phase's environment so the policy is not allowed to forget it)"""

    def __init__(self, sims):
        self.sims = sims; self.n = sum(s.n for s in sims); self.device = sims[0].device
        self._cuts = [s.n for s in sims]

    def reset(self):
        return torch.cat([s.reset() for s in self.sims])

    def step(self, a):
        outs = [s.step(p.contiguous()) for s, p in zip(self.sims, a.split(self._cuts))]
        return tuple(torch.cat([o[k] for o in outs]) for k in range(5))

    def close(self):
        for s in self.sims:
            s.close()


def train(sim, model, opt, iters, n_steps, epochs, minibatch, gamma, lam, clip, log, tag, ent=0.0, reward_clip=None,
          critic_warmup=0, lr_end=None, norm_returns=False, target_kl=None, device_rollout=False, seed=0, env_index_base=0):
    n = sim.n
    dev = sim.device
    pol = None
    if device_rollout:   # rollout side on the HIP kernels of include/brs_policy.h (forward + sample, bootstrap, GAE)
        from balance_robot_mujoco_rl_amd.policy import DevicePolicy, gae as gae_kernel
        pol = DevicePolicy(device=dev.index, seed=seed, env_index_base=env_index_base)
        a_buf = torch.zeros((n, 2), device=dev); start = torch.ones(n, dtype=torch.uint8, device=dev)
        S = torch.zeros((n_steps, n), dtype=torch.uint8, device=dev); gstep = 0
    obs = sim.reset().clone()
    ep_len = torch.zeros(n, device=dev); ep_ret = torch.zeros(n, device=dev)
    done_len_sum = done_ret_sum = done_cnt = 0.0
    B = {k: torch.zeros((n_steps, n) + s, device=dev) for k, s in dict(obs=(6,), act=(2,), logp=(), rew=(), val=(), done=(), boot=()).items()}
    t_start = time.time(); total = 0
    lr0 = opt.param_groups[0]["lr"]
    for it in range(iters):
        if lr_end is not None:   # linear anneal over the phase
            for g_ in opt.param_groups:
                g_["lr"] = lr0 + (lr_end - lr0) * it / max(1, iters - 1)
        with torch.no_grad():
            if pol is not None:
                fp = flat_params(model); pol.use_device_weights(fp)   # the learner's current weights, read in place
            for t in range(n_steps):
                if pol is not None:
                    B["obs"][t] = obs; S[t] = start
                    pol.act(B["obs"][t], gstep, out=(B["act"][t], a_buf, B["logp"][t], B["val"][t])); gstep += 1
                    o, r, te, tr, to = sim.step(a_buf)
                    done = (te | tr).bool()
                    B["rew"][t] = r if reward_clip is None else r.clamp(max=reward_clip)
                    pol.bootstrap(to, te, tr, gamma, B["rew"][t])   # rew += gamma V(terminal_obs) where truncated only
                    B["done"][t] = done.float(); start = (te | tr).clone()
                else:
                    d = model.dist(obs)
                    a = d.sample()
                    B["obs"][t] = obs; B["act"][t] = a; B["logp"][t] = d.log_prob(a).sum(-1); B["val"][t] = model.value(obs)
                    o, r, te, tr, to = sim.step(a.clamp(-1, 1).contiguous())   # SB3 clips actions to the Box before env.step
                    done = (te | tr).bool()
                    # time-limit truncation (not a failure): bootstrap from the terminal observation, like SB3
                    boot = torch.where(tr.bool() & ~te.bool(), model.value(to), torch.zeros_like(r))
                    # learner-side reward clipping (a TransformReward-style wrapper); logged returns stay the env's own
                    B["rew"][t] = r if reward_clip is None else r.clamp(max=reward_clip)
                    B["done"][t] = done.float(); B["boot"][t] = boot
                ep_len += 1; ep_ret += r
                if done.any():
                    done_len_sum += ep_len[done].sum().item(); done_ret_sum += ep_ret[done].sum().item(); done_cnt += done.sum().item()
                    ep_len[done] = 0; ep_ret[done] = 0
                obs = o.clone()
            if pol is not None:
                adv, ret = gae_kernel(B["rew"], B["val"], S, pol.value(obs), start, gamma, lam)
            else:
                last_v = model.value(obs)
                adv = torch.zeros_like(B["rew"]); g = torch.zeros(n, device=dev)
                for t in reversed(range(n_steps)):
                    nv = last_v if t == n_steps - 1 else B["val"][t + 1]
                    nonterm = 1.0 - B["done"][t]
                    delta = B["rew"][t] + gamma * (nv * nonterm + B["boot"][t]) - B["val"][t]
                    g = delta + gamma * lam * nonterm * g
                    adv[t] = g
                ret = adv + B["val"]
            if norm_returns:
                new_scale = _allreduce_mean_(ret.std()).clamp(min=1.0)   # same on all ranks
                if it == 0 and float(model.ret_scale) == 1.0:
                    # the unit is switched on here (e.g. at the start of the second phase): keep the critic's predictions where
                    # they are by folding the change of unit into its last layer
                    k = float(model.ret_scale / new_scale)
                    model.v[4].weight.mul_(k); model.v[4].bias.mul_(k)
                model.ret_scale.lerp_(new_scale, 0.05 if it else 1.0)
        flat = {k: v.reshape((-1,) + v.shape[2:]) for k, v in B.items()}
        fadv = adv.reshape(-1); fret = ret.reshape(-1)
        N = fadv.numel()
        n_upd = 0; kl = 0.0; stop = False
        for _ in range(epochs):
            if stop:
                break
            perm = torch.randperm(N, device=dev)
            for s in range(0, N, minibatch):
                idx = perm[s:s + minibatch]
                d = model.dist(flat["obs"][idx])
                logp = d.log_prob(flat["act"][idx]).sum(-1)
                lr_ = logp - flat["logp"][idx]
                ratio = lr_.exp()
                if target_kl is not None and it >= critic_warmup:   # SB3's target_kl: stop this iteration's updates when the
                    with torch.no_grad():                            # policy has moved far enough from the one that sampled
                        kl = float(_allreduce_mean_(((ratio - 1) - lr_).mean()))
                    if kl > 1.5 * target_kl:
                        stop = True
                        break
                a_ = fadv[idx]; a_ = (a_ - a_.mean()) / (a_.std() + 1e-8)
                pl = -torch.min(ratio * a_, ratio.clamp(1 - clip, 1 + clip) * a_).mean()
                vl = 0.5 * (model.v(flat["obs"][idx]).squeeze(-1) - fret[idx] / model.ret_scale).pow(2).mean()
                if it < critic_warmup:   # new env: let the critic catch up before the actor moves
                    loss = 0.5 * vl
                else:
                    loss = pl + 0.5 * vl - ent * d.entropy().sum(-1).mean()
                opt.zero_grad(set_to_none=True); loss.backward(); allreduce_mean_grads(model)
                # actor and critic are separate towers: clip them separately so large value targets cannot starve the actor
                nn.utils.clip_grad_norm_(list(model.pi.parameters()) + [model.log_std], 0.5)
                nn.utils.clip_grad_norm_(model.v.parameters(), 0.5); opt.step(); n_upd += 1
        total += n * n_steps * _world()
        if _world() > 1:
            st = torch.tensor([done_len_sum, done_ret_sum, done_cnt], dtype=torch.float64, device=dev)
            dist.all_reduce(st); done_len_sum, done_ret_sum, done_cnt = st.tolist()
        if done_cnt > 0:
            row = dict(tag=tag, iter=it, env_steps=total, wall_s=round(time.time() - t_start, 2),
                       mean_ep_len=done_len_sum / done_cnt, mean_ep_ret=done_ret_sum / done_cnt, episodes=int(done_cnt),
                       log_std=round(float(model.log_std.detach().mean()), 3), updates=n_upd, kl=round(kl, 4))
            log.append(row)
            if (it % 10 == 0 or it == iters - 1) and (not dist.is_initialized() or dist.get_rank() == 0):
                print(json.dumps(row), flush=True)
            done_len_sum = done_ret_sum = done_cnt = 0.0
    return total


@torch.no_grad()
def evaluate(env_id, model, n, steps, seed=123, device=0):
    """deterministic policy (mean action) on fresh envs: episode-length statistics and the share of episodes that
    run into the time limit (= balanced for the whole episode)"""
    sim = BatchedSim(env_id, n, device=0, seed=seed, auto_reset=True)
    obs = sim.reset().clone()
    ep_len = torch.zeros(n, device=sim.device); ep_ret = torch.zeros(n, device=sim.device)
    lens, rets, ntrunc, nterm = [], [], 0, 0
    ever_done = torch.zeros(n, dtype=torch.bool, device=sim.device)
    for _ in range(steps):
        o, r, te, tr, _to = sim.step(model.pi(obs).clamp(-1, 1).contiguous())
        ep_len += 1; ep_ret += r
        done = (te | tr).bool()
        if done.any():
            lens.append(ep_len[done].clone()); rets.append(ep_ret[done].clone())
            ntrunc += int((tr.bool() & ~te.bool()).sum()); nterm += int(te.bool().sum())
            ep_len[done] = 0; ep_ret[done] = 0; ever_done |= done
        obs = o.clone()
    still = int((~ever_done).sum())   # first episode still running after `steps` steps
    sim.close()
    lens = torch.cat(lens) if lens else torch.zeros(0); rets = torch.cat(rets) if rets else torch.zeros(0)
    return dict(env=env_id, envs=n, steps=steps, episodes=int(lens.numel()), first_episode_still_running=still, reached_time_limit=ntrunc, fell=nterm,
                frac_reached_time_limit=ntrunc / max(1, ntrunc + nterm), mean_ep_len=float(lens.mean()) if lens.numel() else None,
                median_ep_len=float(lens.median()) if lens.numel() else None, mean_ep_ret=float(rets.mean()) if rets.numel() else None)


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
    ap.add_argument("--lr2", type=float, default=None, help="learning rate of the second phase (default: --lr)")
    ap.add_argument("--log-std2", type=float, default=None, help="reset the policy log-std to this at the second phase")
    ap.add_argument("--fix-log-std2", action="store_true", help="do not train the log-std in the second phase (keeps exploring)")
    ap.add_argument("--critic-warmup2", type=int, default=0, help="value-only iterations at the start of the second phase")
    ap.add_argument("--lr2-end", type=float, default=None, help="anneal the second phase's learning rate linearly to this")
    ap.add_argument("--norm-returns", action="store_true", help="critic in units of the running std of the returns")
    ap.add_argument("--norm-returns2", action="store_true", help="... from the second phase on")
    ap.add_argument("--target-kl", type=float, default=None, help="stop an iteration's updates at 1.5x this approximate KL")
    ap.add_argument("--target-kl2", type=float, default=None, help="... in the second phase only")
    ap.add_argument("--seed", type=int, default=0, help="seeds the initial weights, the action noise and the env streams")
    ap.add_argument("--mix2", type=float, default=0.0, help="share of the envs kept in --env during the second phase")
    ap.add_argument("--reward-clip", type=float, default=None, help="learner-side upper clip of the per-step reward")
    ap.add_argument("--eval-steps", type=int, default=0, help="after training: deterministic evaluation for this many steps")
    ap.add_argument("--eval-envs", type=int, default=4096)
    ap.add_argument("--device-rollout", action="store_true",
                    help="act, bootstrap and GAE with the HIP kernels of include/brs_policy.h instead of torch ops")
    ap.add_argument("--obs-init-scale", default="", help="comma-separated factors on the INITIAL first-layer weights per observation "
                    "channel (both towers), e.g. 1,0.1,1,1,1,1: Env01-v2's obs[1] is a finite difference of two noisy pitch samples "
                    "(+-10 rad/s of noise, envs/env01_v2.py:16-20 with RobotBaseEnv.py:142-157) and saturates freshly initialised tanh units")
    ap.add_argument("--save", default="", help="write the policy/value weights (torch state_dict) here")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BRS_PPO_ONE_DEVICE") == "1":   # rehearsal of N ranks on a one-GPU box (with BRS_PPO_BACKEND=gloo)
        local = 0
    if world > 1:   # python -m torch.distributed.run --nproc-per-node N tools/train_ppo_torch.py ...: --envs is per rank
        torch.cuda.set_device(local)
        if os.environ.get("BRS_PPO_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(os.environ["BRS_PPO_BACKEND"])
    torch.manual_seed(a.seed)   # identical initial weights on every rank
    dev = torch.device("cuda", local)
    model = ActorCritic(a.log_std_init).to(dev)
    if a.obs_init_scale:
        sc = torch.tensor([float(x) for x in a.obs_init_scale.split(",")], device=dev)
        with torch.no_grad():
            model.pi[0].weight.mul_(sc); model.v[0].weight.mul_(sc)
    torch.manual_seed(1000 * (a.seed + 1) + rank)   # ... different action noise
    base = rank * a.envs
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    log = []
    for phase, (env_id, iters) in enumerate(((a.env, a.iters), (a.then, a.iters2))):
        if not env_id:
            continue
        warm = 0
        if phase == 1:
            warm = a.critic_warmup2
            if a.lr2 is not None:
                for g in opt.param_groups:
                    g["lr"] = a.lr2
            if a.log_std2 is not None:
                with torch.no_grad():
                    model.log_std.fill_(a.log_std2)
            if a.fix_log_std2:
                model.log_std.requires_grad_(False)
        if phase == 1 and a.mix2 > 0:
            keep = int(a.envs * a.mix2) // 64 * 64
            sim = MixedSim([BatchedSim(a.env, keep, device=local, seed=2 * a.seed + 1, env_index_base=base, auto_reset=True),
                            BatchedSim(env_id, a.envs - keep, device=local, seed=2 * a.seed, env_index_base=base, auto_reset=True)])
            env_id = f"{a.env}+{env_id}"
        else:
            sim = BatchedSim(env_id, a.envs, device=local, seed=2 * a.seed, env_index_base=base, auto_reset=True)
        train(sim, model, opt, iters, a.n_steps, a.epochs, a.minibatch, a.gamma, a.lam, 0.2, log, env_id, a.ent, a.reward_clip,
              warm, a.lr2_end if phase == 1 else None, a.norm_returns or (phase == 1 and a.norm_returns2),
              a.target_kl2 if (phase == 1 and a.target_kl2 is not None) else a.target_kl, a.device_rollout, 1000 * (a.seed + 1) + phase, base)
        sim.close()
    evals = []
    if world > 1:
        dist.barrier()
    if rank != 0:
        dist.destroy_process_group(); return
    if a.eval_steps > 0:
        for env_id in dict.fromkeys(e for e in (a.env, a.then) if e):
            evals.append(evaluate(env_id, model, a.eval_envs, a.eval_steps)); print(json.dumps(evals[-1]), flush=True)
    if a.save:
        torch.save(model.state_dict(), a.save)
    if a.out:
        json.dump(dict(args=vars(a), world_size=world, log=log, eval=evals), open(a.out, "w"), indent=1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
