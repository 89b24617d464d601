"""worker of test_ppo_tool.py: world_size-2 gloo run of tools/train_ppo_torch.py's learner on a toy CPU env.
usage: python gloo_ppo_worker.py <rank> <world> <port> <outdir> <sync:0|1>"""
import json, os, sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import train_ppo_torch as T


class ToySim:
    """BatchedSim's tensor surface on the CPU: 6-d state pulled towards 0 by the action, reward 1 - |x0|"""

    def __init__(self, n, seed):
        self.n, self.device = n, torch.device("cpu")
        self.g = torch.Generator().manual_seed(seed)
        self.x = torch.zeros(n, 6); self.t = torch.zeros(n)

    def reset(self):
        self.x = torch.rand(self.n, 6, generator=self.g) - 0.5; self.t.zero_()
        return self.x.clone()

    def step(self, a):
        self.x[:, :2] += 0.1 * a; self.x[:, 2:] *= 0.9; self.t += 1
        r = 1 - self.x[:, 0].abs()
        te = (self.x[:, 0].abs() > 1.5).to(torch.uint8); tr = (self.t >= 20).to(torch.uint8)
        tob = self.x.clone()
        d = (te | tr).bool()
        if d.any():
            self.x[d] = torch.rand(int(d.sum()), 6, generator=self.g) - 0.5; self.t[d] = 0
        return self.x.clone(), r, te, tr, tob


def main():
    rank, world, port, outdir, sync = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1"
    if sync:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = T.ActorCritic(-0.5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    torch.manual_seed(1000 + rank)
    log = []
    T.train(ToySim(64, seed=rank), model, opt, 3, 16, 2, 256, 0.99, 0.95, 0.2, log, "toy", norm_returns=True)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [model.ret_scale.reshape(1)])
    json.dump(dict(rank=rank, params=flat.tolist(), rows=len(log), env_steps=log[-1]["env_steps"] if log else 0),
              open(os.path.join(outdir, f"ppo_rank{rank}.json"), "w"))
    if sync:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
