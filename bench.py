#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched balance-robot simulator (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 65536] [--env Env03-v2]

Workload (BASELINE.json configs[2]): Env03-v2, 65,536 env instances PER GPU, random policy
(actions ~ U(-1,1)^2, device generator seed 1234), auto-reset on; one "step" = one env step of every instance =
250 physics substeps each.  Inputs (state, pre-generated action batches) are resident in HBM when timing starts.
N > 1: launched by torch.distributed.run, one rank per GPU; instances shard by global index, no collective on the
step path (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md chip table; 6.29 TB/s measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3


def cpu_baseline(env_id, seconds_budget=20.0):
    """oracle/ ("port": this repo's fp64 CPU restatement, NOT MuJoCo -- MuJoCo is not installable here) timed on
    the host cores on a bounded sample of the same workload"""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 64)
    n = 16 * cores
    orc = O.Oracle(env_id, n, seed=0, auto_reset=True, threads=cores)
    orc.reset()
    rng = np.random.default_rng(1234)
    act = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
    orc.step(act)  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.step(rng.uniform(-1, 1, size=(n, 2)).astype(np.float32))
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 400:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{env_id}, {n} envs x {steps} steps, random actions, auto-reset, oracle/brs_oracle.c fp64 "
                      f"with OpenMP over envs ({el:.1f} s); own CPU restatement, not MuJoCo"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="env instances per GPU")
    ap.add_argument("--env", default="Env03-v2")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("BRS_BENCH_BACKEND", "nccl")  # "gloo" + BRS_BENCH_ONE_DEVICE=1: rehearse N ranks on one GPU
    if os.environ.get("BRS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    # the HIP library is built in-tree by __graft_entry__.build(); if this checkout has none yet, compile it now (still
    # the HIP path -- there is no other): rank 0 builds, the others wait
    from balance_robot_mujoco_rl_amd import _lib
    if rank == 0:
        _lib.build()
    if dist is not None:
        dist.barrier()
    from balance_robot_mujoco_rl_amd import BatchedSim
    n = args.envs
    sim = BatchedSim(args.env, n, device=dev.index, seed=0, env_index_base=rank * n, auto_reset=True,
                     block_threads=args.block_threads)
    sim.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = 16
    actions = [(torch.rand((n, 2), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(pool)]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        sim.step(actions[k % pool])
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()
        sim.step(actions[k % pool])
        ev1[k].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))  # HIP events on the launch stream
    n_done = int(sim.terminated.sum().item())  # touches the outputs (also proves the last step ran)

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        bytes_per_launch = sim.step_bytes_per_env() * n
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, {n} env instances per GPU ({total_envs} total), random policy "
                                   f"U(-1,1)^2, auto-reset on, 250 substeps of 2e-5 s per env step",
                       "envs_per_gpu": n, "substeps": 250, "parallelism": f"env-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": sim.step_kernel_name(), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_env_step": sim.step_bytes_per_env(),
                         "note": "state stays in registers for 250 fused substeps, so HBM traffic is ~0.1% of peak by "
                                 "construction; the binding resource is fp32 VALU issue (see DESIGN.md)"},
            "substeps_per_s": value * 250,
            "last_step_terminated": n_done,
        }
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process; when the committed
        # rocprofv3 summary of this exact workload exists, report its per-launch figure (FETCH_SIZE x2 gfx950 correction
        # + WRITE_SIZE, separate --pmc passes), else null
        try:
            if args.env == "Env03-v2" and n == 65536:
                summ = json.load(open(os.path.join(ROOT, "profiles", "r01_env03_summary.json")))
                prof = summ["hbm_traffic"]
                out["roofline"]["traffic"] = prof["fetch_bytes_x2_corrected"] + prof["write_bytes"]
                # the resource that does bind (same PMC summary): share of wave cycles with the VALU busy, one wave per SIMD
                out["roofline"]["valu"] = {"busy_frac": summ["valu_busy_frac"], "wait_frac": summ["wait_frac"],
                                           "valu_insts_per_wave_per_step": summ["per_wave_step"]["SQ_INSTS_VALU"]}
                out["roofline"]["traffic_source"] = "profiles/r01_env03_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        except Exception:
            pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.env)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
