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
# committed rocprofv3 summaries of the default command per env id (tools/profile_round.sh); each carries the build id it measured
PROFILE_SUMMARY = {"Env03-v2": "r03_env03_summary.json", "Env01-v2": "r03_env01_summary.json"}


def cpu_baseline(env_id, seconds_budget=12.0):
    """oracle/ ("port": this repo's fp64 CPU restatement, NOT MuJoCo -- MuJoCo is not installable here) timed on
    the host cores on a bounded sample of the same workload"""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 64)
    n = 16 * cores
    orc = O.Oracle(env_id, n, seed=0, auto_reset=True, threads=cores)
    orc.reset()
    rng = np.random.default_rng(1234)
    for _ in range(40):  # past the 2 cm drop that starts every episode: wheels on the floor, first blocks arriving
        orc.step(rng.uniform(-1, 1, size=(n, 2)).astype(np.float32))
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.step(rng.uniform(-1, 1, size=(n, 2)).astype(np.float32))
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 400:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "us_per_substep_per_core": el * cores / (n * steps * 250) * 1e6,
            "sample": f"{env_id}, {n} envs x {steps} steps after 40 untimed, random actions, auto-reset, "
                      f"oracle/brs_oracle.c fp64 (general body tree, numeric Jacobians, MuJoCo-style Newton) with "
                      f"OpenMP over envs ({el:.1f} s); own CPU restatement, not MuJoCo"}


def cpu_baseline_closed_form(env_id, seconds_budget=10.0):
    """second CPU figure: the kernel's own source (closed-form gyrostat dynamics, active-set Newton) compiled for the
    host in double with OpenMP over envs (tests/hostsim -- test infrastructure, timed only here).  SURVEY.md section 6
    estimates MuJoCo itself at 2-6 us per substep per core on this model."""
    from tests.hostsim.hostsim import HostSim
    cores = min(os.cpu_count() or 1, 64)
    n = 32 * cores
    hs = HostSim(env_id, n, seed=0, auto_reset=True, double=True, threads=cores)
    hs.reset()
    rng = np.random.default_rng(1234)
    for _ in range(40):
        hs.step(rng.uniform(-1, 1, size=(n, 2)).astype(np.float32))
    t0 = time.perf_counter()
    steps = 0
    while True:
        hs.step(rng.uniform(-1, 1, size=(n, 2)).astype(np.float32))
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 2000:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "us_per_substep_per_core": el * cores / (n * steps * 250) * 1e6,
            "sample": f"{env_id}, {n} envs x {steps} steps after 40 untimed, random actions, auto-reset, host build of "
                      f"the kernel source in fp64 (tests/hostsim) with OpenMP over envs ({el:.1f} s); not MuJoCo"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD before anything here has
    touched the GPU (a process that initialised HIP must never exec), relay rank 0's JSON line and exit code"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return proc.returncode if line is not None or proc.returncode else 1


class SubBatches:
    """--streams S: the envs of one GPU as S handles of n/S envs (contiguous global env indices, same Philox streams as
    the single handle), each stepping on its own HIP stream.  step(k) enqueues one launch per handle and returns; there
    is no barrier between the handles until the caller synchronises the device."""

    def __init__(self, BatchedSim, args, n, S, dev, rank, actions):
        m = n // S
        self.m, self.S, self.dev = m, S, dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        self.sims = [BatchedSim(args.env, m, device=dev.index, seed=0, env_index_base=rank * n + s * m, auto_reset=True,
                                block_threads=args.block_threads, lane_grouping=not args.no_lane_grouping) for s in range(S)]
        self.acts = [[a[s * m:(s + 1) * m].contiguous() for a in actions] for s in range(S)]
        self.last_events = None

    def reset(self):
        for sim in self.sims:
            sim.reset()
        torch.cuda.synchronize(self.dev)

    def step(self, k, timed=None):
        for s, sim in enumerate(self.sims):
            with torch.cuda.stream(self.streams[s]):
                if timed is not None:
                    timed[0][s].record()
                sim.step(self.acts[s][k])
                if timed is not None:
                    timed[1][s].record()

    @property
    def terminated(self):
        return torch.cat([sim.terminated for sim in self.sims])

    def step_bytes_per_env(self):
        return self.sims[0].step_bytes_per_env()

    def step_kernel_name(self):
        return self.sims[0].step_kernel_name()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="env instances per GPU")
    ap.add_argument("--env", default="Env03-v2")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--preroll", type=int, default=300,
                    help="minimum untimed env steps after reset(), on top of --warmup: all episodes start in phase (2 cm "
                         "drop, no contacts, first block in flight); the pre-roll runs until falls and auto-resets have "
                         "de-phased them and the per-step kernel time is stationary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the envs of a GPU over this many handles, each stepping on its own HIP stream with no barrier "
                         "between them until the end (asynchronous sub-batches: a finished sub-batch's SIMDs are back-filled "
                         "by the next launch of another); 1 = one launch per step")
    ap.add_argument("--no-lane-grouping", action="store_true", help="A/B: keep env i on lane i (BRS_FLAG_NO_LANE_GROUPING)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
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
    S = args.streams
    if S < 1 or n % S:
        print(f"bench.py: --envs {n} is not divisible by --streams {S}", file=sys.stderr)
        sys.exit(2)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = 16
    actions = [(torch.rand((n, 2), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(pool)]
    if S > 1:
        sim = SubBatches(BatchedSim, args, n, S, dev, rank, actions)
        actions = list(range(pool))  # SubBatches.step takes the index into its own per-handle action slices
    else:
        sim = BatchedSim(args.env, n, device=dev.index, seed=0, env_index_base=rank * n, auto_reset=True,
                         block_threads=args.block_threads, lane_grouping=not args.no_lane_grouping)
    sim.reset()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- untimed pre-roll to steady state (independent of --warmup): windows of 50 steps, HIP-event time per window;
    # stop once `--preroll` steps are done AND two consecutive windows agree within 2 %, or at the cap
    kstep = 0
    win, cap = 50, max(args.preroll, 1500)
    win_ms = []
    while True:
        if S > 1:  # no single stream sees all launches: wall clock around a synchronised window
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            for _ in range(win):
                sim.step(actions[kstep % pool]); kstep += 1
            torch.cuda.synchronize(dev)
            win_ms.append((time.perf_counter() - tw) * 1e3 / win)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(win):
                sim.step(actions[kstep % pool]); kstep += 1
            e1.record()
            e1.synchronize()
            win_ms.append(e0.elapsed_time(e1) / win)
        stationary = len(win_ms) >= 2 and abs(win_ms[-1] - win_ms[-2]) <= 0.02 * win_ms[-2]
        if (kstep >= args.preroll and stationary) or kstep >= cap:
            break
    preroll = kstep
    for _ in range(args.warmup):
        sim.step(actions[kstep % pool]); kstep += 1
    if S > 1:
        ev0 = [[torch.cuda.Event(enable_timing=True) for _ in range(S)] for _ in range(args.steps)]
        ev1 = [[torch.cuda.Event(enable_timing=True) for _ in range(S)] for _ in range(args.steps)]
    else:
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if S > 1:
            sim.step(actions[kstep % pool], timed=(ev0[k], ev1[k])); kstep += 1
        else:
            ev0[k].record()
            sim.step(actions[kstep % pool]); kstep += 1
            ev1[k].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # HIP events on the stream the kernel is launched on (BatchedSim launches on torch's current stream)
    if S > 1:  # per-launch durations of the sub-batch launches (they overlap: not additive)
        kms = np.array([a.elapsed_time(b) for r0, r1 in zip(ev0, ev1) for a, b in zip(r0, r1)])
    else:
        kms = np.array([a.elapsed_time(b) for a, b in zip(ev0, ev1)])
    kern_ms = float(kms.mean())
    n_done = int(sim.terminated.sum().item())  # touches the outputs (also proves the last step ran)

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        bytes_per_launch = sim.step_bytes_per_env() * (n // S)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, {n} env instances per GPU ({total_envs} total), random policy "
                                   f"U(-1,1)^2, auto-reset on, 250 substeps of 2e-5 s per env step",
                       "envs_per_gpu": n, "substeps": 250, "parallelism": f"env-sharded x{world}, no collective",
                       "lane_grouping": not args.no_lane_grouping, "streams": S, "preroll_steps": preroll, "preroll_window_ms_per_step": [round(x, 4) for x in win_ms]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": sim.step_kernel_name(), "kernel_ms": kern_ms,
                         "kernel_ms_min_median_max": [float(kms.min()), float(np.median(kms)), float(kms.max())],
                         "algorithmic_bytes_per_env_step": sim.step_bytes_per_env(),
                         "note": "state stays in registers for 250 fused substeps, so HBM traffic is ~0.1% of peak by "
                                 "construction; the binding resource is fp32 VALU issue (see DESIGN.md)"},
            "substeps_per_s": value * 250,
            "last_step_terminated": n_done,
        }
        # HBM bytes per launch and the VALU instruction counts come from PMC counters, which cannot be collected from inside
        # this process: they are DERIVED FROM THE COMMITTED PROFILE of this exact command (separate rocprofv3 --pmc passes,
        # FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes) -- and only when that profile carries the build id of
        # the library that ran just now (include/brs.h: brs_build_id = hash of the kernel sources and flags); else null
        out["build_id"] = _lib.build_id()
        prof_path = os.path.join(ROOT, "profiles", PROFILE_SUMMARY.get(args.env, ""))
        out["roofline"]["derived_from_profile"] = None
        if n == 65536 and S == 1 and os.path.isfile(prof_path):
            try:
                summ = json.load(open(prof_path))
                if summ.get("build_id") != out["build_id"]:
                    out["roofline"]["stale_profile"] = (f"{os.path.relpath(prof_path, ROOT)} was measured on build {summ.get('build_id')}, this "
                                                        f"library is {out['build_id']}: traffic / valu withheld (re-run tools/profile_round.sh)")
                else:
                    out["roofline"]["traffic"] = summ["hbm_traffic"]["bytes_per_launch"]
                    out["roofline"]["derived_from_profile"] = os.path.relpath(prof_path, ROOT)
                    pv = summ.get("valu")
                    if pv:
                        # the resource that binds: VALU issue.  Instructions per wave-step from the profile's PMC run, time from THIS
                        # run.  Two ceilings at the 2.4 GHz boost clock over 1,024 SIMDs: the CHIP's (one wave64 VALU instruction
                        # per 2 clocks per SIMD, reachable only with >= 2 resident waves: MI355X_MICROARCH.md, wave scheduling)
                        # and what ONE resident wave per SIMD can issue (one per 4 clocks) -- 65,536 envs are 1,024 waves
                        v = dict(pv)
                        waves = (n + 63) // 64
                        rate = v["valu_insts_per_wave_per_step"] * waves / (kern_ms * 1e-3)
                        v["valu_wave_insts_per_s"] = rate
                        v["chip_peak_valu_wave_insts_per_s"] = 1024 * 2.4e9 / 2
                        v["issue_frac_of_chip_peak"] = rate / v["chip_peak_valu_wave_insts_per_s"]
                        v["issue_frac_of_one_wave_per_simd_ceiling"] = rate / (1024 * 2.4e9 / 4)
                        out["roofline"]["valu"] = v
                        out["roofline"]["binding_resource"] = "valu_issue"
            except Exception as e:
                out["roofline"]["stale_profile"] = f"profile summary unreadable: {e}"
        if not args.no_cpu_baseline:  # rank 0's host cores, whatever the world size
            out["cpu_baseline"] = cpu_baseline(args.env)
            try:
                out["cpu_baseline_closed_form"] = cpu_baseline_closed_form(args.env)
            except Exception as e:  # the host build is test infrastructure; its absence must not break the bench line
                out["cpu_baseline_closed_form"] = {"error": str(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
