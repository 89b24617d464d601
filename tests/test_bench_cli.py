"""bench.py's launcher logic, without a GPU: the contract says `--gpus N` must never silently time one GPU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra):
    env = dict(os.environ); env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_world_size_mismatch_exits_2_before_touching_the_gpu():
    p = _run(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "--gpus 4 but WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""
    p = _run(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2


def test_spawn_builds_a_torch_distributed_run_child_with_the_same_arguments(monkeypatch):
    """`python bench.py --gpus N` with no launcher: a CHILD `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    on 127.0.0.1 (never an exec of this process); rank 0's JSON line is relayed, everything else goes to stderr"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class P:
        returncode = 0
        stdout = 'noise from a rank\n{"metric": "env-steps/sec", "value": 1.0, "n_gpus": 2}\n'

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return P()

    import subprocess as sp
    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "7", "--streams", "2"])

    class A:
        gpus = 2
    rc = bench.spawn_ranks(A())
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "7", "--streams", "2"] and os.path.abspath(cmd[-7]) == BENCH
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


import pytest  # noqa: E402


@pytest.mark.gpu
def test_spawned_two_ranks_on_one_gpu_report_the_whole_job():
    """the whole `--gpus 2` path with the real kernels: bench.py spawns `torch.distributed.run` as a child, both ranks step their own
    shard (global env indices 0..n-1 / n..2n-1), barrier + synchronize around the timed loop, MAX over ranks, rank 0 prints ONE line
    whose value is the aggregate.  One GPU is all a test box has, so both ranks share device 0 and the timing reduction goes over
    gloo (BRS_BENCH_ONE_DEVICE / BRS_BENCH_BACKEND: rehearsal switches; the driver's 8-GPU run uses RCCL and one device per rank)."""
    import json
    env = dict(os.environ, BRS_BENCH_ONE_DEVICE="1", BRS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = ["--gpus", "2", "--env", "Env03-v2", "--envs", "4096", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 4096 and "8192 total" in d["config"]["workload"]
    # aggregate over both ranks: 2 x 4096 envs per step of the slower rank
    assert abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["value"] > 1e5
