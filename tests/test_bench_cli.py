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
