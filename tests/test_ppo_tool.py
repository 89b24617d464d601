"""tools/train_ppo_torch.py: the data-parallel learner (one all-reduce of the flattened gradient per optimiser step) keeps
the ranks' weights identical -- world_size 2 over gloo on the CPU with a toy env (the GPU sim is not needed for this)."""
import json, os, socket, subprocess, sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, sync):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gloo_ppo_worker.py"), str(r), "2", str(port),
                               str(tmp_path), "1" if sync else "0"]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [json.load(open(tmp_path / f"ppo_rank{r}.json")) for r in range(2)]


def test_ranks_stay_in_sync_with_gradient_allreduce(tmp_path):
    a, b = _run(tmp_path, sync=True)
    assert a["rows"] >= 2 and a["env_steps"] == 3 * 16 * 64 * 2, "env_steps counts the whole job"
    np.testing.assert_array_equal(np.array(a["params"]), np.array(b["params"]))


def test_without_allreduce_the_ranks_diverge(tmp_path):
    a, b = _run(tmp_path, sync=False)
    assert a["env_steps"] == 3 * 16 * 64
    assert np.abs(np.array(a["params"]) - np.array(b["params"])).max() > 1e-6
