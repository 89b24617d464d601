"""Closed-loop behavioural fixture: the balance/move policy the reference ships
(ref: envs/RobotMovePolicy.tflite -- an int8 export of an SB3 PPO MlpPolicy trained against MuJoCo; driven at
ref: envs/RobotMoveBaseEnv.py:178-208 with the observation layout of ref: envs/RobotBaseEnv.py:221-246) has to keep OUR
simulated robot on its wheels and follow the wheel-speed schedule of Env01-v3 (ref: envs/env01_v3.py:28-36: target =
dts, -dts, 2 dts, 3 dts with |dts| in [10, 20] rad/s, switched at t = 1, 3, 4.5, 5.5 s; pitch sensor offset of up to 2 deg).

Not a parity test -- MuJoCo itself is not installable here (physics parity UNPINNED) -- but an independent
check of the dynamics: a controller tuned on MuJoCo's version of this robot transfers to the oracle, to the kernel source
and to the HIP path without retuning.  Weights: tests/golden/robot_move_policy.npz (tools/gen_policy_fixture.py);
evaluator: tests/quant_policy.py (integer-exact accumulations; parity with the TFLite interpreter unpinned).
"""
import os, sys
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from quant_policy import QuantMovePolicy  # noqa: E402

PHASE_ENDS = (600, 900, 1100, 1400)  # env steps (5 ms each) at which the schedule of env01_v3.py:28-36 switches / the run ends


def drive(sim_step, obs, act, steps):
    """-> alive [steps, n], |obs[:, 4]| (normalised wheel-speed error) and |pitch| per step"""
    n = obs.shape[0]
    alive = np.ones(n, bool)
    A, E, P = [], [], []
    for _ in range(steps):
        obs, _, te, _, _ = sim_step(act(obs))
        alive &= ~te
        A.append(alive.copy()); E.append(np.abs(obs[:, 4])); P.append(np.abs(obs[:, 0]) * 0.25)
    return np.array(A), np.array(E), np.array(P)


def check_tracking(A, E, P, ends, err_max, what):
    assert A[-1].all(), f"{what}: {int((~A[-1]).sum())} of {A.shape[1]} robots fell"
    for e in ends:
        err, pit = E[e - 20:e].mean(axis=0), P[e - 20:e].max(axis=0)
        # obs[4] = (target - wheel_speed) / 170 * 4 (RobotBaseEnv.py:236): 0.1 = 4.25 rad/s
        assert err.max() < err_max and np.median(err) < 0.6 * err_max, f"{what}: wheel-speed error before step {e}: max {err.max():.3f} median {np.median(err):.3f}"
        # (the 2x target is held for 1 s only: the robot is still leaning into the acceleration when it ends)
        assert pit.max() < 0.3, f"{what}: pitch {pit.max():.3f} rad before step {e}"


def np_policy(pol, which):
    return lambda obs: pol.act(torch.from_numpy(obs), which).numpy()


def test_fixture_is_the_sb3_mlp_shape():
    z = np.load(os.path.join(ROOT, "tests", "golden", "robot_move_policy.npz"))
    assert z["fc0_weight_q"].shape == (64, 6) and z["fc1_weight_q"].shape == (64, 64) and z["fc2_weight_q"].shape == (2, 64)
    assert z["vf2_weight_q"].shape == (1, 64) and z["fc0_weight_q"].dtype == np.int8 and z["fc0_bias_q"].dtype == np.int32
    from balance_robot_mujoco_rl_amd._lib import POLICY_NPARAM
    assert QuantMovePolicy().float_params().shape == (POLICY_NPARAM,)


def test_mujoco_trained_policy_balances_the_kernel_source_and_follows_the_schedule():
    """kernel source on the host in double, 7 s of Env01-v3: nobody falls, the wheel speed sits on every target of the schedule"""
    from hostsim.hostsim import HostSim
    sim = HostSim("Env01-v3", 24, seed=3, auto_reset=False, double=True, threads=8)
    A, E, P = drive(sim.step, sim.reset(), np_policy(QuantMovePolicy(), "mean"), 1400)
    check_tracking(A, E, P, PHASE_ENDS, 0.15, "host double / mean")


def test_the_output_the_reference_reads_also_balances():
    """output[1] of the export carries a frozen exploration-noise sample in its bias (-0.345, +0.250): the robot still
    balances through the first three targets, with the constant wheel-speed offset that bias implies"""
    from hostsim.hostsim import HostSim
    sim = HostSim("Env01-v3", 16, seed=5, auto_reset=False, double=True, threads=8)
    A, E, P = drive(sim.step, sim.reset(), np_policy(QuantMovePolicy(), "actions"), 1100)
    check_tracking(A, E, P, PHASE_ENDS[:3], 0.35, "host double / actions")


def test_mujoco_trained_policy_balances_the_oracle():
    """the fp64 oracle (general body tree, MuJoCo-style solver) under the same policy: 4.5 s, the first two targets"""
    from oracle import oracle as O
    sim = O.Oracle("Env01-v3", 8, seed=3, auto_reset=False, threads=8)
    A, E, P = drive(sim.step, sim.reset(), np_policy(QuantMovePolicy(), "mean"), 900)
    check_tracking(A, E, P, PHASE_ENDS[:2], 0.15, "oracle / mean")
    sim.close()


@pytest.mark.gpu
def test_mujoco_trained_policy_balances_the_hip_path():
    """4,096 Env01-v3 envs on the GPU, 7 s: (a) the int8 evaluator in torch on the device, (b) the same network with
    dequantised weights through the policy KERNEL (brs_policy_act, deterministic) -- rollout without leaving the GPU"""
    from balance_robot_mujoco_rl_amd import BatchedSim
    from balance_robot_mujoco_rl_amd.policy import DevicePolicy
    n = 4096
    qp = QuantMovePolicy(device="cuda")
    for mode in ("int8 evaluator", "policy kernel"):
        sim = BatchedSim("Env01-v3", n, device=0, seed=11, auto_reset=False)
        pol = DevicePolicy(device=0, seed=0)
        pol.set_weights(qp.float_params("mean"))
        obs = sim.reset()
        alive = torch.ones(n, dtype=torch.bool, device="cuda")
        E, P = [], []
        for k in range(1400):
            if mode == "int8 evaluator":
                a = qp.act(obs, "mean").contiguous()
            else:
                a = pol.act(obs, k, deterministic=True)[0]  # the unclipped mean, as RobotMoveBaseEnv applies it
            obs, _, te, _, _ = sim.step(a)
            alive &= ~te.bool()
            if any(e - 20 <= k < e for e in PHASE_ENDS):
                E.append(obs[:, 4].abs().clone()); P.append(obs[:, 0].abs().clone() * 0.25)
        alive = alive.cpu().numpy()
        E, P = torch.stack(E).cpu().numpy(), torch.stack(P).cpu().numpy()
        # (the host build of the kernel source keeps 256 of 256 up with median error 0.03, max 0.11: the bounds leave room
        # for the tails of 4,096 draws of start tilt, target and sensor offset)
        assert alive.mean() > 0.998, f"{mode}: {int((~alive).sum())} of {n} fell"
        for j, e in enumerate(PHASE_ENDS):
            err = E[20 * j:20 * j + 20].mean(axis=0)[alive]
            assert np.median(err) < 0.06 and np.quantile(err, 0.99) < 0.15, f"{mode}: step {e}: median {np.median(err):.3f} p99 {np.quantile(err, 0.99):.3f}"
            assert P[20 * j:20 * j + 20].max(axis=0)[alive].max() < 0.3
        sim.close(); pol.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/balance_robot/envs/RobotMovePolicy.tflite"),
                    reason="the reference checkout is only present in the build container")
def test_fixture_regenerates_from_the_reference_file():
    """tools/gen_policy_fixture.py run again on the reference's .tflite gives the committed arrays (nothing hand-edited)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tflite_reader import read_tflite
    m = read_tflite("/root/reference/src/balance_robot/envs/RobotMovePolicy.tflite")
    z = np.load(os.path.join(ROOT, "tests", "golden", "robot_move_policy.npz"))
    T = m["tensors"]
    ops = {o["outputs"][0]: o for o in m["operators"]}
    # walk output[1] (the actions the reference reads) back to the input
    t, fcs = m["outputs"][1], []
    while t != m["inputs"][0]:
        o = ops[t]
        if o["op"] == "FULLY_CONNECTED":
            fcs.append(o)
        t = o["inputs"][0]
    fcs.reverse()
    assert len(fcs) == 3
    for k, o in enumerate(fcs):
        np.testing.assert_array_equal(z[f"fc{k}_weight_q"], T[o["inputs"][1]]["data"])
        np.testing.assert_array_equal(z[f"fc{k}_bias_q"], T[o["inputs"][2]]["data"])
        np.testing.assert_allclose(z[f"fc{k}_weight_scale"], np.asarray(T[o["inputs"][1]]["scale"], np.float64))
        np.testing.assert_allclose(z[f"fc{k}_out_scale"], np.asarray(T[o["outputs"][0]]["scale"], np.float64))
    np.testing.assert_allclose(z["input_scale"], np.asarray(T[m["inputs"][0]]["scale"], np.float64))
    assert int(z["input_zero_point"][0]) == T[m["inputs"][0]]["zero_point"][0]
