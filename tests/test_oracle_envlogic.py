"""Pins oracle/ (env-logic half) against tests/golden/envlogic.json, which tools/gen_golden.py
produced from the reference's own env classes (physics stubbed).  CPU only.

Covers SURVEY.md §8 rows a1, a4-a10, a12-a14."""
import math

import numpy as np
import pytest

from oracle import oracle as O

CLS2VARIANT = {"Env01": "Env01-v1", "Env01_v2": "Env01-v2", "Env03": "Env03-v1", "Env03_v2": "Env03-v2"}


def test_registry_table(golden):
    reg = golden["registry"]
    assert reg["Env01-v2"]["max_episode_steps"] == 6000
    assert reg["Env03-v2"]["max_episode_steps"] == 1200
    assert reg["Env01-v1"]["max_episode_steps"] == 6000 and reg["Env03-v1"]["max_episode_steps"] == 6000
    c = golden["constants"]
    assert (c["PITCH_MAX"], c["PITCH_DOT_MAX"], c["WHEEL_SPEED_MAX"], c["WHEEL_SPEED_DELTA_MAX"], c["YAW_MAX"]) == \
        (0.25, 1, 170.0, 4.0, 45.0)


def test_pitch_yaw(golden):
    for c in golden["pitch_yaw"]:
        p, y = O.pitch_yaw(c["xquat"])
        assert abs(p - c["pitch"]) < 1e-12, c
        assert abs(y - c["yaw"]) < 1e-12, c


def _angle_diff(a, b):
    return abs((a - b + np.pi) % (2 * np.pi) - np.pi)


def test_pitch_yaw_at_gimbal_lock(golden):
    """scipy's as_euler('xyz') within 1e-7 rad of gimbal lock (wheel axis vertical) zeroes the yaw and folds the rotation about the
    vertical into the pitch; the vectors come from the reference's get_pitch / get_yaw with the real scipy (tools/gen_golden.py)"""
    locked = 0
    for c in golden["pitch_yaw_gimbal"]:
        p, y = O.pitch_yaw(c["xquat"])
        locked += c["yaw"] == 0.0 and abs(c["pitch"]) > 1e-3
        # (angles are compared on the circle: at +-pi the wrap is decided by the last bit)
        assert _angle_diff(p, c["pitch"]) < 1e-9 and _angle_diff(y, c["yaw"]) < 1e-9, (c, p, y)
    assert locked >= 12  # the fixture does contain locked poses


def test_reward(golden):
    o = O.Oracle("Env01-v1", 1)
    for c in golden["reward"]:
        o.set_state(qpos=np.array([[0, 0, 0, 1, 0, 0, 0, 0, 0.0]]), qvel=np.array([c["qvel"]]))
        o.set_xpose(xquat=np.array([c["xquat"]]), xpos=np.zeros((1, 3)))
        nq, nv = o.nq, o.nv
        o.stub_physics(0, np.zeros(nq), np.zeros(nv), [1, 0, 0, 0], [0, 0, 0])
        _, rew, _, _, _ = o.step(np.zeros((1, 2)))
        assert abs(rew[0] - c["reward"]) < 1e-6 * max(1, abs(c["reward"])), c


def _reset_script(rec_reset):
    gym_u = (np.array(rec_reset["gym_uniform"]) + 0.01) / 0.02
    return np.concatenate([gym_u, np.array(rec_reset["uniforms"])])


@pytest.mark.parametrize("idx", range(8))
def test_sequences(golden, idx):
    rec = golden["sequences"][idx]
    variant = CLS2VARIANT[rec["cls"]]
    o = O.Oracle(variant, 1, seed=5)
    assert (o.nq, o.nv) == (rec["nq"], rec["nv"])
    if "attack_side_front" in rec:
        aux = o.get_aux()
        aux[0, 4] = 1.0 if rec["attack_side_front"] else 0.0
        o.set_aux(aux)
        assert rec["block_delay"] == (0.5 if variant == "Env03-v2" else 0.0)
    # ---- reset
    r = rec["reset"]
    o.script_uniforms(0, _reset_script(r))
    obs = o.reset()
    assert o.script_remaining(0) == 0, "reset consumed a different number of uniforms than the reference"
    qpos, qvel, _, tm = o.get_state()
    np.testing.assert_allclose(qpos[0], r["qpos"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(qvel[0], r["qvel"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs[0], np.array(r["obs"], np.float32), rtol=1e-6, atol=1e-7)
    xq, xp = o.get_xpose()
    np.testing.assert_allclose(xq[0], r["xquat"], atol=1e-12)
    assert tm[0] == 0.0
    # ---- steps (physics replaced by the scripted post-step state, exactly as the generator did)
    for k, st in enumerate(rec["steps"]):
        qpos, qvel, _, tm = o.get_state()
        np.testing.assert_allclose(qvel[0], st["pre"]["qvel"], atol=1e-12)
        assert abs(tm[0] - st["pre"]["time"]) < 1e-15
        o.stub_physics(0, st["post"]["qpos"], st["post"]["qvel"], st["post"]["xquat"], st["post"]["xpos"])
        o.script_uniforms(0, st["uniforms"])
        obs, rew, term, trunc, tob = o.step(np.array([st["action"]], np.float32))
        assert o.script_remaining(0) == 0, (k, "uniform count mismatch")
        aux = o.get_aux()
        np.testing.assert_allclose(aux[0, 8:10], st["ctrl"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(obs[0], np.array(st["obs"], np.float32), rtol=2e-6, atol=1e-6, err_msg=f"step {k}")
        assert abs(rew[0] - st["reward"]) < 1e-6 * max(1.0, abs(st["reward"])), k
        assert bool(term[0]) == st["terminated"], k
        assert not trunc[0] and st["truncated"] is False
        qpos, qvel, _, tm = o.get_state()
        np.testing.assert_allclose(qpos[0], st["qpos_after"], atol=1e-12, err_msg=f"step {k}")
        np.testing.assert_allclose(qvel[0], st["qvel_after"], atol=1e-12, err_msg=f"step {k}")
        assert abs(tm[0] - st["time"]) < 1e-15
        if "block_timer" in st and rec["nq"] == 16:
            if st["block_timer"] is None:
                assert math.isnan(aux[0, 1]), k
            else:
                assert aux[0, 1] == st["block_timer"], k
        assert st["calls"] == [["mj_step", 250], ["mj_rnePostConstraint"]]
    # ---- a second reset mid-run: last_pitch survives, first pitch_dot is 0
    r2 = rec["reset2"]
    o.script_uniforms(0, _reset_script(r2))
    obs = o.reset()
    assert o.script_remaining(0) == 0
    np.testing.assert_allclose(obs[0], np.array(r2["obs"], np.float32), rtol=1e-6, atol=1e-7)
    assert obs[0, 1] == 0.0
    qpos, _, _, _ = o.get_state()
    np.testing.assert_allclose(qpos[0], r2["qpos"], atol=1e-12)


def test_block_timer_timeline(golden):
    """Env03-v2: remove when slow, respawn when time - t_removed > 0.5 s; the boundary is decided by the
    fp64 accumulation of 250 x 2e-5 per step, which the oracle reproduces (env03_v1.py:39-49)."""
    tl = golden["block_timer_timeline"]
    o = O.Oracle("Env03-v2", 1, seed=1)
    o.reset()
    aux = o.get_aux()
    aux[0, 4] = 1.0 if tl["attack_side_front"] else 0.0
    o.set_aux(aux)
    nthrows = 0
    for k, row in enumerate(tl["rows"]):
        qpos, qvel, _, _ = o.get_state()
        qvel[0, 8:11] = 0.0
        o.stub_physics(0, qpos[0], qvel[0], row["pre"]["xquat"], row["pre"]["xpos"])
        o.script_uniforms(0, row["uniforms"])
        o.step(np.zeros((1, 2), np.float32))
        assert o.script_remaining(0) == 0, k
        aux = o.get_aux()
        _, _, _, tm = o.get_state()
        assert tm[0] == row["time"], k
        if row["timer"] is None:
            assert math.isnan(aux[0, 1]), k
        else:
            assert aux[0, 1] == row["timer"], k
        qpos, qvel, _, _ = o.get_state()
        np.testing.assert_allclose(qpos[0, 9:12], row["block_qpos"][:3], atol=1e-12, err_msg=f"row {k}")
        np.testing.assert_allclose(qvel[0, 8:14], row["block_qvel"], atol=1e-12, err_msg=f"row {k}")
        # block orientation: scipy quaternion in MuJoCo's slot (sign-insensitive)
        qa, qb = qpos[0, 12:16], np.array(row["block_qpos"][3:7])
        if row["uniforms"]:  # only a throw rewrites the block orientation
            assert min(np.abs(qa - qb).max(), np.abs(qa + qb).max()) < 1e-12, k
            nthrows += 1
    assert nthrows == 2


def test_time_limit_and_autoreset():
    o = O.Oracle("Env03-v2", 2, seed=3, auto_reset=True, max_episode_steps=3)
    o.reset()
    for k in range(3):
        obs, rew, term, trunc, tob = o.step(np.zeros((2, 2), np.float32))
        assert trunc.tolist() == [k == 2] * 2
    aux = o.get_aux()
    assert aux[:, 2].tolist() == [0, 0]  # elapsed reset
    _, _, _, tm = o.get_state()
    assert tm.tolist() == [0.0, 0.0]
    assert (obs[:, 1] == 0).all() and not np.array_equal(obs, tob)


def test_philox_kat():
    # Random123 known-answer vectors for philox4x32-10
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


CLS2VARIANT_F3 = {"Env02": "Env02-v1", "Env01_v3": "Env01-v3"}


def _reset_script_f3(rec, key_r, key_s):
    r = rec[key_r]
    pre = []
    if rec["cls"] == "Env01_v3":  # two scalar draws of the seeded generator come first (env01_v3.py:44-52)
        a, b = rec[key_s]
        pre = [(a + 10.0) / 20.0, (b + 0.0349066) / (2 * 0.0349066)]
    return np.concatenate([pre, (np.array(r["gym_uniform"]) + 0.01) / 0.02, np.array(r["uniforms"])])


@pytest.mark.parametrize("idx", range(7))
def test_sequences_f3(golden, idx):
    """Env02-v1 (per-episode friction) and Env01-v3 (target-speed schedule, pitch offset, shaped reward): SURVEY §8 f3"""
    rec = golden["sequences_f3"][idx]
    o = O.Oracle(CLS2VARIANT_F3[rec["cls"]], 1, seed=5)
    r = rec["reset"]
    o.script_uniforms(0, _reset_script_f3(rec, "reset", "reset_gym_scalars"))
    obs = o.reset()
    assert o.script_remaining(0) == 0
    qpos, qvel, _, _ = o.get_state()
    np.testing.assert_allclose(qpos[0], r["qpos"], atol=1e-12)
    np.testing.assert_allclose(obs[0], np.array(r["obs"], np.float32), rtol=1e-6, atol=1e-7)
    aux = o.get_aux()
    ex = r["extras"]
    if rec["cls"] == "Env02":
        assert abs(aux[0, 10] - ex["friction"]) < 1e-12 and ex["friction"] == ex["floor_friction"]
    else:
        np.testing.assert_allclose(aux[0, 11:14], [ex["delay_target_speed"], ex["pitch_offset"], ex["target_wheel_speed"]], atol=1e-9)
    o.set_state(time=np.array([rec["time0"]]))
    for k, st in enumerate(rec["steps"]):
        o.stub_physics(0, st["post"]["qpos"], st["post"]["qvel"], st["post"]["xquat"], st["post"]["xpos"])
        o.script_uniforms(0, st["uniforms"])
        obs, rew, term, trunc, _ = o.step(np.array([st["action"]], np.float32))
        assert o.script_remaining(0) == 0
        ref = np.array(st["obs"], np.float32)
        cols = [0, 2, 3, 4, 5] if (k == 0 and rec["time0"] > 0) else [0, 1, 2, 3, 4, 5]  # (time0 hack: dt of the first finite difference)
        np.testing.assert_allclose(obs[0, cols], ref[cols], rtol=2e-6, atol=1e-6, err_msg=f"step {k}")
        assert abs(rew[0] - st["reward"]) < 1e-6 * max(1.0, abs(st["reward"])), (k, rew[0], st["reward"])
        assert bool(term[0]) == st["terminated"]
        aux = o.get_aux()
        np.testing.assert_allclose(aux[0, 8:10], st["ctrl"], rtol=1e-12, atol=1e-12)
        assert abs(aux[0, 13] - st["extras"]["target_wheel_speed"]) < 1e-9, k
    o.script_uniforms(0, _reset_script_f3(rec, "reset2", "reset2_gym_scalars"))
    obs = o.reset()
    np.testing.assert_allclose(obs[0], np.array(rec["reset2"]["obs"], np.float32), rtol=1e-6, atol=1e-7)
    aux = o.get_aux()
    ex = rec["reset2"]["extras"]
    if rec["cls"] == "Env02":
        assert abs(aux[0, 10] - ex["friction"]) < 1e-12
    else:
        np.testing.assert_allclose(aux[0, 11:14], [ex["delay_target_speed"], ex["pitch_offset"], 0.0], atol=1e-9)


def test_env02_friction_enters_the_contact_model():
    """lower friction => smaller tangential force capacity: the wheel pyramid rows use the episode's mu"""
    o = O.Oracle("Env02-v1", 1, seed=1)
    o.reset()
    aux = o.get_aux(); aux[0, 10] = 0.5; o.set_aux(aux)
    q = np.array([[0, 0, -0.0203, 1, 0, 0, 0, 0, 0.0]]); v = np.zeros((1, 8)); v[0, 1] = 0.3
    o.set_state(q, v)
    f = o.forward()
    assert f["ncon"] == 4 and all(abs(c["mu"] - 0.5) < 1e-12 for c in f["contacts"])
