#!/usr/bin/env python3
"""Turns the policy the reference ships (ref: src/balance_robot/envs/RobotMovePolicy.tflite -- an int8-quantised export of an SB3
PPO MlpPolicy, loaded by ref:envs/RobotMoveBaseEnv.py:81-98 and driven at :178-208) into a plain weights fixture:

    python tools/gen_policy_fixture.py  ->  tests/golden/robot_move_policy.npz

The file is DATA of the reference (quantised weights, biases, scales, zero points of the actor path input -> output[1],
the tensor RobotMoveBaseEnv reads), parsed by tools/tflite_reader.py -- nothing from the file is executed, no TFLite runtime
is involved (none is installed).  tests/quant_policy.py evaluates it; tests/test_move_policy_closed_loop.py uses it as the
closed-loop behavioural fixture SURVEY.md 8 f4 names: a policy trained against MuJoCo has to balance OUR robot.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tflite_reader import read_tflite

SRC = "/root/reference/src/balance_robot/envs/RobotMovePolicy.tflite"
m = read_tflite(SRC)
T = m["tensors"]
out_actions = m["outputs"][1]  # "the second output is the one that includes the actions needed" (RobotMoveBaseEnv.py:94-97)
# walk back from the action tensor to the input through FULLY_CONNECTED / TANH
by_out = {o["outputs"][0]: o for o in m["operators"]}
chain, t = [], out_actions
while t != m["inputs"][0]:
    o = by_out[t]
    assert o["op"] in ("FULLY_CONNECTED", "TANH"), o
    chain.append(o); t = o["inputs"][0]
chain.reverse()
assert [o["op"] for o in chain] == ["FULLY_CONNECTED", "TANH", "FULLY_CONNECTED", "TANH", "FULLY_CONNECTED"], chain
out = {}
def q(name, t):
    out[name + "_scale"] = np.asarray(T[t]["scale"], np.float64); out[name + "_zero_point"] = np.asarray(T[t]["zero_point"], np.int64)
q("input", m["inputs"][0])
k = 0
for o in chain:
    if o["op"] == "FULLY_CONNECTED":
        a, w, b = o["inputs"]
        assert o["fused_activation"] is None and T[w]["dtype"] == "int8" and T[b]["dtype"] == "int32" and T[w]["qdim"] == 0
        out[f"fc{k}_weight_q"] = T[w]["data"]            # [out, in] int8, per-output-channel scales, zero point 0
        q(f"fc{k}_weight", w)
        out[f"fc{k}_bias_q"] = T[b]["data"]              # int32, scale = input scale x weight scale
        q(f"fc{k}_bias", b)
        q(f"fc{k}_out", o["outputs"][0])
        k += 1
    else:
        q(f"tanh{k - 1}_out", o["outputs"][0])
# the distribution MEAN (PartitionedCall:... tensor feeding the log-prob branch): same weights, another bias -- the exported
# "actions" output carries one frozen sample of the exploration noise in its bias (b_actions - b_mean = (-0.345, +0.250))
mean_op = [o for o in m["operators"] if o["op"] == "FULLY_CONNECTED" and o["inputs"][:2] == chain[-1]["inputs"][:2] and o is not chain[-1]]
assert len(mean_op) == 1
out["fc2_mean_bias_q"] = T[mean_op[0]["inputs"][2]]["data"]
q("fc2_mean_bias", mean_op[0]["inputs"][2])
q("fc2_mean_out", mean_op[0]["outputs"][0])
# the value tower (output[0]): fills the rest of an SB3 MlpPolicy parameter vector (include/brs_policy.h layout)
tv, vchain = m["outputs"][0], []
while tv != m["inputs"][0]:
    o = by_out[tv]
    assert o["op"] in ("FULLY_CONNECTED", "TANH"), o
    vchain.append(o); tv = o["inputs"][0]
vchain.reverse()
k = 0
for o in vchain:
    if o["op"] == "FULLY_CONNECTED":
        a, w, b = o["inputs"]
        out[f"vf{k}_weight_q"] = T[w]["data"]; q(f"vf{k}_weight", w)
        out[f"vf{k}_bias_q"] = T[b]["data"]; q(f"vf{k}_bias", b)
        q(f"vf{k}_out", o["outputs"][0])
        k += 1
assert k == 3
dst = os.path.join(ROOT, "tests", "golden", "robot_move_policy.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes;", {k: v.shape for k, v in out.items() if k.endswith("_q")})
