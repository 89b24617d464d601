"""Evaluator of tests/golden/robot_move_policy.npz: the int8 actor path of the policy the reference ships
(ref: envs/RobotMovePolicy.tflite; fixture made by tools/gen_policy_fixture.py).  Test infrastructure.

The integer accumulations are exact (float64 holds them); requantisation divides by the output scale in floating point
where the TFLite kernels use a fixed-point multiplier, and tanh is evaluated in floating point where TFLite uses a table:
either can differ from the TFLite interpreter by one int8 step on a rounding tie.  No TFLite runtime is installed, so
parity with the interpreter itself is UNPINNED; the fixture is used for closed-loop BEHAVIOUR (a policy trained against
MuJoCo has to keep our robot on its wheels and follow the wheel-speed target), not for bit parity.

Input/output handling follows ref: envs/RobotMoveBaseEnv.py:178-203 (quantise the observation with the input scale /
zero point, clip to int8, read output[1], dequantise)."""
import os
import numpy as np
import torch

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robot_move_policy.npz")


class QuantMovePolicy:
    def __init__(self, device="cpu", path=FIXTURE):
        z = np.load(path)  # allow_pickle stays False
        t = lambda k: torch.as_tensor(np.asarray(z[k], np.float64), device=device)
        self.in_s, self.in_z = t("input_scale"), t("input_zero_point")
        self.mean_b, self.mean_os, self.mean_oz = t("fc2_mean_bias_q"), t("fc2_mean_out_scale"), t("fc2_mean_out_zero_point")
        self.layers = []
        for k in range(3):
            self.layers.append(dict(W=t(f"fc{k}_weight_q"), b=t(f"fc{k}_bias_q"), ws=t(f"fc{k}_weight_scale"), bs=t(f"fc{k}_bias_scale"),
                                    os=t(f"fc{k}_out_scale"), oz=t(f"fc{k}_out_zero_point"),
                                    ts=t(f"tanh{k}_out_scale") if k < 2 else None, tz=t(f"tanh{k}_out_zero_point") if k < 2 else None))

        self._z = z

    def float_params(self, which="mean"):
        """the same network with DEQUANTISED weights as one flat float32 vector in include/brs_policy.h order (pi tower, vf
        tower, log_std = 0): what DevicePolicy.set_weights takes.  Activations are then plain floats (no int8 rounding between
        the layers), so outputs differ from act() by the activation quantisation noise"""
        z = self._z
        def lin(pfx, bias=None):
            w = np.asarray(z[pfx + "_weight_q"], np.float64)
            ws = np.asarray(z[pfx + "_weight_scale"], np.float64)
            w = w * (ws[:, None] if ws.size == w.shape[0] else ws)
            bk = bias or pfx + "_bias"
            b = np.asarray(z[bk + "_q"], np.float64) * np.asarray(z[bk + "_scale"], np.float64)
            return [w.ravel(), b.ravel()]
        parts = lin("fc0") + lin("fc1") + lin("fc2", "fc2_mean_bias" if which == "mean" else "fc2_bias") + lin("vf0") + lin("vf1") + lin("vf2") + [np.zeros(2)]
        return np.concatenate(parts).astype(np.float32)

    @staticmethod
    def _q(x, s, z):
        return torch.clamp(torch.round(x / s) + z, -128, 127)

    def act(self, obs, which="actions"):
        """obs [N, 6] float tensor (the env's observation) -> [N, 2] float32.  which = "actions": output[1] of the export, the
        tensor the reference reads (its bias carries one frozen sample of the exploration noise); "mean": the distribution mean"""
        x = obs.to(torch.float64)
        q, s, z = self._q(x, self.in_s, self.in_z), self.in_s, self.in_z
        for k, L in enumerate(self.layers):
            mean = k == 2 and which == "mean"
            acc = (q - z) @ L["W"].T + (self.mean_b if mean else L["b"])   # exact int32 accumulator
            real = acc * L["bs"]                               # bias scale = input scale x per-channel weight scale
            os_, oz_ = (self.mean_os, self.mean_oz) if mean else (L["os"], L["oz"])
            q, s, z = self._q(real, os_, oz_), os_, oz_
            if L["ts"] is not None:
                y = torch.tanh((q - z) * s)
                q, s, z = self._q(y, L["ts"], L["tz"]), L["ts"], L["tz"]
        return ((q - z) * s).to(torch.float32)
