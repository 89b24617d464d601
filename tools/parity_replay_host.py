#!/usr/bin/env python3
"""CPU proxy of tools/parity_replay_gpu.py: replays the outliers a `parity_locate.py --student host --teacher hostdouble` run
dumped -- the kernel source on the host in FLOAT, one substep at a time, against the oracle -- and reports for each the first
substep at which the velocity difference jumps and whether the oracle's contact list (body pairs) changes within two
substeps of it.  No GPU needed: float-vs-double on the host has the same switching mechanism as GPU-vs-oracle,
minus the GPU's own rounding (fast-math reciprocal / rsqrt, FMA contraction).

    python tools/parity_locate.py --student host --teacher hostdouble --env Env01-v2 --envs 2048 --steps 150 --tol 1e-5 --out /tmp/o.json
    python tools/parity_replay_host.py /tmp/o.json
"""
import collections, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim.hostsim import HostSim  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    rep = json.load(open(sys.argv[1]))
    env_id = rep["env"]
    F = HostSim(env_id, 1, noise=False, double=False)
    orc = O.Oracle(env_id, 1, seed=0, auto_reset=False, noise=False)

    def pairs(ctrl):
        fw = orc.forward(env=0, ctrl=(float(ctrl[0]), float(ctrl[1])))
        return sorted((int(c["body1"]), int(c["body2"])) for c in fw["contacts"])

    cls = collections.Counter()
    for o in rep["outliers"]:
        pre = o["pre"]
        qpos, qvel, warm = (np.array(pre[k], dtype=np.float64)[None] for k in ("qpos", "qvel", "warm"))
        tm, ctrl = np.array([pre["time"]]), np.array(pre["ctrl"])
        F.set_state(qpos, qvel, warm, tm); orc.set_state(qpos, qvel, warm, tm)
        first, prev, P = None, 0.0, []
        for k in range(250):
            P.append(pairs(ctrl))
            F.physics(ctrl[None], 1); orc.physics(ctrl[None], 1)
            ev = float(np.abs(F.get_state()[1] - orc.get_state()[1]).max())
            if first is None and ev > 1e-4 and ev > 20 * max(prev, 1e-8):
                first = k
            prev = ev
        P.append(pairs(ctrl))
        chg = None if first is None else any(P[k] != P[k + 1] for k in range(max(0, first - 2), min(250, first + 2)))
        cls["contact list changes within 2 substeps of the jump" if chg else
            ("jump with the contact list unchanged (friction rows)" if first is not None else "no jump one substep at a time")] += 1
        print(o["env"], o["step"], "tilt", round(o["tilt_deg"], 1), {k: float("%.2g" % v) for k, v in o["per_group"].items()},
              "jump at substep", first, "| contact list changes:", chg)
    print(dict(cls))


if __name__ == "__main__":
    main()
