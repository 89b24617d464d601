#!/usr/bin/env python3
"""Parity report of the round (run on the GPU box): BASELINE.json config 2 in full and config 3 at the size the oracle
finishes in minutes, HIP kernel vs oracle/ (fp64), teacher-forced per env step -- tools/parity_locate.py does the work
(per-coordinate-group maxima, every env-step above 1e-4 dumped with its pre-step state and a substep-level replay).

    python tools/parity_report.py [--quick] [--large] [--policy] [--ids] [--only=name,name] [--scratch: write under gpurun_out/parity/ instead of profiles/]
        ->  profiles/r03_parity_config2.json, _config3.json, _config3_noreset.json [, _config3_large.json, _config3_policy.json, _ids_*.json]
Oracle = oracle/brs_oracle.c (own fp64 restatement; MuJoCo is not installable here: physics parity vs MuJoCo UNPINNED)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
quick = "--quick" in sys.argv
outdir = os.path.join(ROOT, "gpurun_out", "parity") if "--scratch" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(outdir, exist_ok=True)
runs = [("config2", ["--env", "Env01-v2", "--envs", "4096", "--steps", "100" if quick else "1000", "--actions", "zero", "--auto-reset", "0"]),
        ("config3", ["--env", "Env03-v2", "--envs", "1024", "--steps", "60" if quick else "250", "--actions", "random", "--auto-reset", "1"]),
        ("config3_noreset", ["--env", "Env03-v2", "--envs", "1024", "--steps", "60" if quick else "250", "--actions", "random", "--auto-reset", "0"])]
if "--large" in sys.argv:  # 8x the sample of config3: rates of a few per million need millions of env-steps (~5 min)
    runs.append(("config3_large", ["--env", "Env03-v2", "--envs", "4096", "--steps", "500", "--actions", "random", "--auto-reset", "1"]))
if "--policy" in sys.argv:  # robots that stay up while blocks keep hitting them: actions from the reference's MuJoCo-trained policy
    runs.append(("config3_policy", ["--env", "Env03-v2", "--envs", "2048", "--steps", "500", "--actions", "policy", "--auto-reset", "1"]))
if "--ids" in sys.argv:     # the other registered ids, ~250 k env-steps each
    for e in ("Env01-v1", "Env01-v3", "Env02-v1", "Env03-v1"):
        runs.append((f"ids_{e}", ["--env", e, "--envs", "1024", "--steps", "250", "--actions", "random", "--auto-reset", "1"]))
only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]  # e.g. --only=config3_large,config3_policy (one GPU call has 20 minutes)
if only:
    runs = [r for r in runs if r[0] in only[0]]
for name, args in runs:
    out = os.path.join(outdir, f"r03_parity_{name}.json")
    print(f"== {name}", flush=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "parity_locate.py"), "--student", "gpu", "--teacher", "oracle",
                           "--out", out] + args)
    if name.startswith("config3") or name.startswith("ids_Env03"):  # second-level replay of the dumped outliers on the HIP path itself, one substep per launch
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "parity_replay_gpu.py"), out])
