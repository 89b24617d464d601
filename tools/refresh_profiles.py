#!/usr/bin/env python3
"""Regenerate the per-round evidence files under profiles/ that are plain bench / tool runs, in ONE call on the GPU box and
with ONE build (each file records the build id): all six ids at 65,536 envs, launch sizes up to 524,288 envs, the SB3
VecEnv rate, the soak.  The rocprofv3 summaries come from tools/profile_round.sh, the parity files from tools/parity_report.py.

    python tools/refresh_profiles.py [outdir = gpurun_out/refresh]      (then copy outdir/r03_*.json to profiles/)
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "refresh")
os.makedirs(out, exist_ok=True)
ROUND = "r03"


def bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + list(args), capture_output=True, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    raise RuntimeError(p.stderr[-2000:])


def main():
    from balance_robot_mujoco_rl_amd import _lib
    bid = _lib.build_id()
    ids = {}
    for e in ("Env01-v1", "Env01-v2", "Env01-v3", "Env02-v1", "Env03-v1", "Env03-v2"):
        j = bench("--env", e, "--steps", "200", "--warmup", "10")
        ids[e] = dict(env_steps_per_s=round(j["value"]), ms_per_step=round(j["ms_per_step"], 4), kernel=j["roofline"]["kernel"],
                      kernel_ms_min_median_max=[round(x, 4) for x in j["roofline"]["kernel_ms_min_median_max"]])
        print(e, ids[e]["env_steps_per_s"], flush=True)
    json.dump(dict(note="python bench.py --env ID --steps 200 --warmup 10 --no-cpu-baseline, 65,536 envs, one MI355X, one box, steady state "
                        "(pre-roll >= 300 steps), random policy, auto-reset", build_id=bid, ids=ids), open(os.path.join(out, f"{ROUND}_bench_all_ids.json"), "w"), indent=1)
    sizes = {"Env01_v2_two_waves_per_simd": {}, "Env03_v2": {}}
    for key, e, steps in (("Env01_v2_two_waves_per_simd", "Env01-v2", "100"), ("Env03_v2", "Env03-v2", "60")):
        for n in (65536, 131072, 262144, 524288):
            sizes[key][str(n)] = round(bench("--env", e, "--envs", str(n), "--steps", steps, "--warmup", "10")["value"])
            print(key, n, sizes[key][str(n)], flush=True)
    sizes["Env03_v2_two_sub_batches_on_two_streams_65536"] = round(bench("--env", "Env03-v2", "--streams", "2", "--steps", "100", "--warmup", "10")["value"])
    sizes.update(note="python bench.py --env E --envs N --steps 100 (Env01) / 60 (Env03) --warmup 10 --no-cpu-baseline, one MI355X, one box; env-steps/s", build_id=bid)
    json.dump(sizes, open(os.path.join(out, f"{ROUND}_bench_batch_sizes.json"), "w"), indent=1)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vecenv_rate.py")], capture_output=True, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line); j["build_id"] = bid
    json.dump(j, open(os.path.join(out, f"{ROUND}_vecenv_rate.json"), "w"), indent=1)
    print("vecenv", j.get("arrays_env_steps_per_s"), flush=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--out", os.path.join(out, f"{ROUND}_soak.json")], capture_output=True, text=True)
    print("soak rc", p.returncode, p.stdout[-300:], p.stderr[-300:], flush=True)


if __name__ == "__main__":
    main()
