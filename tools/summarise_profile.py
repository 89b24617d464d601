#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory (rocprofv3's rocpd SQLite files) into one JSON -- what is committed
under profiles/.  PMC values are reported per shader-engine instance by rocprofv3: a dispatch's figure is the SUM over
its instances."""
import glob, json, os, sqlite3, sys
import numpy as np

out, env = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "Env03-v2")
kname = "brs_step_kernel<true, *>" if env.startswith("Env03") else "brs_step_kernel_occ2<*>"
res = {"command": f"python3 bench.py --env {env} --steps 300 --warmup 20 (rocprofv3 --kernel-trace --stats); PMC: the same command with --steps 40 "
                  "--warmup 10, one rocprofv3 --pmc run per counter set, mean over the LAST 30 dispatches of the step kernel "
                  "(bench.py pre-rolls >= 300 env steps, so all of them are steady state)",
       "kernel": kname}
try:
    res["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    # the build these counters belong to (include/brs.h: brs_build_id): bench.py uses them only for a library with this id
    res["build_id"] = res["bench"].get("build_id")
    res["bench"]["roofline"].pop("valu", None); res["bench"]["roofline"].pop("traffic", None)   # (derived from an OLDER summary, if at all)
except Exception as e:
    res["bench_error"] = str(e)
tr = os.path.join(out, "trace", "trace_results.db")
if os.path.exists(tr):
    con = sqlite3.connect(tr)
    top = [dict(name=r[0][:60], calls=r[1], total_ns=r[2], average_ns=r[3], percentage=r[4]) for r in con.execute("select * from top_kernels")][:4]
    res["kernel_trace_stats_top"] = top
    d = [r[0] / 1e6 for r in con.execute("select duration from kernels where name like '%brs_step_kernel%' order by start")]
    if d:
        tail = np.array(d[-300:])
        res["kernel_trace_ms"] = {"dispatches": len(d), "timed_300_mean": float(tail.mean()), "timed_300_median": float(np.median(tail)),
                                  "timed_300_min": float(tail.min()), "timed_300_max": float(tail.max()),
                                  "first_20_after_reset_mean": float(np.mean(d[:20])), "all_mean": float(np.mean(d))}
    ran = {r[0] for r in con.execute("select distinct name from kernels where name like '%brs_step_kernel%'")}
    vg = [r for r in con.execute("select kernel_name, sgpr_count, arch_vgpr_count, private_segment_size, group_segment_size, display_name from kernel_symbols where kernel_name like '%brs_step_kernel%'") if r[5] in ran or r[0] in ran]
    if vg:
        res["kernel_symbol"] = dict(name=vg[0][0][:80], sgpr=vg[0][1], arch_vgpr=vg[0][2], scratch_bytes_per_lane=vg[0][3], static_lds=vg[0][4])
pmc = {}
for f in glob.glob(os.path.join(out, "pmc_*", "pmc_results.db")):
    con = sqlite3.connect(f)
    per = {}
    for name, val, disp in con.execute("select counter_name, counter_value, dispatch_id from pmc_events where name like '%brs_step_kernel%'"):
        per.setdefault(name, {}).setdefault(disp, 0.0)
        per[name][disp] += val
    for name, dd in per.items():
        vals = [dd[k] for k in sorted(dd)]
        pmc[name] = float(np.mean(vals[-30:]))
res["pmc_per_dispatch"] = pmc
if "SQ_WAVE_CYCLES" in pmc:
    waves = pmc.get("SQ_WAVES", 1024.0)
    wc = pmc["SQ_WAVE_CYCLES"]
    res["per_wave_step"] = {k: v / waves for k, v in pmc.items() if k.startswith("SQ_")}
    res["valu"] = {"busy_frac": pmc.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_frac": pmc.get("SQ_WAIT_ANY", 0) / wc,
                   "wait_inst_frac": pmc.get("SQ_WAIT_INST_ANY", 0) / wc,
                   "valu_insts_per_wave_per_step": pmc.get("SQ_INSTS_VALU", 0) / waves,
                   "note": "SQ_WAVE_CYCLES counts in units of 4 clocks (mean wave lifetime = value x 4 / waves)"}
if "SQC_ICACHE_REQ" in pmc:
    res["icache"] = {"req": pmc["SQC_ICACHE_REQ"], "hits": pmc.get("SQC_ICACHE_HITS"), "misses": pmc.get("SQC_ICACHE_MISSES"),
                     "miss_rate": pmc.get("SQC_ICACHE_MISSES", 0) / max(1.0, pmc["SQC_ICACHE_REQ"])}
if "FETCH_SIZE" in pmc or "WRITE_SIZE" in pmc:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies the 128-B
    # requests of wide coalesced reads at 64 B -> x2.  This kernel reads 4- and 8-byte-per-lane SoA columns, a width the
    # guide calls uncalibrated: raw and corrected are both given; `bytes_per_launch` uses the corrected read side.
    f, w = pmc.get("FETCH_SIZE", 0.0) * 1024.0, pmc.get("WRITE_SIZE", 0.0) * 1024.0
    res["hbm_traffic"] = {"fetch_bytes_raw": f, "fetch_bytes_x2_corrected": 2 * f, "write_bytes": w, "bytes_per_launch": 2 * f + w,
                          "bytes_per_launch_uncorrected": f + w}
print(json.dumps(res, indent=1))
