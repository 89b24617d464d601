#!/usr/bin/env python3
"""Diagnostic: per-phase wave cycles of the step kernel (needs a -DBRS_TIMING build; run on the GPU box).
    BRS_EXTRA_HIPCC_FLAGS=-DBRS_TIMING python tools/phase_timing.py [Env03-v2]
    BRS_HIP_LIB=ab/libbrs_hip_timing.so python tools/phase_timing.py [Env03-v2]     (prebuilt: tools/ab_build.py timing -DBRS_TIMING)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import _lib
if not os.environ.get("BRS_HIP_LIB"):   # (or point BRS_HIP_LIB at a prebuilt -DBRS_TIMING variant: tools/ab_build.py timing -DBRS_TIMING)
    os.environ.setdefault("BRS_EXTRA_HIPCC_FLAGS", "-DBRS_TIMING")
    _lib.build(force=True)
from balance_robot_mujoco_rl_amd import BatchedSim
env = sys.argv[1] if len(sys.argv) > 1 else "Env03-v2"
n = 65536
sim = BatchedSim(env, n, seed=0)
sim.reset()
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
g = torch.Generator(device="cuda"); g.manual_seed(1234)
acts = [torch.rand((n, 2), generator=g, device="cuda") * 2 - 1 for _ in range(16)]
for k in range(60): sim.step(acts[k % 16])
torch.cuda.synchronize(); L.brs_debug_counters(buf)
steps = 60
for k in range(steps): sim.step(acts[k % 16])
torch.cuda.synchronize(); L.brs_debug_counters(buf)
names = ["kin+smooth", "collide robot-floor", "collide block-floor", "collide coupled", "assemble", "cholesky", "passA/verify", "integrate", "whole trip", "trips", "  coupled: torso patch", "  coupled: wheels"]
waves = n // 64
tot = buf[8]
print(f"{env}: trips per wave-step {buf[9] / waves / steps:.1f}")
for i, nm in [(j, names[j]) for j in (0, 1, 2, 3, 10, 11, 4, 5, 6, 7, 8)]:
    print(f"  {nm:22s} {buf[i] / waves / steps:12.0f} cycles/wave-step  {100.0 * buf[i] / tot:5.1f}% of trip time   {buf[i] / max(1, buf[9]):8.0f} cycles/trip")

# load balance: with one wave per SIMD the launch lasts as long as its slowest wave
ratios = []
for k in range(20):
    sim.step(acts[k % 16]); torch.cuda.synchronize(); L.brs_debug_counters(buf)
    mean_c, max_c, mean_t, max_t = buf[8] / waves, buf[12], buf[9] / waves, buf[13]
    ratios.append((max_c / mean_c, max_t / mean_t, mean_t, max_t))
import statistics as st_
print(f"  slowest wave / mean wave: cycles x{st_.mean(r[0] for r in ratios):.3f}, trips x{st_.mean(r[1] for r in ratios):.3f} "
      f"(mean trips {st_.mean(r[2] for r in ratios):.1f}, max trips {st_.mean(r[3] for r in ratios):.1f})")

# where did the slow waves of the last launch run?
wb = (C.c_ulonglong * 4096)()
L.brs_debug_waves(wb)
import numpy as np
w = np.array(list(wb), dtype=np.uint64).reshape(1024, 4)
cyc, trips, hw, xcc = w[:, 0].astype(float), w[:, 1].astype(float), w[:, 2], w[:, 3] & 0xF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
print(f"  last launch: cycles/trip mean {np.mean(cyc / trips):.0f} min {np.min(cyc / trips):.0f} max {np.max(cyc / trips):.0f}; corr(cycles, trips) {np.corrcoef(cyc, trips)[0, 1]:.2f}")
for x in range(8):
    m = xcc == x
    if m.any(): print(f"    XCC {x}: waves {int(m.sum())} mean cycles {cyc[m].mean():.0f} max {cyc[m].max():.0f} cycles/trip {np.mean(cyc[m] / trips[m]):.0f}")
# lanes are grouped by collision cost class in slot order (brs_group_kernel): wave index ~ bucket
print("  by wave index (32 groups of 32 waves): mean cycles / max cycles / mean trips / cycles per trip")
for gi in range(32):
    sl = slice(32 * gi, 32 * gi + 32)
    print(f"    waves {32 * gi:4d}-{32 * gi + 31:4d}: {cyc[sl].mean():10.0f} {cyc[sl].max():10.0f} {trips[sl].mean():7.1f} {np.mean(cyc[sl] / trips[sl]):7.0f}")
order = np.argsort(-cyc)[:12]
print("  slowest waves: " + ", ".join(f"#{int(i)}: {cyc[i]:.0f} cyc {trips[i]:.0f} trips" for i in order))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "wave_records.npy"), w)
