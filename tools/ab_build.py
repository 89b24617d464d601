#!/usr/bin/env python3
"""Build a VARIANT of libbrs_hip.so next to the product library for same-box A/B runs (boxes differ by +-5 % in clock, so only
runs on one box compare): extra -D flags, output under ab/ (git-ignored *.so, travels with gpurun); select it with
BRS_HIP_LIB=ab/libbrs_hip_<name>.so python bench.py ...

    python tools/ab_build.py novel64 -DBRS_VEL64=0
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import _lib  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
out = os.path.join(ROOT, "ab", f"libbrs_hip_{name}.so")
print(_lib.build(out=out, extra_flags=flags, verbose="--verbose" in flags))
