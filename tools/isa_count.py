#!/usr/bin/env python3
"""Static instruction mix of one step-kernel instantiation (works without a GPU: hipcc cross-compiles gfx950 to assembly
with the product's flags).  Used to compare A/B variants of brs_core.hpp before spending GPU time on them.

    python tools/isa_count.py [-DFLAG ...] [--kernel 'brs_step_kernelILb1ELi3E'] [--markers]

--markers: build with -DBRS_MARKERS and print the instruction count between consecutive `; BRS_MARK name` comments (the
phases of one trip in source order; compiler scheduling moves a few instructions across the fences' neighbours).
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balance_robot_mujoco_rl_amd import _lib  # noqa: E402


def main():
    args = sys.argv[1:]
    kern = "brs_step_kernelILb1ELi3E"
    if "--kernel" in args:
        k = args.index("--kernel"); kern = args[k + 1]; del args[k:k + 2]
    markers = "--markers" in args
    if markers:
        args.remove("--markers"); args.append("-DBRS_MARKERS")
    flags = ["-Xarch_device", "-ffast-math", "-Xarch_device", "-fgpu-flush-denormals-to-zero", "-Xarch_device", "-fno-slp-vectorize",
             "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] + args
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([_lib.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out] + flags +
                              [_lib.SRC], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    m = re.search(r"^(_Z\w*" + re.escape(kern) + r"\w*):.*\n", txt, re.M)
    if not m:
        sys.exit(f"no kernel matching {kern}")
    body = txt[m.end():]
    body = body[:body.index(".Lfunc_end")]
    ops, phase, per_phase = collections.Counter(), "(entry)", collections.OrderedDict()
    for ln in body.splitlines():
        t = ln.strip()
        mm = re.match(r";\s*BRS_MARK\s+(\S+)", t)
        if mm:
            phase = mm.group(1)
            continue
        if not ln.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        ops[op] += 1
        per_phase.setdefault(phase, collections.Counter())[op] += 1
    def summary(c):
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        return dict(total=sum(c.values()), valu=valu, packed=sum(v for k, v in c.items() if k.startswith("v_pk_")),
                    f64=sum(v for k, v in c.items() if k.endswith("_f64") or "_f64_" in k), lds=sum(v for k, v in c.items() if k.startswith("ds_")),
                    mov=c.get("v_mov_b32_e32", 0) + sum(v for k, v in c.items() if k.startswith("v_accvgpr")), cndmask=sum(v for k, v in c.items() if k.startswith("v_cndmask")),
                    branch=sum(v for k, v in c.items() if k.startswith("s_cbranch") or k == "s_branch"), nop=c.get("s_nop", 0))
    print(m.group(1)); print(" ", summary(ops))
    if markers:
        for ph, c in per_phase.items():
            s = summary(c)
            print(f"  {ph:28s} total {s['total']:6d} valu {s['valu']:6d} packed {s['packed']:5d} f64 {s['f64']:4d} lds {s['lds']:4d} mov {s['mov']:4d} cndmask {s['cndmask']:4d}")


if __name__ == "__main__":
    main()
