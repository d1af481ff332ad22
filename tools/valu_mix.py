#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, priced with the measured issue costs of tools/ubench/valu_ops.hip
(profiles/r02_ubench_valu_ops.txt: ns per wave-instruction and SIMD with four waves resident).

    python3 tools/valu_mix.py multi_frame_super_resolution_amd/csrc/accumulate_fast.hip 'k_accumulate2xTile.*Li228ELi4ELi4E'

compiles the file for gfx950 to assembly (device side only, the Makefile's flags), takes the kernels whose mangled name
matches the regex, classifies every VALU instruction:

    fast   v_add/sub/mul/fma/fmac/mac _f32, v_and/or/xor/not, v_add_u32/sub_u32, v_bitop3, v_mov           2.8 cycles
    slow   v_cndmask, v_cmp*, v_cvt*, shifts, v_bfe/v_bfi, v_add3/v_lshl_add/v_lshl_or/v_and_or, v_mad_*,
           v_mul_lo/hi, v_min/max/med3, v_floor/trunc/rndne/fract, SDWA / DPP forms, and any fp32 op with an
           SGPR or literal operand                                                                        4.4 cycles
    trans  v_exp/log/rcp/rsq/sqrt/sin/cos                                                                  8.3 cycles
    pk     v_pk_*                                                                                          5.2 cycles

and prints the static mix, the weighted mean cycles per VALU instruction, and JSON for profiles/fuse_traffic.json.  The
static mix weights every instruction once: loops (the per-frame bodies are unrolled in these kernels) and divergent paths
(the straight-arithmetic fallback) make it an approximation of the dynamic mix -- the number says how far the nominal
"4 cycles per instruction" floor is from the issue cost of THIS instruction stream, not more.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

COST = {"fast": 2.8, "slow": 4.4, "trans": 8.3, "pk": 5.2}
FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|fmaak|fmamk)_f32$|^v_(and|or|xor|not|xnor)_b32$|^v_(add|sub|subrev)_u32$|"
                  r"^v_bitop3_b32$|^v_mov_b32$|^v_add_co_u32$|^v_addc_co_u32$|^v_(add|sub)_nc_u32$")
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")


def classify(mn, ops):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)
    if base.startswith("v_pk_"):
        return "pk"
    if TRANS.match(base):
        return "trans"
    if mn.endswith("_sdwa") or mn.endswith("_dpp") or "dpp" in ops or "sdwa" in ops or "row_" in ops or "wave_sh" in ops:
        return "slow"
    if FAST.match(base):
        # an fp32 op with an SGPR / literal source operand issues in the slow class (integer / logic ops do not)
        if base.endswith("_f32"):
            srcs = [o.strip() for o in ops.split(",")[1:]]
            for o in srcs:
                o = o.lstrip("-|").rstrip("|")
                if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|exec|0x[0-9a-f]+|ttmp\d+|m0)", o):
                    return "slow"
                if re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", o) and o not in ("0.5", "1.0", "2.0", "4.0", "-0.5", "-1.0", "-2.0", "-4.0"):
                    return "slow"
        return "fast"
    return "slow"


def main():
    src, rex = sys.argv[1], re.compile(sys.argv[2])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-function",
               "--cuda-device-only", "-S", "-o", out, os.path.join(root, src) if not os.path.isabs(src) else src] + sys.argv[3:]
        subprocess.check_call(cmd)
        text = open(out).read()
    res = {}
    cur, counts, other = None, None, None
    for ln in text.splitlines():
        m = re.match(r"^(\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith("."):
            name = m.group(1)
            if rex.search(name):
                cur, counts, other = name, {k: 0 for k in COST}, {"salu": 0, "vmem": 0, "lds": 0, "other": 0}
                res[cur] = (counts, other)
            else:
                cur = None
            continue
        if cur is None:
            continue
        if ln.strip().startswith(".end_amdhsa_kernel") or re.match(r"^\s*s_endpgm", ln):
            pass
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = parts[1].split(";")[0] if len(parts) > 1 else ""
        if mn.startswith("v_"):
            counts[classify(mn, ops)] += 1
        elif mn.startswith("s_"):
            other["salu"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif mn.startswith("ds_"):
            other["lds"] += 1
        else:
            other["other"] += 1
    for name, (c, o) in res.items():
        n = sum(c.values())
        if n == 0:
            continue
        mean = sum(c[k] * COST[k] for k in c) / n
        print(f"{name[:110]}")
        print("  VALU %d: " % n + ", ".join(f"{k} {c[k]} ({c[k] / n:.1%})" for k in c) + f"; weighted mean {mean:.2f} cycles / instruction "
              f"(nominal 4); other: {o}")
        print("  " + json.dumps({"valu_static": n, "mix": {k: round(c[k] / n, 4) for k in c}, "cycles_per_inst_weighted": round(mean, 3)}))


if __name__ == "__main__":
    main()
