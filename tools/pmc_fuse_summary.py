#!/usr/bin/env python3
"""Summarise the PMC passes of tools/gpu_pmc_workloads.sh: per workload, HBM bytes and VALU wave-instructions per warp+fuse
LAUNCH (= one tile/strip kernel dispatch + the margin-kernel dispatches of the frames it fuses), and write
<dir>/summary.txt + <dir>/fuse_traffic.json (the entries bench.py reads from profiles/fuse_traffic.json).

gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; HBM bytes = 2 * FETCH_SIZE * 1024 +
WRITE_SIZE * 1024 (the fetch counter sees half of the read traffic on this part)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUSE_SOURCES = ["multi_frame_super_resolution_amd/csrc/accumulate_fast.hip", "multi_frame_super_resolution_amd/csrc/accumulate_common.hpp"]


def fuse_source_sha16():
    """stamp of the kernel sources the counters describe: bench.py reports traffic / the VALU floor only while it matches"""
    h = hashlib.sha256()
    for f in FUSE_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


# the kernel instance each workload's group launches run (static instruction mix: tools/valu_mix.py)
MIX_KERNEL = {"4k16_rggb_x2": "k_accumulate2xTileILi148ELi4ELi4E", "8k8_rggb_x2": "k_accumulate2xTileILi148ELi4ELi4E",
              "8k64_rggb_x2": "k_accumulate2xTileILi148ELi4ELi4E", "4k16_rggb_x4": "k_accumulate4xTileILi148ELi4E",
              "1080p5_gray_x2": "k_accumulate2xTileILi85ELi2ELi2E"}


def static_mix(names):
    out = {}
    try:
        txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_mix.py"), FUSE_SOURCES[0], "|".join(sorted(set(names)))],
                             capture_output=True, text=True, timeout=900).stdout
        cur = None
        for ln in txt.splitlines():
            if ln.startswith("_Z"):
                cur = ln.strip()
            elif ln.strip().startswith("{") and cur:
                for n in names:
                    if n in cur:
                        out[n] = json.loads(ln.strip())
    except Exception as e:   # the summary must not depend on the compiler being there
        print("valu_mix failed:", e)
    return out

root, wls = sys.argv[1], sys.argv[2:]
res, lines = {}, []
for wl in wls:
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values per dispatch
    for f in glob.glob(f"{root}/{wl}/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    main = [k for k in per if "Margin" not in k]
    margin = [k for k in per if "Margin" in k]
    if not main:
        continue
    k = max(main, key=lambda n: sum(per[n].get("SQ_INSTS_VALU", [0])))
    n_main = sum(len(per[m]["SQ_INSTS_VALU"]) for m in main)   # every tile / strip dispatch (an odd burst ends with a 1-frame one)
    n_margin = sum(len(per[m]["SQ_INSTS_VALU"]) for m in margin)

    def per_launch(counter):
        tot = sum(sum(per[m].get(counter, [])) for m in main) + sum(sum(per[m].get(counter, [])) for m in margin)
        return tot / max(n_main, 1)

    fetch, write, valu = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE"), per_launch("SQ_INSTS_VALU")
    hbm = 2 * fetch * 1024 + write * 1024
    # read requests by size (pass 4): exact bytes at the L2's memory side (Infinity-Cache hits included)
    r32, r64, r128, rall = (per_launch("TCC_EA0_RDREQ_32B_sum"), per_launch("TCC_EA0_RDREQ_64B_sum"), per_launch("TCC_EA0_RDREQ_128B_sum"),
                            per_launch("TCC_EA0_RDREQ_sum"))
    w64, wall = per_launch("TCC_EA0_WRREQ_64B_sum"), per_launch("TCC_EA0_WRREQ_sum")
    read_exact = 32 * r32 + 64 * r64 + 128 * r128 if rall else None
    write_exact = 64 * w64 + 32 * (wall - w64) if wall else None
    # the tile kernel's dispatches in two classes: the first launch of a burst reads no accumulators (fresh), the others do
    first_later = None
    try:
        vals = sorted(per[k].get("TCC_EA0_RDREQ_128B_sum") or per[k].get("FETCH_SIZE") or [])
        if len(vals) >= 4:
            lo = [v for v in vals if v < 0.75 * vals[-1]]
            hi = [v for v in vals if v >= 0.75 * vals[-1]]
            if lo and hi:
                first_later = (sum(lo) / len(lo), len(lo), sum(hi) / len(hi), len(hi))
    except Exception:
        first_later = None
    mk = re.search(r"k_\w+", k)
    kname = mk.group(0) if mk else k
    # frames per launch of the profiled run (bench.py's own line in the pass log): bench.py applies the entry only to runs
    # with the same grouping
    fpl = None
    try:
        for ln in open(f"{root}/{wl}/p1.log"):
            if ln.startswith("{") and '"roofline"' in ln:
                fpl = json.loads(ln)["roofline"].get("frames_per_launch")
    except Exception:
        fpl = None
    res[wl] = {"hbm_bytes_per_launch": int(hbm), "valu_wave_insts_per_launch": int(valu), "frames_per_launch": fpl,
               "kernel_source_sha16": fuse_source_sha16(),
               "source": f"rocprofv3 --pmc passes of tools/gpu_pmc_workloads.sh ({kname}: {n_main} dispatches + "
                         f"{n_margin} margin dispatches per burst)"}
    if read_exact:
        res[wl]["hbm_bytes_per_launch_2xfetch"] = res[wl]["hbm_bytes_per_launch"]
        res[wl]["read_bytes_by_request_size"] = int(read_exact)
        res[wl]["write_bytes_by_request_size"] = int(write_exact) if write_exact else int(write * 1024)
        res[wl]["hbm_bytes_per_launch"] = int(read_exact + (write_exact if write_exact else write * 1024))
        res[wl]["traffic_note"] = ("reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B, writes = 64 x WRREQ_64B + 32 x the rest, at the L2's "
                                   "memory side (Infinity-Cache hits included); hbm_bytes_per_launch_2xfetch = 2 x FETCH_SIZE + WRITE_SIZE")
    lines.append(f"{wl}: kernel {k[:90]}")
    if read_exact:
        lines.append(f"  read requests per launch: 32 B {r32:.4g}, 64 B {r64:.4g}, 128 B {r128:.4g} (all {rall:.4g}) -> {read_exact / 1e9:.3f} GB read; "
                     f"write requests {wall:.4g} ({w64:.4g} of 64 B) -> {(write_exact or 0) / 1e9:.3f} GB written; 2 x FETCH_SIZE would say {2 * fetch * 1024 / 1e9:.3f} GB read")
    if first_later:
        lines.append(f"  tile kernel, 128-B read requests (or FETCH_SIZE KB) per dispatch: first launch of a burst {first_later[0]:.4g} (n={first_later[1]}), "
                     f"later launches {first_later[2]:.4g} (n={first_later[3]})")
    lines.append(f"  dispatches per burst: {n_main} (+{n_margin} margin); per launch: FETCH_SIZE {fetch:.0f} KB, WRITE_SIZE {write:.0f} KB, "
                 f"HBM bytes (2*F+W) {hbm / 1e9:.3f} GB, SQ_INSTS_VALU {valu:.4g}, waves {per_launch('SQ_WAVES'):.0f}")
mix = static_mix([MIX_KERNEL[w] for w in res if w in MIX_KERNEL])
for wl in res:
    m = mix.get(MIX_KERNEL.get(wl, ""))
    if m:
        res[wl]["cycles_per_inst_weighted"] = m["cycles_per_inst_weighted"]
        res[wl]["valu_mix_static"] = m["mix"]
        lines.append(f"{wl}: static VALU mix of {MIX_KERNEL[wl]}: {m['mix']} -> {m['cycles_per_inst_weighted']} cycles per instruction (nominal 4)")
open(f"{root}/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(f"{root}/fuse_traffic.json", "w"), indent=1)
print("\n".join(lines))
