#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the summary of the round's scripts/r0N_profile_configs.sh (lines `<tag> fetch <kernel> {...}` /
`<tag> write <kernel> {...}` written by scripts/r0N_pmc.sh): HBM bytes per launch = 2 * FETCH_SIZE KB (gfx950 correction of
MI355X_MICROARCH.md for 16-B/lane streaming reads) + WRITE_SIZE KB, against bench.py's algorithmic bytes of the same
workload.  The kernel label is the full instance name rocprofv3 reports (all template arguments).
usage: scripts/make_pmc_traffic.py profiles/r03/r03p_summary.txt"""
import ast
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpuacceleratedtracking_amd.benchmarks as gb  # algorithmic_bytes (host arithmetic only)

WORK = {  # tag -> (workload key of bench.py's stored_traffic, bytes per complex sample)
    "c2": (["GPSL1", 20000, 4, 3, 1, 4096, "planar"], 8),
    "c3": (["GPSL5", 50000, 4, 5, 12, 1024, "planar"], 8),
    "c4": (["GPSL1", 50000, 16, 3, 4, 512, "planar"], 8),
    "c5": (["GPSL1", 2000000, 64, 3, 64, 1, "planar"], 8),
    "c5_i16": (["GPSL1", 2000000, 64, 3, 64, 1, "i16"], 4),  # configs[4] from int16 pairs: the two-term split
    "c1shape": (["GPSL1", 4000, 1, 3, 1, 16384, "planar"], 8),
    "c2_i16": (["GPSL1", 20000, 4, 3, 1, 4096, "i16"], 4),
    "c2_i8": (["GPSL1", 20000, 4, 3, 1, 4096, "i8"], 2),
    "c4x32": (["GPSL1", 50000, 16, 3, 32, 512, "planar"], 8),  # configs[3] as a whole on one GPU (bench.py constellation_config3 at N = 1)
}


def main(path):
    vals = {}
    for line in open(path):
        m = re.match(r"^(\w+) (fetch|write|sq1) (.*?) (\{.*\})\s*$", line)
        if m:
            vals.setdefault(m.group(1), {})[m.group(2)] = (m.group(3), ast.literal_eval(m.group(4)))
    entries = []
    for tag, (key, sample_bytes) in WORK.items():
        if tag not in vals or "fetch" not in vals[tag] or "write" not in vals[tag]:
            continue
        kernel, fd = vals[tag]["fetch"]
        f = fd["FETCH_SIZE"]
        w = vals[tag]["write"][1]["WRITE_SIZE"]
        _, N, M, L, K, B, _ = key
        alg = gb.algorithmic_bytes(B, N, M, L, K, sample_bytes)
        hbm = int(round((2 * f + w) * 1024))
        e = {"workload_key": key, "kernel": kernel.strip(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
             "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 4)}
        if "sq1" in vals[tag] and "SQ_INSTS_VALU" in vals[tag]["sq1"][1]:  # vector wave-instructions per launch (bench.py: issue-rate term)
            e["valu_insts_per_launch"] = int(vals[tag]["sq1"][1]["SQ_INSTS_VALU"])
        entries.append(e)
    out = {"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 6 --warmup 40 "
                  "--settle 64 [shape flags]` (scripts/r0N_pmc.sh, driven by scripts/r0N_profile_configs.sh of the round -> "
                  + os.path.basename(path) + "); per-dispatch counter values of the correlator kernel named in `kernel`, mean of the "
                  "last four dispatches; FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md prescribes for gfx950 16-B/lane streaming "
                  "reads, WRITE_SIZE (KB) as reported; valu_insts_per_launch = SQ_INSTS_VALU of the sq1 pass (scripts/make_pmc_traffic.py)",
           "source": os.path.relpath(os.path.abspath(path), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")),
           "entries": entries}
    json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json"), "w"), indent=1)
    for e in entries:
        print(e["workload_key"], e["kernel"], e["ratio"])


if __name__ == "__main__":
    main(sys.argv[1])
