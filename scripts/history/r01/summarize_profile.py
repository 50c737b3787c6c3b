#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small summary + JSON.

usage: summarize_profile.py <dir produced by scripts/history/r01/profile_r.sh> <tag>
FETCH_SIZE is doubled for the 16-B/lane streaming reads of this kernel, as
MI355X_MICROARCH.md (HBM section) prescribes for gfx950; WRITE_SIZE is taken as is.
Both counters are in KiB-units of 1024 B? -> rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KB.
"""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def main():
    out, tag = sys.argv[1], sys.argv[2]
    res = {"tag": tag}
    # ---- kernel stats
    rows = []
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for r in rows:
        name = r.get("Name", "")[:90]
        print(f"{name:90s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
    res["kernel_stats"] = rows
    # per-dispatch durations of the dominant kernel from the trace
    durs = {}
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                durs.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    hot = {k: v for k, v in durs.items() if "dc_kernel" in k or "mfma_" in k} or durs
    dom = max(hot, key=lambda k: sum(hot[k])) if hot else None
    if dom:
        d = durs[dom]
        # skip warm-up dispatches (first 2 of the bench + any earlier)
        tail = d[-10:]
        res["dominant_kernel"] = dom
        res["dominant_avg_ns_last10"] = sum(tail) / len(tail)
        res["dominant_min_ns"] = min(d)
        print(f"== dominant kernel: {dom[:100]}\n   dispatches={len(d)} avg(last10)={sum(tail)/len(tail):.0f} ns min={min(d)} ns")
    # ---- PMC
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        vals = {}
        for f in find(os.path.join(out, sub), "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == name:
                        vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        if dom and dom in vals:
            v = vals[dom][-10:]
            res[name + "_per_launch_raw"] = sum(v) / len(v)
            print(f"== {name} (dominant kernel, avg of last {len(v)} dispatches): {sum(v)/len(v):.1f} (counter units: KB)")
    bj = os.path.join(out, "bench_trace.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                b = json.loads(line)
                res["bench_under_trace"] = {k: b[k] for k in ("value", "ms_per_step", "roofline")}
                alg = b["roofline"]["algorithmic_bytes_per_launch"]
                res["algorithmic_bytes_per_launch"] = alg
                if "FETCH_SIZE_per_launch_raw" in res:
                    fetch_b = res["FETCH_SIZE_per_launch_raw"] * 1024 * 2  # x2: gfx950 correction
                    wr_b = res.get("WRITE_SIZE_per_launch_raw", 0.0) * 1024
                    res["hbm_bytes_per_launch"] = fetch_b + wr_b
                    res["traffic_over_algorithmic"] = (fetch_b + wr_b) / alg
                    print(f"== HBM traffic per launch: read {fetch_b/1e6:.1f} MB (FETCH_SIZE x2) + write {wr_b/1e6:.3f} MB"
                          f" = {(fetch_b+wr_b)/alg:.3f} x algorithmic ({alg/1e6:.1f} MB)")
                if dom:
                    res["achieved_GBps_from_trace"] = alg / res["dominant_avg_ns_last10"]
                    print(f"== achieved (algorithmic bytes / traced avg duration): {alg / res['dominant_avg_ns_last10']:.1f} GB/s")
    with open(os.path.join(out, "summary.json"), "w") as fh:
        json.dump(res, fh, indent=1, default=str)


if __name__ == "__main__":
    main()
