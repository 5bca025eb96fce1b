#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/prof.sh to what profiles/rNN keeps: the kernel-stats CSV,
per-launch rows of the dominant kernel for every counter, and summary.json (per-launch averages over
the profiled launches of the dominant solve kernel, HBM traffic corrected as the MI355X guide
prescribes: FETCH_SIZE x2 on gfx950 for these 8-byte-per-lane loads, KiB units)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
DOM = ("k_wave2", "k_wave", "k_phase")


def find(pat):
    r = glob.glob(os.path.join(out, pat), recursive=True)
    return r[0] if r else None


summary = {}
st = find("trace/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration", 0)) or 0))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[:12])
    top = rows[0]
    summary["kernel_stats_top"] = {k: top[k] for k in top}
tr = find("trace/**/*kernel_trace.csv")
dom_name = None
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if any(d in r["Kernel_Name"] for d in DOM)]
    if rows:
        dom_name = max(set(r["Kernel_Name"] for r in rows), key=lambda n: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == n))
        rows = [r for r in rows if r["Kernel_Name"] == dom_name]
        with open(os.path.join(out, "kernel_trace_dominant.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
        summary["dominant_kernel"] = dom_name
        summary["launch_ms"] = d
        summary["launch_ms_timed_mean"] = sum(d[2:]) / max(1, len(d[2:]))   # bench: 2 warm-up + 3 timed launches
        for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Grid_Size", "Workgroup_Size"):
            if k in rows[0]:
                summary[k] = rows[0][k]
counters = {}
for name in ("sq", "sqw", "fetch", "write"):
    f = find(f"{name}/**/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if dom_name and r["Kernel_Name"] == dom_name]
    if not rows:
        continue
    with open(os.path.join(out, f"pmc_{name}.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    per = {}
    for r in rows:
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for cn, dv in per.items():
        vals = [dv[k] for k in sorted(dv, key=int)]
        counters[cn] = sum(vals[2:]) / max(1, len(vals[2:])) if len(vals) > 2 else sum(vals) / len(vals)
summary["counters_per_launch"] = counters
c = counters
if "SQ_INSTS_VALU_FMA_F64" in c:
    summary["executed_fp64_flop_per_launch"] = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    summary["hbm_traffic_bytes_per_launch"] = 1024.0 * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"])
    summary["hbm_traffic_note"] = "FETCH_SIZE (KiB) x2 gfx950 correction + WRITE_SIZE (KiB), MI355X_MICROARCH.md HBM section"
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    summary["shares_of_wave_cycles"] = {k: c[k] / wc for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS") if k in c}
# what bench.py needs to tie the counters to a build (it recomputes the hash at run time)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    summary["kernel_source_sha"] = bench.kernel_source_sha()
except Exception as e:  # noqa: BLE001
    summary["kernel_source_sha"] = None
hc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".head_commit")
summary["commit"] = open(hc).read().strip() if os.path.exists(hc) else None
summary["batch"] = 65536
summary["horizon"] = 50
summary["command"] = "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-e2e --no-stress (one rocprofv3 run per counter group, tools/prof.sh)"
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernel_stats_top"}, indent=1))
