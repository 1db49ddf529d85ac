#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (kernel-trace stats + PMC csv) into one small JSON for profiles/."""
import csv, glob, json, os, sys, statistics

out = sys.argv[1]
res = {"dir": out}

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [
        {k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")}
        for r in rows[:8]]
# per-dispatch durations of the attention kernel (skip warm-up dispatches).  A gated double launch (head size 128)
# runs two attention kernels per call: the one that does the work is reported, the one that leaves at once beside it
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(k in r.get("Kernel_Name", "") for k in ("pa_v1_", "pa_q_", "pa_stage_", "pa_split"))]
    groups = {}
    for r in rows:
        groups.setdefault(r["Kernel_Name"], []).append(r)
    ranked = sorted(groups.values(), key=lambda g: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g))
    for gi, grp in enumerate(ranked[:2]):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp]
        tail = d[10:] if len(d) > 20 else d
        res["pa_v1_dispatches" if gi == 0 else "gated_out_kernel_dispatches"] = {
            "kernel": grp[0]["Kernel_Name"][:80], "n": len(d), "mean_us_all": statistics.mean(d) / 1e3,
            "mean_us_after_warmup": statistics.mean(tail) / 1e3,
            "median_us": statistics.median(tail) / 1e3, "min_us": min(tail) / 1e3,
            "vgpr": grp[0].get("VGPR_Count"), "sgpr": grp[0].get("SGPR_Count"),
            "lds": grp[0].get("LDS_Block_Size"), "wg": grp[0].get("Workgroup_Size"),
            "grid": grp[0].get("Grid_Size")}
MAIN = res.get("pa_v1_dispatches", {}).get("kernel", "pa_")
# PMC
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r.get("Kernel_Name", "").startswith(MAIN[:60]) and r.get("Counter_Name") == name]
        if vals:
            tail = vals[10:] if len(vals) > 20 else vals
            res[name] = {"n": len(vals), "mean_raw_KiB_units": statistics.mean(tail),
                         "mean_bytes_uncorrected": statistics.mean(tail) * 1024}
# SQ / MFMA passes: mean per dispatch of every counter collected for the attention kernel
for sub in ("pmc_sq", "pmc_mfma"):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            if r.get("Kernel_Name", "").startswith(MAIN[:60]):
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            tail = v[10:] if len(v) > 20 else v
            res.setdefault("sq_counters_mean_per_dispatch", {})[k] = statistics.mean(tail)
c = res.get("sq_counters_mean_per_dispatch", {})
if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
    res["sq_fractions_of_wave_cycles"] = {k: c[k] / c["SQ_WAVE_CYCLES"] for k in
                                          ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    f = res["FETCH_SIZE"]["mean_bytes_uncorrected"] * 2      # gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream
    w = res["WRITE_SIZE"]["mean_bytes_uncorrected"]
    res["traffic_bytes_corrected"] = {"read": f, "write": w, "total": f + w}
try:
    b = json.load(open(os.path.join(out, "bench_under_trace.json")))
    res["kernel_variant"] = b["config"]["kernel_variant"]
    res["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_launch"]
    res["bench_event_us_under_trace"] = b["paged_attention_v1_us_per_step"]
    res["library_sha16"] = b.get("library_sha16")     # the binary this pass profiled (bench.py drops a figure taken from another)
    res["profile_tag"] = os.path.basename(out.rstrip("/")).replace("prof_", "")
except Exception as e:  # noqa: BLE001
    res["bench_json_error"] = str(e)
for k in res.get("kernel_stats", []):
    if len(k["Name"]) > 120:
        k["Name"] = k["Name"][:117] + "..."
print(json.dumps(res, indent=1))
