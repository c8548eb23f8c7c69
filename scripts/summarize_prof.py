#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel stats + separate PMC passes) into the small text /
json summaries committed under profiles/.

usage: summarize_prof.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_sq_dir> [key=value ...]
Corrections applied exactly as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE
and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read stream at 1/2, so reads
are doubled (upper estimate for this kernel's narrow reads, which the guide calls uncalibrated);
WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import sys


def counters(d, kernel):
    out = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                 # the most recent run only
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and float(r["Grid_Size"]) > 4096:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # median over the launches: a dispatch now and then reports a doubled counter (seen on SQ_WAVES)
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir, sq_dir = sys.argv[1:6]
    extra = dict(kv.split("=", 1) for kv in sys.argv[6:])
    kernel = extra.get("kernel", "fdrows_lane_kernel")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(root, exist_ok=True)
    lines = []
    stats = sorted(glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for f in stats[-1:]:
        for i, r in enumerate(csv.reader(open(f))):
            if i == 0 or "socp::" in r[0]:
                lines.append(",".join('"%s"' % c if i and j == 0 else c for j, c in enumerate(r)))
    open(os.path.join(root, tag + "_kernel_stats.csv"), "w").write("\n".join(lines) + "\n")
    # per-dispatch durations of the dominant kernel from the kernel trace: the average WITHOUT the first launch (code-object load,
    # cold caches) beside the one rocprofv3's own stats file gives, and the HIP-event times of the un-profiled and of the profiled
    # run of the same command on the same box (bench.py: roofline.kernel_ms) -- what a reader needs to recompute the line's frac
    traces = sorted(glob.glob(os.path.join(stats_dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    timing = {"kernel": kernel}
    for f in traces[-1:]:
        durs = []
        for r in csv.DictReader(open(f)):
            grid = float(r.get("Grid_Size", 0) or 0) or float(r.get("Grid_Size_X", 0) or 0)      # (the trace has one column per dimension)
            if kernel in r.get("Kernel_Name", "") and grid > 4096:
                durs.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
        durs = [d for _, d in sorted(durs)]
        if durs:
            timing.update({"calls": len(durs), "ms_each": durs, "avg_ms_all_calls": sum(durs) / len(durs),
                           "avg_ms_without_first": sum(durs[1:]) / max(1, len(durs) - 1) if len(durs) > 1 else durs[0],
                           "min_ms": min(durs), "max_ms": max(durs)})
    for key in ("unprofiled", "profiled"):
        path = extra.pop(key, None)
        if path and os.path.exists(path):
            try:
                line = [l for l in open(path) if l.startswith("{")][-1]
                d = json.loads(line)
                timing[key + "_hip_event_kernel_ms"] = d["roofline"]["kernel_ms"]
                timing[key + "_ms_per_step"] = d["ms_per_step"]
                timing[key + "_frac"] = d["roofline"]["frac"]
            except Exception as exc:
                timing[key + "_error"] = str(exc)
    json.dump(timing, open(os.path.join(root, tag + "_timing.json"), "w"), indent=1)
    c = {}
    for d in (fetch_dir, write_dir, sq_dir):
        c.update(counters(d, kernel))
    fetch_b = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    write_b = c.get("WRITE_SIZE", 0.0) * 1024.0
    summ = {"tag": tag, "kernel": kernel, "counters_median_per_launch": c,
            "hbm_read_bytes_per_launch(2x FETCH_SIZE KiB)": fetch_b,
            "hbm_write_bytes_per_launch(WRITE_SIZE KiB)": write_b,
            "hbm_bytes_per_launch": fetch_b + write_b}
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        summ["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if "GRBM_GUI_ACTIVE" in c:
        summ["gpu_cycles_per_launch(GRBM_GUI_ACTIVE/8 XCDs)"] = c["GRBM_GUI_ACTIVE"] / 8
        if "SQ_ACTIVE_INST_VALU" in c:
            # SQ_ACTIVE_INST_* count quad-cycles summed over SIMDs (guide: cycle-constants table)
            summ["valu_active_frac_of_1024_simds"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    summ.update(extra)
    json.dump(summ, open(os.path.join(root, tag + "_pmc.json"), "w"), indent=1)
    tj = {"hbm_bytes_per_launch": fetch_b + write_b}
    for k in ("starts", "rk4_steps"):
        if k in extra:
            tj[k] = int(extra[k])
    if "variant" in extra:
        tj["variant"] = extra["variant"]
    if "source" in extra:
        tj["source"] = extra["source"]          # the committed summary the figure comes from (bench.py: roofline.traffic_source)
    tpath = os.path.join(root, "traffic_latest.json")      # one entry per (variant, starts, rk4_steps); bench.py looks its own up
    try:
        entries = json.load(open(tpath))
        entries = entries if isinstance(entries, list) else [entries]
    except Exception:
        entries = []
    key = lambda e: (e.get("variant"), e.get("starts"), e.get("rk4_steps"))
    entries = [e for e in entries if key(e) != key(tj)] + [tj]
    json.dump(entries, open(tpath, "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
