#!/usr/bin/env python3
"""Turn one tools/profile_r02.sh output directory (gpurun_out/<tag>) into the tracked files under profiles/:
    <prefix>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table (all kernels of the command)
    <prefix>_summary.txt        the condensed text summary
    <prefix>_bench.json         what the command printed when run without the profiler
    <prefix>_pmc.json           per-launch counter means of ONE kernel (+ HBM bytes with the gfx950 FETCH correction, f64 instruction mix)
usage: collect_profile.py gpurun_out/<tag> <kernel substring> profiles/<prefix> "<workload description>" [max]
"max": the command launches the kernel more than once with different batch sizes (a small warm-up launch, then the measured one):
take the LARGEST dispatch (per-counter maximum, longest duration in the kernel trace) instead of the mean over dispatches."""
import csv, glob, json, os, shutil, sys, collections
out, kern, prefix, workload = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
pick_max = len(sys.argv) > 5 and sys.argv[5] == "max"
stats = glob.glob(os.path.join(out, "stats/**/*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], prefix + "_kernel_stats.csv")
shutil.copy(os.path.join(out, "summary.txt"), prefix + "_summary.txt")
if os.path.exists(os.path.join(out, "bench.json")) and os.path.getsize(os.path.join(out, "bench.json")):
    shutil.copy(os.path.join(out, "bench.json"), prefix + "_bench.json")
d = {"round": int(os.environ.get("PK_ROUND", "3")), "kernel_match": kern, "workload": workload}
for r in csv.DictReader(open(stats[0])):
    if kern in r["Name"]:
        d["kernel"] = r["Name"].replace("void ", "").split("(")[0]
        d["kernel_avg_ns_rocprof"] = float(r["AverageNs"]); d["kernel_calls_rocprof"] = int(r["Calls"])
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if pick_max:
    tr = glob.glob(os.path.join(out, "stats/**/*kernel_trace.csv"), recursive=True)
    durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if kern in r["Kernel_Name"]]
    d["kernel_ns_largest_dispatch"] = max(durs); d["dispatch_durations_ns"] = durs
    d["note_pick"] = "largest dispatch of the command (the others are warm-up launches at a smaller batch)"
for k, v in sorted(acc.items()):
    d[k] = max(v) if pick_max else sum(v) / len(v)
    d.setdefault("dispatches_per_counter", {})[k] = len(v)
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    d["note_hbm"] = "FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH doubled per MI355X_MICROARCH.md (gfx950 counts a coalesced 128-B read as 64 B)"
if "SQ_INSTS_VALU_FMA_F64" in d:
    d["f64_flops_executed_per_launch"] = 64.0 * (2.0 * d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_ADD_F64"] + d["SQ_INSTS_VALU_MUL_F64"])
    d["note_f64"] = "wave-instruction counts x 64 lanes, FMA = 2 flops; TRANS (v_rcp_f64 ...) not counted as flops"
json.dump(d, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
