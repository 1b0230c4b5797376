"""Condense rocprofv3 CSV output (kernel stats + PMC passes) of tools/profile_bench.sh into a short text summary."""
import csv, glob, sys, os, collections
out = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("stats/**/*kernel_stats.csv"):
    print("  %-90s calls %5s  avg %12.1f ns  total %14s ns  %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"], r["Percentage"]))
for name, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv"), ("SQ", "pmc_sq/**/*counter_collection.csv"), ("F64 mix", "pmc_f64/**/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(pat):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print("== PMC %s (per dispatch mean) ==" % name)
        for k, d in acc.items():
            for c, v in d.items():
                print("  %-70s %-22s mean %16.1f  n %d" % (k, c, sum(v) / len(v), len(v)))
