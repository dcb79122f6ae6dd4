"""Sum rocprofv3 counter_collection.csv per kernel and counter: python tools/pmc_sum.py <dir>"""
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:64]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for (k, c), v in sorted(acc.items()):
        if any(n in k for n in ("k_trace", "k_shade", "k_gen", "k_fold", "k_descend")):
            print(f"{k:66s} {c:36s} {v:16.0f}  per-dispatch {v/len(n[k]):14.0f}  ({len(n[k])} dispatches)")
