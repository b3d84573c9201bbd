"""Prints the per-kernel mean of every counter in a rocprofv3 --pmc output directory (counter_collection.csv)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "solve_kernel"
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
