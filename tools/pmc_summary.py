"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")[:64]
        if not (name.startswith("void k_") or name.startswith("k_")):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
