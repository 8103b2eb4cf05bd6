"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, mean of each counter."""
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "mlp" not in k and "wgrad" not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
