"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per (kernel, grid size), the mean
of each counter over the dispatches.  FETCH_SIZE / WRITE_SIZE are in KiB (x1024 for bytes); on
gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)."""
import csv, sys, glob, collections, os, re
FILTER = re.compile(os.environ.get("PMC_FILTER", "mlp|wgrad|quad|sample"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if not FILTER.search(name):
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
            acc[(short, row.get("Grid_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, grid), d in sorted(acc.items()):
    print(f"{k}  grid={grid}")
    for c, v in sorted(d.items()):
        print(f"   {c:36s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
