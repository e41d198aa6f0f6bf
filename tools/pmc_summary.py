"""Aggregate a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name (GPU box helper)."""
import csv, sys, collections, glob
paths = [p for a in sys.argv[1:] for p in glob.glob(a, recursive=True)]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in paths:
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for row in rd:
            name = row.get("Kernel_Name") or row.get("Kernel Name") or row.get("Name")
            cn = row.get("Counter_Name") or row.get("Counter Name")
            cv = row.get("Counter_Value") or row.get("Counter Value")
            if name is None or cn is None:
                continue
            a = acc[(name.split("(")[0], cn)]
            a[0] += float(cv); a[1] += 1
print("kernel,counter,dispatches,mean_value")
for (name, cn), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{name},{cn},{n},{s / n:.6g}")
