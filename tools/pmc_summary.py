"""Aggregate a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name (GPU box helper)."""
import csv, sys, collections, glob, re
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
            name = name.replace("(anonymous namespace)::", "")
            name = re.sub(r"^void ", "", name)
            depth, cut = 0, len(name)          # cut at the parenthesis that opens the parameter list (template args may hold none)
            for i, ch in enumerate(name):
                if ch == "<": depth += 1
                elif ch == ">": depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i
                    break
            a = acc[(name[:cut], cn)]
            a[0] += float(cv); a[1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_value"])
for (name, cn), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    w.writerow([name, cn, n, f"{s / n:.6g}"])
