import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hist = collections.Counter()
tot = collections.Counter()
for i, r in enumerate(rows):
    if "fillBuffer" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        nxt = rows[i + 1]["Kernel_Name"][:50] if i + 1 < len(rows) else "-"
        b = "<10us" if d < 10 else "<30us" if d < 30 else "<100us" if d < 100 else "<1ms" if d < 1000 else ">=1ms"
        hist[(b, nxt)] += 1
        tot[(b, nxt)] += d
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{k[0]:7s} next={k[1]:50s} calls {hist[k]:6d} total {v/1e3:8.2f} ms")
