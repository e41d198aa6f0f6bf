"""From a rocprofv3 kernel trace (csv): how much of the GPU's busy time had two or more kernels in flight, and the busy
time itself - the measure of what the two-stream preconditioner application gains (tools/gpu_r4_c.sh)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, busy, over = 0, None, 0, 0
for t, d in ev:
    if last is not None and depth > 0:
        busy += t - last
        if depth > 1:
            over += t - last
    depth += d
    last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{len(rows)} kernels: sum of kernel durations {tot / 1e6:.1f} ms, GPU busy (union) {busy / 1e6:.1f} ms, of which >= 2 kernels in flight {over / 1e6:.1f} ms "
      f"({100.0 * over / max(busy, 1):.1f} %)")
