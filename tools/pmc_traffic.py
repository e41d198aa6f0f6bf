"""Per-kernel HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_driver.py.

The counters are KiB at the L2's memory side (MI355X_MICROARCH.md §HBM).  Their scale depends on the access width, so
the driver first streams a known number of bytes with 4-, 8-, 16- and 32-byte loads and 4-, 8-, 16-byte stores per lane
(k_cal_read<T> / k_cal_write<T>); the factor bytes / (counter * 1024) of the matching width corrects every other kernel.

    python tools/pmc_traffic.py FETCH_summary.csv WRITE_summary.csv CAL_BYTES [CAL_FETCH_summary.csv CAL_WRITE_summary.csv]
        > profiles/r03_pmc_traffic.json
(the calibration kernels may come from a separate, short run of tools/pmc_driver.py: last two arguments)
"""
import csv, json, sys

if len(sys.argv) > 2 and sys.argv[1] == "--show":          # one line per kernel group of a file this script wrote
    j = json.load(open(sys.argv[2]))
    for k, g in j.get("groups", {}).items():
        if isinstance(g, dict):
            print("%-22s launches %6d (trace %6d)  PMC %9.1f MB / launch  algorithmic %9.1f MB  ratio %.2f" % (
                k, g["launches"], g["trace_launches"], g["pmc_bytes_per_launch"] / 1e6, g["algorithmic_bytes_per_launch"] / 1e6,
                g["traffic_over_algorithmic"]))
    sys.exit(0)

def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["kernel"]] = (float(row["mean_value"]), int(row["dispatches"]))
    return out

fetch, write, cal_bytes = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
cal_fetch, cal_write = (load(sys.argv[4]), load(sys.argv[5])) if len(sys.argv) > 5 else (fetch, write)
# the bench line of the profiled command (7th argument): which workload the counters belong to - bench.py only quotes them for
# a run of the same size
workload = {}
if len(sys.argv) > 6:
    try:
        j = json.loads(open(sys.argv[6]).read().strip().splitlines()[-1])
        workload = {"tets": j["config"]["tets"], "dofs": j["config"]["dofs"], "steps": j["steps"], "warmup": j["warmup"],
                    "storage": j["config"].get("storage", "mixed")}
    except Exception as e:                      # the counters are still valid; only the tag is missing
        workload = {"workload_error": str(e)}
def factor(table, kern, typ):
    for name, (v, n) in table.items():
        if kern in name and typ in name and v > 0:
            return cal_bytes / (v * 1024.0)
    return None
cal = {"read4": factor(cal_fetch, "k_cal_read", "<float>"), "read8": factor(cal_fetch, "k_cal_read", "<double>"),
       "read16": factor(cal_fetch, "k_cal_read", "<float, 4"), "read32": factor(cal_fetch, "k_cal_read", "<double, 4"),
       "write4": factor(cal_write, "k_cal_write", "<float>"), "write8": factor(cal_write, "k_cal_write", "<double>"),
       "write16": factor(cal_write, "k_cal_write", "<float, 4")}
# dominant load / store width of the solver kernels (bytes per lane and instruction)
WIDTH = [("k_gcr_dots<double", "read32", "write8"), ("k_gcr_axpy<double", "read32", "write16"),
         ("k_gcr_dots<float", "read16", "write8"), ("k_gcr_axpy<float", "read16", "write16"),
         ("k_gcr_flush", "read16", "write8"), ("k_gcr_update", "read8", "write8"), ("k_spmv_node6", "read8", "write8"),
         ("k_spmv<", "read8", "write8"), ("k_spmv_tiled_f32", "read4", "write16"), ("k_sweep_sb_b3", "read16", "write16"),
         ("k_sweep_csr_f32", "read4", "write4"), ("k_residual", "read8", "write8"), ("k_jacobian", "read8", "write8")]
per_launch, detail = {}, {}
for name in sorted(set(fetch) | set(write)):
    if "k_cal_" in name:
        continue
    rk, wk = "read8", "write8"
    for key, r_, w_ in WIDTH:
        if key in name:
            rk, wk = r_, w_
            break
    f = fetch.get(name, (0.0, 0))[0] * 1024.0 * (cal.get(rk) or 1.0)
    w = write.get(name, (0.0, 0))[0] * 1024.0 * (cal.get(wk) or 1.0)
    short = name.split("fsi::")[-1]
    per_launch[short] = f + w
    detail[short] = {"fetch_bytes": f, "write_bytes": w, "raw_fetch_KiB": fetch.get(name, (0.0, 0))[0],
                     "raw_write_KiB": write.get(name, (0.0, 0))[0], "launches": fetch.get(name, (0, 0))[1], "widths": [rk, wk]}
# Per kernel GROUP of bench.py's table: PMC bytes summed over every launch of the group's kernels, divided
# by the launches the bench line of the SAME run counts for the group, next to the algorithmic bytes per launch of that line -
# the profiled command runs with --warmup 0, so the library's counters and the trace cover the same launches.
groups = {}
if len(sys.argv) > 6:
    try:
        sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
        from bench import KERNEL_GROUPS
        bench_line = json.loads(open(sys.argv[6]).read().strip().splitlines()[-1])
        for key, names in KERNEL_GROUPS.items():
            rows = [(k, v) for k, v in detail.items() if any(k == nm or k.startswith(nm) for nm in names)]
            bk = next((k for k in bench_line["kernels"] if k.startswith(key) or (key == "Schur" and k.startswith("Schur"))), None)
            if not rows or bk is None or not bench_line["kernels"][bk]["launches"]:
                continue
            tot = sum((v["fetch_bytes"] + v["write_bytes"]) * v["launches"] for _, v in rows)
            bl = bench_line["kernels"][bk]["launches"]
            groups[key] = {"kernels": {k: v["launches"] for k, v in rows}, "trace_launches": sum(v["launches"] for _, v in rows),
                           "launches": bl, "pmc_bytes_total": tot, "pmc_bytes_per_launch": tot / bl,
                           "algorithmic_bytes_per_launch": bench_line["kernels"][bk]["algorithmic_bytes_per_launch"],
                           "traffic_over_algorithmic": tot / bl / max(bench_line["kernels"][bk]["algorithmic_bytes_per_launch"], 1.0)}
    except Exception as e:
        groups = {"error": str(e)}
print(json.dumps({**workload, "groups": groups, "unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB counters x 1024 x calibration factor of the access width)",
                  "calibration_bytes": cal_bytes, "calibration_factors": cal, "per_launch_bytes": per_launch, "detail": detail}, indent=1))
