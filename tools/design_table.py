"""Print the measurement table of DESIGN.md §4 from profiles/<round>_final_bench.json, <round>_final_kernel_stats.csv and
<round>_pmc_traffic.json (so that the document quotes the committed files, not a transcription).

    python tools/design_table.py [r03]"""
import csv, json, sys
from pathlib import Path
P = Path(__file__).resolve().parent.parent / "profiles"
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = json.loads((P / f"{R}_final_bench.json").read_text().strip().splitlines()[-1])
tr = {r["Name"].replace("void fsi::", "").replace("fsi::", "").replace("(anonymous namespace)::", "").split("(")[0]: float(r["AverageNs"]) / 1e3
      for r in csv.DictReader(open(P / f"{R}_final_kernel_stats.csv"))}
pm = {k: (v["fetch_bytes"] + v["write_bytes"]) / 1e6 for k, v in json.loads((P / f"{R}_pmc_traffic.json").read_text())["detail"].items()}
steps, ms = d["steps"], d["ms_per_step"]
apps = d["phase_calls"]["precond_calls"]
print(f"{d['value']:.2f} Newton-it/s, {ms:.1f} ms/step, {d['newton_iterations']} Newton / {d['krylov_iterations']} Krylov, timed region {ms * steps:.0f} ms, "
      f"precond {d['phase_ms']['precond_ms'] / apps:.2f} ms per application")
def t(*names): return " + ".join(f"{tr[n]:,.0f}".replace(",", " ") if tr[n] >= 100 else f"{tr[n]:.1f}" for n in names if n in tr)
def p(*names): return sum(pm.get(n, 0.0) for n in names)
for k, v in d["kernels"].items():
    print(f"| {k[:60]} | {v['launches']} | {v['avg_launch_ms'] * 1e3:.1f} | {v['gpu_ms']:.0f} | {v['share_of_timed_region'] * 100:.1f} % | "
          f"{v['algorithmic_bytes_per_launch'] / 1e6:.0f} | {v['achieved_GBps']:.0f} | {v['frac_of_hbm_peak']:.2f} |")
print("trace [us]: dots/axpy f32", t("k_gcr_dots<float, 8>", "k_gcr_axpy<float>", "k_gcr_dots<float, 8, true>", "k_gcr_axpy<float, true>"), "| f64", t("k_gcr_dots<double, 4>", "k_gcr_dots<double, 8>", "k_gcr_axpy<double>", "k_gcr_dots<double, 8, false>", "k_gcr_axpy<double, false>"),
      "| node6p + tail", t("k_spmv_node6p<true>", "k_spmv<0, float>", "k_spmv_prow<float>"), "| node6 f64 + tail", t("k_spmv_node6<double, true>", "k_spmv<0, double>", "k_spmv_prow<double>"))
names = ["k_sweep_tiled_h<1>", "k_sweep_tiled_h<3>", "k_sweep_tiled_h<1, 256>", "k_sweep_tiled_h<3, 256>", "k_sweep_sb_h", "k_sweep_sb_b3<1>",
         "k_sweep_schur_tiled", "k_sweep_schur_tiled<64>", "k_sweep_sc_f32", "k_gcr_flush<4>", "k_residual<2>", "k_residual_gather",
         "k_jacobian<2, 2>", "k_vel_correct", "k_vel_correct32", "k_pres_rhs32", "k_spmv_db", "k_db_rows_sub", "k_pres_rows"]
for n in [n for n in names if n in tr or n in pm]:
    print(f"   {n:24s} trace {tr.get(n, float('nan')):9.1f} us   PMC {pm.get(n, float('nan')):9.1f} MB")
print("   PMC outer product f32", round(p("k_spmv_node6p<true>", "k_spmv<0, float>", "k_spmv_prow<float>")), "f64", round(p("k_spmv_node6<double, true>", "k_spmv<0, double>", "k_spmv_prow<double>")),
      "ortho f32 dots/axpy", round(pm.get("k_gcr_dots<float, 8>", 0) + pm.get("k_gcr_dots<float, 8, true>", 0)), round(pm.get("k_gcr_axpy<float>", 0) + pm.get("k_gcr_axpy<float, true>", 0)))
if "value_fp64_storage" in d:
    f = d["fp64_storage"]
    print(f"all-FP64 storage: {d['value_fp64_storage']:.2f} Newton-it/s, {d['ms_per_step_fp64_storage']:.1f} ms/step, {f['newton_iterations']} Newton / "
          f"{f['krylov_iterations']} Krylov; phases {({k: round(v) for k, v in f['phase_ms'].items()})}")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"cpu baseline: lifetime {c['value']:.3f} it/s, steady {c.get('steady_state', {}).get('value')}, refresh step {c.get('refresh_step', {}).get('seconds')} s, cores {c['cores']}")
