"""Print ms/step and the sampled per-launch averages of the sweep kernels from bench JSON files."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    ph = d["phase_ms"]
    print(f"{f}: {d['value']:.3f} it/s {d['ms_per_step']:.1f} ms/step N{d['newton_iterations']} K{d['krylov_iterations']} "
          f"prec {ph['precond_ms']:.0f} spmv {ph['spmv_ms']:.0f} ortho {ph['ortho_ms']:.0f}")
    for k, v in d["kernels"].items():
        if any(t in k for t in ("tiled", "Schur", "sb_b3", "sb_h", "node6")):
            print(f"     {k[:34]:34s} {v['avg_launch_ms'] * 1e3:8.1f} us x {v['launches']:6d} = {v['gpu_ms']:7.1f} ms  {v['achieved_GBps']:6.0f} GB/s")
