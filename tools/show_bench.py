"""Print the key fields of bench.py output files (the JSON line may be preceded by transport chatter)."""
import json
import sys
for path in sys.argv[1:]:
    try:
        line = [l for l in open(path, errors="replace") if l.startswith("{")][-1]
        d = json.loads(line)
        keys = ("value", "n_gpus", "ms_per_step", "newton_iterations", "krylov_iterations", "setup_s")
        print(path, {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in keys}, d["config"]["parallelism"],
              {k: round(v, 1) for k, v in d["phase_ms"].items()}, "roofline", d["roofline"]["kernel"][:24],
              round(d["roofline"]["frac"], 3), d["config"].get("storage"),
              {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k.startswith(("value_", "ms_per_step_"))},
              {k: v for k, v in d.get("solver_events", {}).items() if v},
              "precond ms/app %.3f" % (d["phase_ms"]["precond_ms"] / max(1, d["phase_calls"]["precond_calls"])))
    except Exception as e:
        print(path, "no bench line:", e)
