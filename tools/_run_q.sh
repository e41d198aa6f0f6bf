for t in 50000 100000; do
python bench.py --steps 10 --warmup 2 --tets $t --no-cpu-baseline --no-fp64-line > gpurun_out/q_$t.json 2> gpurun_out/q_$t.err
python - <<PY
import json
d=json.loads(open("gpurun_out/q_$t.json").read().strip().splitlines()[-1])
print(d["config"]["tets"], "tets:", round(d["value"],2), "it/s", round(d["ms_per_step"],1), "ms/step", "newton", d["newton_iterations"], "krylov", d["krylov_iterations"], d["solver_events"])
print("   ", d["krylov_per_solve"])
PY
done
python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest_gpu_q.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu_q.log
python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/r03_bench_q.json 2>gpurun_out/r03_bench_q.err
python tools/show_bench.py gpurun_out/r03_bench_q.json | cut -c1-200
