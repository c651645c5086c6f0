cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03y_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r03y_pytest.log | tail -3
python - <<'PY'
import subprocess, sys, time
t = time.perf_counter()
r = subprocess.run([sys.executable, "bench.py"], capture_output=True, text=True)
open("gpurun_out/r03y_bench.json", "w").write(r.stdout); open("gpurun_out/r03y_bench.err", "w").write(r.stderr)
print("bench wall", round(time.perf_counter() - t, 1), "s rc", r.returncode)
PY
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03y_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "setup_s", d.get("setup_s"), "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
print(d["roofline"]["memory_classes"]["arena"]["class_gib"], d["roofline"]["memory_classes"].get("vectors"))
for e in d.get("extra_configs", []):
    print({k: e[k] for k in e if k in ("ms", "gflops", "ms_per_iteration", "pc_setup_s", "setup_s", "ms_per_part", "iterations")}, e["workload"][:40])
for e in d.get("general_csr", []):
    print(e["ms"], e["frac_moved"], e["bit_identical_to_headline_product"])
print(d.get("cg_loop", {}).get("ms_per_iteration_opt_cg"))
PY
