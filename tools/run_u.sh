cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03y_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r03y_pytest.log | tail -3
python - <<'PY'
import subprocess, sys, time
t = time.perf_counter()
r = subprocess.run([sys.executable, "bench.py"], capture_output=True, text=True)
open("gpurun_out/r03y_bench.json", "w").write(r.stdout); open("gpurun_out/r03y_bench.err", "w").write(r.stderr)
print("bench wall", round(time.perf_counter() - t, 1), "s rc", r.returncode)
PY
cut -c1-900 gpurun_out/r03y_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03y_bench.json").read().strip().splitlines()[-1])
print("setup_s", d.get("setup_s"), "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
for e in d.get("extra_configs", []):
    print({k: e[k] for k in e if k in ("workload", "ms", "gflops", "ms_per_iteration", "pc_setup_s", "setup_s", "ms_per_part")})
PY
