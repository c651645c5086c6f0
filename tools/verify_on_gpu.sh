# The checks run on the GPU box before a commit that touches the path: smoke, the whole -m gpu suite, two fuzzers, the default
# bench line.  gpurun --timeout 3600 -- 'bash tools/verify_on_gpu.sh'
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PA_TEST_EXTENDED=1 timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/verify_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/verify_pytest.log | tail -3
timeout 900 python tests/fuzz/fuzz_fem.py 300 962000 2>&1 | tail -1
timeout 900 python tests/fuzz/fuzz_mul.py 200 963000 2>&1 | tail -1
python - <<'PY'
import subprocess, sys, time, json
t = time.perf_counter()
r = subprocess.run([sys.executable, "bench.py"], capture_output=True, text=True)
open("gpurun_out/verify_bench.json", "w").write(r.stdout); open("gpurun_out/verify_bench.err", "w").write(r.stderr)
print("bench wall", round(time.perf_counter() - t, 1), "s rc", r.returncode)
d = json.loads(r.stdout.strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "setup_s", d.get("setup_s"), "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for e in d.get("extra_configs", []):
    print({k: e[k] for k in e if k in ("ms", "gflops", "ms_per_iteration", "pc_setup_s", "setup_s", "ms_per_part", "iterations")}, e["workload"][:40])
PY
