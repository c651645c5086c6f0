cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03n_pytest.log 2>&1
tail -3 gpurun_out/r03n_pytest.log
python - <<'PY'
import subprocess, time, sys
t = time.time()
r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5"], stdout=open("gpurun_out/r03n_bench.json", "w"), stderr=open("gpurun_out/r03n_bench.err", "w"))
print("default bench: rc", r.returncode, "wall seconds", round(time.time() - t, 1))
PY
