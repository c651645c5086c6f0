cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, sys, time, json
t = time.perf_counter()
r = subprocess.run([sys.executable, "bench.py"], capture_output=True, text=True)
open("gpurun_out/r03y_bench.json", "w").write(r.stdout); open("gpurun_out/r03y_bench.err", "w").write(r.stderr)
print("bench wall", round(time.perf_counter() - t, 1), "s rc", r.returncode)
d = json.loads(r.stdout.strip().splitlines()[-1])
print(d["value"], d["roofline"]["traffic"], d["roofline"]["moved_bytes_per_launch"], d["roofline"]["traffic_source"])
print(r.stderr[-600:])
PY
