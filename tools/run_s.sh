cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "arena or generated_on_the_device or row_subsets" > gpurun_out/r03z_pytest.log 2>&1
tail -5 gpurun_out/r03z_pytest.log | cut -c1-300
PA_SETUP_TIMING=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err
grep -E "pa arena\] [+-]" gpurun_out/r03z_bench.err | cut -c1-200 | head -30
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03z_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["setup_s"])
for e in d.get("extra_configs", []):
    print(e["workload"][:50], e.get("ms"), e.get("gflops"), e.get("ms_per_iteration"), e.get("pc_setup_s"))
for e in d.get("general_csr", []):
    print(e)
print(d["cg_loop"] if "cg_loop" in d else "")
PY
