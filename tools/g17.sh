cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/g17_bench.json 2> gpurun_out/g17_bench.err
echo bench_s $(( $(date +%s) - S )) rc $?
tail -3 gpurun_out/g17_bench.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/g17_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "lib defaults", d.get("ms_per_step_library_defaults"))
print(json.dumps(d["value_dictionary_mode"].get("seven_values"), indent=0)[:1500])
PY
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head
