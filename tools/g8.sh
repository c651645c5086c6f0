cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/g8_pytest.log 2>&1
echo pytest_s $(( $(date +%s) - S ))
grep -E "passed|failed|error" gpurun_out/g8_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/g8_pytest.log | head
bash profiles/run_profiles.sh 2>&1 | tail -2 | cut -c1-300
python profiles/summarize.py r06 > gpurun_out/g8_summary_print.json 2>&1
cp profiles/r06_summary.json profiles/r06_kernel_stats.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_kt gpurun_out/prof_kth gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_tcc
for i in 1 2; do timeout 600 python tools/hpcg_driver.py 1 256 30 > gpurun_out/g8_hpcg_$i.log 2>&1; grep -iE "official|GFLOP|setup|rating" gpurun_out/g8_hpcg_$i.log | tail -6; done
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/g8_bench.json 2> gpurun_out/g8_bench.err
echo bench_s $(( $(date +%s) - S )) rc $?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/g8_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "counter", d["roofline"].get("frac_counter"), "traffic", d["roofline"]["traffic"], "lib defaults", d.get("ms_per_step_library_defaults"))
PY
du -sh gpurun_out
