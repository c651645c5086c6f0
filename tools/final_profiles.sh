# What the round's committed profiles come from (run on the GPU box from the repo root; gpurun merges at most 64 MiB back, so the raw rocprofv3
# directories are removed once summarised):   gpurun --timeout 3400 -- 'bash tools/final_profiles.sh r06'
TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/final_pytest.log 2>&1
echo pytest_s $(( $(date +%s) - S ))
grep -E "passed|failed|error" gpurun_out/final_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/final_pytest.log | head
bash profiles/run_profiles.sh 2>&1 | tail -1 | cut -c1-200
python profiles/summarize.py $TAG > gpurun_out/final_summary_print.json 2>&1
cp profiles/${TAG}_summary.json profiles/${TAG}_kernel_stats.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_kt gpurun_out/prof_kth gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_tcc
rm -rf gpurun_out/k1_sq
bash profiles/run_k1_sq.sh > /dev/null 2>&1
python profiles/summarize_k1_sq.py ${TAG}_lean > gpurun_out/final_sq_print.json 2>&1; cp profiles/${TAG}_lean_k1_sq.json gpurun_out/
rm -rf gpurun_out/k1_sq
for i in 1 2; do timeout 600 python tools/hpcg_driver.py 1 256 30 > gpurun_out/final_hpcg_$i.log 2>&1; grep -h "hpcg_driver\] rating" gpurun_out/final_hpcg_$i.log | cut -c1-220; done
bash tools/probe/mg_kstats2.sh > gpurun_out/final_mgk.log 2>&1
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo bench_s $(( $(date +%s) - S )) rc $?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "counter", d["roofline"].get("frac_counter"), "traffic", d["roofline"]["traffic"], "lib defaults", d.get("ms_per_step_library_defaults"))
PY
du -sh gpurun_out
