cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/g4_pytest.log 2>&1
echo pytest_s $(( $(date +%s) - S ))
grep -E "passed|failed|error" gpurun_out/g4_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/g4_pytest.log | head
rm -rf gpurun_out/k1_sq
bash profiles/run_k1_sq.sh 2>&1 | tail -12
python profiles/summarize_k1_sq.py r06_lean > gpurun_out/g4_sq_summary.json 2>&1; cp profiles/r06_lean_k1_sq.json gpurun_out/
