cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python tests/fuzz/fuzz_spmv.py 60 424200 2>&1 | tail -4
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/g1_pytest.log 2>&1
echo pytest_s $(( $(date +%s) - S ))
grep -E "passed|failed|error" gpurun_out/g1_pytest.log | tail -3
