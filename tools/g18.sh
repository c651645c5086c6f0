cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py tests/test_gpu_blas1_cg.py tests/test_gpu_hpcg_mg.py tests/test_gpu_value_dict.py -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
timeout 600 python tools/probe/pell_lean_time.py 256 mg 2>&1 | grep "MG-PCG" | grep "lean=1"
