cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export PA_HIP_LIBRARY=$PWD/partitionedarrays.jl_amd/libpa_hip.so.e2
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py tests/test_gpu_hpcg_mg.py tests/test_gpu_value_dict.py tests/test_gpu_exchange_chain.py -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
timeout 900 python tools/probe/pell_lean_time.py 256 128 mg > gpurun_out/g13_lean.log 2>&1; grep -v "^{" gpurun_out/g13_lean.log | sed 's/->.*: min/: min/' | cut -c1-150 | grep -v "lean=0"
