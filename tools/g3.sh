cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py -q -k "drop or colour_sweeps" 2>&1 | grep -v "^$" | cut -c1-400 | tail -80
