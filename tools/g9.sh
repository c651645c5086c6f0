cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_exchange.py tests/test_gpu_exchange_chain.py -q 2>&1 | tail -30 | cut -c1-300
