cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_exchange.py -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
