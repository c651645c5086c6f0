cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_f32.py -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
PA_CTX_PER_PART=1 timeout 600 python -m pytest tests/test_gpu_exchange.py -q -k "hand_partition or doc_examples" 2>&1 | grep -E "passed|failed|FAILED|Error" | head
