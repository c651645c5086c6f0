cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -q -k "config_4_at_full_size" 2>&1 | tail -40 | cut -c1-300
