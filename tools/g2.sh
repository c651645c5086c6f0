cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py -q -x 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_spmv_kernels.py tests/test_gpu_value_dict.py tests/test_gpu_hpcg_mg.py tests/test_gpu_mul.py tests/test_gpu_exchange_chain.py tests/test_gpu_blas1_cg.py tests/test_gpu_f32.py -q -x 2>&1 | tail -15
timeout 600 python tests/fuzz/fuzz_spmv.py 40 424200 2>&1 | tail -3
timeout 900 python tools/probe/pell_lean_time.py 256 128 mg > gpurun_out/g2_lean.log 2>&1; tail -22 gpurun_out/g2_lean.log
