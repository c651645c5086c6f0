cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03e_pytest.log 2>&1
tail -3 gpurun_out/r03e_pytest.log
PA_SETUP_TIMING=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cg-iters 0 --no-value-dict > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
