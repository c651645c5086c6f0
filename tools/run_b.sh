cd $GRAFT_REPO_ROOT
(cd tools/probe && timeout 600 ./extent_probe 16 14) > gpurun_out/r03b_extent.log 2>&1
PA_SETUP_TIMING=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --cg-iters 0 --no-value-dict > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest.log 2>&1
tail -3 gpurun_out/r03b_pytest.log
