cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -k "arena_places or device_side or eight_ranks or column_encodings or compacted_column or config5_fem or fem_matrix or laplacian or fdm" > gpurun_out/r03f_pytest.log 2>&1
tail -3 gpurun_out/r03f_pytest.log
PA_SETUP_TIMING=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cg-iters 0 --no-value-dict > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
PA_ARENA_PLAIN_VECTORS=1 PA_SETUP_TIMING=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cg-iters 0 --no-value-dict --no-extra > gpurun_out/r03f_bench_plain.json 2> gpurun_out/r03f_bench_plain.err
