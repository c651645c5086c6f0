cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "device_side_encoding or eight_ranks or arena_places or column_encodings or compacted_column" > gpurun_out/r03d_pytest.log 2>&1
tail -3 gpurun_out/r03d_pytest.log
PA_SETUP_TIMING=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cg-iters 0 --no-value-dict > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
timeout 1500 bash profiles/run_utcl.sh > gpurun_out/r03d_utcl.log 2>&1
python profiles/summarize_utcl.py r03 > gpurun_out/r03d_utcl_summary.log 2>&1
cp profiles/r03_utcl.json gpurun_out/ 2>/dev/null
