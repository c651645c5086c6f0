cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03l_pytest.log 2>&1
tail -3 gpurun_out/r03l_pytest.log
S=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err
E=$(date +%s.%N)
echo "default bench wall seconds: $(echo "$E - $S" | bc)" | tee gpurun_out/r03l_bench_wall.log
timeout 600 python tests/fuzz/fuzz_spmv.py 300 940000 2>&1 | tail -2 > gpurun_out/r03l_fuzz_spmv.log
tail -1 gpurun_out/r03l_fuzz_spmv.log
