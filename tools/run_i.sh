cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03i_pytest.log 2>&1
tail -3 gpurun_out/r03i_pytest.log
PA_SETUP_TIMING=1 timeout 600 python tools/probe/setup_profile.py 256 > gpurun_out/r03i_setup_profile.log 2>&1
grep "function calls" gpurun_out/r03i_setup_profile.log
/usr/bin/time -v -o gpurun_out/r03i_bench_time.log python bench.py --steps 20 --warmup 5 > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err
grep "Elapsed" gpurun_out/r03i_bench_time.log
bash profiles/run_profiles.sh > gpurun_out/r03i_profiles.log 2>&1
python profiles/summarize.py r03 > gpurun_out/r03i_summarize.log 2>&1
cp profiles/r03_summary.json profiles/r03_kernel_stats.csv profiles/r03_bench_n1.json gpurun_out/ 2>/dev/null
ls profiles | grep r03
