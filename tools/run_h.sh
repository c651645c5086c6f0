cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03h_pytest.log 2>&1
tail -3 gpurun_out/r03h_pytest.log
for f in "fuzz_mul.py 500 930000" "fuzz_spmv.py 200 931000" "fuzz_fem.py 300 932000" "fuzz_hpcg.py 120 933000" "fuzz_cg.py 50 934000" "fuzz_partitions.py 1500 935000"; do
  set -- $f
  timeout 900 python tests/fuzz/$1 $2 $3 2>&1 | tail -2 > gpurun_out/r03h_$1.log
  tail -1 gpurun_out/r03h_$1.log
done
PA_SETUP_TIMING=1 timeout 600 python tools/probe/setup_profile.py 256 > gpurun_out/r03h_setup_profile.log 2>&1
