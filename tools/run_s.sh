cd $GRAFT_REPO_ROOT
for f in "fuzz_hpcg.py 200 953000" "fuzz_mul.py 300 950000" "fuzz_spmv.py 150 951000" "fuzz_fem.py 200 952000" "fuzz_cg.py 40 954000" "fuzz_partitions.py 1000 955000"; do
  set -- $f
  timeout 900 python tests/fuzz/$1 $2 $3 2>&1 | tail -2 > gpurun_out/r03late_$1.log
  tail -1 gpurun_out/r03late_$1.log | cut -c1-300
done
