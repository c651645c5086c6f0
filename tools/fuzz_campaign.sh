# A long run of the fuzzers on the final tree of a round, default routes, the round-4 switches and round 6's (pattern-ELL's masked form, the row-split one-byte stream, few-valued blocks): gpurun --timeout 3000 -- 'bash tools/fuzz_campaign.sh'
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
L=gpurun_out/r06_fuzz_campaign.log
: > $L
run() { echo "== $*" >> $L; ( "$@" 2>&1 | grep -v amdgpu.ids | tail -2 ) >> $L; }
run timeout 900 python tests/fuzz/fuzz_spmv.py 150 990000
run timeout 900 python tests/fuzz/fuzz_mul.py 300 991000
run timeout 900 python tests/fuzz/fuzz_fem.py 300 992000
run timeout 900 python tests/fuzz/fuzz_exchange.py 1500 993000
run timeout 900 python tests/fuzz/fuzz_partitions.py 1000 994000
run timeout 900 python tests/fuzz/fuzz_hpcg.py 100 995000
run timeout 900 python tests/fuzz/fuzz_cg.py 40 996000
for sw in PA_SPMV_VALUE_DICT=1 PA_CTX_PER_PART=1 PA_PUSH=0 PA_MUL_GHOST_FROM_BUFFER=0 PA_SPMV_COLSPLIT=3; do
  run env $sw timeout 900 python tests/fuzz/fuzz_spmv.py 60 997000
  run env $sw timeout 900 python tests/fuzz/fuzz_mul.py 150 997100
  run env $sw timeout 900 python tests/fuzz/fuzz_fem.py 150 997200
  run env $sw timeout 900 python tests/fuzz/fuzz_exchange.py 500 997300
done
for sw in PA_SPMV_PELL_LEAN=0 PA_SPMV_PELL_BYTES=0 PA_SPMV_PELL_CLASSES=0; do
  run env $sw PA_SPMV_VALUE_DICT=1 timeout 900 python tests/fuzz/fuzz_spmv.py 60 997600 --few-values
  run env $sw timeout 900 python tests/fuzz/fuzz_hpcg.py 40 997700
done
run env PA_SPMV_VALUE_DICT=1 timeout 900 python tests/fuzz/fuzz_spmv.py 150 997800 --few-values
run env PA_SPMV_VALUE_DICT=1 timeout 900 python tests/fuzz/fuzz_spmv.py 80 997900 --two-values
run env PA_SPMV_VALUE_DICT=1 timeout 900 python tests/fuzz/fuzz_hpcg.py 60 997400
run env PA_SPMV_VALUE_DICT=1 timeout 900 python tests/fuzz/fuzz_cg.py 20 997500
run env PA_TRANSPORT=ipc timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29741 tests/fuzz/fuzz_dist_driver.py 60 998000
grep -c "== " $L; grep -i "mismatch\|error\|Traceback" $L | sort | uniq -c | head -20
