#!/bin/bash
# Reproduces the profiles committed under profiles/ (run on the GPU box from the repo root):
#   bash profiles/run_profiles.sh && python profiles/summarize.py r02
# Pass 1: kernel trace + stats of the default bench command.  Pass 2: the headline part only (no CG loop, no value
# dictionary) so that the last `steps` launches of k_spmv_rowsplit are exactly the timed region.  Passes 3-5: PMC
# counters, one set per pass, never combined with other trace domains.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
H="python $R/bench.py --no-cpu-baseline --no-value-dict --no-extra --cg-iters 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python $R/bench.py > $O/bench_kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_kth -o kth -- $H > $O/bench_kth.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $H --steps 10 > $O/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $H --steps 10 > $O/bench_write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/prof_tcc -o t -- $H --steps 10 > $O/bench_tcc.log 2>&1
tail -1 $O/bench_kt.log | cut -c1-300
