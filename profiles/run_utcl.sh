#!/bin/bash
# Address-translation counters of the product kernel at 256^3 and 512^3 rows per part ("settle the large-part
# loss with counters").  One counter set per pass, never combined with trace domains.  Run on the GPU box from the repo root:
#   bash profiles/run_utcl.sh && python profiles/summarize_utcl.py r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/utcl
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
for n in 256 512; do
  for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_UTCL2_BUSY"; do
    tag=$(echo $set | tr ' ' '+' | cut -c1-60)
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/n${n}_$tag -o c -- python $R/tools/probe/big_part.py $n > $O/n${n}_$tag.log 2>&1
    tail -1 $O/n${n}_$tag.log | cut -c1-200
  done
done
