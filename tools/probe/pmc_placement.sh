cd /tmp && export TMPDIR=/tmp
P=/root/repo/tools/probe/placement_probe
O=/root/repo/gpurun_out/pmcp
rm -rf $O; mkdir -p $O
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 250 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i --output-format csv -- $P 256 8 pmc > $O/p$i.log 2>&1
  grep "copy" $O/p$i.log | head -16
done
ls -R $O | head -30
