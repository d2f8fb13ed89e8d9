set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 $GRAFT_REPO_ROOT/tools/labbin/stream_lab > $O/stream_lab.txt 2>&1
rocprofv3 --list-avail > $O/counters_avail.txt 2>&1 || rocprofv3 -L > $O/counters_avail.txt 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$n -- $GRAFT_REPO_ROOT/tools/labbin/gemm_x3_lab 272115 200 0 200 2 > $O/pmc_$n.log 2>&1
  find /tmp/pmc_$n -name "*counter_collection.csv" -exec cp {} $O/pmc_$n.csv \;
done
ls -la $O
