# vector-memory-path counters on the batch workload, with and without the epilogue stores / the MFMA loops
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
for V in "A=1" "ZV_DBG=16" "ZV_DBG=2"; do
 echo "===== $V" >> $O/pmc2.txt
 ( echo "# pass a"; bash scripts/pmc.sh "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" $V
   echo "# pass b"; bash scripts/pmc.sh "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr" $V
   echo "# pass c"; bash scripts/pmc.sh "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" $V ) 2>&1 | grep -E "^#|resblock_pair_kernel<(64|128), ., false>" >> $O/pmc2.txt
done
cat $O/pmc2.txt
