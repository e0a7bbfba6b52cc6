cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MC=60 MK=120 MH=256 MS=2
rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/p$i -o p$i -- python $R/tools/micro_conv.py wgrad 5 > /tmp/p$i.log 2>&1 || tail -5 /tmp/p$i.log
  db=$(find /tmp/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db wgrad_kernel >> $R/gpurun_out/pmc_s2wgrad.txt 2>&1
done
