cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MN=16 MC=${MC:-480} MK=${MK:-960} MH=${MH:-32} MS=2 MR=3
OUT=$R/gpurun_out/r05_pmc_pl.txt; : > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/p$i -o p$i -- python $R/tools/micro_conv.py fwd 5 > /tmp/p$i.log 2>&1 || { echo "pass $i failed/timeout" >> $OUT; tail -3 /tmp/p$i.log >> $OUT; }
  db=$(find /tmp/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db gconv_pl >> $OUT 2>&1
done
cat $OUT
