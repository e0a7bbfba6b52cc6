# SQ / HBM counters for one micro_conv shape (default: the 60->120 stride-2 weight gradient), one counter family per
# rocprofv3 pass, every pass under `timeout`.  Only counter sets known to work on this pool: a pass with TCP_*/TCC_*/TA_*
# names (or FETCH_SIZE and WRITE_SIZE together) hung for 10 minutes in round 1 - do not add them back untested.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MC=${MC:-60} MK=${MK:-120} MH=${MH:-256} MS=${MS:-2}
OUT=$R/gpurun_out/pmc_micro.txt; : > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/p$i -o p$i -- python $R/tools/micro_conv.py ${1:-wgrad} 5 > /tmp/p$i.log 2>&1 || { echo "pass $i failed/timeout"; tail -3 /tmp/p$i.log; }
  db=$(find /tmp/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db >> $OUT 2>&1
done
