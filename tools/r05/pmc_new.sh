# SQ counters (the known-good sets of pmc_pl.sh) of the round-5 kernels on their benchmark shapes:
#   gconv_wr_kernel (first Encoder layer, split form: 60 <- 9, 7x7 @256), wgrad_c3_kernel (3 -> 60 7x7 weight gradient)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_pmc_new.txt; : > $OUT
run() {  # $1 = tag, $2 = kernel substring, $3 = micro_conv mode; shape from env
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
             "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_WAVES"; do
    i=$((i+1))
    rm -rf /tmp/p$1$i
    timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/p$1$i -o p -- python $R/tools/micro_conv.py $3 5 > /tmp/p$1$i.log 2>&1 || { echo "$1 pass $i failed/timeout" >> $OUT; }
    db=$(find /tmp/p$1$i -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $2 2>&1 | grep -v "^\['id'" >> $OUT
  done
}
export MN=16 MC=9 MK=60 MH=256 MR=7 MS=1 MPAD=3,3,3,3; run wr gconv_wr fwd
export MN=16 MC=3 MK=60 MH=256 MR=7 MS=1 MPAD=3,3,3,3; run c3 wgrad_c3_kernel wgrad
cat $OUT
