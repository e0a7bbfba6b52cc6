# item 8: one rank through RCCL - CU budget of the collective kernels (NCCL_MAX_NCHANNELS) x gradient payload, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run10; mkdir -p $O; cd $R
timeout 200 python bench.py --steps 6 --warmup 3 --no-extras > $O/plain.json 2>$O/plain.err
python -c "import json;d=json.load(open('$O/plain.json'));print('no reducer:', d['ms_per_step'])"
for pay in f32 bf16; do
  for ch in default 2 4 8 16; do
    if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
    HIFIC_FORCE_DIST=1 HIFIC_GRAD_PAYLOAD=$pay timeout 200 python bench.py --steps 6 --warmup 3 --no-extras > $O/r_${pay}_$ch.json 2>$O/r_${pay}_$ch.err
    python -c "
import json
d=json.load(open('$O/r_${pay}_$ch.json'))
print('payload $pay, NCCL_MAX_NCHANNELS=$ch:', d['ms_per_step'], 'ms; exposed', d.get('rccl',{}).get('exposed_comm_ms'), 'one-rank eff', d.get('rccl',{}).get('weak_scaling_eff'))"
  done
done
echo "--- practical peak (MFMA-only twin) vs the real kernel, trunk launch"
MOPS=fwd timeout 120 python tools/micro_sp9.py 40 2>&1 | grep sp9
MOPS=fwd HIFIC_LIB_PATH=$R/high-fidelity-generative-compression_amd/libhific_hip_mfma_only.so timeout 120 python tools/micro_sp9.py 40 2>&1 | grep sp9
