R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_elementwise.py -q -p no:cacheprovider -x -k "discriminator or upcat or upsample or model_losses or branch" 2>&1 | tail -2
HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_PROF_DUMP=0 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/uc -o uc -- python bench.py --steps 3 --warmup 2 --no-extras > /tmp/uc.log 2>&1
db=$(find /tmp/uc -name "*.db" | head -1); python tools/rocpd_stats.py $db | grep -i "upcat"
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
