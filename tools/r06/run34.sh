R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run34; mkdir -p $O; cd $R
n=8
HIFIC_BENCH_REHEARSAL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus $n --steps 2 --warmup 1 > $O/n$n.out 2> $O/n$n.err
echo "n=$n rc=$? stdout lines: $(wc -l < $O/n$n.out)"
python - <<PY
import json
d=json.loads(open("$O/n$n.out").read().strip().splitlines()[-1])
r=d["rccl"]
print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["global_batch"], r["rccl_ranks"], r["buckets"], r["payload_sweep_ms_per_step"], r["one_rank_same_box_ms_per_step"], r["weak_scaling_eff"])
PY
grep -i "error\|Traceback" $O/n$n.err | head -5
