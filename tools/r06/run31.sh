# multi-rank control flow of bench.py rehearsed on one GPU (2 and 4 ranks on cuda:0, gloo transport)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run31; mkdir -p $O; cd $R
for n in 2 4; do
HIFIC_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus $n --steps 3 --warmup 2 > $O/n$n.out 2> $O/n$n.err
echo "n=$n rc=$? stdout lines: $(wc -l < $O/n$n.out)"
python - <<PY
import json
try:
    d=json.loads(open("$O/n$n.out").read().strip().splitlines()[-1])
    r=d["rccl"]
    print(d["n_gpus"], d["value"], d["ms_per_step"], r["backend"], r["rccl_ranks"], r["exposed_comm_ms"], r["payload_sweep_ms_per_step"], r["one_rank_same_box_ms_per_step"], r["weak_scaling_eff"])
except Exception as e:
    print("parse failed", e)
PY
tail -5 $O/n$n.err | cut -c1-300
done
