set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/r03_bench_w1.json 2> $O/r03_bench_w1.err; echo "world1 rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/r03_bench_w1.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','n_gpus','scaling','comm','ms_per_step')})"
OSC_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/r03_bench_w2.out 2> $O/r03_bench_w2.err; echo "world2-on-one-device rc=$? (expected non-zero, no JSON line)"; cat $O/r03_bench_w2.out | head -3; grep "bench.py: rank" $O/r03_bench_w2.err | head -4
