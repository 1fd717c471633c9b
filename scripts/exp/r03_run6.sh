set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_t3.log 2>&1; tail -8 $O/r03_t3.log
timeout -k 10 300 python scripts/shard_local_times.py 1 2 4 8 > $O/r03_shard2.txt 2>&1; cut -c1-150 $O/r03_shard2.txt
timeout -k 10 400 python scripts/config_times.py c3 c4 c5 > $O/r03_config_times.txt 2>&1; cut -c1-260 $O/r03_config_times.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/r03_bench0.json 2> $O/r03_bench0.err; tail -c 1500 $O/r03_bench0.json
