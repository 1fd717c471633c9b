set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -rf $O/r03
bash scripts/profile_gpu.sh r03 > $O/r03_profile.log 2>&1; tail -1 $O/r03_profile.log
timeout -k 10 400 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err; tail -c 400 $O/r03_bench.json
timeout -k 10 600 python scripts/config_times.py > $O/r03_config_times.txt 2>&1
timeout -k 10 300 python scripts/locality_demo.py > $O/r03_locality_demo.txt 2>&1
timeout -k 10 300 python scripts/request_latency.py > $O/r03_request_latency.txt 2>&1
timeout -k 10 300 python scripts/mid_size_probe.py > $O/r03_mid_size_probe.txt 2>&1
timeout -k 10 400 python scripts/shard_local_times.py c3 1 2 4 8 > $O/r03_shard3.txt 2>&1; timeout -k 10 400 python scripts/shard_local_times.py c5 1 4 >> $O/r03_shard3.txt 2>&1; timeout -k 10 500 python scripts/shard_local_times.py c4 1 8 >> $O/r03_shard3.txt 2>&1
cut -c1-150 $O/r03_shard3.txt
