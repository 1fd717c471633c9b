set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_api_surface.py -x -q 2>&1 | tail -1
timeout -k 10 600 python scripts/config_times.py > $O/r03_config_times.txt 2>&1; cut -c1-140 $O/r03_config_times.txt
timeout -k 10 400 python scripts/shard_local_times.py c3 1 2 4 8 > $O/r03_shard3.txt 2>&1; timeout -k 10 400 python scripts/shard_local_times.py c5 1 4 >> $O/r03_shard3.txt 2>&1; timeout -k 10 500 python scripts/shard_local_times.py c4 1 8 >> $O/r03_shard3.txt 2>&1
cut -c1-110 $O/r03_shard3.txt
timeout -k 10 300 python scripts/mid_size_probe.py > $O/r03_mid.txt 2>&1; cut -c1-70 $O/r03_mid.txt
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'], 4), round(d['value'], 2), d['roofline']['traffic'])"; done
