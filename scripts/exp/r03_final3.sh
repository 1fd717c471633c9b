set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_t14.log 2>&1; tail -3 $O/r03_t14.log
rm -rf $O/r03
bash scripts/profile_gpu.sh r03 > $O/r03_profile.log 2>&1; tail -1 $O/r03_profile.log
bash scripts/pmc_configs.sh r03 > $O/r03_pmc_configs.txt 2>&1; grep "k_apply_blocked\|k_spmm<16, 1, 0" $O/r03_pmc_configs.txt | cut -c1-220
timeout -k 10 600 python scripts/config_times.py > $O/r03_config_times.txt 2>&1; cut -c1-140 $O/r03_config_times.txt
timeout -k 10 400 python scripts/shard_local_times.py c3 1 2 4 8 > $O/r03_shard3.txt 2>&1; timeout -k 10 400 python scripts/shard_local_times.py c5 1 4 >> $O/r03_shard3.txt 2>&1; timeout -k 10 500 python scripts/shard_local_times.py c4 1 8 >> $O/r03_shard3.txt 2>&1
cut -c1-110 $O/r03_shard3.txt
timeout -k 10 300 python scripts/mid_size_probe.py > $O/r03_mid.txt 2>&1; cut -c1-70 $O/r03_mid.txt
