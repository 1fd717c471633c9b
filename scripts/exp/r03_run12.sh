set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03_t6.log 2>&1; tail -4 $O/r03_t6.log
OSC_REORDER=1 timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03_t6_reorder.log 2>&1; tail -4 $O/r03_t6_reorder.log
bash scripts/profile_gpu.sh r03 > $O/r03_profile.log 2>&1; tail -2 $O/r03_profile.log
bash scripts/pmc_configs.sh r03 > $O/r03_pmc_configs.txt 2>&1; tail -14 $O/r03_pmc_configs.txt
timeout -k 10 600 python scripts/config_times.py > $O/r03_config_times.txt 2>&1; cut -c1-220 $O/r03_config_times.txt
timeout -k 10 300 python scripts/mid_size_probe.py > $O/r03_mid_size_probe.txt 2>&1
timeout -k 10 300 python scripts/locality_demo.py > $O/r03_locality_demo.txt 2>&1; cat $O/r03_locality_demo.txt
timeout -k 10 300 python scripts/request_latency.py > $O/r03_request_latency.txt 2>&1; tail -4 $O/r03_request_latency.txt
