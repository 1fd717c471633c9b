set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -k "panel or knn or config or random_shapes or fullsize or sharded" > $O/r03_t8.log 2>&1; tail -4 $O/r03_t8.log
timeout -k 10 300 python scripts/locality_demo.py 2>&1 | cut -c1-200
OSC_KNN_PANEL_SCATTER=0 timeout -k 10 300 python scripts/locality_demo.py 2>&1 | cut -c1-200
timeout -k 10 300 python scripts/config_times.py c3 c4 2>&1 | cut -c1-130
