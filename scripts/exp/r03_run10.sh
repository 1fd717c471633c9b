set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
( for S in 256 384 512 768; do echo "== OSC_SPMM_SLAB=$S"; OSC_SPMM_SLAB=$S timeout -k 10 300 python scripts/locality_demo.py 2>&1 | tail -1; done ) > $O/r03_locality3.txt 2>&1
cat $O/r03_locality3.txt
( for NRG in 1 2; do echo "== OSC_KNN_PANEL_NRG=$NRG"; OSC_KNN_PANEL_NRG=$NRG timeout -k 10 300 python scripts/config_times.py c4 2>&1 | cut -c1-200
  OSC_KNN_PANEL_NRG=$NRG timeout -k 10 300 python scripts/exp/settle_loop.py 200000 384 32 nochain 2 2>&1 | tail -1 | cut -c1-120
  OSC_KNN_PANEL_NRG=$NRG timeout -k 10 300 python scripts/exp/settle_loop.py 100000 256 16 nochain 2 2>&1 | tail -1 | cut -c1-120; done ) > $O/r03_nrg.txt 2>&1
cat $O/r03_nrg.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or knn or config4 or random_shapes or block_major" > $O/r03_t5.log 2>&1; tail -5 $O/r03_t5.log
