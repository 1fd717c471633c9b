set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
( for S in 0 256 64; do echo "== OSC_SPMM_SLAB=$S"; OSC_SPMM_SLAB=$S timeout -k 10 300 python scripts/locality_demo.py 2>&1 | tail -2; done
  echo "== OSC_SPMM_DEEP=0"; OSC_SPMM_DEEP=0 timeout -k 10 300 python scripts/locality_demo.py 2>&1 | tail -2 ) > $O/r03_locality2.txt 2>&1
cat $O/r03_locality2.txt
OSC_REORDER=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_t4_reorder.log 2>&1; tail -5 $O/r03_t4_reorder.log
