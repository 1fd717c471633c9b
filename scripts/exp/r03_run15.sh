set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_t7.log 2>&1; tail -4 $O/r03_t7.log
python3 scripts/exp/small_trace.py
