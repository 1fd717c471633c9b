set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "knn or panel or graph or config or wide_neighbor or end_to_end or degenerate or ties" > $O/r03_t9.log 2>&1; tail -3 $O/r03_t9.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_knn_c5b -- python3 $R/scripts/knn_only.py 200000 1536 64 > $O/r03_knn_c5b.log 2>&1
f=$(find $O/r03_knn_c5b -name "*kernel_stats.csv" | head -1); grep "k_mutual_ell\|k_knn_pref" $f | cut -c1-140
