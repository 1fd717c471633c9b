set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/exp/small_trace.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_small -- python3 $R/scripts/exp/small_trace.py > $O/r03_small.log 2>&1
f=$(find $O/r03_small -name "*kernel_stats.csv" | head -1); grep -i "settle_small\|fillBuffer\|copyBuffer" $f | cut -c1-160
