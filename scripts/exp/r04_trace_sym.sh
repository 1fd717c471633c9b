set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_sym; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "20000 600 32" "100000 768 32"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $O/tr_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$tag -- python3 $R/scripts/knn_only.py $cfg > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
  f=$(find $O/tr_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; python3 $R/scripts/exp/kernel_stats_table.py $f 22
  rm -rf $O/tr_$tag
done
