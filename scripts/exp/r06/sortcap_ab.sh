ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
for rep in 1 2; do
for c in 2560 5120 10240 1280; do
  export OSC_KNN_SORT_CAP=$c
  O=$ROOT/gpurun_out/sortcap_${c}_$rep; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $ROOT/scripts/knn_only.py $1 $2 $3 > $O/run.log 2>&1
  f=$(find $O/t -name "*kernel_stats.csv" | sort | tail -1)
  echo "cap $c (rep $rep): $(grep build_ms $O/run.log)  select: $(grep panel_select $f | sed 's/(.*)"/"/' | cut -d, -f4)"
done
done
