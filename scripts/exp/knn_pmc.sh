# FETCH_SIZE / L2 hit of the prefilter kernel for a given OSC_KNN_SPLITS (arg 1) and optional library variant (arg 2)
S=$1; V=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export OSC_KNN_SPLITS=$S
if [ -n "$V" ]; then export OSC_LIB_PATH=$ROOT/oscillink_amd/liboscillink_hip_$V.so; fi
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo S${S}_${V}_$C | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/gpurun_out/knn_pmc/$n -- python3 $ROOT/scripts/knn_only.py > /dev/null 2>&1
done
