set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
( timeout -k 10 300 python scripts/shard_local_times.py c3 1 2 4 8 2>&1 | cut -c1-150
  timeout -k 10 400 python scripts/shard_local_times.py c5 1 4 2>&1 | cut -c1-150
  timeout -k 10 500 python scripts/shard_local_times.py c4 1 8 2>&1 | cut -c1-150 ) > $O/r03_shard3.txt 2>&1
cat $O/r03_shard3.txt
( for L in "" _u4 _u8; do echo "== lib$L"; OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip$L.so timeout -k 10 300 python scripts/locality_demo.py 2>&1; done ) > $O/r03_locality_u.txt 2>&1
cat $O/r03_locality_u.txt
