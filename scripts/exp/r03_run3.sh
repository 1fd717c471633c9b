set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
( for LIB in nb48 nb48nt; do
  export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_$LIB.so
  for B in 12 16 20 24 28; do echo "== C5 $LIB OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 200000 1536 64 chain 6 2>&1 | tail -1 | cut -c1-160; done
  for B in 7 9 10 12; do echo "== C3 $LIB OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 8 2>&1 | tail -1 | cut -c1-160; done
  for B in 6 8 9 10; do echo "== 40000x256x32 $LIB OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 40000 256 32 nochain 12 2>&1 | tail -1 | cut -c1-160; done
  done
  for LIB in nb48nt nb48nt_tu; do export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_$LIB.so; echo "== windows $LIB"; timeout -k 10 200 python3 $R/scripts/shard_local_times.py 1 4 8 2>&1 | cut -c1-150; done
) > $O/r03_nt_sweep.txt 2>&1
cat $O/r03_nt_sweep.txt
