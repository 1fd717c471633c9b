set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_nb48.so
( for B in 24 28 32 40 48; do echo "== C5 nb48 lib OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 200000 1536 64 chain 6 2>&1 | tail -1 | cut -c1-200; done
  for B in 9 12 14 16 20 24 32; do echo "== C3 nb48 lib OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 8 2>&1 | tail -1 | cut -c1-200; done
  for B in 4 6 8 12 16; do echo "== 40000x256x32 OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 40000 256 32 nochain 12 2>&1 | tail -1 | cut -c1-200; done
  for B in 0 16 32 48; do echo "== C4 OSC_SPMM_XS=1 OSC_SPMM_BLOCKED=$B"; OSC_SPMM_XS=1 OSC_SPMM_BLOCKED=$B timeout -k 10 200 python3 $R/scripts/exp/settle_loop.py 1000000 384 16 nochain 4 2>&1 | tail -1 | cut -c1-200; done
) > $O/r03_nb_sweep.txt 2>&1
cat $O/r03_nb_sweep.txt
