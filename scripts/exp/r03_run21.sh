set -u
R=$GRAFT_REPO_ROOT
for L in sl3 sl5; do export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_$L.so
  for B in 7 8 9 10 12 14 16; do echo "== $L nb=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 8 2>&1 | tail -1 | cut -c1-140; done; done
