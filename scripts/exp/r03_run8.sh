set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
( for L in "" _u4 _u8; do echo "== lib$L"; export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip$L.so
  timeout -k 10 200 python scripts/mid_size_probe.py 2>&1 | cut -c1-175
  timeout -k 10 200 python scripts/exp/settle_loop.py 1000000 384 16 nochain 4 2>&1 | tail -1 | cut -c1-150
  OSC_SPMM_XS=0 timeout -k 10 200 python scripts/exp/settle_loop.py 100000 768 32 nochain 6 2>&1 | tail -1 | cut -c1-150
  timeout -k 10 200 python scripts/exp/settle_loop.py 50000 64 16 nochain 8 2>&1 | tail -1 | cut -c1-150
  timeout -k 10 200 python scripts/exp/settle_loop.py 300000 256 16 nochain 6 2>&1 | tail -1 | cut -c1-150
  done ) > $O/r03_u_general.txt 2>&1
cat $O/r03_u_general.txt
