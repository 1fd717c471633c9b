set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
( for B in -1 0 8 12 16; do echo "== OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 200000 1536 64 chain 6 2>&1 | tail -1; done
  for B in 16 18 20 24; do echo "== nb24 lib OSC_SPMM_BLOCKED=$B"; OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_nb24.so OSC_SPMM_BLOCKED=$B timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 200000 1536 64 chain 6 2>&1 | tail -1; done
  for G in 2 4 8; do echo "== OSC_XS_GROUPS=$G"; OSC_XS_GROUPS=$G timeout -k 10 120 python3 $R/scripts/exp/settle_loop.py 200000 1536 64 chain 6 2>&1 | tail -1; done
) > $O/r03_c5_hyp.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for W in 8 4; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_win$W -- python3 $R/scripts/exp/trace_window.py $W > $O/r03_win$W.log 2>&1; done
cd $R && bash scripts/pmc_configs.sh r03 > $O/r03_pmc_configs.txt 2>&1
cat $O/r03_c5_hyp.txt | cut -c1-260
cat $O/r03_pmc_configs.txt
