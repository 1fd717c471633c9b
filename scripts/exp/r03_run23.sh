set -u
R=$GRAFT_REPO_ROOT
cd $R
export OSC_XS_MIN_GROUPS=2
for CFG in "400000 512 32" "500000 384 16" "300000 768 64"; do
  for B in 0 -1 8 12 16 20 24 32; do echo "== $CFG nb=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 4 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-20; done; done
