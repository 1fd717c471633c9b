set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "300000 768 32" "400000 512 32" "500000 384 16" "700000 384 16" "1000000 384 16"; do
  for MG in 4 2 1; do echo "== $CFG OSC_XS_MIN_GROUPS=$MG"; OSC_XS_MIN_GROUPS=$MG timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 4 2>&1 | tail -1 | cut -c1-400 | sed 's/.*settle_ms=/settle_ms=/' ; done; done
