set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "5000 768 16" "4000 1024 32" "3000 1536 16" "5000 384 16" "6000 512 32" "2500 768 32" "3500 2048 16"; do
  for MR in "6144,32768" "2048,32768"; do echo "== $CFG OSC_XS_MIN_ROWS=$MR"; OSC_XS_MIN_ROWS=$MR timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 30 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-250; done; done
