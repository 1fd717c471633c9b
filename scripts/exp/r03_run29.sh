set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "40000 256 32" "60000 256 16" "100000 192 16" "50000 512 16" "100000 384 32" "30000 768 32"; do
  for T in 0 200 400 800 100000; do echo "== $CFG OSC_TEMPORAL_MB=$T"; OSC_TEMPORAL_MB=$T timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 16 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-16; done; done
