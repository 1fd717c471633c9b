set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "12000 128 16" "16384 128 16" "20000 128 16" "20000 64 16" "25000 128 32" "30000 192 16" "12000 512 16" "10000 768 32"; do
  for MR in "16384,32768" "8192,8192"; do for MB in 2.0 0.5; do echo "== $CFG OSC_XS_MIN_ROWS=$MR OSC_BLK_MB=$MB"; OSC_XS_MIN_ROWS=$MR OSC_BLK_MB=$MB timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 20 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-18; done; done; done
