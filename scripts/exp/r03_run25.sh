set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "100000 64 16" "100000 32 16" "200000 64 32" "60000 64 32" "100000 48 16"; do
  for MC in 96 32; do echo "== $CFG OSC_XS_MIN_COLS=$MC"; OSC_XS_MIN_COLS=$MC timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 8 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-300; done; done
echo "== c4 window of 8 ranks"; for MC in 96 32; do OSC_XS_MIN_COLS=$MC OSC_XS_MIN_GROUPS=1 timeout -k 10 300 python scripts/shard_local_times.py c4 8 2>&1 | cut -c1-140; done
